"""What the REFERENCE does with birefringence on a ray that is reflected off the bottom of the ice shelf
(analyticraytracing.py:2369-2445 with n_reflections > 0, :2118-2130) -- the combination nuradiomc_amd refuses.

    cp -r /root/reference /tmp/refcopy
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/probe_bire_reflection.py \
        > tests/golden/ref_bire_reflection_probe.txt

get_pulse_propagation_birefringence takes acc = int(path length / m) steps of `get_path(i_solution, n_points=acc)`.  For a solution
with k bottom reflections get_path returns the concatenation of its k + 1 segments with n_points EACH (one point repeated at every
reflection), i.e. (k + 1) acc - k points about 1 / (k + 1) m apart -- and the loop still runs over the first acc - 1 steps only:
the pulse is propagated along the first 1 / (k + 1) of the path (down to about the first reflection), in steps of half (a third ...)
the intended length, and the repeated point gives a zero-length step (NaN direction, skipped with a warning) whenever it falls
inside that range.  The result is therefore not the birefringent propagation along the ray; it depends on where the first segment
ends.  The printout below records exactly that for one vertex / receiver pair in Moore's Bay ice."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402

ice = medium.get_ice_model('mooresbay_simple')
config = {'propagation': dict(attenuate_ice=False, focusing_limit=2, focusing=False, birefringence=True,
                              birefringence_model='southpole_A', birefringence_propagation='analytical', n_reflections=1)}
r = ray.ray_tracing(ice, n_reflections=1, config=config)
x1, x2 = np.array([300., 0., -300.]), np.array([0., 0., -100.])
r.set_start_and_end_point(x1, x2)
r.find_solutions()
print('medium mooresbay_simple, reflective layer at z = %.0f m; vertex %s, receiver %s; %d solutions' % (ice.reflection, x1, x2, r.get_number_of_solutions()))
for iS in range(r.get_number_of_solutions()):
    res = r.get_results()[iS]
    D = r.get_path_length(iS)
    acc = int(D / units.m)
    path = r.get_path(iS, n_points=acc)
    ln = np.linalg.norm(np.diff(path, axis=0), axis=1)
    covered = ln[:acc - 1].sum()
    print('solution %d: type %d, bottom reflections %d: path length %.2f m -> acc = %d steps requested; get_path returns %d points, '
          'step lengths %.3f .. %.3f m, %d zero-length steps (first at step %s); the loop `for i in range(acc - 1)` covers %.2f m = %.1f %% of the path'
          % (iS, res['type'], res['reflection'], D, acc, len(path), ln[ln > 0].min(), ln.max(), int((ln == 0).sum()),
             (np.flatnonzero(ln == 0)[0] if (ln == 0).any() else '-'), covered, 100 * covered / ln.sum()))
