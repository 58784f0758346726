"""Host-side helpers of the drop-in classes that need no GPU."""
import numpy as np
from conftest import golden


def test_ray_path_vs_reference_get_path():
    """nuradiomc_amd.propagation.analytic_ray_path (what ray_tracing.get_path returns) against paths drawn by the reference's
    ray_tracing.get_path (analyticraytracing.py:2148-2162) for its own solutions -- both orders of the end points, direct,
    refracted and reflected rays (tests/golden/gen/gen_paths.py)."""
    from nuradiomc_amd.propagation import analytic_ray_path
    g = golden('ref_paths.npz')
    n = 0
    for i in range(len(g['x1'])):
        for iS in range(g['n_sol'][i]):
            ref = g['path'][i, iS]
            p = analytic_ray_path(g['x1'][i], g['x2'][i], g['C0'][i, iS], *g['ice'], n_points=ref.shape[0])
            assert p.shape == ref.shape and np.max(np.abs(p - ref)) < 1e-4      # metres, paths of ~1 km (observed 3e-6)
            lo, hi = (g['x1'][i], g['x2'][i]) if g['x2'][i][2] >= g['x1'][i][2] else (g['x2'][i], g['x1'][i])
            assert np.max(np.abs(p[0] - lo)) < 1e-6 and np.max(np.abs(p[-1] - hi)) < 1e-3   # ends on the receiver (C0 to 1e-7)
            n += 1
    assert n >= 30


def test_bire_with_bottom_reflections_is_ill_defined_in_the_reference():
    """The one combination of the ray tracer the product refuses (birefringence along bottom-reflected paths): the committed
    printout of the reference itself (tests/golden/gen/probe_bire_reflection.py) shows why -- its propagation loop walks acc - 1
    steps of a path that get_path returns with 2 acc - 1 points, i.e. it stops part way (56 ... 65 % here), with half-size steps and
    a zero-length step at the reflection; on paths without a bottom reflection it covers the whole path."""
    import os
    import re
    rows = [q for q in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_bire_reflection_probe.txt'))
            if q.startswith('solution')]
    assert len(rows) == 4
    for q in rows:
        refl = int(re.search(r'bottom reflections (\d+)', q).group(1))
        acc = int(re.search(r'acc = (\d+)', q).group(1))
        pts = int(re.search(r'returns (\d+) points', q).group(1))
        frac = float(re.search(r'= ([0-9.]+) % of the path', q).group(1))
        if refl == 0:
            assert pts == acc and frac == 100.0
        else:
            assert pts == (refl + 1) * acc - refl and frac < 70. and '1 zero-length steps' in q
