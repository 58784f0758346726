"""Golden vectors for the focusing factor, produced by the reference's ray_tracing.get_focusing
(NuRadioMC/SignalProp/analyticraytracing.py:2778-2888, numerical branch: second trace to the receiver moved by dz).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_focusing.py
"""
import os
import sys
import logging
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
ice = medium.get_ice_model('southpole_2015')
r = ray.ray_tracing(ice, attenuation_model='SP1', log_level=logging.ERROR, use_cpp=False, compile_numba=False)
rng = np.random.default_rng(41)
n = 160
rr, ph = np.sqrt(rng.uniform(0, 2500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
x1 = np.stack([rr * np.cos(ph), rr * np.sin(ph), rng.uniform(-2500., -5., n)], axis=1)
x2 = np.stack([np.zeros(n), np.zeros(n), rng.choice([-3., -100., -200., -1500.], n)], axis=1)
foc = np.full((n, 2), np.nan)
n_sol = np.zeros(n, np.int32)
for i in range(n):
    r.set_start_and_end_point(x1[i], x2[i])
    r.find_solutions()
    n_sol[i] = r.get_number_of_solutions()
    for iS in range(n_sol[i]):
        foc[i, iS] = r.get_focusing(iS, dz=-0.01, limit=2.)
np.savez_compressed(os.path.join(OUT, 'ref_focusing.npz'), x1=x1, x2=x2, n_sol=n_sol, focusing=foc,
                    ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), dz=-0.01, limit=2.)
print('pairs', n, 'solutions', int(n_sol.sum()), 'at the limit', int(np.sum(foc == 2.)), 'range', np.nanmin(foc), np.nanmax(foc))
