"""Golden ray paths from the reference's ray_tracing.get_path (analyticraytracing.py:2148-2162 -> get_path_reflections :1293 ->
get_path :1239-1291), 40 points per solution, for 30 pairs of the south-pole geometry (both orders of the end points).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_paths.py
"""
import logging
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
ice = medium.get_ice_model('southpole_2015')
r = ray.ray_tracing(ice, log_level=logging.ERROR, use_cpp=False, compile_numba=False)
rng = np.random.default_rng(77)
n, P = 30, 40
x1 = np.stack([rng.uniform(-1500, 1500, n), rng.uniform(-1500, 1500, n), rng.uniform(-2000, -50, n)], axis=1)
x2 = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-200, -2, n)], axis=1)
swap = np.arange(n) % 3 == 0
a, b = np.where(swap[:, None], x2, x1), np.where(swap[:, None], x1, x2)
n_sol = np.zeros(n, np.int32)
C0 = np.full((n, 2), np.nan)
path = np.full((n, 2, P, 3), np.nan)
for i in range(n):
    r.set_start_and_end_point(a[i], b[i])
    r.find_solutions()
    n_sol[i] = r.get_number_of_solutions()
    for iS in range(n_sol[i]):
        C0[i, iS] = r.get_results()[iS]['C0']
        path[i, iS] = r.get_path(iS, n_points=P)
print('pairs', n, 'solutions', n_sol.sum())
np.savez_compressed(os.path.join(OUT, 'ref_paths.npz'), x1=a, x2=b, n_sol=n_sol, C0=C0, path=path,
                    ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]))
