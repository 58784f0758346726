"""Golden vectors for the GL3 attenuation model (Greenland 2021; NuRadioMC/utilities/attenuation.py:206-221 with the
depth table NuRadioMC/utilities/data/GL3_params.csv) and the speed-optimised path integration the reference uses for it
(analyticraytracing.py:998-1064: 10 m segment sums, QUADPACK on ds only around the turning point).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_gl3.py

The depth table itself (300 rows: depth, slope, offset) is stored as INPUT DATA of the fixture: the model is defined by it.
"""
import os
import sys
import logging
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium, attenuation  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
ice = medium.get_ice_model('greenland_simple')
r = ray.ray_tracing(ice, attenuation_model='GL3', n_frequencies_integration=25, log_level=logging.ERROR, use_cpp=False,
                    compile_numba=False)
rng = np.random.default_rng(51)
n = 140
rr, ph = np.sqrt(rng.uniform(0, 2500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
x1 = np.stack([rr * np.cos(ph), rr * np.sin(ph), rng.uniform(-2800., -5., n)], axis=1)
x2 = np.stack([np.zeros(n), np.zeros(n), rng.choice([-3., -60., -100., -400.], n)], axis=1)
# a few receivers right below the turning depth of their ray (fallback window reaching past the end point)
ff = np.fft.rfftfreq(4096, 0.5)
fcoarse = np.linspace(ff[1], ff[-1], 25)
C0 = np.full((n, 2), np.nan)
att = np.full((n, 2, 25), np.nan)
n_sol = np.zeros(n, np.int32)
for i in range(n):
    r.set_start_and_end_point(x1[i], x2[i])
    r.find_solutions()
    n_sol[i] = r.get_number_of_solutions()
    for iS in range(n_sol[i]):
        C0[i, iS] = r.get_results()[iS]['C0']
        att[i, iS] = r.get_attenuation(iS, fcoarse, fcoarse[-1])  # the requested grid IS the integration grid
zz = -np.concatenate([np.linspace(0., 3100., 400), [4.5, 4.50149850149, 2994.9, 3000., 2.0]])
Lz = np.array([[attenuation.get_attenuation_length(float(z), float(f), 'GL3') for f in (0.05, 0.2, 0.6, 1.0)] for z in zz])
np.savez_compressed(os.path.join(OUT, 'ref_gl3.npz'), x1=x1, x2=x2, n_sol=n_sol, C0=C0, att=att, fcoarse=fcoarse,
                    ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), gl3_table=attenuation.gl3_parameters,
                    z_probe=zz, f_probe=np.array([0.05, 0.2, 0.6, 1.0]), L_probe=Lz)
print('pairs', n, 'solutions', int(n_sol.sum()), 'att range', np.nanmin(att), np.nanmax(att))
