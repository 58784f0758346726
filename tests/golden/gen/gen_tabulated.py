"""Golden vectors for TABULATED antenna patterns: a synthetic vector-effective-length table (ours: smooth analytic
functions sampled on a (frequency, theta, phi) grid) is written in the reference's pickle format and run through the
reference's own AntennaPattern (tri-linear complex interpolation, antennapattern.py:1338-1577, orientation handling
:1190-1307) and through the whole chain (refharness.simulate_event).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_tabulated.py

The real antenna tables (bicone_v8, RNOG_vpol_..., createLPDA_...) are downloaded by the reference on demand and are not
available offline; the table itself is stored in the fixture, so nothing of the reference travels.
"""
import os
import sys
import pickle
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402
import NuRadioReco.detector.antennapattern as ap  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
NAME = 'synthetic_table_v1'
deg = np.pi / 180

# ---- the synthetic table: grid order of the reference's flat index iF * nT * nP + iP * nT + iT
freqs = np.linspace(0.03, 1.23, 41)
thetas = np.linspace(0., 180., 19) * deg
phis = np.linspace(0., 360., 37) * deg
F, P, T = np.meshgrid(freqs, phis, thetas, indexing='ij')
ff, pp, tt = F.ravel(), P.ravel(), T.ravel()
shape = (ff / 0.25) ** 1.5 / (1 + (ff / 0.3) ** 3)
H_theta = 0.25 * shape * np.sin(tt) * (1 + 0.3 * np.cos(pp)) * np.exp(-2j * np.pi * ff * (14. + 3. * np.cos(tt)))
H_phi = 0.08 * shape * np.cos(tt) * np.sin(pp) * np.exp(-2j * np.pi * ff * (11. + 2. * np.sin(pp)) + 0.4j)
orientation = (0., 0., 90 * deg, 0.)   # boresight +z, tine-plane normal +x (like the LPDA tables)
model_dir = os.path.join(os.path.dirname(ap.__file__), 'AntennaModels', NAME)
os.makedirs(model_dir, exist_ok=True)
with open(os.path.join(model_dir, NAME + '.pkl'), 'wb') as f:
    pickle.dump([orientation[0], orientation[1], orientation[2], orientation[3], ff, tt, pp, H_phi, H_theta], f, protocol=4)

# ---- (1) raw responses of the reference for a set of directions / orientations on an L grid
pat = ap.AntennaPatternProvider().load_antenna_pattern(NAME)
fgrid = np.fft.rfftfreq(1500, 0.5)
rng = np.random.default_rng(61)
dirs = np.stack([np.arccos(rng.uniform(-1, 1, 16)), rng.uniform(0, 2 * np.pi, 16)], axis=1)
oris = np.array([[0., 0., 90 * deg, 90 * deg], [0., 0., 90 * deg, 0.], [180 * deg, 0., 90 * deg, 60 * deg],
                 [90 * deg, 30 * deg, 0., 0.], [45 * deg, 200 * deg, 90 * deg, 290 * deg]])
resp = np.zeros((len(oris), len(dirs), 2, len(fgrid)), complex)
for io, o in enumerate(oris):
    for idr, (zen, az) in enumerate(dirs):
        v = pat.get_antenna_response_vectorized(fgrid, zen, az, *o)
        resp[io, idr, 0], resp[io, idr, 1] = v['theta'], v['phi']

# ---- (2) whole chain with five differently oriented copies of the table antenna
import gen_chain  # noqa: E402
ori5 = [[0., 0., 90 * deg, 0.], [0., 0., 90 * deg, 90 * deg], [180 * deg, 0., 90 * deg, 0.], [90 * deg, 0., 90 * deg, 90 * deg],
        [90 * deg, 120 * deg, 0., 0.]]
gen_chain.run('N256_tab', n_events=160, seed=26, N=256, full_rays=60, full_events=6, antenna=NAME,
              cable_delay=[0., 2.2, 0., 0., 5.5], rmax=2500., orientation=ori5, energy=2e17)
g = dict(np.load(os.path.join(OUT, 'chain_N256_tab.npz'), allow_pickle=True))
g.update(tab_freqs=freqs, tab_thetas=thetas, tab_phis=phis, tab_H_theta=H_theta, tab_H_phi=H_phi,
         tab_orientation=np.array(orientation), resp_fgrid=fgrid, resp_dirs=dirs, resp_oris=oris, resp=resp)
np.savez_compressed(os.path.join(OUT, 'chain_N256_tab.npz'), **g)
print('table', len(ff), 'entries; responses', resp.shape)
