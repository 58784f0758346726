"""Golden vectors for the ARZ time-domain Askaryan model, produced by the reference's NuRadioMC/SignalGen/ARZ/ARZ.py
(get_vector_potential :36-275, ARZ.get_time_trace :500-673) and askaryan.get_time_trace(model='ARZ2020').

The reference downloads its shower library (library_v1.2.pkl); there is no network here, so a small library of the same
layout ({type: {energy: {'depth', 'charge_excess': [...]}}}, A01preprocess_shower_library.py) is built from the one
charge-excess profile the reference ships (shower_library/nue_1EeV_CC_1_s0001.t1005/.t1006, electrons minus positrons,
depth offset 1000 g/cm^2 removed) plus Gaisser-Hillas shaped ones, and the download check is switched off.  The library
travels inside the fixture.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_arz.py
"""
import os
import sys
import pickle
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalGen.ARZ import ARZ  # noqa: E402
from NuRadioMC.SignalGen import askaryan  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
lib_dir = os.path.join(os.path.dirname(ARZ.__file__), 'shower_library')
_, depth_e, N_e = np.loadtxt(os.path.join(lib_dir, 'nue_1EeV_CC_1_s0001.t1005'), unpack=True)
_, depth_p, N_p = np.loadtxt(os.path.join(lib_dir, 'nue_1EeV_CC_1_s0001.t1006'), unpack=True)
assert np.all(depth_e == depth_p)
depth = depth_e * units.g / units.cm ** 2 - 1000 * units.g / units.cm ** 2
ce_aires = N_e - N_p


def gaisser_hillas(x_gcm2, n_max, x_max, lam=60.):
    x = np.maximum(x_gcm2, 1e-9)
    return n_max * (x / x_max) ** (x_max / lam) * np.exp((x_max - x) / lam)


xg = depth / (units.g / units.cm ** 2)
library = {'EM': {1e18: {'depth': depth, 'charge_excess': [ce_aires, gaisser_hillas(xg, 5.5e8, 1100., 75.)]},
                  1e16: {'depth': depth, 'charge_excess': [gaisser_hillas(xg, 6e6, 800., 60.)]}},
           'HAD': {1e18: {'depth': depth, 'charge_excess': [gaisser_hillas(xg, 4.5e8, 900., 65.),
                                                           gaisser_hillas(xg, 4.2e8, 1000., 70.),
                                                           gaisser_hillas(xg, 4.8e8, 850., 62.)]},
                   1e17: {'depth': depth, 'charge_excess': [gaisser_hillas(xg, 4.6e7, 820., 63.)]}}}
lib_path = '/tmp/arz_library_fixture.pkl'
with open(lib_path, 'wb') as fout:
    pickle.dump(library, fout)
ARZ.ARZ._ARZ__check_and_get_library = lambda self: True   # no download
default_path = os.path.join(lib_dir, 'library_v1.2.pkl')   # inside the COPY of the reference tree (/tmp/refcopy)
assert default_path.startswith('/tmp/'), default_path
with open(default_path, 'wb') as fout:
    pickle.dump(library, fout)

cher = np.arccos(1 / 1.78)
out = dict(lib_depth=depth, lib_EM_1e18=np.array(library['EM'][1e18]['charge_excess']),
           lib_EM_1e16=np.array(library['EM'][1e16]['charge_excess']),
           lib_HAD_1e18=np.array(library['HAD'][1e18]['charge_excess']),
           lib_HAD_1e17=np.array(library['HAD'][1e17]['charge_excess']))

# ---- 1. vector potentials (module-level function, explicit profile)
a = ARZ.ARZ(seed=1234, library=lib_path, arz_version='ARZ2020', use_numba=False)
par_e = dict(Af=a._Af_e, t0_pos=a._t0_e_pos, freq_pos=a._freq_e_pos, exp_pos=a._exp_e_pos, t0_neg=a._t0_e_neg,
             freq_neg=a._freq_e_neg, exp_neg=a._exp_e_neg)
par_p = dict(Af=a._Af_p, t0_pos=a._t0_p_pos, freq_pos=a._freq_p_pos, exp_pos=a._exp_p_pos, t0_neg=a._t0_p_neg,
             freq_neg=a._freq_p_neg, exp_neg=a._exp_p_neg)
vp_cases = []
for (typ, E, th_deg, N, dt, R, f1, f2, shift, emf) in [
        ('EM', 1e18, 55., 512, 0.1, 1000., 1., 100., False, 1.), ('EM', 1e18, 56., 512, 0.1, 1000., 1., 100., True, 1.),
        ('EM', 1.24e18, 50., 512, 0.1, 200., 1., 100., False, 1.), ('EM', 1e18, 62., 256, 0.5, 1000., 1., 100., False, 1.),
        ('EM', 1e18, 55.8, 256, 0.5, 1000., 1., 1., False, 1.), ('EM', 1e18, 57., 256, 0.2, 500., 3., 100., False, 1.),
        ('HAD', 3e17, 54., 512, 0.1, 1000., 1., 100., False, 0.93), ('HAD', 1e18, 58., 256, 0.5, 3000., 1., 100., True, 0.9),
        ('EM', 1e18, 56.1, 1024, 0.5, 1000., 1., 100., False, 1.)]:
    prof = ce_aires if typ == 'EM' else library['HAD'][1e18]['charge_excess'][0]
    vp = ARZ.get_vector_potential(E, th_deg * units.deg, N, dt, depth, prof, shower_type=typ, n_index=1.78, distance=R,
                                  interp_factor=f1, interp_factor2=f2, shift_for_xmax=shift, em_factor=emf,
                                  **(par_e if typ == 'EM' else par_p))
    vp_cases.append((typ, E, th_deg * units.deg, N, dt, R, f1, f2, shift, emf))
    out['vp_%d' % (len(vp_cases) - 1)] = vp
    print('vp', typ, th_deg, N, dt, np.abs(vp).max(), flush=True)
out['vp_cases'] = np.array([(0 if c[0] == 'EM' else 1,) + c[1:] for c in vp_cases], float)

# ---- 2. ARZ.get_time_trace from the library (random profile choice, rescaling, 20 deg cut, theta')
rng = np.random.default_rng(5)
tr_cases, tr = [], []
a.set_seed(77)
for k in range(14):
    typ = ['HAD', 'EM'][k % 2]
    E = 10 ** rng.uniform(16.5, 18.5) if k != 7 else 2e18
    th = cher + rng.uniform(-12, 12) * units.deg if k != 5 else cher + 23 * units.deg
    R = 10 ** rng.uniform(2.3, 3.5)
    same = bool(k in (3, 4, 9))
    iN = 1 if k == 7 else None
    trace = a.get_time_trace(E, th, 256, 0.5, typ, 1.78, R, same_shower=same, iN=iN)
    tr_cases.append((k % 2 == 0, E, th, R, same, -1 if iN is None else iN, a.get_last_shower_profile_id()[typ]))
    tr.append(trace)
    print('trace', typ, E, th / units.deg, tr_cases[-1][-1], np.abs(trace).max(), flush=True)
out['tr_cases'] = np.array(tr_cases, float)
out['tr'] = np.array(tr)
out['tr_seed'] = 77

# ---- 3. through askaryan.get_time_trace / get_frequency_spectrum (what simulation.py calls)
ask_cases, ask_tr, ask_spec = [], [], []
for k in range(6):
    typ = ['EM', 'HAD'][k % 2]
    E = 10 ** rng.uniform(17., 18.3)
    th = cher + rng.uniform(-6, 6) * units.deg
    kw = dict(seed=4321)
    if k >= 4:
        kw['iN'] = 0
    trace, add = askaryan.get_time_trace(E, th, 256, 0.5, typ, 1.78, 1500., 'ARZ2020', full_output=True, **kw)
    spec = askaryan.get_frequency_spectrum(E, th, 256, 0.5, typ, 1.78, 1500., 'ARZ2020', iN=add['iN'], seed=4321)
    ask_cases.append((k % 2 == 1, E, th, kw.get('iN', -1), add['iN']))
    ask_tr.append(trace)
    ask_spec.append(spec)
    print('askaryan', typ, add['iN'], np.abs(trace).max(), flush=True)
out['ask_cases'] = np.array(ask_cases, float)
out['ask_tr'] = np.array(ask_tr)
out['ask_spec'] = np.array(ask_spec)
out['ask_seed'] = 4321
np.savez_compressed(os.path.join(OUT, 'ref_arz.npz'), **out)
os.remove(default_path)
