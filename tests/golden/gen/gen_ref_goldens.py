"""Convert the reference's own golden vectors (pickle / npy) to .npz, with the inputs that produce them.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_ref_goldens.py

* NuRadioMC/test/SignalProp/reference_C0.pkl      (T05unit_test_C0_SP.py:14-48, rtol 1e-7)
* NuRadioMC/test/SignalGen/reference_v2.npy       (U01unit_test.py:15-51; rows 0-99 Alvarez2009, 200-299 Alvarez2000)
"""
import os
import pickle
import numpy as np

REF = os.environ.get('NRMC_REFCOPY', '/tmp/refcopy')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')

# --- C0 golden (inputs regenerated exactly as T05 does) ---------------------------------
np.random.seed(10)
n_events = 1000
rr = np.random.triangular(50., 3000., 3000., n_events)
phiphi = np.random.uniform(0, 2 * np.pi, n_events)
xx = rr * np.cos(phiphi)
yy = rr * np.sin(phiphi)
zz = np.random.uniform(0., -3000., n_events)
points = np.array([xx, yy, zz]).T
with open(os.path.join(REF, 'NuRadioMC/test/SignalProp/reference_C0.pkl'), 'rb') as fin:
    C0_ref = pickle.load(fin, encoding='latin1')
np.savez_compressed(os.path.join(OUT, 'ref_C0_SP.npz'), points=points, x_receiver=np.array([0., 0., -5.]),
                    C0_ref=np.asarray(C0_ref, dtype=float), ice=np.array([1.78, 0.426, 71.]))
print('ref_C0_SP', points.shape, np.shape(C0_ref))

# --- Askaryan golden ------------------------------------------------------------------------
ref = np.load(os.path.join(REF, 'NuRadioMC/test/SignalGen/reference_v2.npy'))
n_index = 1.78
deg = np.pi / 180.
Es = 10 ** np.linspace(15, 19, 5)
thetas = np.arccos(1. / n_index) + np.linspace(-5, 5, 10) * deg
rows = []
i = -1
for model in ['Alvarez2009', 'ARZ2019', 'Alvarez2000', 'ARZ2020']:
    for E in Es:
        for st in ['EM', 'HAD']:
            for th in thetas:
                i += 1
                rows.append((model, E, st, th, i))
sel = [r for r in rows if r[0] in ('Alvarez2009', 'Alvarez2000')]
np.savez_compressed(os.path.join(OUT, 'ref_askaryan_v2.npz'),
                    model=np.array([r[0] for r in sel]), energy=np.array([r[1] for r in sel]),
                    shower_type=np.array([r[2] for r in sel]), theta=np.array([r[3] for r in sel]),
                    trace=ref[[r[4] for r in sel]], n_index=n_index, dt=0.5, N=256, R=1000., seed=1234)
print('ref_askaryan_v2', len(sel))
