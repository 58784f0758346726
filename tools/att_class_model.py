"""How many rule rounds the dense quadrature kernel wastes by advancing the rays of a wave in lockstep, for different orders of the
ray list ("lists in the order of the predicted work").  CPU study with the checker (test infrastructure): rays of
the first n events of the bench list, QUADPACK's evaluation counts per (ray, frequency) from oracle/nrmc_oracle.c, rounds of a ray =
ceil(max_f neval / 42); a wave pair holds five rays (wave A: rays 0, 1 and half of 2; wave B: the other half of 2, 3, 4) and spends
max(rounds of its rays) per pair.     python tools/att_class_model.py [n_events]
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from oracle import raytrace_oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
v, z, a = bench.make_events(n, 10)
chan = np.asarray(bench.CHANNELS, float)
x1, x2 = np.repeat(v, len(chan), axis=0), np.tile(chan, (n, 1))
o = orc.raytrace_batch(x1, x2, bench.ICE)
m = np.isfinite(o['C0'])
X1, X2 = np.repeat(x1, 2, axis=0).reshape(-1, 2, 3)[m], np.repeat(x2, 2, axis=0).reshape(-1, 2, 3)[m]
C, T = o['C0'][m], o['type'][m]
ff = np.fft.rfftfreq(4096, 0.5)
att, nev = orc.attenuation_batch(X1, X2, C, bench.ICE, 'SP1', np.linspace(ff[1], ff[-1], 25), return_neval=True)
R = np.ceil(nev.max(axis=1) / 42.).astype(int)
n_ice, dn, z0 = bench.ICE
zt_true = np.log((n_ice - 1. / C) / dn) * z0
zt = np.minimum(zt_true, 0.)
zlo, zhi = np.minimum(X1[:, 2], X2[:, 2]), np.maximum(X1[:, 2], X2[:, 2])
h = np.where(T == 1, zt_true - zhi, zt_true)
span, up = np.maximum(zt - zlo, 1.), np.maximum(zt - zhi, 1e-3)
rounds2 = 35. + 1.3 * np.log2(up) - 3.2 * np.log2(span)
cls = np.where(T == 2, 10 + np.clip(np.floor((rounds2 - 6.75) * 2.), 0, 8).astype(int),
               np.digitize(-h, [-30., -22., -14., -10., -6., -4., -3., -2., -1.]))   # spectral.hip: quad_class


def lockstep(Rr):
    P = Rr[:len(Rr) // 5 * 5].reshape(-1, 5)
    spent = np.maximum.reduce([P[:, 0], P[:, 1], P[:, 2]]).sum() + np.maximum.reduce([P[:, 2], P[:, 3], P[:, 4]]).sum()
    return spent / ((P[:, 0] + P[:, 1] + 2 * P[:, 2] + P[:, 3] + P[:, 4]).sum() / 3.)


print('%d rays, mean rounds %.2f; rounds by type:' % (len(R), R.mean()), {int(t): round(float(R[T == t].mean()), 2) for t in (1, 2, 3)})
for name, key in (('list order', None), ('by solution type (rounds 1-2)', np.where(T == 1, 0, np.where(T == 3, 1, 2))),
                  ('nineteen predicted-work classes', cls), ('perfect sort', R)):
    Rr = R if key is None else R[np.argsort(key, kind='stable')]
    print('%-34s rounds spent / rounds needed = %.3f' % (name, lockstep(Rr)))
