import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd
rng = np.random.default_rng(10)
n = 40000
r = np.sqrt(rng.uniform(0, 4000. ** 2, n)); ph = rng.uniform(0, 2 * np.pi, n)
vert = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, 0, n)], axis=1)
chan = np.array([[0., 0., -100. - i] for i in range(5)])
ctx = nuradiomc_amd.Context((1.78, 0.423, 77.))
o = ctx.find_solutions_batch(vert, chan, outer=True)
x1 = np.repeat(np.repeat(vert, 5, axis=0), 2, axis=0); x2 = np.repeat(np.tile(chan, (n, 1)), 2, axis=0)
C0 = o['C0'].reshape(-1); typ = o['type'].reshape(-1)
m = np.isfinite(C0)
x1, x2, C0, typ = x1[m], x2[m], C0[m], typ[m]
ff = np.fft.rfftfreq(4096, 0.5); fc = np.linspace(ff[1], ff[-1], 25)
t = time.time(); att, nev = ctx.attenuation_batch(x1, x2, C0, fc, return_neval=True); dt = time.time() - t
print('rays', len(C0), 'time %.3f s' % dt, 'mean neval', nev.mean(), 'hist', np.unique(nev, return_counts=True))
flat = nev.reshape(-1)
nw = len(flat) // 64
w = flat[:nw * 64].reshape(nw, 64)
print('sum neval', w.sum(), 'sum over waves of 64*max', (64 * w.max(axis=1)).sum(), 'waste factor', (64 * w.max(axis=1)).sum() / w.sum())
print('per-ray: frac of rays with all freqs == 21 or 42:', np.mean((nev.max(axis=1) <= 42)), ' mean of per-ray max', nev.max(axis=1).mean(), 'mean of per-ray mean', nev.mean(axis=1).mean())
for ty in (1, 2, 3):
    mm = typ == ty
    print('type', ty, 'n', mm.sum(), 'mean neval', nev[mm].mean(), 'frac heavy(>200)', (nev[mm] > 200).mean())
# neval vs frequency index
print('mean neval per freq index', nev.mean(axis=0).round(1))
