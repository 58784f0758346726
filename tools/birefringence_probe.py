"""Birefringent propagation on the GPU vs the oracle's numpy loop (usage: birefringence_probe.py [n_rays] [N])."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import nuradiomc_amd
from oracle import birefringence_oracle as bo
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ref_birefringence.npz'))
tck = [(g['tck_southpole_A_%d_t' % j], g['tck_southpole_A_%d_c' % j]) for j in range(3)]
ice = g['sp_ice']
n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(2)
r, ph = np.sqrt(rng.uniform(0, 2500. ** 2, 4 * n)), rng.uniform(0, 2 * np.pi, 4 * n)
x1 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2500, -50, 4 * n)], axis=1)
x2 = np.tile([0., 0., -100.], (4 * n, 1))
ctx = nuradiomc_amd.Context(tuple(ice), 'SP1', device=0)
t = ctx.find_solutions_batch(x1, x2)
sel = np.flatnonzero(t['n_sol'] > 0)[:n]
x1, x2, C0, D = x1[sel], x2[sel], t['C0'][sel, 0], t['D'][sel, 0]
n_f = N // 2 + 1
spec = np.ones((len(sel), 2, n_f), complex)
ctx.birefringence_batch(x1[:8], x2[:8], C0[:8], D[:8], spec[:8], 2.0, tck)
t0 = time.time()
out = ctx.birefringence_batch(x1, x2, C0, D, spec, 2.0, tck)
t_gpu = time.time() - t0
m = 3
t0 = time.time()
for i in range(m):
    st = bo.path_steps(x1[i], x2[i], C0[i], D[i], ice, tck)
    e = bo.propagate(spec[i][0], spec[i][1], 2.0, st)
t_cpu = (time.time() - t0) / m
steps = int(np.sum(D.astype(int) - 1))
print('GPU: %d rays, %d steps, %d bins in %.3f s (host buffers, PCIe included) = %.0f rays/s, %.2e step-bins/s; oracle numpy %.2f s/ray; last dev %.1e'
      % (len(sel), steps, n_f, t_gpu, len(sel) / t_gpu, steps * n_f / t_gpu, t_cpu, np.max(np.abs(out[m - 1] - e)) / np.max(np.abs(e))))
