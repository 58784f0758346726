import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import nuradiomc_amd
from oracle import raytrace_oracle as orc
rng = np.random.default_rng(123)
n = 20000
r = np.sqrt(rng.uniform(0, 4000. ** 2, n)); ph = rng.uniform(0, 2 * np.pi, n)
vert = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, 0, n)], axis=1)
chan = np.array([[0., 0., -100. - i] for i in range(5)])
ice = (1.78, 0.423, 77.)
ctx = nuradiomc_amd.Context(ice)
o = ctx.find_solutions_batch(vert, chan, outer=True)
ref = orc.raytrace_batch(np.repeat(vert, 5, axis=0), np.tile(chan, (n, 1)), ice)
ok = o['n_sol'] == ref['n_sol']
print('count mismatch', (~ok).sum())
for k in ('C0', 'D', 'T'):
    rel = np.abs(o[k] - ref[k]) / np.abs(ref[k]); rel[~np.isfinite(rel)] = 0; rel[~ok] = 0
    srt = np.sort(rel.ravel())[::-1]
    print(k, 'top', srt[:6], 'n>1e-6', (rel > 1e-6).sum(), 'n>1e-7', (rel > 1e-7).sum(), 'median', np.median(rel[rel > 0]))
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print('   worst', i, o[k][i], ref[k][i], 'C0', o['C0'][i], ref['C0'][i], 'type', o['type'][i], np.repeat(vert, 5, axis=0)[i[0]], np.tile(chan, (n, 1))[i[0]])
