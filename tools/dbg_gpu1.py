import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import nuradiomc_amd
from conftest import golden
from oracle import raytrace_oracle as orc
for name in 'ABC':
    g = golden('raytrace_%s.npz' % name)
    ctx = nuradiomc_amd.Context(g['ice'], str(g['att_model']))
    o = ctx.find_solutions_batch(g['x1'], g['x2'])
    r = orc.raytrace_batch(g['x1'], g['x2'], g['ice'])
    for k in ('C0', 'D', 'T'):
        for nm, ref in (('fixture', g), ('oracle', r)):
            m = np.isfinite(o[k]) & np.isfinite(ref[k])
            rel = np.abs(o[k] - ref[k]) / np.abs(ref[k])
            rel[~m] = 0
            i = np.unravel_index(np.argmax(rel), rel.shape)
            print(name, k, nm, 'max rel %.3e' % rel.max(), 'at', i, 'gpu %.12g ref %.12g' % (o[k][i], ref[k][i]),
                  'C0 gpu %.12g fixture %.12g oracle %.12g' % (o['C0'][i], g['C0'][i], r['C0'][i]), 'type', o['type'][i], 'x1', g['x1'][i[0]], 'x2', g['x2'][i[0]])
    print(name, 'nsol mismatch vs fixture', (o['n_sol'] != g['n_sol']).sum(), 'vs oracle', (o['n_sol'] != r['n_sol']).sum())
