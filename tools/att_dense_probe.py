"""attenuation_dense_kernel against attenuation_group_kernel on a survey sample: bit equality of the factors and of QUADPACK's
evaluation counts, overflow rate, time.    python tools/att_dense_probe.py [n_events]   (GPU box)"""
import os
import sys
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
v, z, a = bench.make_events(n, 10)
chan = bench.CHANNELS
for ice, model in ((bench.ICE, 'SP1'), (bench.ICE_GREENLAND, 'GL1')):
    ctx = nuradiomc_amd.Context(ice, attenuation_model=model)
    o = ctx.find_solutions_batch(v, chan, outer=True)
    x1 = np.repeat(np.repeat(v, 5, axis=0), 2, axis=0)
    x2 = np.repeat(np.tile(chan, (n, 1)), 2, axis=0)
    C0 = o['C0'].reshape(-1)
    m = np.isfinite(C0)
    x1, x2, C0 = x1[m], x2[m], C0[m]
    ff = np.fft.rfftfreq(4096, 0.5)
    for nf in (25, 21, 30, 12):
        fc = np.linspace(ff[1], ff[-1], nf)
        res = {}
        for mode in ('dense', 'legacy'):
            if mode == 'legacy':
                os.environ['NRHIP_ATT_LEGACY'] = '1'
            else:
                os.environ.pop('NRHIP_ATT_LEGACY', None)
            ctx.attenuation_batch(x1[:1000], x2[:1000], C0[:1000], fc)
            t = time.time()
            att, nev = ctx.attenuation_batch(x1, x2, C0, fc, return_neval=True)
            res[mode] = (att, nev, time.time() - t, ctx.attenuation_last_overflow())
        os.environ.pop('NRHIP_ATT_LEGACY', None)
        (a1, n1, t1, o1), (a2, n2, t2, o2) = res['dense'], res['legacy']
        same = np.array_equal(a1.view(np.int64), a2.view(np.int64))
        print('%s n_freq %d rays %d: factors bit-equal %s, neval equal %s, overflow %d (%.2f %%), wall dense %.3f s legacy %.3f s, mean neval %.1f'
              % (model, nf, len(C0), same, np.array_equal(n1, n2), o1, 100. * o1 / len(C0), t1, t2, n1.mean()), flush=True)
        if not same:
            bad = np.argwhere(a1.view(np.int64) != a2.view(np.int64))
            print('  first differences', bad[:10].tolist(), a1[tuple(bad[0])], a2[tuple(bad[0])], n1[tuple(bad[0])], n2[tuple(bad[0])])
