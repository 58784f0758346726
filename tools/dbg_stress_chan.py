import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd, bench
n = 1000000
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1', device=0)
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='analytic_VPol', n_samples=4096, sampling_rate=2.0, n_freq=25)
v, z, a = bench.make_events(n, 10)
t, s = st.simulate_events(v, z, a, np.full(n, bench.ENERGY), np.zeros(n, np.int32), np.ones(n))
mv, ie = st.fetch('item_maxV'), st.fetch('item_event')
need = st.fetch('item_need').view(np.int32)[:len(mv)].reshape(-1, 5)
sel = ie[need.any(axis=1)][:int(sys.argv[2])]
print('selected', len(sel), 'events')
vs, zs, az = v[sel], z[sel], a[sel]
m = len(sel)
args = (vs, zs, az, np.full(m, bench.ENERGY), np.zeros(m, np.int32), np.ones(m))
ref = None
for k in range(int(sys.argv[1])):
    t, s = st.simulate_events(*args, dump_traces=True)
    mvk = st.fetch('item_maxV').copy()
    if ref is None:
        ref = mvk
        print('items', len(ref), 'triggered', t.sum())
        continue
    d = np.flatnonzero(mvk != ref)
    if len(d):
        tr, off = st.fetch('trace'), st.fetch('trace_offset')
        for it in d[:3]:
            x = tr[off[it]:off[it + 1]]
            big = np.flatnonzero(np.abs(x) > 10 * ref[it])
            print('call', k, 'item', it, 'event', st.fetch('item_event')[it // 5], 'ch', it % 5, 'maxV', mvk[it], 'ref', ref[it], 'L', len(x),
                  'n_big', len(big), 'big range', (big.min(), big.max()) if len(big) else None, 'nan', np.isnan(x).sum())
            np.save('gpurun_out/glitch_trace_%d_%d.npy' % (k, it), x)
print('done')
