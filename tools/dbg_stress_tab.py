import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd, bench
n = 1000000
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1', device=0)
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='analytic_VPol', n_samples=4096, sampling_rate=2.0, n_freq=25)
vertex, zenith, azimuth = bench.make_events(n, 10)
d_in = [ctx.to_device(a) for a in (vertex, zenith, azimuth, np.full(n, bench.ENERGY), np.zeros(n, np.int32), np.ones(n))]
d_trig = ctx.malloc(n)
ref = {}
bad = 0
G_STRIDE, NT = 8200, 5
for k in range(int(sys.argv[1])):
    s = st.simulate_events_dev(n, *d_in, d_trig, askaryan_model='Alvarez2009', want_stats=True)
    lens = st.fetch('lengths')
    G = st.fetch('tab_G').view(np.uint64).reshape(-1, NT, G_STRIDE, 2)[:len(lens), 0, :8193]   # VPol table only
    hn = st.fetch('tab_hnorm').view(np.uint64).reshape(-1, NT)[:len(lens), 0]
    for nm, a in (('G', G), ('hnorm', hn)):
        if nm not in ref:
            ref[nm] = a.copy()
        elif not np.array_equal(a, ref[nm]):
            d = np.argwhere(a != ref[nm])
            bad += 1
            print('call', k, nm, 'differs at', len(d), 'entries; lengths', np.unique(lens[d[:, 0]]), 'first', d[:4].tolist(), 'n_trig', s['n_triggered'])
    if s['n_triggered'] != 9053:
        print('call', k, 'n_trig', s['n_triggered'])
print('calls', sys.argv[1], 'deviations', bad)
