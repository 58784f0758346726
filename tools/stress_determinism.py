"""Repeat the bench workload N times on one GPU and report any call whose trigger mask, candidate count or active-ray
count deviates from the first call (how the barrier race of the per-event early exit was found):
    python tools/stress_determinism.py 400
"""
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
ref = None
bad = 0
for k in range(int(sys.argv[1])):
    s = st.simulate_events_dev(n, *d_in, d_trig, askaryan_model='Alvarez2009', want_stats=True)
    t = np.zeros(n, np.uint8); ctx.to_host(t, d_trig)
    idx = np.flatnonzero(t)
    if ref is None:
        ref = idx
    d = np.setxor1d(ref, idx)
    if len(d) or s['n_candidate_events'] != 83263 or s['n_active_rays'] != 1204897:
        bad += 1
        mv, ie = st.fetch('item_maxV'), st.fetch('item_event')
        print('call', k, len(idx), s['n_candidate_events'], s['n_active_rays'], 'differs at', d,
              [mv[np.flatnonzero(ie == e)[0] * 5:np.flatnonzero(ie == e)[0] * 5 + 5] for e in d[:2]])
print('calls', sys.argv[1], 'deviating', bad, 'reference count', len(ref))
