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
ref_counts = None
bad = 0
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for k in range(n_calls):
    s = st.simulate_events_dev(n, *d_in, d_trig, askaryan_model='Alvarez2009', want_stats=True)
    t = np.zeros(n, np.uint8); ctx.to_host(t, d_trig)
    idx = np.flatnonzero(t)
    if ref is None:
        ref = idx
    d = np.setxor1d(ref, idx)
    counts = (s['n_candidate_events'], s['n_active_rays'], s['n_rays'], s['n_integrand_evals'])
    if k == 4:   # (calls 1-3 try the one- and the two-stage attenuation and the two block sizes of the convolution: other ray counts)
        ref_counts = counts
    if len(d) or s['n_candidate_events'] != 83263 or (ref_counts is not None and counts != ref_counts):
        bad += 1
        mv, ie = st.fetch('item_maxV'), st.fetch('item_event')
        print('call', k, len(idx), s['n_candidate_events'], s['n_active_rays'], 'differs at', d,
              [mv[np.flatnonzero(ie == e)[0] * 5:np.flatnonzero(ie == e)[0] * 5 + 5] for e in d[:2]])
print('calls', n_calls, 'deviating', bad, 'reference count', len(ref), 'counts', ref_counts)
