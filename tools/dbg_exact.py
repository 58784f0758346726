import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd, bench
n = 1000000
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1')
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='analytic_VPol', n_samples=4096, sampling_rate=2.0, n_freq=25)
v, z, a = bench.make_events(n, 10)
args = (v, z, a, np.full(n, bench.ENERGY), np.zeros(n, np.int32), np.ones(n))
tp, sp = st.simulate_events(*args)
mvp, iep = st.fetch('item_maxV').copy(), st.fetch('item_event').copy()
candp = st.fetch('ev_candidate').copy()
te, se = st.simulate_events(*args, no_pruning=True)
mve, iee = st.fetch('item_maxV').copy(), st.fetch('item_event').copy()
cande = st.fetch('ev_candidate').copy()
print('production', tp.sum(), sp['n_candidate_events'], 'exact', te.sum(), se['n_candidate_events'])
print('candidate flags equal', np.array_equal(candp, cande), 'differing events', np.flatnonzero(tp != te))
thr = 3 * st.vrms
for e in np.flatnonzero(tp != te)[:5]:
    i0, i1 = np.flatnonzero(iep == e)[0], np.flatnonzero(iee == e)[0]
    print(e, 'production maxV', mvp[i0 * 5:i0 * 5 + 5], 'exact', mve[i1 * 5:i1 * 5 + 5], 'thr', thr)
