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
masks = []
for it in range(int(os.environ.get('NRUNS', 3))):
    t, s = st.simulate_events(*args)
    masks.append(t.copy())
    if it < 3 or t.sum() != masks[0].sum():
        print('run', it, t.sum(), s['n_candidate_events'], s['n_active_rays'], np.flatnonzero(t != masks[0]))
    if it == 0:
        mv0, ie0 = st.fetch('item_maxV').copy(), st.fetch('item_event').copy()
        en0 = st.fetch('ray_e_norm').copy(); me0 = st.fetch('ray_max_efield').copy()
mv1, ie1 = st.fetch('item_maxV'), st.fetch('item_event')
en1 = st.fetch('ray_e_norm'); me1 = st.fetch('ray_max_efield')
d = np.flatnonzero(masks[0] != masks[-1])
print('differing events', d)
print('e_norm differs at', np.sum(~((en0 == en1) | (np.isnan(en0) & np.isnan(en1)))), 'max_efield differs at', np.sum(~((me0 == me1) | (np.isnan(me0) & np.isnan(me1)))))
for e in d[:5]:
    i0, i1 = np.flatnonzero(ie0 == e), np.flatnonzero(ie1 == e)
    print(e, 'run0 maxV', mv0[i0[0] * 5:i0[0] * 5 + 5] if len(i0) else None, 'last', mv1[i1[0] * 5:i1[0] * 5 + 5] if len(i1) else None, 'thr', 3 * st.vrms)
