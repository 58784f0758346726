import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd, bench
n = 400000
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1')
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='analytic_VPol', n_samples=4096, sampling_rate=2.0, n_freq=25)
v, z, a = bench.make_events(n, 10)
out = st.simulate_events(v, z, a, np.full(n, bench.ENERGY), np.zeros(n, np.int32), np.ones(n), askaryan_model='Alvarez2009')
L = st.fetch('ev_L'); c = st.fetch('ev_candidate').astype(bool)
Lc = L[c]
print('candidates', c.sum(), 'L min/median/max', Lc.min(), np.median(Lc), Lc.max(), 'frac > 8192', (Lc > 8192).mean(), 'frac > 5544', (Lc > 5544).mean())
print('percentiles', np.percentile(Lc, [50, 90, 99, 99.9]))
mv = st.fetch('item_maxV'); print('items', len(mv), 'prefiltered (neg)', (mv < 0).sum())
print('items evaluated (>=0)', (mv >= 0).sum(), 'NaN (skipped after trigger)', np.isnan(mv).sum(), 'bounded', (mv < 0).sum(), 'stage ms', out[1]['stage_ms'] if isinstance(out, tuple) else None)
