import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import bench, nuradiomc_amd
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1')
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=4096, sampling_rate=2.0)
print('inv_lmax', st.att_bound_inv_length[:5], st.att_bound_inv_length[-3:], 'Lmax', 1/st.att_bound_inv_length[[0,5,12,24]])
n = 100000
v, z, a = bench.make_events(n, 10)
trig, stats = st.simulate_events(v, z, a, bench.ENERGY, 'HAD')
b = st.fetch('ray_bound'); D = st.fetch('ray_D'); act = st.fetch('ray_active')[:len(b)]
print(stats['n_rays'], stats['n_active_rays'], 'bound quantiles', np.quantile(b, [0.1, 0.5, 0.9, 0.99]), 'thr', 2*st.vrms_efield, 'frac rays bound>thr', (b > 2*st.vrms_efield).mean())
me = st.fetch('ray_max_efield')
ex = me >= 0
print('exact frac', ex.mean(), 'ratio exact/bound median', np.median(me[ex]/b[ex]))
