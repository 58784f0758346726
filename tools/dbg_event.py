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
e = int(sys.argv[1])
mv, ie = st.fetch('item_maxV'), st.fetch('item_event')
i = np.flatnonzero(ie == e)
print('candidate', st.fetch('ev_candidate')[e], 'L', st.fetch('ev_L')[e], 'n_rays', st.fetch('ev_n_rays')[e], 'pos in list', i, 'maxV', mv[i[0]*5:i[0]*5+5] if len(i) else None, 'thr', 3*st.vrms)
rb = st.fetch('ev_ray_begin')[e]; nr = st.fetch('ev_n_rays')[e]
print('ray ch', st.fetch('ray_channel')[rb:rb+nr], 'e_norm', st.fetch('ray_e_norm')[rb:rb+nr], 'max_ef', st.fetch('ray_max_efield')[rb:rb+nr])
need = st.fetch('item_need').view(np.int32)[:len(mv)]
print('need', need[i[0]*5:i[0]*5+5] if len(i) else None, 'n heavy items', need.sum(), 'events with heavy', len(np.unique(np.flatnonzero(need)//5)))
