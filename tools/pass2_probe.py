import sys; sys.path.insert(0,'/root/repo')
import numpy as np, bench, nuradiomc_amd
wl = bench.make_workload(2, 1000000, 10)
ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'])
st = bench.build_array(ctx, wl)
d = bench.upload_events(ctx, wl)
for _ in range(3):
    s1 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], n_groups=d['n_groups'])
    s2, _, nk = st.triggered_pass_dev(d['n'], *d['in'], d['trig'], n_groups=d['n_groups'])
print({k: round(v, 2) for k, v in s2['stage_ms'].items()})
print({k: v for k, v in s2.items() if k != 'stage_ms'})
