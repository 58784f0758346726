"""bench.py's exact call sequence (device-resident inputs, warm-up without stats), dumping the trigger mask"""
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
out = []
for k in range(4):
    s = st.simulate_events_dev(n, *d_in, d_trig, askaryan_model='Alvarez2009', want_stats=(k >= 1))
    t = np.zeros(n, np.uint8); ctx.to_host(t, d_trig)
    out.append(np.flatnonzero(t))
    print('call', k, len(out[-1]), None if s is None else s['n_triggered'], np.setxor1d(out[0], out[-1]))
np.save(sys.argv[1], out[-1])
