"""Repeat a two-lane array step N times and compare the OR mask, the per-station masks and the counters with the one-lane run
(threads + two streams: any lost update would show as a deviating mask).   python tools/stress_lanes.py [config] [events] [repeats]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import nuradiomc_amd
import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 20
wl = bench.make_workload(cfg, n_ev, seed=10)
ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'], device=0)
arr = bench.build_array(ctx, wl)
n_st = len(wl['centres'])
d = bench.upload_events(ctx, wl)
ng = d['n_groups']
d_st = ctx.malloc(n_st * ng)


def run():
    s_ = arr.simulate_events_dev(d['n'], *d['in'], d['trig'], d_max_distance=d['md'], n_groups=ng, d_group_begin=d['gb'],
                                 d_station_triggered=d_st, **wl['sim_kw'])
    a, b = np.zeros(ng, np.uint8), np.zeros(n_st * ng, np.uint8)
    ctx.to_host(a, d['trig'])
    ctx.to_host(b, d_st)
    return s_, a, b


s1, t1, st1 = run()
ctx2 = nuradiomc_amd.Context(wl['ice'], wl['att_model'], device=0)
arr.add_lane(bench.build_array(ctx2, wl).station)
bad = 0
for k in range(rep):
    s2, t2, st2 = run()
    ok = np.array_equal(t1, t2) and np.array_equal(st1, st2) and all(s1[q] == s2[q] for q in ('n_rays', 'n_candidate_events', 'n_triggered'))
    bad += not ok
print('config %d, %d events, %d two-lane repeats: %d deviating; %d triggers' % (cfg, n_ev, rep, bad, int(t1.sum())))
