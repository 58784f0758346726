"""Where the host side of output.simulate_to_output spends its time (cProfile of bench.end_to_end on the headline list).
usage (GPU box): python tools/e2e_profile.py [n_events]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nuradiomc_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
wl = bench.make_workload(2, n, 10)
ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'], device=0)
st = bench.build_array(ctx, wl)
bench.end_to_end(st, wl)   # warm-up (tables, allocations)
pr = cProfile.Profile()
pr.enable()
r = bench.end_to_end(st, wl)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
