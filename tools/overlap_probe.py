"""Two halves of the config-2 list on two contexts (two HIP streams, two workspaces) from two host threads: do the barrier-bound
channel stage of one half and the VALU-bound ray stages of the other overlap on the GPU?   python tools/overlap_probe.py [n_parts]"""
import os
import sys
import threading
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 1000000
wl = bench.make_workload(2, n, 10)
cuts = np.linspace(0, n, P + 1).astype(int)
parts = []
for k in range(P):
    ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'])
    st = bench.build_array(ctx, wl)
    d = bench.upload_events(ctx, wl, (int(cuts[k]), int(cuts[k + 1])))
    parts.append((ctx, st, d))


def run(k, out):
    ctx, st, d = parts[k]
    out[k] = st.simulate_events_dev(d['n'], *d['in'], d['trig'], want_stats=True, n_groups=d['n_groups'])


def step(offset=0.):
    out = [None] * P
    th = [threading.Thread(target=run, args=(k, out)) for k in range(P)]
    for k, t in enumerate(th):
        t.start()
        if offset and k + 1 < P:
            time.sleep(offset)
    for t in th:
        t.join()
    return out


for off in (0., 0.004, 0.008, 0.012):
    for _ in range(3):
        step(off)
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        out = step(off)
    dt = (time.perf_counter() - t0) / K
    print('parts %d, start offset %.0f ms: %.2f ms per 1e6 events, triggers %d' % (P, off * 1e3, dt * 1e3, sum(o['n_triggered'] for o in out)), flush=True)
# sequential reference on the same objects
for _ in range(2):
    for k in range(P):
        run(k, [None] * P)
t0 = time.perf_counter()
for _ in range(10):
    for k in range(P):
        run(k, [None] * P)
print('sequential: %.2f ms per 1e6 events' % ((time.perf_counter() - t0) / 10 * 1e3))
