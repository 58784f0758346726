"""How many host cores does the GPU box really give us?  cgroup quota, affinity, and the oracle's throughput at several worker
counts (5 s each) -- decides `cores` of bench.py's cpu_baseline."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import bench

if __name__ == '__main__':   # the worker processes are spawned: they re-import this file
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us', '/proc/loadavg'):
        try:
            print(f, open(f).read().strip())
        except Exception as e:
            print(f, 'n/a', e)
    print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
    wl = bench.make_workload(2, 100000, 10)
    for cores in [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 256]:
        t = time.time()
        base, n_done, flags = bench.cpu_baseline(wl, 6., 100000, cores=cores)
        print(cores, 'workers:', '%.0f events/s' % base['value'], n_done, 'wall %.1f' % (time.time() - t), flush=True)
