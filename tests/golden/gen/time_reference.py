"""CPU baseline, leg 1 of BASELINE.md section 3: the REFERENCE's pure-Python path timed in the build container on the
benchmark workload (bench.py config 2: station S5, southpole_2015, SP1, Alvarez2009, 4096 samples, 3e17 eV hadronic showers,
bench.make_events(n, 10)), as 1 process and as 8 processes over an 8-way split of the event list (the reference's own scale-out
model, NuRadioMC/utilities/runner.py:9-15).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/time_reference.py [n_events]

Prints events/s and the number of triggers; the numbers are recorded in BASELINE.md.
"""
import multiprocessing as mp
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..', '..'))


def work(args):
    lo, hi = args
    import refharness as rh
    import bench
    det = rh.StationS5(n_samples=4096, fs=2.0)
    cfg = rh.default_config()
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    v, z, a = bench.make_events(hi, 10)
    t0 = time.time()
    n_trig = 0
    for i in range(lo, hi):
        sh = rh.make_shower(i, v[i], z[i], a[i], bench.ENERGY, 'HAD')
        n_trig += rh.simulate_event(i, sh, det, prop, ice, cfg, vrms, vrms_e)['triggered']
    return hi - lo, n_trig, time.time() - t0


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    for procs in (8, 1):
        m = n if procs > 1 else max(n // 8, 1)     # the single process gets an eighth of the list (same per-process work)
        cuts = np.linspace(0, m, procs + 1).astype(int)
        t0 = time.time()
        with mp.Pool(procs) as pool:
            res = pool.map(work, list(zip(cuts[:-1], cuts[1:])))
        wall = time.time() - t0
        busy = max(r[2] for r in res)
        print('%d process(es): %d events, %d triggered, %.1f s in the event loops (%.1f s wall incl. start-up) = %.2f events/s'
              % (procs, m, sum(r[1] for r in res), busy, wall, m / busy), flush=True)
