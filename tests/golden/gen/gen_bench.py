"""The headline workload pinned on the reference: the first K event groups of bench.py's config-2 list (bench.make_events(1e6, 10):
station S5, southpole_2015, SP1, Alvarez2009, 4096 samples @ 2 GHz, 3e17 eV hadronic showers) through the REFERENCE's own
functions (refharness.simulate_event = simulation.calculate_sim_efield :93, apply_det_response_sim :465, apply_det_response
:530, trigger/simpleThreshold.py), 8 processes over an 8-way split.  Unlike time_reference.py (which only times the loop) this
keeps what the reference computed: trigger mask, candidate flag, common trace length, start time, per-channel max |V|, ray
counts and the per-ray table.

    cp -r /root/reference /tmp/refcopy
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_bench.py [K]

Writes tests/golden/chain_bench_N4096.npz (data: inputs + expected outputs).
"""
import multiprocessing as mp
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..', '..'))
N_LIST = 1000000
RAY_KEYS = ('channel', 'iS', 'type', 'C0', 'C1', 'D', 'T', 'view', 'zenith', 'azimuth', 't0', 'max_efield', 'max_amp_ray',
            'signal_time')


def work(args):
    lo, hi = args
    import refharness as rh
    import bench
    det = rh.StationS5(n_samples=4096, fs=2.0)
    cfg = rh.default_config()
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    v, z, a = bench.make_events(N_LIST, 10)
    t0 = time.time()
    n = hi - lo
    ev = dict(candidate=np.zeros(n, bool), triggered=np.zeros(n, bool), L=np.zeros(n, np.int64), t_min=np.full(n, np.nan),
              n_rays=np.zeros(n, np.int32), maxV=np.zeros((n, 5)), argmaxV=np.zeros((n, 5), np.int64))
    rays = []
    for i in range(lo, hi):
        sh = rh.make_shower(i, v[i], z[i], a[i], bench.ENERGY, 'HAD')
        o = rh.simulate_event(i, sh, det, prop, ice, cfg, vrms, vrms_e)
        k = i - lo
        ev['candidate'][k] = o['candidate']
        ev['triggered'][k] = o['triggered']
        ev['L'][k] = o['L']
        ev['t_min'][k] = o['t_min']
        ev['n_rays'][k] = len(o['rays'])
        if 'V' in o:
            ev['maxV'][k] = np.max(np.abs(o['V']), axis=1)
            ev['argmaxV'][k] = np.argmax(np.abs(o['V']), axis=1)
        for r in o['rays']:
            rays.append([i] + [float(r.get(key, np.nan)) for key in RAY_KEYS])
    return lo, hi, ev, np.array(rays).reshape(-1, 1 + len(RAY_KEYS)), time.time() - t0, vrms, vrms_e, \
        np.array([ice.n_ice, ice.delta_n, ice.z_0])


if __name__ == '__main__':
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    procs = 8
    cuts = np.linspace(0, K, 8 * procs + 1).astype(int)   # small chunks: the per-event cost varies a lot with the vertex
    t0 = time.time()
    with mp.Pool(procs) as pool:
        res = pool.map(work, list(zip(cuts[:-1], cuts[1:])), chunksize=1)
    res.sort(key=lambda r: r[0])
    import bench
    v, z, a = bench.make_events(N_LIST, 10)
    out = dict(N=4096, fs=2.0, n_list=N_LIST, seed=10, energy=bench.ENERGY, vrms=res[0][5], vrms_efield=res[0][6], ice=res[0][7],
               att_model='SP1', n_freq=25, askaryan_model='Alvarez2009', trigger_sigma=3.0, min_efield_amplitude=2.0,
               delta_C_cut=0.698, vertex=v[:K], zenith=z[:K], azimuth=a[:K])
    for key in res[0][2]:
        out['ev_' + key] = np.concatenate([r[2][key] for r in res])
    R = np.concatenate([r[3] for r in res])
    out['ray_event'] = R[:, 0].astype(np.int64)
    for j, key in enumerate(RAY_KEYS):
        col = R[:, 1 + j]
        out['ray_' + key] = col.astype(np.int32) if key in ('channel', 'iS', 'type') else col
    np.savez_compressed(os.path.join(HERE, '..', 'chain_bench_N4096.npz'), **out)
    print('%d events, %d rays, %d candidates, %d triggered, %.1f s wall, %.1f s of event loops summed'
          % (K, len(R), out['ev_candidate'].sum(), out['ev_triggered'].sum(), time.time() - t0, sum(r[4] for r in res)))
