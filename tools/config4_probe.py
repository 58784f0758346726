"""BASELINE config 4 ingredients inside simulate_events: ARZ2020 time-domain emission + birefringence (southpole_A), one 5-channel
station (-100 .. -104 m), southpole_2015 ice + SP1, 4096 samples at 2 GHz, 1e18 eV showers.  The ARZ shower library of the
reference is a download; the library of tests/golden/ref_arz.npz (the AIRES profile the reference ships + Gaisser-Hillas
shaped ones) stands in.  usage: config4_probe.py [n_events] [chunk] [mode: arz+bire | arz | bire]"""
import sys, time, os
import numpy as np
R = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import nuradiomc_amd, bench
from nuradiomc_amd import arz as arz_mod
from test_oracle_golden import _arz_library

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
mode = sys.argv[3] if len(sys.argv) > 3 else 'arz+bire'
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1', device=0)
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=4096, sampling_rate=2.0)
kw = {}
types = np.array(['HAD', 'EM'])[np.arange(n) % 2]
energy = np.full(n, 1e18)
if 'arz' in mode:
    a = arz_mod.ARZ(seed=1, library=_arz_library(np.load(os.path.join(R, 'tests', 'golden', 'ref_arz.npz'))))
    st.set_arz(a)
    kw = dict(askaryan_model='ARZ2020', arz_iN=a.draw_profile_numbers(energy, list(types)))
if 'bire' in mode:
    b = np.load(os.path.join(R, 'tests', 'golden', 'ref_birefringence.npz'))
    st.set_birefringence([(b['tck_southpole_A_%d_t' % j], b['tck_southpole_A_%d_c' % j]) for j in range(3)])
vertex, zenith, azimuth = bench.make_events(n, 10)
kL = np.full(n, 10 ** 1.5)
def run(sl):
    k = dict(kw)
    if 'arz_iN' in k:
        k['arz_iN'] = k['arz_iN'][sl]
    return st.simulate_events(vertex[sl], zenith[sl], azimuth[sl], energy[sl], types[sl], kL[sl], **k)
run(slice(0, min(n, chunk)))   # warm-up of a full call: clocks, allocations, table caches
t0 = time.time()
n_trig = n_rays = n_cand = 0
steps_all = steps_prop = rays_prop = 0
stage = {}
for a0 in range(0, n, chunk):
    trig, stats = run(slice(a0, min(n, a0 + chunk)))
    n_trig += int(trig.sum()); n_rays += stats['n_rays']; n_cand += stats['n_candidate_events']
    for k_, v_ in stats['stage_ms'].items():
        stage[k_] = stage.get(k_, 0.) + v_
    if 'bire' in mode and stats['n_rays']:   # work of the birefringent propagation: path steps of the rays it was run for
        ns = st.fetch('gen_n_steps')[:stats['n_rays']].astype(np.int64)
        pr = st.fetch('ray_propagated')[:stats['n_rays']] != 0
        steps_all += int(ns.sum()); steps_prop += int(ns[pr].sum()); rays_prop += int(pr.sum())
dt = time.time() - t0
print('config 4 ingredients (%s): %d events x 5 channels, %d rays through emission + propagation, %d candidate events, %d triggered; '
      '%.2f s wall = %.0f events/s, %.0f rays/s' % (mode, n, n_rays, n_cand, n_trig, dt, n / dt, n_rays / dt))
print('stage ms (sum over calls):', {k_: round(v_, 1) for k_, v_ in stage.items()})
if steps_all:
    print('birefringence: %d of %d rays propagated, %.3g of %.3g path steps x 2049 bins = %.3g step-bins'
          % (rays_prop, n_rays, steps_prop, steps_all, steps_prop * 2049.))
