"""Golden vectors for the analytic ray tracer, produced by the reference's pure-Python path.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_raytrace.py

Per (vertex x1, receiver x2) pair, through the reference's public API
(NuRadioMC/SignalProp/analyticraytracing.py: set_start_and_end_point :2057, find_solutions :2118,
get_solution_type :2132, get_launch_vector :2560, get_receive_vector :2593, get_reflection_angle :2626,
get_path_length :2650, get_travel_time :2697, get_attenuation :2744):
  n_sol, and per solution type, C0, C1, D, T, launch[3], receive[3], reflection angle (NaN = None),
  attenuation on the 25-point coarse grid of a N=4096 @ 2 GHz trace (for the first `n_att` pairs).
"""
import os
import sys
import time
import logging
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
MAXS = 2


def run_set(name, ice_name, att_model, x1s, x2s, n_att, n_freq=25, N=4096, fs=2.0):
    ice = medium.get_ice_model(ice_name)
    r = ray.ray_tracing(ice, attenuation_model=att_model, n_frequencies_integration=n_freq,
                        log_level=logging.ERROR, use_cpp=False, compile_numba=False)
    n = len(x1s)
    ff = np.fft.rfftfreq(N, 1. / fs)
    fcoarse = np.linspace(ff[1], ff[-1], n_freq)
    o = dict(x1=np.array(x1s, float), x2=np.array(x2s, float), n_sol=np.zeros(n, np.int32),
             type=np.zeros((n, MAXS), np.int32), C0=np.full((n, MAXS), np.nan), C1=np.full((n, MAXS), np.nan),
             D=np.full((n, MAXS), np.nan), T=np.full((n, MAXS), np.nan),
             launch=np.full((n, MAXS, 3), np.nan), receive=np.full((n, MAXS, 3), np.nan),
             refl_angle=np.full((n, MAXS), np.nan), att=np.full((n_att, MAXS, n_freq), np.nan),
             fcoarse=fcoarse, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), att_model=att_model,
             ice_model=ice_name)
    t0 = time.time()
    for i in range(n):
        r.set_start_and_end_point(x1s[i], x2s[i])
        r.find_solutions()
        ns = r.get_number_of_solutions()
        o['n_sol'][i] = ns
        for iS in range(ns):
            res = r.get_results()[iS]
            o['type'][i, iS] = r.get_solution_type(iS)
            assert res['type'] == o['type'][i, iS]
            o['C0'][i, iS] = res['C0']
            o['C1'][i, iS] = res['C1']
            o['D'][i, iS] = r.get_path_length(iS)
            o['T'][i, iS] = r.get_travel_time(iS)
            o['launch'][i, iS] = r.get_launch_vector(iS)
            o['receive'][i, iS] = r.get_receive_vector(iS)
            ra = r.get_reflection_angle(iS)
            ra = np.asarray(ra).ravel()[0]  # np.squeeze([None]) is a 0-d object array
            o['refl_angle'][i, iS] = np.nan if ra is None else float(ra)
            if i < n_att:
                o['att'][i, iS] = r.get_attenuation(iS, fcoarse, fcoarse[-1])
    print(name, n, 'pairs', 'n_sol histogram', np.bincount(o['n_sol']), '%.1f s' % (time.time() - t0))
    np.savez_compressed(os.path.join(OUT, 'raytrace_%s.npz' % name), **o)


if __name__ == '__main__':
    which = sys.argv[1:] or ['A', 'B', 'C']
    if 'A' in which:  # S5 survey geometry (BASELINE config 2): southpole_2015 / SP1
        ev = rh.random_events(300, seed=11)
        det = rh.StationS5()
        x1 = np.repeat(ev['vertex'], 5, axis=0)
        x2 = np.tile(det.pos, (300, 1))
        run_set('A', 'southpole_2015', 'SP1', x1, x2, n_att=400)
    if 'B' in which:  # the reference's own T05 geometry (shallow receiver -> reflected rays): southpole_simple
        g = np.load(os.path.join(OUT, 'ref_C0_SP.npz'))
        run_set('B', 'southpole_simple', 'SP1', g['points'], np.tile(g['x_receiver'], (1000, 1)), n_att=150)
    if 'C' in which:  # Greenland, receivers at assorted depths, vertices also ABOVE the receiver (swap branch)
        rng = np.random.default_rng(5)
        n = 500
        r_ = rng.uniform(20., 3000., n)
        ph = rng.uniform(0, 2 * np.pi, n)
        x1 = np.stack([r_ * np.cos(ph), r_ * np.sin(ph), rng.uniform(-2900., -0.5, n)], axis=1)
        x2 = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n),
                       rng.choice([-2., -15., -60., -97.5, -200., -450.], n)], axis=1)
        run_set('C', 'greenland_simple', 'GL1', x1, x2, n_att=150)
