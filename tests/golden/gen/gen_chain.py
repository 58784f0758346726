"""Golden vectors for the whole per-event hot path, produced by the reference's own functions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_chain.py

Every event goes through refharness.simulate_event, i.e. simulation.calculate_sim_efield (simulation.py:93),
apply_det_response_sim (:465), apply_det_response (:530) and trigger/simpleThreshold.py, on station S5
(SURVEY.md section 8d): ray tracing -> Askaryan (Alvarez2009) -> attenuation / Fresnel -> per-efield voltages
-> combined channel voltages on the event's common time grid -> Butterworth 80-500 MHz -> 3 Vrms threshold.
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def run(name, n_events, seed, N, full_rays, full_events, energy=3e17, em_every=4, model='Alvarez2009',
        antenna='analytic_VPol', cable_delay=0., rmax=4000., orientation=None, focusing=False, ice_model='southpole_2015',
        att_model='SP1', n_reflections=0, zmin=-2700., z_top=-100., zmax=0., fs=2.0):
    det = rh.StationS5(n_samples=N, fs=fs, antenna=antenna, cable_delay=cable_delay, orientation=orientation, z_top=z_top)
    cfg = rh.default_config(model=model, ice_model=ice_model, attenuation_model=att_model, fs=fs)
    cfg['propagation']['focusing'] = bool(focusing)
    cfg['propagation']['n_reflections'] = int(n_reflections)
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    ev = rh.random_events(n_events, seed, energy=energy, rmax=rmax, zmin=zmin, zmax=zmax)
    stype = np.array(['EM' if (em_every and i % em_every == 0) else 'HAD' for i in range(n_events)])
    rays = []
    evo = dict(candidate=np.zeros(n_events, bool), triggered=np.zeros(n_events, bool), L=np.zeros(n_events, np.int64),
               t_min=np.full(n_events, np.nan), k_L=np.full(n_events, np.nan), n_rays=np.zeros(n_events, np.int32),
               maxV=np.zeros((n_events, 5)), argmaxV=np.zeros((n_events, 5), np.int64), sumV2=np.zeros((n_events, 5)))
    V_list, V_ev = [], []
    spec_list, sim_list, full_idx = [], [], []
    t0 = time.time()
    for i in range(n_events):
        sh = rh.make_shower(i, ev['vertex'][i], ev['zenith'][i], ev['azimuth'][i], ev['energy'][i], stype[i])
        o = rh.simulate_event(i, sh, det, prop, ice, cfg, vrms, vrms_e)
        evo['candidate'][i] = o['candidate']
        evo['triggered'][i] = o['triggered']
        evo['L'][i] = o['L']
        evo['t_min'][i] = o['t_min']
        evo['k_L'][i] = o['k_L']
        evo['n_rays'][i] = len(o['rays'])
        for r in o['rays']:
            r['event'] = i
            if len(full_idx) < full_rays:
                full_idx.append(len(rays))
                spec_list.append(r['spec'][1:])          # eTheta, ePhi after propagation effects
                sim_list.append(r['simch_spec'])         # per-efield voltage spectrum after filters (N grid)
            rays.append(r)
        if 'V' in o:
            V = o['V']
            evo['maxV'][i] = np.max(np.abs(V), axis=1)
            evo['argmaxV'][i] = np.argmax(np.abs(V), axis=1)
            evo['sumV2'][i] = np.sum(V ** 2, axis=1)
            if len(V_ev) < full_events:
                V_ev.append(i)
                V_list.append(V)
    print(name, '%d events, %d rays, %d candidates, %d triggered, %.1f s' % (
        n_events, len(rays), evo['candidate'].sum(), evo['triggered'].sum(), time.time() - t0))
    R = {k: np.array([r[k] for r in rays]) for k in
         ('event', 'channel', 'iS', 'C0', 'C1', 'type', 'zenith', 'azimuth', 'D', 'T', 'view', 'pol_angle', 'launch',
          't0', 'r_theta', 'r_phi', 'max_efield', 'simch_t0', 'max_amp_ray', 'signal_time', 'reflection', 'reflection_case')}
    out = dict(N=N, fs=fs, vrms=vrms, vrms_efield=vrms_e, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]),
               att_model=att_model, n_freq=25, askaryan_model=model, antenna=antenna, cable_delay=cable_delay,
               n_reflections=int(n_reflections), z_reflection=float(getattr(ice, 'reflection', None) or 0.),
               reflection_coefficient=float(getattr(ice, 'reflection_coefficient', None) or 1.),
               reflection_phase_shift=float(getattr(ice, 'reflection_phase_shift', None) or 0.),
               focusing=bool(focusing), focusing_limit=2.,
               det_pos=det.pos, det_orientation=np.array(det.orientation if orientation is None else orientation),
               delta_C_cut=0.698,
               trigger_sigma=3.0, min_efield_amplitude=2.0,
               vertex=ev['vertex'], zenith=ev['zenith'], azimuth=ev['azimuth'], energy=ev['energy'],
               shower_type=stype, full_ray_index=np.array(full_idx, np.int64),
               full_spec=np.array(spec_list), full_simch=np.array(sim_list),
               V_events=np.array(V_ev, np.int64), V_offsets=np.cumsum([0] + [v.shape[1] for v in V_list]),
               V_concat=np.concatenate(V_list, axis=1) if V_list else np.zeros((5, 0)))
    out.update({'ev_' + k: v for k, v in evo.items()})
    out.update({'ray_' + k: v for k, v in R.items()})
    np.savez_compressed(os.path.join(OUT, 'chain_%s.npz' % name), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['N256', 'N4096', 'N256_hpol', 'N256_lpda', 'N256_focus']
    if 'N256' in which:
        run('N256', n_events=300, seed=21, N=256, full_rays=400, full_events=12)
    if 'N4096' in which:
        run('N4096', n_events=120, seed=22, N=4096, full_rays=6, full_events=3, rmax=2500.)
    if 'N1280' in which:   # the trace grid of the reference's own example (examples/01_Veff_simulation: 256 detector samples at 1 GHz
        # simulated at 5 GHz -> 1280 samples, not a power of two); cable delays exercise the sub-sample shift on that grid
        run('N1280', n_events=260, seed=31, N=1280, full_rays=40, full_events=4, rmax=2500., fs=5.0, cable_delay=[0., 3.3, 7.77, 12.2, 19.8])
    if 'N3200' in which:   # an RNO-G read-out (2048 samples at 3.2 GHz) on the 5 GHz simulation grid
        run('N3200', n_events=120, seed=32, N=3200, full_rays=10, full_events=2, rmax=2500., fs=5.0)
    if 'N10240' in which:   # 2048 ns at 5 GHz: N / 2 = 5 * 1024 (an odd-radix pass in the device's transforms), unequal cable delays
        run('N10240', n_events=100, seed=34, N=10240, full_rays=4, full_events=1, rmax=2500., fs=5.0, cable_delay=[0., 3.3, 7.77, 12.2, 19.8])
    if 'N256_hpol' in which:  # HPol antennas + unequal cable delays: exercises ePhi, Fresnel r_s and the sub-sample shift
        run('N256_hpol', n_events=150, seed=23, N=256, full_rays=200, full_events=8, antenna='analytic_HPol',
            cable_delay=[0., 3.3, 7.77, 12.2, 19.8], rmax=2500.)
    if 'N256_lpda' in which:  # LPDAs in five orientations (both VEL components, three phase regimes) + one cable delay
        d = np.pi / 180
        ori = [[0., 0., 90 * d, 0.], [0., 0., 90 * d, 90 * d], [180 * d, 0., 90 * d, 0.], [90 * d, 0., 90 * d, 90 * d],
               [90 * d, 120 * d, 0., 0.]]
        run('N256_lpda', n_events=200, seed=24, N=256, full_rays=200, full_events=8, antenna='analytic_LPDA',
            cable_delay=[0., 0., 4.4, 0., 0.], rmax=2500., orientation=ori, energy=1e17)
    if 'N256_mb' in which:   # Moore's Bay: reflective bottom at -576 m (mooresbay_simple), MB1, one bottom reflection, station at -5 .. -9 m;
        # vertices in the lower 200 m of the shelf: the direct and the bottom-reflected signals of one event then arrive within
        # 2.5 us of each other (the device path takes common traces of up to 12288 samples)
        run('N256_mb', n_events=260, seed=27, N=256, full_rays=120, full_events=8, rmax=900., ice_model='mooresbay_simple',
            att_model='MB1', n_reflections=1, zmin=-570., zmax=-370., z_top=-5., energy=1e18)
    if 'N4096_mb' in which:   # the same shelf with the headline's traces (4096 samples at 2 GHz): common traces of ~ 19 000 samples
        run('N4096_mb', n_events=60, seed=28, N=4096, full_rays=4, full_events=1, rmax=900., ice_model='mooresbay_simple',
            att_model='MB1', n_reflections=1, zmin=-570., zmax=-370., z_top=-5., energy=1e18)
    if 'N256_mb_focus' in which:   # the shelf of N256_mb with propagation.focusing on: the second trace of get_focusing lists the
        # bottom-reflected solutions too (its tracer is built with the same n_reflections)
        run('N256_mb_focus', n_events=140, seed=29, N=256, full_rays=60, full_events=4, rmax=900., ice_model='mooresbay_simple',
            att_model='MB1', n_reflections=1, zmin=-570., zmax=-370., z_top=-5., energy=1e18, focusing=True)
    if 'N256_focus' in which:  # propagation.focusing on (ray convergence factor from a second trace, limit 2)
        run('N256_focus', n_events=200, seed=25, N=256, full_rays=100, full_events=6, rmax=2500., focusing=True)
