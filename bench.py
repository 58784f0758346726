#!/usr/bin/env python3
"""bench.py -- the per-event hot path on the BASELINE workloads, one process per GPU.

    python bench.py                                      # BASELINE configs[1]: 1e6 events, 5-channel station, one MI355X
    python bench.py --config 3|4|5                       # the array workloads (35 x 24 RNO-G-like, + ARZ2020 / birefringence, 200 x 5)
    python bench.py --flavour mixed                      # configs[1] with nu_e CC / NC event groups instead of fixed hadronic showers
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--scaling strong]

One step = one pass of the whole hot path (ray tracing -> delta_C cut -> attenuation -> Askaryan emission -> candidate cut ->
antenna + filter response on the event's common time grid -> trigger; for arrays: over every station) over one batch of
synthetic events that is already resident in HBM.  Events shard across ranks with no exchange inside the compute; the one
collective is the all-gather of the per-rank triggered masks over xGMI after the timed loop (RCCL bound through the C ABI,
nuradiomc_amd.comm -- no PyTorch anywhere; torch.distributed.run is only the process launcher).  `--scaling weak` (default):
every rank owns `--events` events; `--scaling strong`: ONE list of `--events` events is cut with shard_range.

Prints ONE JSON line (rank 0).  `roofline` prices the stage that takes longest in the measured steps (HIP-event times on the
stream the kernels run on, averaged over the timed steps); `cpu_baseline` times the oracle on ALL host cores on a bounded
sample of the same event list and the GPU's trigger mask of that sample must equal the oracle's (parity_check).
"""
import argparse
import hashlib
import json
import re
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ICE = (1.78, 0.423, 77.)            # southpole_2015 (NuRadioMC/utilities/medium.py:69)
ICE_GREENLAND = (1.78, 0.51, 37.25)  # greenland_simple (medium.py:145)
N_SAMPLES, FS = 4096, 2.0
CHANNELS = np.array([[0., 0., -100. - i] for i in range(5)])
ENERGY = 3e17                       # shower energy [eV] of a 1 EeV neutrino at <y> ~ 0.3
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s
FP64_PEAK_TFLOPS = 78.6             # same guide: dense FP64 (vector and matrix alike)
FP32_PEAK_TFLOPS = 157.3            # same guide: FP32 vector
FLOP_PER_OBJECTIVE = 189.           # one evaluation of the ray finder's objective (analyticraytracing.py:204-272): ~90 add / mul, 13 divisions
                                    # and 6 square roots at 1 flop, 1 exp + 3 log at 20 (round 5: the finder without the hybr stage needs no
                                    # logarithm for the depth of the turning point; 209 with it)
# general path (config 4), counted per unit of work the kernels report:
FLOP_PER_ARZ_EVAL = 50.             # one point of the vector-potential integrand (ARZ.py:216-266): retarded time (rsqrt at 1), degree-6
                                    # form-factor polynomial, direction factors, trapezoid weight and the two accumulations
FLOP_PER_BIRE_STEP_BIN = 36.        # one 1 m path step on one frequency bin (analyticraytracing.py:2402-2445): two real 2 x 2 by complex
                                    # products (24), the phase factor on one component (6), the recurrence of the phase (6)
FLOP_PER_BIRE_STEP = 250.           # one step record: two path points, three splines, effective indices, two eigen-polarisations
# algorithmic HBM bytes of the un-fused formulation, N = 4096, L = 5296
B_RAY, B_CHANNEL, B_PAIR = 601216, 169504, 320
DCUT = [-1.56434411e+02, 2.54131322e+01, -1.34932379e+00, 2.39984185e-02]   # config_default.yaml speedup.distance_cut_coefficients
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def make_events(n, seed, rmax=4000.):
    """uniform in r^2 <= rmax^2 and z in [-2.7 km, 0], isotropic directions"""
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.uniform(0, rmax ** 2, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    vertex = np.stack([r * np.cos(phi), r * np.sin(phi), rng.uniform(-2700., 0., n)], axis=1)
    zenith = np.arccos(rng.uniform(-1, 1, n))
    azimuth = rng.uniform(0, 2 * np.pi, n)
    return vertex, zenith, azimuth


def _mixed_showers(n_groups, seed, vertex, zenith, azimuth, e_nu):
    """nu_e CC (45 %): hadronic shower y E + electromagnetic shower (1 - y) E at one vertex; everything else: one hadronic shower
    y E.  Inelasticity from a steeply falling distribution with <y> ~ 0.25 (y = u^3 scaled to [0.005, 1]); k_L of the EM
    showers 10^N(mean(E), sigma(E)) (parametrizations.py:141-172) from the same synthetic generator."""
    from nuradiomc_amd import sequencing
    rng = np.random.default_rng(seed + 7919)
    y = 0.005 + 0.995 * rng.random(n_groups) ** 3
    cc = rng.random(n_groups) < 0.45
    rep = np.where(cc, 2, 1)
    grp = np.repeat(np.arange(n_groups), rep)
    first = np.concatenate([[True], grp[1:] != grp[:-1]])
    en = np.where(first, (y * e_nu)[grp], ((1 - y) * e_nu)[grp])
    typ = np.where(first, 0, 1).astype(np.int32)
    kL = np.ones(len(grp))
    em = np.flatnonzero(typ == 1)
    ms = np.array([sequencing.alvarez2009_k_L_distribution(e) for e in en[em]]).reshape(-1, 2)
    kL[em] = 10 ** rng.normal(ms[:, 0], ms[:, 1])
    return dict(vertex=vertex[grp], zenith=zenith[grp], azimuth=azimuth[grp], energy=en, shower_type=typ, k_L=kL, group=grp)


def make_workload(config=2, n=1000000, seed=10, flavour='had', trigger='threshold'):
    """The synthetic inputs and detector of one BASELINE configuration (n event groups).  trigger (configs 3 / 4, the RNO-G like
    array): 'threshold' = 3 Vrms on any channel (rounds 1-2), 'pa' = the phased array on the four deep dipoles of the power
    string (11 beams, 16-sample windows, phasedArrayBase.py), 'pa_adc_noise' = the same behind the 8-bit 472 MHz trigger ADC with
    4 x FFT up-sampling and with thermal noise on every channel -- what an RNO-G station triggers on."""
    d = np.pi / 180
    wl = dict(config=config, n=n, seed=seed, N=N_SAMPLES, fs=FS, centres=None, sim_kw={}, flavour=flavour)
    if config == 2:
        v, z, a = make_events(n, seed)
        wl.update(ice=ICE, att_model='SP1', rel_pos=CHANNELS, antenna=['analytic_VPol'] * 5,
                  orientation=np.tile([0., 0., 90 * d, 90 * d], (5, 1)), cable_delay=np.zeros(5), distance_cut=False,
                  name="BASELINE configs[1]: %d events/GPU, 5-ch analytic_VPol station at -100..-104 m, southpole_2015 ice, SP1, "
                       "Alvarez2009, 4096 samples @ 2 GHz, Butterworth 80-500 MHz, 3 Vrms threshold" % n)
        e_nu = 1e18
    elif config in (3, 4):
        lay = np.load(os.path.join(GOLDEN, 'rnog_array_layout.npz'))   # data read from NuRadioReco/detector/RNO_G/RNO_array.json
        centres = lay['centres']
        c0 = centres[:, :2].mean(axis=0)
        rmax = np.max(np.linalg.norm(centres[:, :2] - c0, axis=1)) + 3000.
        v, z, a = make_events(n, seed, rmax)
        v[:, :2] += c0
        v[:, 2] = np.minimum(v[:, 2], -1.)
        wl.update(ice=ICE_GREENLAND, att_model='GL1', rel_pos=lay['rel_pos'], antenna=[str(x) for x in lay['antenna']],
                  orientation=lay['orientation'], cable_delay=lay['cable_delay'], centres=centres, distance_cut=True,
                  N=2048 if config == 3 else 4096)
        wl['name'] = ("BASELINE configs[2]: %d events/GPU x 35 stations (RNO_array.json) x 24 channels (analytic VPol / HPol / LPDA "
                      "stand-ins), greenland_simple ice, GL1, Alvarez2009, speedup.distance_cut, 2048 samples @ 2 GHz, 3 Vrms "
                      "threshold on any channel" % n) if config == 3 else (
                      "BASELINE configs[3]: %d events/GPU x 35 stations x 24 channels, greenland_simple ice, GL1, ARZ2020 time-domain "
                      "emission + birefringence greenland_A, speedup.distance_cut, 4096 samples @ 2 GHz, 3 Vrms threshold" % n)
        if config == 4:
            wl['sim_kw'] = dict(askaryan_model='ARZ2020')
        wl['trigger'] = trigger
        if trigger != 'threshold':
            wl['name'] = wl['name'].replace('3 Vrms threshold on any channel', '').replace('3 Vrms threshold', '').rstrip(', ') + \
                ', trigger: phased array on the 4 deep dipoles' + (' with 8-bit 472 MHz trigger ADC, 4x up-sampling and thermal noise' if trigger == 'pa_adc_noise' else '')
        e_nu = 1e18
    elif config == 5:
        n_st, spacing = 200, 1240.
        side = int(np.ceil(np.sqrt(n_st)))
        centres = np.array([[spacing * (i - (side - 1) / 2), spacing * (j - (side - 1) / 2), 0.]
                            for i in range(side) for j in range(side)])[:n_st]
        rmax = np.max(np.abs(centres[:, :2])) + 3000.
        v, z, a = make_events(n, seed, rmax)
        wl.update(ice=ICE, att_model='SP1', rel_pos=CHANNELS, antenna=['analytic_VPol'] * 5,
                  orientation=np.tile([0., 0., 90 * d, 90 * d], (5, 1)), cable_delay=np.zeros(5), centres=centres,
                  distance_cut=True, N=2048,
                  name="BASELINE configs[4]: %d events/GPU x 200 stations (1.24 km grid) x 5-ch dipole string, southpole_2015 ice, SP1, "
                       "Alvarez2009, showers log-uniform in 1e16..1e20 eV, nu_e CC groups (HAD + EM), speedup.distance_cut, 2048 "
                       "samples @ 2 GHz, 2-of-5 high/low (3 Vrms) coincidence within 30 ns" % n)
        wl['sim_kw'] = dict(trigger='high_low', n_coincidences=2, coinc_window=30.)
        flavour = wl['flavour'] = 'mixed'
        e_nu = 10 ** np.random.default_rng(seed + 1).uniform(16., 20., n)
    else:
        raise ValueError("config must be 2, 3, 4 or 5")
    if flavour == 'mixed':
        ev = _mixed_showers(n, seed, v, z, a, np.broadcast_to(e_nu, (n,)))
    else:
        ev = dict(vertex=v, zenith=z, azimuth=a, energy=np.full(n, ENERGY * (e_nu / 1e18)), shower_type=np.zeros(n, np.int32),
                  k_L=np.ones(n), group=np.arange(n))
    wl['events'] = ev
    return wl


def arz_library():
    """the small shower library of the reference's layout (charge-excess profiles of EM / HAD showers) held as test data"""
    g = np.load(os.path.join(GOLDEN, 'ref_arz.npz'))
    dep = g['lib_depth']
    return {'EM': {1e18: {'depth': dep, 'charge_excess': list(g['lib_EM_1e18'])}, 1e16: {'depth': dep, 'charge_excess': list(g['lib_EM_1e16'])}},
            'HAD': {1e18: {'depth': dep, 'charge_excess': list(g['lib_HAD_1e18'])}, 1e17: {'depth': dep, 'charge_excess': list(g['lib_HAD_1e17'])}}}


def birefringence_splines():
    """the three depth splines of the reference's model file birefringence_greenland_A.npy (held as test data)"""
    b = np.load(os.path.join(GOLDEN, 'ref_birefringence.npz'))
    return [(b['tck_greenland_A_%d_t' % j], b['tck_greenland_A_%d_c' % j]) for j in range(3)]


def build_array(ctx, wl):
    """Station (config 2) or StationArray (configs 3-5) of the workload on this context"""
    import nuradiomc_amd
    c0 = np.zeros(3) if wl['centres'] is None else wl['centres'][0]
    st = nuradiomc_amd.Station(ctx, wl['rel_pos'] + c0, antenna=wl['antenna'], orientation=wl['orientation'],
                               cable_delay=wl['cable_delay'], n_samples=wl['N'], sampling_rate=wl['fs'], n_freq=25)
    trig = wl.get('trigger', 'threshold')
    if trig in ('pa', 'pa_adc_noise'):
        ang = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 11))
        if trig == 'pa':
            rolls = st.set_phased_array([0, 1, 2, 3], ang, window=16, step=8)
            wl['sim_kw'] = dict(wl['sim_kw'], trigger='phased_array', trigger_threshold=2.0 * (2 * st.vrms) ** 2)
            wl['pa'] = dict(channels=[0, 1, 2, 3], rolls=np.array(rolls), window=16, step=8, threshold=2.0 * (2 * st.vrms) ** 2, adc=None)
        else:
            rolls = st.set_phased_array([0, 1, 2, 3], ang, window=24, step=8, upsampling_factor=4,
                                        adc=dict(sampling_frequency=0.472, n_bits=8, noise_count=5, output='counts'))
            amp = st.set_noise(300.)
            # mean noise power of the coherent sum: 4 channels x (5 counts)^2; 6 x that keeps noise-only windows below ~1e-2 per event
            wl['sim_kw'] = dict(wl['sim_kw'], trigger='phased_array', trigger_threshold=6.0 * (2 * 5) ** 2, noise=True, noise_seed=1235)
            wl['pa'] = dict(channels=[0, 1, 2, 3], rolls=np.array(rolls), window=24, step=8, threshold=6.0 * (2 * 5) ** 2,
                            adc=dict(fs=0.472, n_bits=8, noise_count=5, up=4, vrms=float(st.vrms)), noise_amp=np.array(amp), noise_seed=1235)
    if wl['config'] == 4:
        from nuradiomc_amd import arz as arz_mod
        st.set_birefringence(birefringence_splines(), angle_to_iceflow=None)
        st.set_arz(arz_mod.ARZ(seed=1235, library=arz_library()))
    if wl['centres'] is None:
        return st
    return nuradiomc_amd.StationArray(st, wl['centres'], relative_position=wl['rel_pos'])


def upload_events(ctx, wl, sl=None):
    """the shower list (or its slice of event groups) -> HBM; returns dict(in=[6 device pointers], md, gb, trig, n, n_groups)"""
    from nuradiomc_amd.station import distance_cut
    ev = wl['events']
    grp = ev['group']
    if sl is None:
        sel = slice(0, len(grp))
        g0, g1 = 0, int(grp[-1]) + 1 if len(grp) else 0
    elif isinstance(sl, np.ndarray):   # an index set of event groups (interleaved chunks of a strong-scaling run); one shower per group
        if len(grp) and int(grp[-1]) + 1 != len(grp):
            raise SystemExit("bench.py: interleaved shards need one shower per event group (--flavour had)")
        sel = sl
        g0, g1 = 0, len(sl)
        sel = type('IndexSel', (), {'start': 0, 'stop': len(sl), 'idx': sl})()
    else:
        g0, g1 = sl
        lo, hi = np.searchsorted(grp, g0), np.searchsorted(grp, g1)
        sel = slice(lo, hi)
    n = sel.stop - sel.start
    n_groups = g1 - g0
    pick = sel.idx if hasattr(sel, 'idx') else sel
    arrs = [np.ascontiguousarray(ev[k][pick]) for k in ('vertex', 'zenith', 'azimuth', 'energy', 'shower_type', 'k_L')]
    gb = None
    if n != n_groups:
        gsel = grp[pick] - g0
        gb = np.ascontiguousarray(np.concatenate([np.flatnonzero(np.concatenate([[True], gsel[1:] != gsel[:-1]])), [n]]), np.int32)
    d = dict(n=n, n_groups=n_groups, host=arrs, gb_host=gb)
    d['in'] = [ctx.to_device(x) for x in arrs]
    d['md'] = ctx.to_device(distance_cut(arrs[0], arrs[3], gb, DCUT)) if wl['distance_cut'] else None
    d['gb'] = ctx.to_device(gb) if gb is not None else None
    d['trig'] = ctx.malloc(max(n_groups, 1))
    return d


def free_events(ctx, d):
    for p in d['in'] + [d['md'], d['gb'], d['trig']]:
        if p is not None:
            ctx.free(p)


def _oracle_chunk(args):
    """one chunk of the CPU baseline (worker process): the oracle's chain over consecutive event groups -- one station (config 2)
    or the station loop over an array (configs 3 and 5: a group is triggered when any station triggers)"""
    (ev, lo, hi, flavour, arr) = args
    from oracle import spectral_oracle as so   # checker / baseline only
    out = np.zeros(hi - lo, np.uint8)
    grp = ev['group']
    if arr is not None:
        vrms, vrms_e = so.vrms_from_filters(arr['fs'])
        skw = dict(antenna=arr['antenna'], orientation=arr['orientation'], cable_delay=arr['cable_delay'], n_samples=arr['N'], fs=arr['fs'])
        trig = None
        if arr['trigger'] == 'high_low':
            trig = dict(trigger='high_low', n_coincidences=arr['n_coincidences'], threshold_high=3 * vrms, threshold_low=-3 * vrms,
                        high_low_window=5., coinc_window=arr['coinc_window'])
        gen_kw = {}
        stations = np.arange(len(arr['centres']))
        if arr.get('general'):   # config 4: ARZ2020 emission (the GPU run's profile numbers) + birefringent propagation; ONE
            # station per job (a central event group costs the oracle minutes per station: 48 rays x 0.2 s of numpy each)
            from oracle import arz_oracle
            gen_kw = dict(model='ARZ2020', arz=arz_oracle.ARZ(arz_library(), seed=0), birefringence=(birefringence_splines(), None))
            stations = np.array([arr['station']])
        for g in range(lo, hi):
            a, b = np.searchsorted(grp, g), np.searchsorted(grp, g + 1)
            sh = [dict(vertex=ev['vertex'][i], zenith=ev['zenith'][i], azimuth=ev['azimuth'][i], energy=ev['energy'][i],
                       shower_type='HAD' if ev['shower_type'][i] == 0 else 'EM', k_L=ev['k_L'][i], vertex_time=0.,
                       iN=int(ev['iN'][i]) if 'iN' in ev else None) for i in range(a, b)]
            pa = arr.get('pa')
            if pa is not None and 'noise_amp' in pa:
                # thermal noise: keyed by the event group's index in the list, every station its own stream (array.py: the seed of
                # station i is seed + 0x9E3779B97F4A7C15 (i + 1))
                res = []
                for i_st in stations:
                    seed_i = (int(pa['noise_seed']) + 0x9E3779B97F4A7C15 * (int(i_st) + 1)) & 0xffffffffffffffff
                    res += so.simulate_event_group_array(sh, arr['centres'][[i_st]], arr['rel_pos'], arr['ice'], vrms, vrms_e, station_kw=skw,
                                                         att_model=arr['att_model'], n_freq=25, distance_cut_coefficients=DCUT,
                                                         trigger=trig, noise=(seed_i, g, 0, pa['noise_amp']), **gen_kw)
            else:
                res = so.simulate_event_group_array(sh, arr['centres'][stations], arr['rel_pos'], arr['ice'], vrms, vrms_e, station_kw=skw,
                                                    att_model=arr['att_model'], n_freq=25, distance_cut_coefficients=DCUT,
                                                    trigger=trig, **gen_kw)
            if pa is None:
                out[g - lo] = any(o['triggered'] for o in res)
            else:
                # the phased array of phasedArrayBase.py on the oracle's (noisy) traces of the array's channels: analog beams, or the
                # trigger ADC + FFT up-sampling + saturating count sums (the oracle's restatements, pinned by tests/test_gpu_chain.py)
                hit = False
                for o in res:
                    if 'V' not in o:
                        continue
                    Vp = o['V'][pa['channels']]
                    if pa['adc'] is None:
                        hit = hit or bool(so.phased_array_trigger(Vp, pa['rolls'], pa['window'], pa['step'], pa['threshold'])[0])
                    else:
                        a = pa['adc']
                        U = np.array([so.digital_upsampling_fft(so.adc_digital_trace(x, arr['fs'], a['fs'], a['n_bits'], a['vrms'],
                                                                                     a['noise_count'], 'counts'), a['up']) for x in Vp])
                        p = so.phased_array_power_digital(U, pa['rolls'], pa['window'], pa['step'], 'counts')
                        hit = hit or bool(np.any(p > np.trunc(pa['threshold'])))
                out[g - lo] = hit
        return lo, out
    st = so.Station(CHANNELS, n_samples=N_SAMPLES, fs=FS)
    vrms, vrms_e = so.vrms_from_filters(FS)
    for g in range(lo, hi):
        a, b = np.searchsorted(grp, g), np.searchsorted(grp, g + 1)
        if flavour == 'had':
            o = so.simulate_event(ev['vertex'][a], ev['zenith'][a], ev['azimuth'][a], ev['energy'][a], 'HAD', None, st, ICE,
                                  vrms, vrms_e)
        else:
            sh = [dict(vertex=ev['vertex'][i], zenith=ev['zenith'][i], azimuth=ev['azimuth'][i], energy=ev['energy'][i],
                       shower_type='HAD' if ev['shower_type'][i] == 0 else 'EM', k_L=ev['k_L'][i]) for i in range(a, b)]
            o = so.simulate_event_group(sh, st, ICE, vrms, vrms_e)
        out[g - lo] = o['triggered']
    return lo, out


def cpu_baseline(wl, budget_s, n_max, cores=None, arz_iN=None):
    """The oracle (C ray tracer / attenuation + numpy spectral chain) on all host cores over the first event groups of the same
    list; processes chunks until `budget_s` is spent.  Returns (json dict, number of groups done, their triggered flags)."""
    import multiprocessing as mp
    cores = cores or usable_cores()
    ev, grp = wl['events'], wl['events']['group']
    if arz_iN is not None:
        ev = dict(ev, iN=np.asarray(arz_iN))
    arr = None
    if wl['centres'] is not None:   # arrays: every group visits every station (the distance cut drops most of them quickly)
        arr = dict(centres=wl['centres'], rel_pos=wl['rel_pos'], antenna=wl['antenna'], orientation=wl['orientation'],
                   cable_delay=wl['cable_delay'], N=wl['N'], fs=wl['fs'], ice=wl['ice'], att_model=wl['att_model'],
                   trigger=wl['sim_kw'].get('trigger', 'simple'), n_coincidences=wl['sim_kw'].get('n_coincidences', 1),
                   coinc_window=wl['sim_kw'].get('coinc_window', 200.), general=wl['config'] == 4, pa=wl.get('pa'))
    chunk = 125 if arr is None else 4
    n_near = 2   # config 4: the stations nearest to the group's vertex are the jobs (station-events, not whole groups)

    def job(lo, station=None):
        hi = min(lo + chunk, n_max)
        a, b = np.searchsorted(grp, lo), np.searchsorted(grp, hi)
        return ({k: v[a:b] for k, v in ev.items()}, lo, hi, wl['flavour'], arr if station is None else dict(arr, station=station))
    if wl['config'] == 4:
        chunk = 1
        first = np.searchsorted(grp, np.arange(n_max))
        dist = np.linalg.norm(ev['vertex'][first][:, None, :2] - wl['centres'][None, :, :2], axis=2)
        near = np.argsort(dist, axis=1)[:, :n_near]
        pair_list = [(g, int(near[g, k])) for g in range(n_max) for k in range(n_near)]
        jobs = (job(g, i) for g, i in pair_list)
    else:
        jobs = (job(lo) for lo in range(0, n_max, chunk))
    done = {}
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    ctx = mp.get_context('spawn')   # the parent holds a GPU context: never fork it
    with ctx.Pool(cores) as pool:
        pool.map(_noop, range(cores))   # workers up (imports, liboracle.so) before the clock starts
        t0 = time.time()
        it = pool.imap(_oracle_chunk, jobs)
        n_sub = 0
        done_pairs = []
        for lo, out in it:
            done[lo] = out
            done_pairs.append(int(out[0]))
            n_sub += 1
            if time.time() - t0 > budget_s:
                break
        dt = time.time() - t0
        pool.terminate()
    # only the contiguous prefix counts (imap yields in order)
    if wl['config'] == 4:
        # station-events: value = pairs / s / n_stations is what a whole-array loop would reach if every station cost like the
        # near ones (it is a lower bound on the oracle's rate: far stations are cut early)
        flags = np.array([done_pairs[k] for k in range(len(done_pairs))], np.uint8)
        pairs = pair_list[:len(flags)]
        return dict(value=len(flags) / dt / len(wl['centres']), unit="events/s", cores=cores, kind="port",
                    sample="%d (event group, station) pairs -- the %d stations nearest to each of the first %d event groups of the same "
                           "list -- in %.1f s on %d worker processes (oracle: C ray tracer + QUADPACK attenuation, C ARZ integral, numpy "
                           "birefringence and spectral chain); value = pairs / s / %d stations; %d station-events triggered"
                           % (len(flags), n_near, (len(flags) + n_near - 1) // n_near, dt, cores, len(wl['centres']), int(flags.sum()))), pairs, flags
    flags = np.concatenate([done[lo] for lo in sorted(done)]) if done else np.zeros(0, np.uint8)
    n_done = len(flags)
    return dict(value=n_done / dt, unit="events/s", cores=cores, kind="port",
                sample="%d event groups of the same synthetic list in %.1f s on %d worker processes (oracle: C ray tracer + "
                       "QUADPACK attenuation, numpy spectral chain; in-flight chunks of the other workers not counted), %d triggered"
                       % (n_done, dt, cores, int(flags.sum()))), n_done, flags


def end_to_end(st, wl):
    """The whole drop-in for the same list: event-list arrays on the HOST -> the tables the reference's output writer stores
    (output_writer_hdf5.py:215-320) on the HOST -- upload, Earth-absorption weights, the hot path, the traces and per-ray tables of the
    triggered groups, the assembly of the tables.  nuradiomc_amd.output.simulate_to_output, timed as one call."""
    from nuradiomc_amd import output
    ev = wl['events']
    n = len(ev['zenith'])
    data = dict(event_group_ids=ev['group'].astype(np.int64), shower_ids=np.arange(n), xx=ev['vertex'][:, 0], yy=ev['vertex'][:, 1],
                zz=ev['vertex'][:, 2], zeniths=ev['zenith'], azimuths=ev['azimuth'], shower_energies=ev['energy'],
                shower_type=np.where(ev['shower_type'] == 0, 'had', 'em'), energies=np.full(n, 1e18), flavors=np.full(n, 12),
                n_interaction=np.ones(n, int), interaction_type=np.full(n, 'nc'), inelasticity=ev['energy'] / 1e18,
                vertex_times=np.zeros(n), shower_realization_Alvarez2009=ev['k_L'])
    # like the steps: one untimed call first (the dump_traces pass over the triggered groups allocates its 2 GB of trace workspace, the
    # per-length tables of "L = N" are built; on a fresh box that first call takes 0.4 ... 0.6 s), then the timed one
    t0 = time.perf_counter()
    output.simulate_to_output(st, output.EventList(data), station_ids=[101])
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = output.simulate_to_output(st, output.EventList(data), station_ids=[101])
    dt = time.perf_counter() - t0
    trig = out.datasets.get('triggered', np.zeros(0, bool))
    return {"seconds": dt, "first_call_seconds": first, "events_per_s": (int(ev['group'][-1]) + 1) / dt,
            "phases_s": {k: round(v, 3) for k, v in out.timing.items()},
            "n_triggered_showers": int(np.sum(trig)), "n_datasets": len(out.datasets),
            "note": "host arrays in, output tables (all datasets of the reference's HDF5 layout) out; the device-resident step above is "
                    "'pass1' without upload"}


def usable_cores():
    """host cores this process may really use: the CPU affinity mask, capped by the cgroup's CPU quota (the GPU boxes show 256
    logical CPUs but run under a 16-core quota -- 256 busy workers would be throttled to a crawl)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _noop(i):
    from oracle import spectral_oracle  # noqa: F401
    return i


def _json_default(o):
    return o.item() if hasattr(o, 'item') else str(o)


def source_hash():
    """sha256 over the kernel sources: ties a committed profile to the code it was taken on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'nuradiomc_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def launch_ranks(args, argv=None):
    """`--gpus N` means N ranks.  Under a launcher (WORLD_SIZE set: torch.distributed.run, mpirun wrappers) this process IS one of
    them: returns None, unless WORLD_SIZE contradicts --gpus (exit code 2 with a message -- a line that says n_gpus = 8 must not
    have run on one).  Without a launcher and N > 1 this process -- which has made no GPU call yet -- becomes the launcher the
    reference's runner.py:53-87 is: N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one
    device each), rank 0's stdout (the JSON line) is this process's stdout, the other ranks' goes to stderr; returns the exit code
    (non-zero if any child's is; the others are then ended by PID)."""
    import socket
    import subprocess
    ws = os.environ.get('WORLD_SIZE')
    if ws is not None:
        if int(ws) != args.gpus:
            print("bench.py: --gpus %d but the launcher's WORLD_SIZE is %s -- start as many ranks as --gpus says (or drop the "
                  "launcher: bench.py --gpus N starts its own ranks)" % (args.gpus, ws), file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None

    def free_port():
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(('127.0.0.1', 0))
            return s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
               NRHIP_COMM_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                for q in live:   # a rank failed: the others would wait in a collective for ever
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)   # clocks and caches settle over the first calls
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument('--flavour', default='had', choices=['had', 'mixed'])
    ap.add_argument('--events', type=int, default=None, help='event groups per rank and step (weak) or in total (strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--emulate-shard', default=None, metavar='R/W[/CHUNK]',
                    help='one GPU: run the shard rank R of W would get of the --events list under --scaling strong -- contiguous '
                         '(shard_range) or, with CHUNK, interleaved chunks of CHUNK events (shard_chunks); no collectives, no CPU leg')
    ap.add_argument('--chunk', type=int, default=20000, help='config 4: events per call of the general path (spectra and traces of every ray are resident: ~380 KB per ray)')
    ap.add_argument('--cpu-budget', type=float, default=12., help='seconds of CPU baseline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--device', type=int, default=None, help='GPU index of this rank (default: LOCAL_RANK)')
    ap.add_argument('--trigger', default='threshold', choices=['threshold', 'pa', 'pa_adc_noise'],
                    help='configs 3 / 4: threshold on any channel, or the phased array on the four deep dipoles (optionally with the '
                         'trigger ADC and thermal noise)')
    ap.add_argument('--two-pass', action='store_true', help='config 2: the traces of the triggered events from a second pass over them '
                    '(round-2 scheme) instead of emitting them inside pass 1')
    ap.add_argument('--no-traces', action='store_true', help='config 2: time pass 1 only (trigger mask), without the second pass that '
                    'keeps the channel traces of the triggered events')
    ap.add_argument('--lanes', type=int, default=None, help='arrays: station calls side by side on this many streams (one Station object, '
                    'workspace and host thread each; default 2 for config 3, 1 otherwise)')
    ap.add_argument('--allow-tcp', action='store_true', help='if RCCL does not come up on every rank: run the collectives over the TCP '
                    'star instead of exiting non-zero (single-GPU boxes: tools/two_ranks_one_gpu.sh)')
    ap.add_argument('--end-to-end', action='store_true', help='config 2: host event list -> output tables on the host (upload, hot path, '
                    'traces of the triggered events, the tables output_writer_hdf5.py stores), timed as a whole: one extra JSON field')
    ap.add_argument('--no-end-to-end', action='store_true', help='config 2: leave out the `end_to_end` object (the whole drop-in on the same list, '
                    'host event list -> output tables on the host, timed once after the steps)')
    ap.add_argument('--write-expected-sha', action='store_true', help='--scaling strong on ONE rank: record the hash of the trigger mask as the one an N-rank run must gather')
    ap.add_argument('--dry-run', action='store_true', help='launcher check without a GPU: the ranks meet on the TCP star, gather their '
                    '(rank, local rank) pairs and a sharded mask, rank 0 prints a JSON line')
    args = ap.parse_args()
    rc = launch_ranks(args)
    if rc is not None:
        raise SystemExit(rc)
    # this process is a rank: the JSON line alone goes to the stdout it was given -- libraries on the GPU side write to fd 1 as well
    # (RCCL prints a version banner there), so fd 1 becomes stderr and the line is written through a duplicate of the original
    # (not when a caller has replaced sys.stdout to read the line: tools/conv_phase_probe.py runs main() in-process)
    sys.stdout.flush()
    if sys.stdout is sys.__stdout__:
        json_out = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    else:
        json_out = sys.stdout

    def emit(obj):
        json_out.write(json.dumps(obj, default=_json_default) + '\n')
        json_out.flush()
    if args.dry_run:
        from nuradiomc_amd import comm as nrcomm
        rank, local_rank, world = nrcomm.env_rank()
        c = nrcomm.Comm(None, rank, world, backend='tcp')
        n_total = args.events or 1003
        a, b = nrcomm.shard_range(n_total, rank, world)
        full = (np.arange(n_total) % 7 == 0).astype(np.uint8)
        mask = c.allgather_masks(full[a:b].copy(), b - a, n_total)
        seen = c.allreduce_sum([1, rank, local_rank])
        c.barrier()
        if rank == 0:
            emit({"dry_run": True, "n_gpus": world, "ranks": int(seen[0]), "rank_sum": int(seen[1]), "local_rank_sum": int(seen[2]),
                  "mask_ok": bool(np.array_equal(mask, full)), "collectives": c.mode})
        c.close()
        return
    cfgno = args.config
    if args.events is None:
        args.events = {2: 1000000, 3: 1000000, 4: 20000, 5: 1000000}[cfgno]
    if args.steps is None:
        args.steps = {2: 100, 3: 3, 4: 1, 5: 2}[cfgno]   # (config 2: 5 s of GPU work; 400 steps measured the same time per step)
    if args.warmup is None:
        args.warmup = {2: 3, 3: 1, 4: 1, 5: 1}[cfgno]

    import nuradiomc_amd
    from nuradiomc_amd import comm as nrcomm
    rank, local_rank, world = nrcomm.env_rank()
    shard_idx = None
    if args.emulate_shard:
        if world != 1 or cfgno == 4:
            raise SystemExit("bench.py: --emulate-shard is a one-rank option of configs 2, 3, 5")
        q = [int(v) for v in args.emulate_shard.split('/')]
        wl = make_workload(cfgno, args.events, 10, args.flavour, args.trigger)
        g0, g1 = nrcomm.shard_range(args.events, q[0], q[1])
        if len(q) > 2:
            shard_idx = nrcomm.shard_chunks(args.events, q[0], q[1], q[2])
            g0, g1 = 0, len(shard_idx)
        args.no_cpu_baseline = True
        args.no_end_to_end = True
    elif args.scaling == 'strong':
        wl = make_workload(cfgno, args.events, 10, args.flavour, args.trigger)
        g0, g1 = nrcomm.shard_range(args.events, rank, world)
    else:
        wl = make_workload(cfgno, args.events, 10 + rank, args.flavour, args.trigger)
        g0, g1 = 0, args.events
    # one device per rank; with more ranks than devices (two ranks on a one-GPU box: --allow-tcp) the ranks wrap around
    from nuradiomc_amd import _lib as nrlib
    device = args.device if args.device is not None else local_rank % max(int(nrlib.load().nrhip_device_count()), 1)
    ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'], device=device)
    comm = nrcomm.Comm(ctx, rank, world, allow_tcp=True if args.allow_tcp else None)
    det = build_array(ctx, wl)
    is_array = wl['centres'] is not None
    st = det.station if is_array else det
    n_lanes = args.lanes if args.lanes is not None else (2 if cfgno == 3 else 1)   # measured: config 3 -6 %, config 5 +-0
    lane_ctx = []
    if is_array:
        for _ in range(max(n_lanes, 1) - 1):   # (build_array may touch wl['sim_kw']: identical values every time)
            c2 = nuradiomc_amd.Context(wl['ice'], wl['att_model'], device=device)
            lane_ctx.append(c2)
            det.add_lane(build_array(c2, wl).station)
    else:
        n_lanes = 1
    d = upload_events(ctx, wl, shard_idx if shard_idx is not None else (g0, g1))
    n, n_groups = d['n'], d['n_groups']
    dev_kw = dict(d_max_distance=d['md'], n_groups=n_groups, d_group_begin=d['gb'], **wl['sim_kw'])
    arz_iN = None
    if cfgno == 4:   # the profile numbers are shower parameters of the input list (drawn once, resident like k_L)
        arz_iN = iN = st._arz.draw_profile_numbers(d['host'][3], ['HAD' if c == 0 else 'EM' for c in d['host'][4]])
        dev_kw['arz_rows'] = st._arz_shower_profiles(d['host'][3], d['host'][4], iN)

    # traces of the triggered events inside the step: single station (with the second-pass fallback), and the array whose trigger the
    # convolution kernel decides itself (config 3, threshold on any channel: every station call emits the traces of the station-
    # events that trigger there into its buffer; a caller keeps them per station through on_station)
    with_traces = (cfgno == 2 or (cfgno == 3 and args.trigger == 'threshold')) and not args.no_traces

    # config 4 keeps spectra and traces per ray (~380 KB): a list longer than --chunk events (a 1.25e6-event shard of BASELINE
    # configs[3]) is walked in chunks of the RESIDENT list -- pointer offsets, no copies; the counters add up, the mask fills in place
    chunk4 = args.chunk if (cfgno == 4 and n > args.chunk) else None
    if chunk4 and (d['gb'] is not None):
        raise SystemExit("bench.py: --config 4 in chunks needs one shower per event group (--flavour had)")

    def step_chunked():
        from nuradiomc_amd.array import _add_stats
        total = None
        for a in range(0, n, chunk4):
            b = min(n, a + chunk4)
            ins = [d['in'][0] + 24 * a, d['in'][1] + 8 * a, d['in'][2] + 8 * a, d['in'][3] + 8 * a, d['in'][4] + 4 * a, d['in'][5] + 8 * a]
            kw_c = dict(dev_kw, n_groups=b - a, d_max_distance=None if d['md'] is None else d['md'] + 8 * a,
                        arz_rows=tuple(np.ascontiguousarray(r[a:b]) for r in dev_kw['arz_rows']))
            if dev_kw.get('noise'):   # thermal noise is keyed by the group's index in the WHOLE list, not in the chunk
                kw_c['noise_group_offset'] = int(dev_kw.get('noise_group_offset', 0)) + a
            s_ = det.simulate_events_dev(b - a, *ins, d['trig'] + a, want_stats=True, **kw_c)
            if total is None:
                total = dict(s_)
                total['stage_ms'] = dict(s_['stage_ms'])
                total['n_events'], total['n_triggered'] = 0, 0
            else:
                nt = total['n_triggered']
                _add_stats(total, s_)
                total['n_triggered'] = nt
            total['n_triggered'] += s_['n_triggered']
            total['n_events'] = b
        return total

    def step():
        """one pass of the hot path over the resident list; config 2: including the channel traces of the triggered events, what
        the reference writes for them -- the convolution kernel emits all channels of an event the moment it triggers
        (emit_traces); only if an event could not be served that way (buffer full, event decided by another kernel) a second pass
        over the triggered groups (gathered in HBM, dump_traces) follows"""
        if chunk4:
            return step_chunked()
        s1 = det.simulate_events_dev(n, *d['in'], d['trig'], want_stats=True, emit_traces=with_traces and not args.two_pass, **dev_kw)
        if with_traces:
            s1['pass2_ms'] = 0.
            s1['trace_bytes'] = 8 * s1['n_emitted_samples']
            if not is_array and (s1['n_emitted_events'] != s1['n_triggered'] or s1['n_emit_overflow']):
                s2, _, nk = st.triggered_pass_dev(n, *d['in'], d['trig'], n_groups=n_groups, d_group_begin=d['gb'], **wl['sim_kw'])
                s1['pass2_ms'] = s2['stage_ms']['total'] if s2 else 0.
                s1['trace_bytes'] = st.fetch_bytes('trace') if s2 else 0
        return s1

    # untimed steps: the W of the command line, but at least the three calls after which a station object has settled its two
    # per-station choices (one or two stages of the quadrature, block size of the convolution kernel: calls 2 and 3 time one each)
    n_untimed = args.warmup if cfgno == 4 else max(args.warmup, 3)
    for _ in range(n_untimed):
        step()
    comm.barrier()
    t0 = time.perf_counter()
    acc = None
    for k in range(args.steps):
        s = step()
        if acc is None:
            acc = dict(s)
            acc['stage_ms'] = dict(s['stage_ms'])
        else:
            acc['stage_ms'] = {q: acc['stage_ms'][q] + s['stage_ms'][q] for q in s['stage_ms']}
            acc['pass2_ms'] = acc.get('pass2_ms', 0.) + s.get('pass2_ms', 0.)
    comm.barrier()
    elapsed = time.perf_counter() - t0
    stats = s
    if with_traces and is_array and stats.get('n_emit_overflow', 0):
        raise SystemExit("bench.py: %d station-events found the emit buffer full: their traces were NOT written inside the step"
                         % stats['n_emit_overflow'])
    sm = {q: v / max(args.steps, 1) for q, v in acc['stage_ms'].items()}   # average per step (arrays: summed over the stations)

    elapsed = float(comm.allreduce_max([elapsed])[0])
    n_total = args.events if args.scaling == 'strong' else args.events * world
    if args.emulate_shard:
        n_total = n_groups
    if args.scaling == 'strong':
        mask = comm.allgather_masks(d['trig'], n_groups, n_total)   # the one collective: triggered masks over xGMI
        n_trig_total = int(mask.sum())
    else:
        n_trig_total = int(comm.allreduce_sum([stats['n_triggered']])[0])
        if world > 1:   # exercise the collective of the path all the same (equal shards)
            mask = comm.allgather_masks(d['trig'], n_groups, n_total)
            assert int(mask.sum()) == n_trig_total
    mask_sha = hashlib.sha256(np.ascontiguousarray(mask, np.uint8).tobytes()).hexdigest()[:16] if (args.scaling == 'strong' or world > 1) else None
    # a strong-scaling run is self-checking: the gathered mask of N ranks must be the mask of ONE rank over the same list, whose hash
    # is committed (profiles/expected_mask_sha16.json, written by `--scaling strong --gpus 1 --write-expected-sha`)
    sha_check = None
    if rank == 0 and args.scaling == 'strong' and mask_sha is not None:
        key = 'config%d_%s_%s_%d_events' % (cfgno, args.flavour, args.trigger, args.events)
        exp_path = os.environ.get('NRHIP_EXPECTED_SHA_JSON', os.path.join(ROOT, 'profiles', 'expected_mask_sha16.json'))
        known = json.load(open(exp_path)) if os.path.exists(exp_path) else {}
        if args.write_expected_sha and world == 1:
            known[key] = mask_sha
            json.dump(known, open(exp_path, 'w'), indent=1, sort_keys=True)
        if key in known:
            sha_check = 'equal to the one-rank mask (%s)' % exp_path.replace(ROOT + os.sep, '') if known[key] == mask_sha else \
                        'DIFFERS from the one-rank mask %s' % known[key]
        else:
            sha_check = 'no one-rank hash committed for ' + key
    counters = ('n_pairs', 'n_rays', 'n_active_rays', 'n_candidate_events', 'n_integrand_evals')
    tot = dict(zip(counters, comm.allreduce_sum([stats[k] for k in counters])))

    pass2_ms = acc.get('pass2_ms', 0.) / max(args.steps, 1) if with_traces else None   # kernel time of pass 2 inside the step (HIP events)
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = n_total * args.steps / elapsed
        dom = max((k for k in sm if k != 'total'), key=lambda k: sm[k])
        kernel_of = {'raytrace': 'raytrace_roots_fast_kernel + raytrace_records_kernel', 'ray_setup': 'select/scan/ray_setup kernels',
                     'amp_bound': 'amp_bound_kernel', 'attenuation': 'attenuation_dense_kernel',
                     'efield_max': 'efield_bound_kernel + efield_sample_kernel + efield_max_kernel' if cfgno != 4 else
                                   'arz_vector_potential_kernel + bire_steps_kernel + bire_propagate_kernel (general path)',
                     'event_grid': 'event_grid_kernel + candidate lists',
                     'length_tables': 'length_tables_kernel', 'channel': 'channel_prefilter_kernel + channel_conv_kernel'}
        nh = wl['N'] // 2 + 1
        b_field = 2 * nh * 16 + 2 * nh * 16 + 2 * wl['N'] * 8     # write spec_N, c2r N in/out per ray
        scale = wl['N'] / 4096.
        alg = {'raytrace': B_PAIR * stats['n_pairs'], 'ray_setup': 136 * stats['n_rays'],
               'amp_bound': 136 * stats['n_rays'] + 8 * stats['n_rays'],
               'attenuation': (32 + 8 * 25) * stats['n_active_rays'],
               'efield_max': 232 * stats['n_active_rays'] + b_field * stats['n_efield_transforms'],
               'event_grid': 48 * stats['n_rays'], 'length_tables': 0,
               # only the transforms actually carried out are priced (pruned items move no algorithmic bytes)
               'channel': (B_RAY * scale - b_field) * stats['n_ray_transforms'] + B_CHANNEL * scale * stats['n_channel_transforms']}
        if cfgno == 4:   # general path: spectra and traces of every kept ray are materialised
            alg['efield_max'] = stats['n_rays'] * (2 * nh * 16 * 3 + 2 * wl['N'] * 8 * 2)
        alg_bytes = alg.get(dom, 0)
        achieved = alg_bytes / (sm[dom] * 1e-3) / 1e9 if sm[dom] > 0 else 0.
        # HBM bytes per launch from the committed rocprofv3 PMC passes -- quoted only if they were taken on THESE kernel sources
        traffic, traffic_note = None, None
        pmc = os.environ.get('NRHIP_PMC_JSON', os.path.join(ROOT, 'profiles', 'r06_pmc_traffic.json' if cfgno == 2 else
                                                            'r06_pmc_traffic_config%d.json' % cfgno))
        std_size = (cfgno == 2 and args.flavour == 'had' and n == 1000000) or (cfgno == 4 and n == 20000 and args.trigger == 'threshold') or \
                   (cfgno in (3, 5) and n_groups == 1000000 and args.trigger == 'threshold' and args.flavour == 'had')
        if std_size and os.path.exists(pmc):
            pj = json.load(open(pmc))
            if pj.get('source_hash') == source_hash():
                # config 2: the (largest) launch of the dominant kernel; arrays: the kernels of the stage summed over one step
                names = [re.sub(r' .*', '', q) for q in kernel_of[dom].split(' + ')]
                traffic = pj['kernels'].get(names[-1]) if cfgno == 2 else sum(pj['kernels'].get(q, 0.) for q in names)
                traffic_note = "from_profile: GB per %s, rocprofv3 PMC passes (%s), kernel sources %s" % (
                    'launch' if cfgno == 2 else 'step (all launches of the stage\'s kernels)', pj.get('file'), pj['source_hash'])
            else:
                traffic_note = "the PMC profile was taken on other kernel sources (%s != %s): not quoted" % (
                    pj.get('source_hash'), source_hash())
        # FP64 view of the attenuation quadrature: one integrand evaluation = frequency-independent node part (shared
        # by the 25 lanes of a ray, ~110 flop incl. exp, 2 sqrt, 2 div) / 25 + the lane's own part.  SP1 (round 6: table-reduced exp,
        # 12 FP64 instructions = 19 flop instead of 19 instructions = 35 flop): argument 2, exp 19, min / ds / rule sums 8 -> ~29;
        # the other models (a division by the attenuation length instead of the exp): ~45 as before
        flop_per_eval = 110. / 25. + (29. if wl['att_model'] == 'SP1' else 45.)
        fp64 = stats['n_integrand_evals'] * flop_per_eval / (sm['attenuation'] * 1e-3) / 1e12 if sm['attenuation'] > 0 else 0.
        out = {
            "metric": "simulated events/sec (1e6-evt 1 EeV SP survey)", "value": value, "unit": "events/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_steps": n_untimed, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl['name'], "baseline_config_index": cfgno - 1, "flavour": wl['flavour'],
                       "event_groups_per_gpu": n_groups, "showers_per_gpu": n,
                       "n_stations": 1 if not is_array else len(wl['centres']), "station_lanes": n_lanes, "n_channels": len(wl['rel_pos']),
                       "n_pairs": stats['n_pairs'], "n_rays": stats['n_rays'],
                       "n_active_rays": stats['n_active_rays'], "n_candidate_events": stats['n_candidate_events'],
                       "n_channel_items": stats['n_channel_items'], "n_channel_transforms": stats['n_channel_transforms'],
                       "n_ray_transforms": stats['n_ray_transforms'], "n_efield_transforms": stats['n_efield_transforms'],
                       "n_adc_convolution_flops": stats.get('n_adc_convolution_flops', 0),
                       "n_triggered_rank0": stats['n_triggered'], "n_triggered_all": n_trig_total,
                       "all_ranks": {k: int(v) for k, v in tot.items()},
                       "triggered_events_per_s": n_trig_total / (elapsed / max(args.steps, 1)),
                       "step_includes_traces_of_triggered_events": bool(with_traces),
                       "pass2_ms_traces_of_triggered_events": pass2_ms,
                       "traces_emitted_in_pass1": bool(with_traces and not args.two_pass), "n_emitted_events": stats.get('n_emitted_events'),
                       "trace_bytes": stats.get('trace_bytes'), "n_emit_overflow": stats.get('n_emit_overflow', 0),
                       "stage_ms_avg_per_step": {k: round(v, 3) for k, v in sm.items()},
                       "stage_ms_note": ("summed over the stations of the array" + (" and over the %d station lanes that run side by side "
                                         "(the sum exceeds the step)" % n_lanes if n_lanes > 1 else "")) if is_array else None,
                       "gathered_mask_sha16": mask_sha,   # of the all-gathered trigger mask (the same for any number of ranks when strong)
                       "gathered_mask_check": sha_check,
                       "collectives": comm.mode},   # 'local' (one rank), 'rccl', or 'tcp' (RCCL did not come up on every rank)
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": "from_profile" if traffic is not None else None,
                         "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": sm[dom],
                         "attenuation_integrand_evals": stats['n_integrand_evals'],
                         "attenuation_fp64_tflops_est": fp64, "fp64_vector_peak_tflops": FP64_PEAK_TFLOPS,
                         "objective_evals": stats.get('n_objective_evals', 0),
                         "note": "every kernel of the path is FP64-VALU / LDS bound: the fused kernels move ~1e-3 of the "
                                 "un-fused algorithmic bytes of SURVEY 8(d) that the HBM view is priced on"},
        }
        # FP64 view of the channel stage: per channel trace two FFT_MAX-point complex transforms (5 M log2 M flop each) and the
        # spectrum product (14 flop per bin pair), per ray transform an N/2-point complex transform plus ~40 flop per amplitude bin
        M_, nh_ = 8192, wl['N'] // 2
        flop_channel = (stats['n_channel_transforms'] * (2 * 5. * M_ * 13 + 14. * (M_ // 2 + 1)) +
                        stats['n_ray_transforms'] * (5. * nh_ * np.log2(nh_) + 40. * nh_))
        if wl['sim_kw'].get('noise'):   # the noise trace of every channel trace: one more 8192-point transform pair (inverse chirp-z)
            flop_channel += stats['n_channel_transforms'] * (2 * 5. * M_ * 13 + 6. * M_)
        # trigger-ADC chain of the phased array: the chirp convolutions of pa_czt_stage_kernel, M (2 * 5 log2 M + 18) each (transform
        # pair + the three products), summed by the kernel over the convolution lengths it ran
        flop_channel += stats.get('n_adc_convolution_flops', 0)
        # FP64 view of the ray finder: calls of the objective delta_y(log C0) counted by the kernel (hybrd + two Brent searches, ~1e2 per
        # pair) x FLOP_PER_OBJECTIVE (~90 add / mul, 13 divisions and 6 square roots at 1 flop, 1 exp + 4 log at 20)
        flop_of = {'attenuation': stats['n_integrand_evals'] * flop_per_eval, 'channel': flop_channel,
                   'raytrace': stats.get('n_objective_evals', 0) * FLOP_PER_OBJECTIVE}
        if cfgno == 4:   # emission + propagation of the general path, by the work the kernels counted
            flop_of['efield_max'] = (stats.get('n_arz_evals', 0) * FLOP_PER_ARZ_EVAL + stats.get('n_bire_step_bins', 0) * FLOP_PER_BIRE_STEP_BIN +
                                     stats.get('n_bire_steps', 0) * FLOP_PER_BIRE_STEP)
            out["config"].update({k: stats.get(k, 0) for k in ('n_arz_evals', 'n_bire_steps', 'n_bire_step_bins')})
        out["roofline"]["fp64_frac_by_stage"] = {k: (v / (sm[k] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if sm[k] > 0 else 0.) for k, v in flop_of.items()}
        # The pruning stages work in single precision (bounds that are inflated to stay bounds): priced in counted FP32 operations
        # against the FP32 vector peak.  Per ray, from the kernels' code (spectral.hip):
        #   amp_bound_kernel      two-sided sums over groups of 4 bins: (N / 2 - 1) / 4 nodes x 27 (node evaluation 12 + group sums 12
        #                         + 3 shared) + the depth-bin path lengths (2 legs x 64 edges x 22) + the bin sums (63 bins x n_fc x 2)
        #   efield_decide_kernel  per active ray N / 2 bins x (amplitude 9, attenuation interpolation 3, sum of squares 2, total
        #                         variation 2, and the 32 columns of the matrix product (sum_k v_k, 31 samples) x 2) = 16 + 64
        #                         (round 5: efield_bound_kernel 15 per bin of every active ray + efield_sample_kernel 123 per bin of
        #                         the rays it left open; NRHIP_EFIELD_TWO_KERNELS=1 still runs that pair -- price it with 15 / 123)
        # and the N / 2-point transforms of efield_max_kernel in FP64 (5 M log2 M + 40 per amplitude bin).  `vector_time_frac_by_stage`:
        # the time the counted operations would take at the FP64 / FP32 vector peaks over the stage's measured time -- one number
        # per stage, nothing unpriced, and the same for the whole step
        nhb = wl['N'] // 2
        n_fc_ = len(st.att_freq)   # coarse frequencies of the attenuation (analyticraytracing.py:933-960)
        f32_of = {'amp_bound': stats['n_rays'] * ((nhb - 1) / 4. * 27. + 2 * 64 * 22. + 63 * n_fc_ * 2.),
                  'efield_max': (stats['n_active_rays'] * (nhb - 1) * 15. + stats.get('n_efield_sampled', 0) * nhb * (15. + 24 * 4. + 12.)
                                 if os.environ.get('NRHIP_EFIELD_TWO_KERNELS') else stats['n_active_rays'] * nhb * (16. + 64.))}
        f64_of = dict(flop_of)
        if cfgno != 4:
            f64_of['efield_max'] = stats['n_efield_transforms'] * (5. * nhb * np.log2(nhb) + 40. * nhb)
        if cfgno == 4:
            f32_of.pop('efield_max')
        vt = {k: ((f64_of.get(k, 0.) / (FP64_PEAK_TFLOPS * 1e12) + f32_of.get(k, 0.) / (FP32_PEAK_TFLOPS * 1e12)) / (sm[k] * 1e-3) if sm[k] > 0 else 0.)
              for k in ('raytrace', 'amp_bound', 'attenuation', 'efield_max', 'channel')}
        vt['whole_step'] = ((sum(f64_of.values()) / (FP64_PEAK_TFLOPS * 1e12) + sum(f32_of.values()) / (FP32_PEAK_TFLOPS * 1e12)) / (sm['total'] * 1e-3)
                            if sm['total'] > 0 else 0.)
        out["roofline"]["vector_time_frac_by_stage"] = {k: round(v, 4) for k, v in vt.items()}
        out["roofline"]["fp32_flops_by_stage"] = {k: float(v) for k, v in f32_of.items()}
        out["roofline"]["fp32_vector_peak_tflops"] = FP32_PEAK_TFLOPS
        out["config"]["n_efield_sampled"] = stats.get('n_efield_sampled', 0)
        if dom in flop_of:
            # the quadrature and the channel kernels move next to no HBM bytes (everything lives in registers / LDS): priced in
            # algorithmic FP64 flops against the dense FP64 peak of the MI355X (78.6 TFLOP/s, vector and matrix alike; nothing on
            # this path is MFMA-shaped).  attenuation: integrand evaluations (counted by the kernel) x flop_per_eval (DESIGN.md
            # section 4); channel: the transform flops above (that kernel is bound by LDS passes and barriers, not by the VALU)
            tf = flop_of[dom] / (sm[dom] * 1e-3) / 1e12 if sm[dom] > 0 else 0.
            out["roofline"].update({"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": flop_of[dom],
                                    "hbm_view_GBs": achieved})
        out["cpu_baseline"] = None
        if args.trigger != 'threshold' and cfgno == 4:
            pass   # (config 4 with the phased array: GPU-only line; the general path with it is pinned by tests/test_gpu_chain.py)
        elif world == 1 and not args.no_cpu_baseline and cfgno == 4:
            # the oracle needs minutes per central event group here: the check is per (event group, station) -- the per-station masks
            # of one more GPU pass against the oracle's single-station runs
            n_st = len(wl['centres'])
            d_st = ctx.malloc(n_st * n_groups)
            det.simulate_events_dev(n, *d['in'], d['trig'], d_station_triggered=d_st, want_stats=True, **dev_kw)
            st_mask = np.zeros(n_st * n_groups, np.uint8)
            ctx.to_host(st_mask, d_st)
            ctx.free(d_st)
            st_mask = st_mask.reshape(n_st, n_groups)
            base, pairs, flags = cpu_baseline(wl, args.cpu_budget, min(n_groups, 2000), arz_iN=arz_iN)
            got = np.array([st_mask[i, g] for g, i in pairs], np.uint8)
            mism, n_done = int(np.sum(got != flags)), len(pairs)
            base['parity_check'] = ("GPU per-station trigger flags == oracle on the %d sampled (event group, station) pairs (%d of them "
                                    "triggered): %d mismatches" % (n_done, int(flags.sum()), mism))
        elif world == 1 and not args.no_cpu_baseline:
            base, n_done, flags = cpu_baseline(wl, args.cpu_budget, min(n_groups, 200000), arz_iN=arz_iN)
            host_mask = np.zeros(n_groups, np.uint8)
            ctx.to_host(host_mask, d['trig'])
            mism = int(np.sum(host_mask[:n_done] != flags))
            base['parity_check'] = "GPU trigger mask == oracle on the %d sampled event groups: %d mismatches" % (n_done, mism)
        if world == 1 and not args.no_cpu_baseline and (args.trigger == 'threshold' or cfgno != 4):
            out["cpu_baseline"] = base
            if mism:
                emit(out)
                raise SystemExit("bench.py: the GPU trigger mask differs from the oracle's on %d of %d sampled events" % (mism, n_done))
        if sha_check and sha_check.startswith('DIFFERS'):
            emit(out)
            raise SystemExit("bench.py: the gathered trigger mask of %d ranks %s" % (world, sha_check))
        if (args.end_to_end or not args.no_end_to_end) and cfgno == 2 and world == 1 and args.flavour == 'had':
            out["end_to_end"] = end_to_end(st, wl)
        emit(out)
    comm.barrier()
    free_events(ctx, d)
    comm.close()


if __name__ == '__main__':
    main()
