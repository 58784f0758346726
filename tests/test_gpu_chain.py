"""HIP spectral stages and the whole hot path through the C ABI vs the oracle and the reference fixtures."""
import os
import numpy as np
import pytest
from conftest import golden, ROOT
from oracle import raytrace_oracle as rto
from oracle import spectral_oracle as so
import nuradiomc_amd

pytestmark = pytest.mark.gpu


def test_chirp_z_in_lds(gpu_ctx_factory):
    """the in-LDS Bluestein transform against a direct DFT, forward (modulus m) and inverse (modulus L) shapes"""
    ctx = gpu_ctx_factory((1.78, 0.423, 77.))
    rng = np.random.default_rng(3)
    for n_in, n_out, Q, sgn in [(128, 1719, 1719, -1.), (2048, 2648, 2648, -1.), (2649, 5296, 5296, +1.),
                                (2048, 6145, 6145, -1.), (4501, 3691, 9000, +1.), (17, 5, 7, -1.),
                                (128, 8064, 8064, -1.), (8065, 128, 16128, +1.)]:   # the longest common trace at N = 256
        x = rng.normal(size=(3, n_in)) + 1j * rng.normal(size=(3, n_in))
        got = ctx.debug_czt(x, n_out, Q, sgn)
        j = np.arange(n_in)[:, None]
        k = np.arange(n_out)[None, :]
        ref = x @ np.exp(sgn * 2j * np.pi * ((j * k) % Q) / Q)
        assert np.max(np.abs(got - ref)) < 1e-10 * np.max(np.abs(ref)), (n_in, n_out, Q)


def test_wave_sums_without_the_lds_crossbar(gpu_ctx_factory):
    """csrc/wave_reduce.h (v_permlane32_swap / v_permlane16_swap folds + DPP row sums) against plain sums: the FP64 sums to
    rounding, the FP32 ones to FP32 rounding, integers exactly; every lane sees the same wave_sum; the lane-below shift"""
    ctx = gpu_ctx_factory((1.78, 0.423, 77.))
    rng = np.random.default_rng(11)
    x = rng.uniform(0.1, 1., size=(37, 8, 64))
    x[1] = rng.integers(0, 1000, size=(8, 64))          # integers: exact in any order
    x[2] = np.arange(8 * 64).reshape(8, 64)             # every (value, lane) distinguishable
    x[3] = 0.
    x[3, 5, 17] = 1.                                     # one lane of one value
    got = ctx.debug_wave_sums(x)
    ref = x.sum(axis=2)
    assert np.max(np.abs(got[:, :8] - ref) / np.maximum(ref, 1.)) < 1e-14
    assert np.max(np.abs(got[:, 8:16] - ref) / np.maximum(ref, 1.)) < 1e-6
    for w in (1, 2, 3):
        assert np.array_equal(got[w, :8], ref[w]) and np.array_equal(got[w, 8:16], ref[w])
    assert np.all(got[:, 16:80] == got[:, 16:17]) and np.all(got[:, 80:144] == got[:, 80:81])
    assert np.max(np.abs(got[:, 16] - ref[:, 0]) / np.maximum(ref[:, 0], 1.)) < 1e-14
    assert np.max(np.abs(got[:, 80] - ref[:, 0]) / np.maximum(ref[:, 0], 1.)) < 1e-6
    f1 = x[:, 1, :].astype(np.float32).astype(float)
    assert np.array_equal(got[:, 145:208], f1[:, :63]) and np.all(got[:, 144] == 0.)
    assert np.array_equal(got[:, 208], f1[:, 63])
    f2 = x[:, 2, :].astype(np.float32).astype(float)
    assert np.array_equal(got[:, 241:273], f2[:, :32]) and np.all(got[:, 209:241] == 0.)


def test_askaryan_spectrum_vs_oracle(gpu_ctx_factory):
    ctx = gpu_ctx_factory((1.78, 0.423, 77.))
    n_index = 1.78
    thetas = np.arccos(1. / n_index) + np.linspace(-12, 12, 9) * np.pi / 180
    for model in ('Alvarez2009', 'Alvarez2000', 'ZHS1992'):
        for N, dt in ((256, 0.5), (4096, 0.5), (512, 0.2)):
            for st in ('HAD', 'EM'):
                for E in (1e15, 3e17, 1e19):
                    k_L = 42.0 if st == 'EM' else None
                    got = ctx.askaryan_spectrum_batch(E, thetas, N, dt, st, n_index, 1234.5, model, k_L=k_L)
                    for i, th in enumerate(thetas):
                        ref, _ = so.askaryan_frequency_spectrum(E, th, N, dt, st, n_index, 1234.5, model, k_L=k_L)
                        scale = np.max(np.abs(ref))
                        if scale == 0:
                            assert np.all(got[i] == 0)
                        else:
                            assert np.max(np.abs(got[i] - ref)) < 1e-9 * scale, (model, N, st, E, th)
    with pytest.raises(NotImplementedError):
        ctx.askaryan_spectrum_batch(1e18, 1.0, 256, 0.5, 'HAD', 1.78, 1000., 'HCRB2017')
    with pytest.raises(NotImplementedError):
        ctx.askaryan_spectrum_batch(1e18, 1.0, 256, 0.5, 'TAU', 1.78, 1000., 'Alvarez2009')


def _oracle_antenna(g):
    if 'tab_freqs' in g:  # tabulated pattern: the synthetic table travels inside the fixture
        return dict(freqs=g['tab_freqs'], thetas=g['tab_thetas'], phis=g['tab_phis'], H_theta=g['tab_H_theta'],
                    H_phi=g['tab_H_phi'], orientation=g['tab_orientation'])
    return str(g['antenna'])


def _station(ctx, g):
    antenna = str(g['antenna'])
    if 'tab_freqs' in g:
        antenna = nuradiomc_amd.TabulatedAntenna(g['tab_freqs'], g['tab_thetas'], g['tab_phis'], g['tab_H_theta'],
                                                 g['tab_H_phi'], g['tab_orientation'], name=antenna)
    kw = {}
    if 'hw_amp' in g:   # per-channel analog chains (gaussian_tapered + measured amplifier responses)
        from test_oracle_chain import hw_filters
        chains = hw_filters(g)
        kw = dict(channel_filters=[chains[c] for c in range(len(g['det_pos']))])
    return nuradiomc_amd.Station(ctx, g['det_pos'], antenna=antenna, orientation=tuple(g['det_orientation']),
                                 cable_delay=g['cable_delay'], n_samples=int(g['N']), sampling_rate=float(g['fs']),
                                 n_freq=int(g['n_freq']), **kw)


def _oracle_filters(g):
    if 'hw_amp' in g:
        from test_oracle_chain import hw_filters
        return hw_filters(g)
    return so.DEFAULT_FILTERS


def _run_fixture(gpu_ctx_factory, name, n_events, no_pruning=False, dump_traces=True):
    g = golden('chain_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    assert abs(st.vrms - float(g['vrms'])) <= 1e-12 * st.vrms and abs(st.vrms_efield - float(g['vrms_efield'])) <= 1e-12 * st.vrms_efield
    if 'hw_amp' not in g:
        assert st.vrms == float(g['vrms']) and st.vrms_efield == float(g['vrms_efield'])
    sl = slice(0, n_events)
    kL = np.where(np.isnan(g['ev_k_L'][sl]), 1.0, g['ev_k_L'][sl])
    trig, stats = st.simulate_events(g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl],
                                     g['shower_type'][sl], kL, askaryan_model=str(g['askaryan_model']), dump_traces=dump_traces,
                                     no_pruning=no_pruning)
    return g, ctx, st, trig, stats, kL


@pytest.mark.parametrize('name,n_events', [('N256', 300), ('N256_hpol', 150), ('N256_lpda', 200), ('N256_tab', 160),
                                           ('N4096', 60), ('N256_hw', 220), ('N1280', 260), ('N3200', 60), ('N10240', 100)])
def test_spectral_stages_vs_oracle_on_identical_rays(gpu_ctx_factory, name, n_events):
    """Feed the ORACLE with the ray tables the GPU produced, so that every later stage sees identical
    (C0, D, T, launch, receive) on both sides: kept rays exact, amplitudes / traces to 1e-6."""
    g, ctx, st, trig, stats, kL = _run_fixture(gpu_ctx_factory, name, n_events, no_pruning=True)
    n_ch = len(g['det_pos'])
    ost = so.Station(g['det_pos'], antenna=_oracle_antenna(g), orientation=tuple(g['det_orientation']),
                     cable_delay=g['cable_delay'], n_samples=int(g['N']), fs=float(g['fs']))
    T = {k: st.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                  'slot_refl_angle', 'ray_event', 'ray_channel', 'ray_solution', 'ray_view',
                                  'ray_pol_theta', 'ray_pol_phi', 'ray_zenith', 'ray_azimuth', 'ray_t0', 'ray_r_theta',
                                  'ray_r_phi', 'ray_att', 'ray_max_efield', 'ev_n_rays', 'ev_L', 'ev_candidate',
                                  'ev_t_min', 'ev_ray_begin')}
    assert stats['n_rays'] == len(T['ray_event']) == T['ev_n_rays'].sum()
    item_event = st.fetch('item_event') if stats['n_candidate_events'] else np.zeros(0, np.int32)
    maxV = st.fetch('item_maxV').reshape(-1, n_ch) if len(item_event) else np.zeros((0, n_ch))
    toff = st.fetch('trace_offset') if len(item_event) else None
    trace = st.fetch('trace') if len(item_event) else None
    n_fc = len(st.att_freq)
    att = T['ray_att'].reshape(-1, n_fc)
    n_ray_checked = n_cand = 0
    for ev in range(n_events):
        ps = slice(ev * n_ch, (ev + 1) * n_ch)
        ss = slice(ev * n_ch * 2, (ev + 1) * n_ch * 2)
        rays = dict(n_sol=T['pair_n_sol'][ps], type=T['slot_type'][ss].reshape(n_ch, 2),
                    C0=T['slot_C0'][ss].reshape(n_ch, 2), D=T['slot_D'][ss].reshape(n_ch, 2),
                    T=T['slot_T'][ss].reshape(n_ch, 2), refl_angle=T['slot_refl_angle'][ss].reshape(n_ch, 2),
                    launch=T['slot_launch'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3),
                    receive=T['slot_receive'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3))
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev],
                              str(g['shower_type'][ev]), float(kL[ev]), ost, g['ice'], st.vrms, st.vrms_efield,
                              model=str(g['askaryan_model']), rays=rays, filters=_oracle_filters(g))
        r0 = T['ev_ray_begin'][ev]
        sel = np.arange(r0, r0 + T['ev_n_rays'][ev])
        assert [(r['channel'], r['iS']) for r in o['rays']] == list(zip(T['ray_channel'][sel], T['ray_solution'][sel]))
        for r, k in zip(o['rays'], sel):
            assert T['ray_event'][k] == ev
            assert abs(r['view'] - T['ray_view'][k]) < 1e-12
            assert abs(r['pol'][1] - T['ray_pol_theta'][k]) < 1e-12 and abs(r['pol'][2] - T['ray_pol_phi'][k]) < 1e-12
            assert abs(r['zenith'] - T['ray_zenith'][k]) < 1e-12 and abs(r['azimuth'] - T['ray_azimuth'][k]) < 1e-12
            assert abs(r['t0'] - T['ray_t0'][k]) < 1e-9
            assert abs(r['r_theta'] - T['ray_r_theta'][k]) < 1e-12 and abs(r['r_phi'] - T['ray_r_phi'][k]) < 1e-12
            a_ref = rto.attenuation_batch(g['vertex'][ev][None], g['det_pos'][r['channel']][None], [r['C0']], g['ice'],
                                          str(g['att_model']), st.att_freq)[0]
            assert np.max(np.abs(att[k] - a_ref) / a_ref) < 1e-6
            assert abs(r['max_efield'] - T['ray_max_efield'][k]) <= 1e-6 * r['max_efield']
            n_ray_checked += 1
        assert bool(T['ev_candidate'][ev]) == o['candidate']
        assert bool(trig[ev]) == o['triggered']
        if o['candidate']:
            n_cand += 1
            assert T['ev_L'][ev] == o['L']
            assert abs(T['ev_t_min'][ev] - o['t_min']) < 1e-9
            i = int(np.where(item_event == ev)[0][0])
            scale = np.max(np.abs(o['V']))
            assert np.all(np.abs(maxV[i] - np.max(np.abs(o['V']), axis=1)) <= 1e-6 * scale)
            for ch in range(n_ch):
                tr = trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]]
                assert len(tr) == o['L']
                assert np.max(np.abs(tr - o['V'][ch])) <= 1e-6 * scale, (ev, ch)
    assert n_ray_checked > 100 and n_cand >= 5
    assert stats['n_candidate_events'] == n_cand and stats['n_triggered'] == trig.sum()


@pytest.mark.parametrize('output,method,mode', [('counts', 'lin', 'power_sum'), ('counts', 'fir', 'hilbert_env'),
                                                ('voltage', 'fir', 'power_sum'), ('voltage', 'lin', 'hilbert_env'),
                                                ('counts', 'fft', 'hilbert_ideal'), ('voltage', 'fir', 'hilbert_ideal'),
                                                ('analog', 'fft', 'power_sum'), ('analog', 'fir', 'hilbert_env'),
                                                ('analog', 'lin', 'hilbert_ideal')])
def test_phased_array_upsampling_methods_and_envelope_mode(gpu_ctx_factory, output, method, mode):
    """The other processing options of the digitised phased array inside simulate_events: upsampling_method 'lin' / 'fir'
    (coefficients rounded to 1 / 128) and mode 'hilbert_env' (FIR Hilbert transformer, max + 3/8 min; 'hilbert_ideal' here =
    hilbert_transformer_kwargs ideal_transformer=True: scipy.signal.hilbert, exact magnitude); output 'analog' = no trigger ADC
    (phased_trigger(apply_digitization=False) with upsampling_kwargs / the envelope mode: get_traces :312-321).  GPU vs the oracle's
    restatement (pinned on the reference's own functions in test_phased_array_modes_vs_reference) applied to the channel traces the
    GPU dumped: up-sampled traces (counts: every sample; volts 1e-9 lsb), per-beam maxima (window powers or envelopes) and
    decisions."""
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -96.], [0., 0., -97.], [0., 0., -98.], [0., 0., -99.], [0., 0., -60.], [20., 15., -95.]])
    cable = np.array([1.2, 0., 2.6, 0.7, 0., 3.])
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, cable_delay=cable, n_samples=512, sampling_rate=2.0)
    vrms = st.vrms
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 11))
    window, step, adc_fs, nbits, ncount, up, gain, taps, htaps, hgain = 24, 8, 0.472, 8, 5, 4, 128, 31, 31, 128
    rolls = st.set_phased_array([0, 1, 2, 3], angles, ref_index=1.75, window=window, step=step, upsampling_factor=up,
                                adc=None if output == 'analog' else dict(sampling_frequency=adc_fs, n_bits=nbits, noise_count=ncount, output=output),
                                upsampling_method=method, coeff_gain=gain, filter_taps=taps, mode=mode.replace('_ideal', '_env'),
                                hilbert_transformer_kwargs=dict(hilbert_n_taps=htaps, hilbert_coeff_gain=hgain,
                                                                ideal_transformer=mode == 'hilbert_ideal'))
    lsb = vrms / ncount
    unit = vrms / lsb if output == 'counts' else vrms
    threshold = 2.5 * (2 * unit) ** 2 if mode == 'power_sum' else 7. * unit
    if output == 'analog':
        adc_fs = 2.0
    rng = np.random.default_rng(14)
    n = 120
    r, ph = np.sqrt(rng.uniform(0, 1500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500., -10., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(16.8, 18.2, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold, dump_traces=True)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    dig, dlen = st.fetch('pa_digital_trace'), st.fetch('pa_digital_length').reshape(len(item_event), 4)
    stride = len(dig) // (len(item_event) * 4)
    dig = dig.reshape(len(item_event), 4, stride)
    pa_max = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
    n_ch = len(pos)
    n_trig = 0
    for i, e in enumerate(item_event):
        V = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in range(4)])
        U = np.array([so.digital_upsampling(x if output == 'analog' else so.adc_digital_trace(x, 2.0, adc_fs, nbits, vrms, ncount, output),
                                            adc_fs, method, up, gain, taps) for x in V])
        assert np.all(dlen[i] == U.shape[1])
        got = dig[i, :, :U.shape[1]]
        assert np.max(np.abs(got - U)) <= (0 if output == 'counts' else 1e-9 * lsb), e
        if mode == 'power_sum':
            p = so.phased_array_power_digital(U, rolls, window, step, output.replace('analog', 'voltage'))
        else:
            p = so.phased_array_envelope_digital(U, rolls, output.replace('analog', 'voltage'), 8, htaps, hgain, ideal=mode == 'hilbert_ideal')
        mx = p.max(axis=1)
        assert np.max(np.abs(pa_max[i] - mx)) <= 1e-9 * np.max(np.abs(mx)), e
        t = bool(np.any(p > (np.trunc(threshold) if output == 'counts' else threshold)))
        assert t == bool(trig[e]), e
        n_trig += t
    assert len(item_event) >= 15 and 2 <= n_trig < len(item_event)
    with pytest.raises(NotImplementedError):
        st.set_phased_array([0, 1, 2, 3], angles, upsampling_factor=2, upsampling_method='cubic')
    with pytest.raises(ValueError):
        st.set_phased_array([0, 1, 2, 3], angles, mode='amplitude')


@pytest.mark.gpu
def test_custom_polarization(gpu_ctx_factory):
    """config signal.polarization = 'custom' with signal.ePhi (simulation.calculate_polarization_vector :821-825): every ray
    is polarised along (0, sqrt(1 - ePhi^2), ePhi) in its on-sky basis.  GPU vs oracle on identical rays (HPol + VPol station of
    the LPDA fixture: the phi component matters): polarisation components, candidate flags, traces 1e-6, decisions; an unknown
    option raises like the reference."""
    g = golden('chain_N256_lpda.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    n, e_phi = 120, 0.6
    sl = slice(0, n)
    kL = np.where(np.isnan(g['ev_k_L'][sl]), 1.0, g['ev_k_L'][sl])
    args = (g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl], g['shower_type'][sl], kL)
    trig, stats = st.simulate_events(*args, askaryan_model=str(g['askaryan_model']), dump_traces=True, no_pruning=True,
                                     polarization='custom', ePhi=e_phi)
    n_ch = len(g['det_pos'])
    ost = so.Station(g['det_pos'], antenna=_oracle_antenna(g), orientation=tuple(g['det_orientation']),
                     cable_delay=g['cable_delay'], n_samples=int(g['N']), fs=float(g['fs']))
    T = {k: st.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                  'slot_refl_angle', 'ray_pol_theta', 'ray_pol_phi', 'ev_candidate', 'ev_L')}
    assert np.all(np.abs(T['ray_pol_theta'] - 0.8) < 1e-15) and np.all(np.abs(T['ray_pol_phi'] - 0.6) < 1e-15)
    item_event, toff, trace = st.fetch('item_event'), st.fetch('trace_offset'), st.fetch('trace')
    n_cand = 0
    for ev in range(n):
        ps, ss = slice(ev * n_ch, (ev + 1) * n_ch), slice(ev * n_ch * 2, (ev + 1) * n_ch * 2)
        rays = dict(n_sol=T['pair_n_sol'][ps], type=T['slot_type'][ss].reshape(n_ch, 2),
                    C0=T['slot_C0'][ss].reshape(n_ch, 2), D=T['slot_D'][ss].reshape(n_ch, 2),
                    T=T['slot_T'][ss].reshape(n_ch, 2), refl_angle=T['slot_refl_angle'][ss].reshape(n_ch, 2),
                    launch=T['slot_launch'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3),
                    receive=T['slot_receive'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3))
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev], str(g['shower_type'][ev]),
                              float(kL[ev]), ost, g['ice'], st.vrms, st.vrms_efield, model=str(g['askaryan_model']), rays=rays,
                              filters=_oracle_filters(g), polarization_ephi=e_phi)
        assert bool(T['ev_candidate'][ev]) == o['candidate'] and bool(trig[ev]) == o['triggered'], ev
        if o['candidate']:
            n_cand += 1
            i = int(np.where(item_event == ev)[0][0])
            scale = np.max(np.abs(o['V']))
            for ch in range(n_ch):
                tr = trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]]
                assert len(tr) == o['L'] and np.max(np.abs(tr - o['V'][ch])) <= 1e-6 * scale, (ev, ch)
    assert n_cand >= 5
    trig_auto, _ = st.simulate_events(*args, askaryan_model=str(g['askaryan_model']))
    assert not np.array_equal(trig_auto, trig) or n_cand > 0
    with pytest.raises(ValueError):
        st.simulate_events(*args, polarization='eTheta')


@pytest.mark.parametrize('name,n_events', [('N256', 300), ('N256_hpol', 150), ('N256_lpda', 200), ('N256_tab', 160),
                                           ('N4096', 120), ('N256_hw', 220), ('N1280', 260), ('N3200', 120), ('N10240', 100)])
def test_whole_path_vs_reference_fixture(gpu_ctx_factory, name, n_events):
    """End to end (GPU ray tracing included) against the reference's own outputs.  The reference's first ray
    root carries ~1e-7 of iteration noise (see tests/test_oracle_golden.py), which moves arrival times by up to
    ~1e-3 ns, hence amplitudes only to ~1e-3; decisions (candidate / trigger / trace length) must agree except
    where that noise changes the reference's solution count."""
    g, ctx, st, trig, stats, kL = _run_fixture(gpu_ctx_factory, name, n_events)
    n_rays = st.fetch('ev_n_rays')
    same_rays = n_rays == g['ev_n_rays'][:n_events]
    print(name, 'events with the reference\'s ray count: %d of %d' % (same_rays.sum(), n_events))
    assert same_rays.mean() >= 0.99
    assert np.all(n_rays >= g['ev_n_rays'][:n_events])   # the true solution set: only the reference can be short of a ray
    cand = st.fetch('ev_candidate').astype(bool)
    assert np.array_equal(cand[same_rays], g['ev_candidate'][:n_events][same_rays])
    assert np.array_equal(trig[same_rays], g['ev_triggered'][:n_events][same_rays])
    L = st.fetch('ev_L')
    both = same_rays & cand
    assert np.array_equal(L[both], g['ev_L'][:n_events][both])
    item_event = st.fetch('item_event')
    maxV = st.fetch('item_maxV').reshape(len(item_event), -1)
    worst = 0.
    for i, ev in enumerate(item_event):
        if both[ev]:
            ref = g['ev_maxV'][ev]
            worst = max(worst, float(np.max(np.abs(maxV[i] - ref)) / np.max(ref)))
    print(name, 'channel maxima vs reference: max |dV| / max V = %.2e' % worst)
    assert worst <= 6e-4   # observed <= 2.8e-4 on the eight fixtures (a 1e-7 shift of T is a 3e-3 rad phase at 500 MHz)


def reference_rays_table(g, n_events, n_ch):
    """the reference's own launch parameters as the `given_C0` table [event][channel][2] (NaN = no ray): the fixtures keep the
    rays that passed the reference's delta_C cut, (event, channel, iS, C0)"""
    given = np.full((3, n_events, n_ch, 2), np.nan)
    m = g['ray_event'] < n_events
    for i, k in enumerate(('ray_C0', 'ray_D', 'ray_T')):
        given[i, g['ray_event'][m], g['ray_channel'][m], g['ray_iS'][m]] = g[k][m]
    return given, m      # [C0 | D | T][event][channel][2]


def check_against_reference_on_its_rays(g, st, trig, stats, n_events, ref_rays, what, tol=1e-6, tol_path=1e-6):
    """north_star's contract against the REFERENCE (not the oracle): with the reference's rays handed to the batched path, every
    event's decisions are the reference's; path lengths, travel times, field maxima, channel maxima and the stored channel traces
    agree to 1e-6 relative.  No event is masked out.  Everything is measured first and printed, then asserted."""
    n_ch = len(st.position)
    T = {k: st.fetch(k)[:stats['n_rays']] for k in ('ray_event', 'ray_channel', 'ray_C0', 'ray_D', 'ray_t0', 'ray_view', 'ray_zenith',
                                                   'ray_azimuth', 'ray_max_efield')}
    assert stats['n_rays'] == int(ref_rays.sum())                      # every given ray passes the delta_C cut again, nothing else does
    assert np.array_equal(T['ray_event'], g['ray_event'][ref_rays]) and np.array_equal(T['ray_channel'], g['ray_channel'][ref_rays])
    assert np.array_equal(T['ray_C0'], g['ray_C0'][ref_rays])          # the given launch parameters, bit for bit
    worst = {}
    worst['D'] = float(np.max(np.abs(T['ray_D'] - g['ray_D'][ref_rays]) / g['ray_D'][ref_rays]))
    # travel time: the field's start time is vertex time + T - N / (2 fs) (simulation.py:259-268), so a difference of start times
    # IS the difference of travel times; relative to the reference's T where the fixture holds it, else to D n_ice / c >= T
    t_ref = g['ray_T'][ref_rays] if 'ray_T' in g else g['ray_D'][ref_rays] * 1.3 / 0.299792458
    worst['T'] = float(np.max(np.abs(T['ray_t0'] - g['ray_t0'][ref_rays]) / t_ref))
    worst['angles_rad'] = float(max(np.max(np.abs(T[k] - g[k][ref_rays])) for k in ('ray_view', 'ray_zenith', 'ray_azimuth')))
    me = g['ray_max_efield'][ref_rays]
    worst['max_efield'] = float(np.max(np.abs(T['ray_max_efield'] - me) / me))   # (exhaustive mode: exact numbers for every ray)
    ev_cand = st.fetch('ev_candidate')[:n_events].astype(bool)
    item_event = st.fetch('item_event')
    maxV = st.fetch('item_maxV').reshape(len(item_event), n_ch)
    if np.array_equal(item_event, np.flatnonzero(g['ev_candidate'][:n_events])):
        ref = g['ev_maxV'][item_event]
        worst['maxV'] = float(np.max(np.abs(maxV - ref) / np.max(ref, axis=1, keepdims=True)))
        worst['t_min_ns'] = float(np.max(np.abs(st.fetch('ev_t_min')[:n_events][ev_cand] - g['ev_t_min'][:n_events][ev_cand])))
    n_traces = 0
    if 'V_events' in g:
        toff, trace = st.fetch('trace_offset'), st.fetch('trace')
        row = {int(e): i for i, e in enumerate(item_event)}
        worst['trace'] = 0.
        for j, ev in enumerate(g['V_events']):
            if int(ev) >= n_events or int(ev) not in row:
                continue
            V = g['V_concat'][:, g['V_offsets'][j]:g['V_offsets'][j + 1]]
            i = row[int(ev)]
            for ch in range(n_ch):
                tr = trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]]
                assert len(tr) == V.shape[1]
                worst['trace'] = max(worst['trace'], float(np.max(np.abs(tr - V[ch])) / np.max(np.abs(V))))
            n_traces += 1
    print(what, '%d events, %d rays of the reference, %d candidates, %d triggers, %d stored traces; max rel: ' % (
        n_events, stats['n_rays'], int(ev_cand.sum()), int(trig.sum()), n_traces) + ', '.join('%s %.1e' % kv for kv in worst.items()))
    # decisions: exact on ALL events
    assert np.array_equal(st.fetch('ev_n_rays')[:n_events], g['ev_n_rays'][:n_events])
    assert np.array_equal(ev_cand, g['ev_candidate'][:n_events])
    assert np.array_equal(trig.astype(bool), g['ev_triggered'][:n_events])
    L = st.fetch('ev_L')[:n_events]
    assert np.array_equal(L[ev_cand], g['ev_L'][:n_events][ev_cand])
    # numbers: north_star's 1e-6 relative (angles: 1e-9 rad; the readout's start, a time of ~1e4 ns, to 1e-6 of the travel time)
    for k in ('D', 'T', 'max_efield'):
        assert worst[k] <= tol_path, (k, worst[k])
    assert worst['maxV'] <= tol, worst['maxV']
    assert worst['angles_rad'] <= 1e-9 and worst['t_min_ns'] <= 1e-2
    if n_traces:
        assert worst['trace'] <= tol, worst['trace']
    return worst, n_traces


@pytest.mark.parametrize('given', ['C0', 'C0+D+T'])
@pytest.mark.parametrize('name,n_events', [('N256', 300), ('N256_hpol', 150), ('N256_lpda', 200), ('N256_tab', 160),
                                           ('N4096', 120), ('N256_hw', 220), ('N1280', 260), ('N3200', 120), ('N10240', 100)])
def test_reference_rays_through_the_batched_path(gpu_ctx_factory, name, n_events, given):
    """The parity triangle closed on the GPU: the reference's OWN rays (every chain fixture holds them) go
    into the batched path through nrhip_sim_config.given_C0 -- no root search, everything from raytrace_records_kernel onwards as
    in production -- and the results are held against what the REFERENCE computed from the same rays, on ALL events.
      'C0':     launch parameters only.  Decisions exact; path lengths and travel times to 2e-7 -- the reference's closed forms take
                sqrt(n(z_turn)^2 - beta^2) of a difference that cancels completely at a refracted ray's turning point, and what is
                left of it is the rounding of the libm in use (against 60-digit arithmetic either side is exact or ~1e-8 off, at
                random: DESIGN section 2) --, hence traces only to that noise times omega T: observed 7e-12 ... 2e-4.
      'C0+D+T': the reference's path length and travel time handed over as well (given_D / given_T): field maxima, channel maxima
                and the stored channel traces at north_star's 1e-6 against the reference itself."""
    g = golden('chain_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    n_ch = len(g['det_pos'])
    tab, ref_rays = reference_rays_table(g, n_events, n_ch)
    sl = slice(0, n_events)
    kL = np.where(np.isnan(g['ev_k_L'][sl]), 1.0, g['ev_k_L'][sl])
    kw = dict(given_C0=tab[0]) if given == 'C0' else dict(given_C0=tab[0], given_D=tab[1], given_T=tab[2])
    trig, stats = st.simulate_events(g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl], g['shower_type'][sl], kL,
                                     askaryan_model=str(g['askaryan_model']), dump_traces=True, no_pruning=True, **kw)
    tol, tol_path = (5e-4, 2.5e-7) if given == 'C0' else (1e-6, 1e-9)
    worst, n_traces = check_against_reference_on_its_rays(g, st, trig, stats, n_events, ref_rays, name + ' given ' + given, tol, tol_path)
    assert n_traces >= 1 and stats['n_candidate_events'] >= 5
    assert stats['n_objective_evals'] == 0     # no root search ran


@pytest.mark.parametrize('name,n_events', [('N256', 300), ('N256_lpda', 200), ('N256_tab', 160), ('N4096', 120), ('N256_hw', 220),
                                           ('N1280', 260)])
def test_pruning_changes_no_result(gpu_ctx_factory, name, n_events):
    """Skipping rays of events that provably cannot pass the candidate cut (un-attenuated sum-of-magnitudes bound)
    and skipping transforms whose bound is below the cut must leave every decision and every trace unchanged."""
    g, ctx, st, trig_a, stats_a, kL = _run_fixture(gpu_ctx_factory, name, n_events, no_pruning=True)
    A = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ev_t_min', 'item_event', 'item_maxV', 'trace', 'ray_max_efield')}
    g, ctx, st, trig_b, stats_b, kL = _run_fixture(gpu_ctx_factory, name, n_events, no_pruning=False)
    B = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ev_t_min', 'item_event', 'item_maxV', 'trace', 'ray_max_efield',
                                  'ray_bound', 'ray_active', 'ray_att')}
    assert np.array_equal(trig_a, trig_b)
    for k in ('ev_candidate', 'ev_L', 'item_event', 'item_maxV', 'trace'):
        assert np.array_equal(A[k], B[k]), k
    assert stats_b['n_active_rays'] < stats_a['n_active_rays'] == stats_a['n_rays']
    act = B['ray_active'][:stats_b['n_rays']].astype(bool)
    att = B['ray_att'].reshape(stats_b['n_rays'], -1)
    assert np.all(np.isnan(att[~act])) and not np.any(np.isnan(att[act]))
    # the bound really is an upper bound on the exact maximum (no_pruning run), and pruned rays are below the cut
    assert np.all(B['ray_bound'] * (1 + 1e-6) >= A['ray_max_efield'])
    assert np.all(A['ray_max_efield'][~act] <= 2.0 * st.vrms_efield)
    # production mode (no trace dump): transforms whose bound stays below the cut / threshold are skipped and report the
    # negated bound; decisions are unchanged, bounds really bound, exact values agree where they were evaluated
    g, ctx, st, trig_c, stats_c, kL = _run_fixture(gpu_ctx_factory, name, n_events, no_pruning=False, dump_traces=False)
    C = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'item_event', 'item_maxV', 'ray_max_efield')}
    assert np.array_equal(trig_a, trig_c)
    for k in ('ev_candidate', 'ev_L', 'item_event'):
        assert np.array_equal(A[k], C[k]), k
    # per channel: >= 0 exact, < 0 "at most", NaN = not evaluated because an earlier channel of the event had triggered
    n_ch = len(g['det_pos'])
    nan = np.isnan(C['item_maxV'])
    assert np.all(trig_c[np.repeat(C['item_event'], n_ch)[nan]])
    ex = C['item_maxV'] >= 0
    bd = C['item_maxV'] < 0
    assert np.array_equal(C['item_maxV'][ex], A['item_maxV'][ex])
    assert np.all(-C['item_maxV'][bd] * (1 + 1e-9) >= A['item_maxV'][bd])
    assert np.all(-C['item_maxV'][bd] < 3.0 * st.vrms)
    # per ray: negative = "at most", positive = exact or "at least" (samples next to the pulse centre already above
    # the cut); either way on the same side of the cut as the exact maximum
    exr = C['ray_max_efield'] >= 0
    cut = 2.0 * st.vrms_efield
    assert np.all(C['ray_max_efield'][exr] <= A['ray_max_efield'][exr] * (1 + 1e-9))
    assert np.array_equal(C['ray_max_efield'][exr] > cut, A['ray_max_efield'][exr] > cut)
    assert np.all(-C['ray_max_efield'][~exr] * (1 + 1e-6) >= A['ray_max_efield'][~exr])
    n_lb = (C['ray_max_efield'][exr] < A['ray_max_efield'][exr] * (1 - 1e-9)).sum()
    print(name, 'channels evaluated exactly: %d of %d' % (ex.sum(), len(ex)),
          'rays: %d of %d (of which %d by the lower bound)' % (exr.sum(), len(exr), n_lb))


def test_pruning_rigorous_on_survey_sample(gpu_ctx_factory):
    """2e4 synthetic survey events (bench.py's generator, EM showers mixed in, 512-sample traces): the production
    path (all bounds active) and the exhaustive path must give the same candidate flags, trace lengths and trigger mask,
    and every bound must dominate the exactly evaluated quantity."""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=512, sampling_rate=2.0)
    n = 20000
    v, z, a = bench.make_events(n, 77)
    rng = np.random.default_rng(5)
    typ = np.where(rng.random(n) < 0.3, 'EM', 'HAD')
    kL = 10 ** rng.normal(1.9, 0.05, n)
    en = 10 ** rng.uniform(17, 18.5, n)
    trig_a, sa = st.simulate_events(v, z, a, en, typ, kL, no_pruning=True)
    A = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'item_event', 'item_maxV', 'ray_max_efield')}
    trig_b, sb = st.simulate_events(v, z, a, en, typ, kL)
    B = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'item_event', 'item_maxV', 'ray_max_efield', 'ray_bound')}
    assert trig_a.sum() > 20 and sa['n_candidate_events'] > 300
    assert np.array_equal(trig_a, trig_b)
    assert np.array_equal(A['ev_candidate'], B['ev_candidate']) and np.array_equal(A['ev_L'], B['ev_L'])
    assert np.array_equal(A['item_event'], B['item_event'])
    assert sb['n_active_rays'] < 0.8 * sa['n_active_rays']
    assert np.all(B['ray_bound'] * (1 + 1e-6) >= A['ray_max_efield'])
    skipped = B['ray_max_efield'] < 0
    assert np.all(-B['ray_max_efield'][skipped] * (1 + 1e-6) >= A['ray_max_efield'][skipped])
    cut = 2.0 * st.vrms_efield
    assert np.all(B['ray_max_efield'][~skipped] <= A['ray_max_efield'][~skipped] * (1 + 1e-9))
    assert np.array_equal(B['ray_max_efield'][~skipped] > cut, A['ray_max_efield'][~skipped] > cut)
    sk = B['item_maxV'] < 0
    nan = np.isnan(B['item_maxV'])
    assert sk.mean() > 0.3 and nan.sum() > 0
    assert np.all(trig_b[np.repeat(B['item_event'], len(bench.CHANNELS))[nan]])
    assert np.all(-B['item_maxV'][sk] * (1 + 1e-9) >= A['item_maxV'][sk])
    assert np.array_equal(B['item_maxV'][~sk & ~nan], A['item_maxV'][~sk & ~nan])


@pytest.mark.parametrize('N', [64, 250, 1000, 2048, 8192])
def test_candidate_cut_on_the_matrix_cores_at_other_trace_lengths(gpu_ctx_factory, N):
    """efield_decide_kernel (32 rays per wave, the samples as bfloat16 matrix products of split operands) at trace lengths whose
    halves are odd / no multiple of the operand's 8 bins / fill the LDS tables differently: same candidate flags and triggers as
    the exhaustive path, every "at most" above and every "at least" below the exact maximum, and the samples did decide rays"""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=N, sampling_rate=2.0)
    n = 4000
    v, z, a = bench.make_events(n, 78)
    rng = np.random.default_rng(6)
    typ = np.where(rng.random(n) < 0.3, 'EM', 'HAD')
    kL = 10 ** rng.normal(1.9, 0.05, n)
    en = 10 ** rng.uniform(17, 18.7, n)
    trig_a, sa = st.simulate_events(v, z, a, en, typ, kL, no_pruning=True)
    A = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ray_max_efield')}
    trig_b, sb = st.simulate_events(v, z, a, en, typ, kL)
    B = {k: st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ray_max_efield', 'ray_bound')}
    assert sa['n_candidate_events'] > 20
    assert np.array_equal(trig_a, trig_b)
    assert np.array_equal(A['ev_candidate'], B['ev_candidate']) and np.array_equal(A['ev_L'], B['ev_L'])
    assert sb['n_efield_sampled'] > 100 and sb['n_efield_transforms'] < sb['n_efield_sampled']
    assert np.all(B['ray_bound'] * (1 + 1e-6) >= A['ray_max_efield'])
    skipped = B['ray_max_efield'] < 0
    assert skipped.sum() > 100
    assert np.all(-B['ray_max_efield'][skipped] * (1 + 1e-6) >= A['ray_max_efield'][skipped])
    cut = 2.0 * st.vrms_efield
    assert np.all(B['ray_max_efield'][~skipped] <= A['ray_max_efield'][~skipped] * (1 + 1e-9))
    assert np.array_equal(B['ray_max_efield'][~skipped] > cut, A['ray_max_efield'][~skipped] > cut)
    n_lb = (B['ray_max_efield'][~skipped] < A['ray_max_efield'][~skipped] * (1 - 1e-9)).sum()
    print('N', N, 'rays', len(skipped), 'at most:', int(skipped.sum()), 'at least (samples):', int(n_lb), 'transformed:', sb['n_efield_transforms'])


def test_context_destroyed_before_station():
    """destroy order does not matter at the C ABI: a context takes what its stations hold on the GPU with it, the station
    handle stays valid for nrhip_station_destroy (Python finalises objects of a reference cycle in arbitrary order)"""
    import ctypes
    import bench
    ctx = nuradiomc_amd.Context(bench.ICE, 'SP1', device=0)
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=256, sampling_rate=2.0)
    v, z, a = bench.make_events(50, 3)
    st.simulate_events(v, z, a, np.full(50, 1e18), 'HAD')
    lib = st._lib
    h_ctx, h_st = ctx._h, st._h
    lib.nrhip_ctx_destroy(h_ctx)          # behind Python's back: the context goes first
    ctx._h = None
    with pytest.raises(Exception):
        st.fetch('ev_L')
    lib.nrhip_station_destroy(h_st)       # must neither crash nor touch the dead context
    st._h = None


def test_station_move_to(gpu_ctx_factory):
    """one Station object moved through an array of identical stations gives what separate objects give"""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    v, z, a = bench.make_events(3000, 5)
    en = np.full(3000, 1e18)
    centres = np.array([[0., 0., 0.], [800., -300., 0.], [-1500., 900., 0.]])
    moved = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=512, sampling_rate=2.0)
    n_trig = 0
    for c in centres:
        fresh = nuradiomc_amd.Station(ctx, bench.CHANNELS + c, n_samples=512, sampling_rate=2.0)
        t_ref, s_ref = fresh.simulate_events(v, z, a, en, 'HAD')
        moved.move_to(bench.CHANNELS + c)
        with pytest.raises(Exception):
            moved.fetch('ev_L')       # the tables of the last call belonged to the old positions
        t, s_ = moved.simulate_events(v, z, a, en, 'HAD')
        assert np.array_equal(t, t_ref) and s_['n_rays'] == s_ref['n_rays'] and s_['n_candidate_events'] == s_ref['n_candidate_events']
        assert np.array_equal(moved.fetch('ev_L'), fresh.fetch('ev_L'))
        n_trig += t.sum()
        fresh.close()
    assert n_trig > 30
    with pytest.raises(ValueError):
        moved.move_to(np.zeros((3, 3)))


def test_long_lists_are_cut_into_calls(gpu_ctx_factory):
    """simulate_events cuts long shower lists into several calls at event-group boundaries: same mask, summed counters"""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=512, sampling_rate=2.0)
    n = 4000
    v, z, a = bench.make_events(n, 8)
    en = np.full(n, 1e18)
    gid = np.repeat(np.arange(n // 4 + 400), np.random.default_rng(2).integers(1, 6, n // 4 + 400))[:n]
    vt = np.random.default_rng(3).uniform(0, 50, n)
    first = np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
    v = v[first][np.searchsorted(first, np.arange(n), side='right') - 1] + np.random.default_rng(4).uniform(-3, 3, (n, 3))
    ref, s_ref = st.simulate_events(v, z, a, en, 'HAD', vertex_time=vt, group_id=gid)
    cut, s_cut = st.simulate_events(v, z, a, en, 'HAD', vertex_time=vt, group_id=gid, max_showers_per_call=333)
    assert np.array_equal(ref, cut) and ref.sum() > 20 and len(ref) == len(np.unique(gid))
    for k in ('n_events', 'n_pairs', 'n_rays', 'n_candidate_events', 'n_triggered'):
        assert s_ref[k] == s_cut[k], k
    one, _ = st.simulate_events(v, z, a, en, 'HAD', max_showers_per_call=1000)
    assert np.array_equal(one, st.simulate_events(v, z, a, en, 'HAD')[0])


def test_release_workspace(gpu_ctx_factory):
    """the per-call tables can be handed back (arrays simulated station by station) and come back with the next call"""
    g, ctx, st, trig, stats, kL = _run_fixture(gpu_ctx_factory, 'N256', 100)
    assert len(st.fetch('ev_L')) == 100
    assert st.release_workspace() > 1000 and st.release_workspace() == 0
    with pytest.raises(Exception):
        st.fetch('ev_L')
    trig2, _ = st.simulate_events(g['vertex'][:100], g['zenith'][:100], g['azimuth'][:100], g['energy'][:100],
                                  g['shower_type'][:100], kL, askaryan_model=str(g['askaryan_model']), dump_traces=True)
    assert np.array_equal(trig, trig2) and len(st.fetch('ev_L')) == 100


def test_simulate_events_edge_cases(gpu_ctx_factory):
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=256, sampling_rate=2.0)
    # empty event list
    trig, stats = st.simulate_events(np.zeros((0, 3)), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0, np.int32))
    assert trig.shape == (0,) and stats['n_events'] == 0
    # events without any ray solution (shadow zone) and events whose rays all fail the viewing-angle cut
    v = np.array([[3900., 0., -5.], [3800., 100., -3.], [0., 50., -1500.]])
    zen = np.array([0.3, 1.0, 0.0])  # third: shower along the vertical axis, far off the Cherenkov cone
    trig, stats = st.simulate_events(v, zen, np.zeros(3), 1e18, 'HAD')
    assert not trig.any() and stats['n_candidate_events'] == 0 and stats['n_channel_items'] == 0
    assert np.array_equal(st.fetch('ev_L')[:2], [0, 0]) and np.all(np.isnan(st.fetch('ev_t_min')[:2]))
    # one bright event right on the cone: candidate and triggered; zero-energy shower: rays but no signal
    from oracle import spectral_oracle as so
    v = np.array([[300., 100., -400.], [300., 100., -400.]])
    ost = so.Station(bench.CHANNELS, n_samples=256, fs=2.0)
    vr, ve = so.vrms_from_filters(2.0)
    for zen_, az_ in [(2.2, 0.3), (1.0, 3.4)]:
        trig, stats = st.simulate_events(v, zen_, az_, np.array([1e19, 1e15]), 'HAD')
        ref = [so.simulate_event(v[i], zen_, az_, [1e19, 1e15][i], 'HAD', None, ost, bench.ICE, vr, ve)['triggered']
               for i in range(2)]
        assert list(trig) == ref
    # a station whose traces would exceed the supported common-trace length fails loudly, not silently
    far = nuradiomc_amd.Station(ctx, np.array([[0., 0., -100.], [0., 0., -2400.]]), n_samples=4096, sampling_rate=2.0)
    with pytest.raises(nuradiomc_amd.NrhipError):
        far.simulate_events(np.array([[200., 0., -1200.]]), 1.5, 0.2, 1e20, 'HAD', trigger_threshold=1e-9,
                            min_efield_amplitude=1e-12)
    # unsupported configurations raise like the reference does
    with pytest.raises(NotImplementedError):
        nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='createLPDA_100MHz_InfFirn')
    with pytest.raises(NotImplementedError):
        nuradiomc_amd.Context(bench.ICE, 'GL9')


@pytest.mark.parametrize('name,n_events', [('N256', 200), ('N256_hpol', 150), ('N256_lpda', 200), ('N4096', 60), ('N256_hw', 220)])
def test_channel_kernels_agree(gpu_ctx_factory, name, n_events, monkeypatch):
    """Traces up to 8192 samples go through one real convolution per channel (channel_conv_kernel), longer ones through
    the per-ray chirp-z kernel; NRHIP_CHANNEL_CZT=1 sends everything through the latter.  Same traces, same decisions."""
    g, ctx, st, trig_a, stats_a, kL = _run_fixture(gpu_ctx_factory, name, n_events)
    tr_a, mv_a, off = st.fetch('trace').copy(), st.fetch('item_maxV').copy(), st.fetch('trace_offset').copy()
    monkeypatch.setenv('NRHIP_CHANNEL_CZT', '1')
    g, ctx, st, trig_b, stats_b, kL = _run_fixture(gpu_ctx_factory, name, n_events)
    tr_b, mv_b = st.fetch('trace'), st.fetch('item_maxV')
    assert np.array_equal(trig_a, trig_b)
    assert len(tr_a) == len(tr_b) == off[-1] and len(mv_a) > 0
    assert np.max(np.abs(tr_a - tr_b)) <= 1e-9 * np.max(np.abs(tr_b))
    assert np.max(np.abs(mv_a - mv_b)) <= 1e-9 * np.max(np.abs(mv_b))


def test_convolution_kernel_length_classes(gpu_ctx_factory, monkeypatch):
    """channel_conv_kernel has two instantiations: events of up to 4096 samples (half-capacity LDS buffer, two blocks per CU,
    4096- or 2048-point packed transform) and longer ones (8192 / 4096 points).  A 2048-sample station whose channels are up to
    180 m apart produces common traces on both sides of 4096 samples in one call; every trace and decision must equal the
    per-ray chirp-z kernel's (NRHIP_CHANNEL_CZT=1) to 1e-9, and the one-block mode's (NRHIP_CONV_ONE_BLOCK=1) too."""
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -100.], [0., 0., -103.], [40., 0., -60.], [0., 0., -200.], [10., 5., -20.]])
    rng = np.random.default_rng(5)
    n = 400
    r, ph = np.sqrt(rng.uniform(0, 1200. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1800., -10., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.5, 19., n)

    def run():
        ctx = gpu_ctx_factory(ice, 'SP1')
        st = nuradiomc_amd.Station(ctx, pos, n_samples=2048, sampling_rate=2.0)
        trig, stats = st.simulate_events(v, zen, az, en, 'HAD', dump_traces=True)
        return trig, st.fetch('trace').copy(), st.fetch('trace_offset').copy(), st.fetch('ev_L').copy(), st.fetch('item_event').copy()

    trig_a, tr_a, off_a, L_a, ie_a = run()
    Lc = L_a[ie_a]
    n_small, n_large = np.sum(Lc <= 4096), np.sum((Lc > 4096) & (Lc <= 8192))   # (anything longer takes the chirp-z kernel anyway)
    assert n_small >= 10 and n_large >= 10, (n_small, n_large, Lc.max())
    assert 5 < trig_a.sum() < n
    monkeypatch.setenv('NRHIP_CONV_ONE_BLOCK', '1')
    trig_b, tr_b, off_b, _, _ = run()
    monkeypatch.delenv('NRHIP_CONV_ONE_BLOCK')
    monkeypatch.setenv('NRHIP_CHANNEL_CZT', '1')
    trig_c, tr_c, off_c, _, _ = run()
    assert np.array_equal(trig_a, trig_b) and np.array_equal(trig_a, trig_c)
    assert np.array_equal(off_a, off_b) and np.array_equal(off_a, off_c)
    scale = np.max(np.abs(tr_c))
    assert np.max(np.abs(tr_a - tr_c)) <= 1e-9 * scale and np.max(np.abs(tr_b - tr_c)) <= 1e-9 * scale
    # the two instantiations do the same arithmetic on an event (same transform sizes, rays added in the same order): bit-equal traces
    assert np.array_equal(tr_a, tr_b)


@pytest.mark.parametrize('name', ['groups_N256', 'groups_dcut_N256'])
def test_event_groups(gpu_ctx_factory, name):
    """Multi-shower event groups through nrhip_simulate_event_groups: vs the oracle (same rays bit for bit, hence traces
    to 1e-6 and exact decisions) and vs the reference's own outputs (tests/golden/chain_groups_N256.npz); production and
    exhaustive mode give the same masks."""
    g = golden('chain_%s.npz' % name)
    dcut = g['distance_cut_coefficients'] if ('distance_cut' in g and bool(g['distance_cut'])) else None  # speedup.distance_cut
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(ost.fs)
    kL = np.where(np.isnan(g['k_L']), 1.0, g['k_L'])
    args = (g['vertex'], g['zenith'], g['azimuth'], g['energy'], g['shower_type'], kL)
    trig, stats = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'], dump_traces=True,
                                     distance_cut_coefficients=dcut)
    n_groups = len(g['ev_candidate'])
    assert trig.shape == (n_groups,) and stats['n_events'] == n_groups
    cand, L, t_min, n_rays = (st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ev_t_min', 'ev_n_rays'))
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    n_ch = len(g['det_pos'])
    pos = {int(e): i for i, e in enumerate(item_event)}
    n_cand = n_multi = 0
    for gi in range(n_groups):
        idx = np.flatnonzero(g['group'] == gi)
        showers = [dict(vertex=g['vertex'][i], zenith=float(g['zenith'][i]), azimuth=float(g['azimuth'][i]),
                        energy=float(g['energy'][i]), shower_type=str(g['shower_type'][i]), k_L=float(kL[i]),
                        vertex_time=float(g['vertex_time'][i])) for i in idx]
        o = so.simulate_event_group(showers, ost, g['ice'], vrms, vrms_e, distance_cut_coefficients=dcut)
        assert len(o['rays']) == n_rays[gi] and o['candidate'] == bool(cand[gi]) and o['triggered'] == bool(trig[gi]), gi
        if o['candidate']:
            n_cand += 1
            n_multi += len(idx) > 1
            assert o['L'] == L[gi] and abs(o['t_min'] - t_min[gi]) < 1e-9
            scale = np.max(np.abs(o['V']))
            for ch in range(n_ch):
                it = pos[gi] * n_ch + ch
                assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (gi, ch)
        if n_rays[gi] == g['ev_n_rays'][gi]:  # the reference itself, where its first-root noise kept the ray count
            assert bool(cand[gi]) == bool(g['ev_candidate'][gi]) and bool(trig[gi]) == bool(g['ev_triggered'][gi])
    assert (n_cand >= 20 and n_multi >= 8 and trig.sum() >= 5) if dcut is None else (n_cand >= 10 and trig.sum() >= 1)
    trig_p, _ = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'], distance_cut_coefficients=dcut)
    assert np.array_equal(trig, trig_p) and np.array_equal(cand, st.fetch('ev_candidate'))
    with pytest.raises(ValueError):
        st.simulate_events(*args, group_id=np.roll(g['group'], 1))


@pytest.mark.parametrize('kw', [dict(trigger='high_low', n_coincidences=2, hi=2.0, lo=-2.0, high_low_window=5., coinc_window=30.),
                                dict(trigger='high_low', n_coincidences=1, hi=3.0, lo=-3.0, high_low_window=5., coinc_window=200.),
                                dict(trigger='high_low', n_coincidences=3, hi=1.5, lo=-2.5, high_low_window=3., coinc_window=10.),
                                dict(trigger='simple', n_coincidences=3, thr=2.0, coinc_window=20.),
                                dict(trigger='simple', n_coincidences=2, thr=3.0, coinc_window=5000.)])
def test_trigger_modes(gpu_ctx_factory, kw):
    """high/low threshold and n-fold coincidence triggers (highLowThreshold.py:13-150, simpleThreshold.py) fused into the
    channel kernel: the mask and the first triggered bin must be what the reference's logic (oracle restatement, pinned
    bit-exactly by tests/golden/ref_trigger.npz) gives on the very traces the kernel produced; production mode (bounds,
    no trace dump) gives the same mask."""
    g = golden('chain_N256.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    n = 300
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    args = (g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], 3 * g['energy'][:n], g['shower_type'][:n], kL)
    vr = st.vrms
    opts = dict(trigger=kw['trigger'], n_coincidences=kw['n_coincidences'], coinc_window=kw['coinc_window'])
    okw = dict(trigger=kw['trigger'], n_coincidences=kw['n_coincidences'], coinc_window=kw['coinc_window'])
    if kw['trigger'] == 'high_low':
        opts.update(threshold_high=kw['hi'] * vr, threshold_low=kw['lo'] * vr, high_low_window=kw['high_low_window'])
        okw.update(threshold_high=kw['hi'] * vr, threshold_low=kw['lo'] * vr, high_low_window=kw['high_low_window'])
    else:
        opts.update(trigger_threshold=kw['thr'] * vr)
        okw.update(threshold=kw['thr'] * vr)
    trig, stats = st.simulate_events(*args, dump_traces=True, **opts)
    item_event, tr, off, tbin = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset'), st.fetch('ev_trigger_bin')
    n_ch = len(g['det_pos'])
    expect = np.zeros(n, bool)
    for i, e in enumerate(item_event):
        V = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in range(n_ch)])
        t, bins = so.station_trigger(V, st.sampling_rate, **okw)
        expect[e] = t
        assert tbin[e] == (bins[0] if t else -1), e
    assert np.array_equal(trig, expect) and 3 <= trig.sum() < len(item_event)
    trig_p, _ = st.simulate_events(*args, **opts)
    assert np.array_equal(trig_p, trig)
    if kw['n_coincidences'] > 1:   # coincidence logic in production mode stops an event once it cannot trigger any more (the
        # silent channels are counted): mask AND first bin of the events that do trigger stay those of all channels
        assert np.array_equal(st.fetch('ev_trigger_bin')[:n][trig], tbin[:n][trig])


@pytest.mark.parametrize('name,n_events', [('N256', 300), ('N256_hpol', 150), ('N256_lpda', 200), ('N4096', 60), ('N256_hw', 220),
                                           ('N1280', 260), ('N256_tab', 160)])
def test_amp_per_ray_solution(gpu_ctx_factory, name, n_events):
    """speedup.amp_per_ray_solution: per-efield voltage on the N grid, Hilbert-envelope maximum and its time for every
    ray of the candidate events -- vs the oracle on the same rays (1e-6) and vs the reference's own values (5e-3: they
    carry the reference's first-root noise, see test_whole_path_vs_reference_fixture)."""
    g = golden('chain_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    sl = slice(0, n_events)
    kL = np.where(np.isnan(g['ev_k_L'][sl]), 1.0, g['ev_k_L'][sl])
    trig, stats = st.simulate_events(g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl], g['shower_type'][sl],
                                     kL, askaryan_model=str(g['askaryan_model']), amp_per_ray=True)
    env, tsig = st.fetch('ray_max_amp_envelope'), st.fetch('ray_signal_time')
    rev, rch, rsol = st.fetch('ray_event'), st.fetch('ray_channel'), st.fetch('ray_solution')
    cand = st.fetch('ev_candidate').astype(bool)
    assert np.all(np.isnan(env[~cand[rev]])) and not np.any(np.isnan(env[cand[rev]]))
    ost = so.Station(g['det_pos'], antenna=_oracle_antenna(g), orientation=tuple(g['det_orientation']), cable_delay=g['cable_delay'],
                     n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = st.vrms, st.vrms_efield
    n_checked = n_ref = 0
    worst_ref = 0.
    for e in np.flatnonzero(cand):
        o = so.simulate_event(g['vertex'][e], g['zenith'][e], g['azimuth'][e], g['energy'][e], str(g['shower_type'][e]),
                              float(kL[e]), ost, g['ice'], vrms, vrms_e, filters=_oracle_filters(g))
        mine = np.flatnonzero(rev == e)
        assert len(mine) == len(o['rays'])
        for k, r in zip(mine, o['rays']):
            assert (rch[k], rsol[k]) == (r['channel'], r['iS'])
            assert abs(env[k] - r['max_amp_ray']) <= 1e-6 * r['max_amp_ray'] and abs(tsig[k] - r['signal_time']) < 1e-6
            n_checked += 1
        ref = np.flatnonzero(g['ray_event'] == e)
        if len(ref) == len(mine):
            worst_ref = max(worst_ref, float(np.max(np.abs(env[mine] - g['ray_max_amp_ray'][ref]) / g['ray_max_amp_ray'][ref])))
            n_ref += len(ref)
    print(name, 'envelope maxima of %d rays vs reference: max rel %.2e' % (n_ref, worst_ref))
    assert worst_ref <= 1.5e-3
    assert n_checked >= 30 and n_ref >= 25
    # the per-efield voltages are properties of the simulated fields: thermal noise and the trigger in use do not change them
    st.set_noise(300.)
    st.simulate_events(g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl], g['shower_type'][sl], kL,
                       askaryan_model=str(g['askaryan_model']), amp_per_ray=True, noise=True, noise_seed=5)
    assert np.array_equal(st.fetch('ray_max_amp_envelope'), env, equal_nan=True) and np.array_equal(st.fetch('ray_signal_time'), tsig, equal_nan=True)
    if len(g['det_pos']) >= 4 and np.allclose(g['det_pos'][:4, :2], g['det_pos'][0, :2]):
        ang = np.arcsin(np.linspace(-0.8, 0.8, 5))
        st.set_phased_array([0, 1, 2, 3], ang, window=16, step=8)
        st.simulate_events(g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl], g['shower_type'][sl], kL,
                           askaryan_model=str(g['askaryan_model']), amp_per_ray=True, trigger='phased_array',
                           trigger_threshold=2.0 * (2 * st.vrms) ** 2)
        assert np.array_equal(st.fetch('ray_max_amp_envelope'), env, equal_nan=True)


def test_filter_kinds(gpu_ctx_factory):
    """A chain of rectangular + Chebyshev + |Butterworth| stages (signal_processing.get_filter_response types, pinned
    against the reference by tests/test_oracle_golden.py::test_filter_responses_vs_reference) through the whole path:
    Vrms, candidate cut, channel traces and trigger vs the oracle."""
    g = golden('chain_N256.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    chain = [dict(type='rectangular', passband=(0.09, 0.7), order=0), dict(type='cheby1', passband=(0.12, 0.55), order=4, rp=0.5),
             dict(type='butterabs', passband=(0., 0.6), order=6)]
    st = nuradiomc_amd.Station(ctx, g['det_pos'], n_samples=int(g['N']), sampling_rate=float(g['fs']), filters=chain)
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(ost.fs, chain)
    assert abs(st.vrms - vrms) <= 1e-9 * vrms and abs(st.vrms_efield - vrms_e) <= 1e-9 * vrms_e
    n = 200
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    trig, stats = st.simulate_events(g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], g['energy'][:n], g['shower_type'][:n],
                                     kL, dump_traces=True, min_efield_amplitude=2 * vrms_e, trigger_threshold=3 * vrms)
    cand, item_event, tr, off = st.fetch('ev_candidate'), st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos = {int(e): i for i, e in enumerate(item_event)}
    n_cand = 0
    for e in range(n):
        o = so.simulate_event(g['vertex'][e], g['zenith'][e], g['azimuth'][e], g['energy'][e], str(g['shower_type'][e]),
                              float(kL[e]), ost, g['ice'], vrms, vrms_e, filters=chain)
        assert o['candidate'] == bool(cand[e]) and o['triggered'] == bool(trig[e]), e
        if o['candidate']:
            n_cand += 1
            scale = np.max(np.abs(o['V']))
            for ch in range(5):
                it = pos[e] * 5 + ch
                assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (e, ch)
    assert n_cand >= 8


def test_production_mode_is_deterministic(gpu_ctx_factory):
    """Repeated production-mode calls on 2e5 survey events give one and the same trigger mask, equal to the exhaustive
    mode's (regression: the per-event early exit once let a wave that ran ahead reset the block's "event has triggered"
    flag while slower waves still had to read it)."""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=4096, sampling_rate=2.0)
    n = 200000
    v, z, a = bench.make_events(n, 10)
    args = (v, z, a, np.full(n, bench.ENERGY), 'HAD')
    ref, _ = st.simulate_events(*args, no_pruning=True)
    assert ref.sum() > 1000
    for _ in range(12):
        trig, stats = st.simulate_events(*args)
        assert np.array_equal(trig, ref)


def test_full_size_properties(gpu_ctx_factory):
    """BASELINE config 2 at its full size (1e6 events, 5 channels, 4096 samples; the oracle would need an hour for it):
    properties that do not depend on the size -- the trigger mask is a function of the event alone (any permutation of the
    list permutes the mask; halves and unequal shards of the list give the same entries as the whole: what the multi-GPU
    sharding relies on), triggered events are candidates, every count of the stats is consistent with the tables, and the
    known answer of the benchmark list (9053 triggers) is reproduced."""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=4096, sampling_rate=2.0)
    n = 1000000
    v, z, a = bench.make_events(n, 10)
    en = np.full(n, bench.ENERGY)
    trig, stats = st.simulate_events(v, z, a, en, 'HAD')
    assert trig.sum() == 9053 == stats['n_triggered'] and stats['n_events'] == n and stats['n_pairs'] == 5 * n
    cand = st.fetch('ev_candidate').astype(bool)
    n_rays = st.fetch('ev_n_rays')
    assert cand.sum() == stats['n_candidate_events'] and n_rays.sum() == stats['n_rays'] and n_rays.max() <= 10
    assert np.all(cand[trig]) and np.all(n_rays[cand] > 0)
    assert stats['n_active_rays'] <= stats['n_rays'] and stats['n_channel_items'] == 5 * stats['n_candidate_events']
    L = st.fetch('ev_L')
    assert np.all(L[cand] % 2 == 0) and stats['max_length'] == L[cand].max()
    # permutation
    perm = np.random.default_rng(1).permutation(n)
    trig_p, _ = st.simulate_events(v[perm], z[perm], a[perm], en[perm], 'HAD')
    assert np.array_equal(trig_p, trig[perm])
    # shards of unequal size (what bench.py --gpus N does with contiguous ranges)
    cuts = [0, 137, 400000, 400001, 999999, n]
    parts = [st.simulate_events(v[i:j], z[i:j], a[i:j], en[i:j], 'HAD')[0] for i, j in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), trig)


@pytest.mark.parametrize('att_model', ['GL1', 'GL3'])
def test_rnog_like_station_24_channels(gpu_ctx_factory, att_model):
    """A 24-channel station in the shape of RNO-G (BASELINE configs 3-5: deep VPol / HPol strings plus shallow LPDAs in
    several orientations, unequal cable delays, Greenland ice with GL1 attenuation, 2-fold high/low coincidence):
    decisions and channel traces vs the oracle."""
    ice = (1.78, 0.51, 37.25)  # greenland_simple (medium.py:145)
    d = np.pi / 180
    pos, ant, ori = [], [], []
    for i, z in enumerate([-95., -96., -97., -98., -80., -60., -40.]):           # power string: 4 VPol + 3 more VPol
        pos.append([0., 0., z]); ant.append('analytic_VPol'); ori.append([0., 0., 90 * d, 90 * d])
    for z in (-94., -79.):                                                           # two HPols on the power string
        pos.append([0., 0., z]); ant.append('analytic_HPol'); ori.append([0., 0., 90 * d, 90 * d])
    for x, y in ((-20., 30.), (25., 28.)):                                           # two helper strings
        for z in (-95., -94., -93.):
            pos.append([x, y, z]); ant.append('analytic_VPol' if z != -94. else 'analytic_HPol'); ori.append([0., 0., 90 * d, 90 * d])
    for k in range(9):                                                               # 9 shallow LPDAs: 3 up, 6 tilted down
        a = 2 * np.pi * k / 9
        pos.append([12 * np.cos(a), 12 * np.sin(a), -3.])
        ant.append('analytic_LPDA')
        ori.append([0., 0., 90 * d, (90 + 40 * k) * d] if k % 3 == 0 else [120 * d, a, 90 * d, a + 90 * d])
    pos, ori = np.array(pos), np.array(ori)
    assert len(pos) == 24
    cable = np.linspace(0., 37.3, 24)
    kw = {}
    if att_model == 'GL3':
        kw['gl3_table'] = golden('ref_gl3.npz')['gl3_table']
        rto.set_gl3_table(kw['gl3_table'])
    ctx = gpu_ctx_factory(ice, att_model, **kw)
    st = nuradiomc_amd.Station(ctx, pos, antenna=ant, orientation=ori, cable_delay=cable, n_samples=512, sampling_rate=2.0)
    ost = so.Station(pos, antenna=ant, orientation=ori, cable_delay=cable, n_samples=512, fs=2.0)
    vrms, vrms_e = so.vrms_from_filters(2.0)
    rng = np.random.default_rng(12)
    n = 160
    r, ph = np.sqrt(rng.uniform(0, 1500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500., -10., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(16.5, 18., n)
    opts = dict(trigger='high_low', n_coincidences=2, threshold_high=2.5 * vrms, threshold_low=-2.5 * vrms, coinc_window=60.)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', dump_traces=True, **opts)
    cand, item_event, tr, off = st.fetch('ev_candidate'), st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos_of = {int(e): i for i, e in enumerate(item_event)}
    n_cand = 0
    for e in range(n):
        o = so.simulate_event(v[e], zen[e], az[e], en[e], 'HAD', None, ost, ice, vrms, vrms_e, att_model=att_model)
        assert o['candidate'] == bool(cand[e]), e
        if not o['candidate']:
            continue
        n_cand += 1
        scale = np.max(np.abs(o['V']))
        for ch in range(24):
            it = pos_of[e] * 24 + ch
            assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (e, ch)
        t, _ = so.station_trigger(o['V'], 2.0, 'high_low', n_coincidences=2, threshold_high=2.5 * vrms,
                                  threshold_low=-2.5 * vrms, coinc_window=60.)
        assert t == bool(trig[e]), e
    assert n_cand >= 15 and trig.sum() >= 3
    trig_p, _ = st.simulate_events(v, zen, az, en, 'HAD', **opts)
    assert np.array_equal(trig, trig_p)


def test_focusing_batched(gpu_ctx_factory):
    """propagation.focusing in the batched path (second ray-tracing pass to the receivers moved by 1 cm, factor applied to
    the ray's amplitude): per-ray maxima, candidate flags, traces and triggers vs the oracle (both trace the same rays bit
    for bit, so the finite difference is the same), and the factor really changes the amplitudes."""
    g = golden('chain_N256_focus.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(ost.fs)
    n = 200
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    args = (g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], g['energy'][:n], g['shower_type'][:n], kL)
    trig, stats = st.simulate_events(*args, focusing=True, focusing_limit=2., dump_traces=True, no_pruning=True)
    mx, rev = st.fetch('ray_max_efield').copy(), st.fetch('ray_event').copy()
    cand, item_event, tr, off = st.fetch('ev_candidate'), st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos = {int(e): i for i, e in enumerate(item_event)}
    n_cand = n_rays = 0
    for e in range(n):
        o = so.simulate_event(g['vertex'][e], g['zenith'][e], g['azimuth'][e], g['energy'][e], str(g['shower_type'][e]),
                              float(kL[e]), ost, g['ice'], vrms, vrms_e, focusing=True, focusing_limit=2.)
        mine = np.flatnonzero(rev == e)
        assert len(mine) == len(o['rays']) and o['candidate'] == bool(cand[e]) and o['triggered'] == bool(trig[e]), e
        for k, r in zip(mine, o['rays']):
            assert abs(mx[k] - r['max_efield']) <= 1e-6 * r['max_efield']
            n_rays += 1
        if o['candidate']:
            n_cand += 1
            scale = np.max(np.abs(o['V']))
            for ch in range(5):
                it = pos[e] * 5 + ch
                assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (e, ch)
    assert n_cand >= 20 and n_rays > 500
    trig0, _ = st.simulate_events(*args, dump_traces=True, no_pruning=True)
    mx0 = st.fetch('ray_max_efield')
    ratio = mx / mx0
    assert np.all(ratio <= 2.0 * 1.2) and np.mean(np.abs(ratio - 1) > 1e-3) > 0.9
    trig_p, _ = st.simulate_events(*args, focusing=True, focusing_limit=2.)
    assert np.array_equal(trig_p, trig)


@pytest.mark.parametrize('mode,N', [('birefringence', 512), ('arz', 512), ('arz+birefringence', 512), ('arz+birefringence', 640),
                                    ('arz+focusing', 512), ('birefringence', 4100), ('birefringence', 8192), ('birefringence', 10240)])
def test_general_path_arz_birefringence(gpu_ctx_factory, mode, N):
    """BASELINE config 4 inside simulate_events: time-domain ARZ2020 emission and / or birefringent propagation.  The GPU
    materialises the on-sky spectra and traces of every kept ray; compared with the oracle's chain (pinned against the
    reference piece by piece: ARZ traces, birefringent propagation, efield -> voltage) on identical ray tables: ray spectra,
    maxima, candidate flags, trace lengths, channel voltage traces (1e-6) and trigger decisions.  'arz+focusing': propagation.focusing
    with a time-domain model -- the factor multiplies every bin of the ray's spectrum but DC (analyticraytracing.py:3011-3016)."""
    from nuradiomc_amd import arz as arz_mod
    from oracle import arz_oracle
    from test_oracle_golden import _arz_library
    g = golden('chain_N256.npz')
    ice = g['ice']
    fs = 2.0   # (N = 640: a trace length that is no power of two -- Bluestein transforms in the spectrum / trace / channel kernels;
    #            N = 4100: the same on 8192 points, the amplitude tables of the kernels in HBM scratch; N = 10240: radix 5 x 1024)
    ctx = gpu_ctx_factory(ice, 'SP1')
    pos = g['det_pos']
    st = nuradiomc_amd.Station(ctx, pos, n_samples=N, sampling_rate=fs)
    ost = so.Station(pos, n_samples=N, fs=fs)
    vrms, vrms_e = so.vrms_from_filters(fs)
    assert st.vrms == vrms
    rng = np.random.default_rng(4)
    n = 60
    r, ph = np.sqrt(rng.uniform(0, 900. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1200., -120., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.3, 18.7, n)
    types = np.array(['HAD', 'EM'])[rng.integers(0, 2, n)]
    kw, okw = {}, {}
    bire = None
    if 'birefringence' in mode:
        b = golden('ref_birefringence.npz')
        tck = [(b['tck_southpole_A_%d_t' % j], b['tck_southpole_A_%d_c' % j]) for j in range(3)]
        st.set_birefringence(tck, angle_to_iceflow=25.)
        bire = (tck, 25.)
    model = 'Alvarez2009'
    iN = np.zeros(n, int)
    oarz = None
    if 'arz' in mode:
        lib = _arz_library(golden('ref_arz.npz'))
        a = arz_mod.ARZ(seed=3, library=lib)
        st.set_arz(a)
        iN = a.draw_profile_numbers(en, list(types))
        model = 'ARZ2020'
        kw = dict(arz_iN=iN)
        oarz = arz_oracle.ARZ(lib, seed=3)
    if 'focusing' in mode:
        kw = dict(kw, focusing=True, focusing_limit=2.)
        okw = dict(focusing=True, focusing_limit=2.)
    kL = np.where(types == 'EM', 10 ** 1.5, 1.0)
    # the eigen-polarisations divide by n^2 - n_i^2 ~ 1e-3 and ~2000 steps multiply up: two IEEE implementations of the
    # birefringent propagation agree to ~1e-6..1e-5 (the reference's own T07 allows 2e-3 of the pulse amplitude)
    tol = 3e-5 if bire else 2e-6
    trig, stats = st.simulate_events(v, zen, az, en, types, kL, askaryan_model=model, dump_traces=True, **kw)
    T = {k: st.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                  'slot_refl_angle', 'ray_channel', 'ray_solution', 'ray_max_efield', 'ev_n_rays', 'ev_L',
                                  'ev_candidate', 'ev_ray_begin', 'ray_spectra')}
    n_ch, n_f = len(pos), N // 2 + 1
    spectra = T['ray_spectra'].view(np.complex128).reshape(-1, 2, n_f)
    item_event = st.fetch('item_event') if stats['n_candidate_events'] else np.zeros(0, np.int32)
    toff = st.fetch('trace_offset') if len(item_event) else None
    trace = st.fetch('trace') if len(item_event) else None
    n_cand = n_rays = n_trig = 0
    for ev in range(n):
        ps = slice(ev * n_ch, (ev + 1) * n_ch)
        ss = slice(ev * n_ch * 2, (ev + 1) * n_ch * 2)
        rays = dict(n_sol=T['pair_n_sol'][ps], type=T['slot_type'][ss].reshape(n_ch, 2), C0=T['slot_C0'][ss].reshape(n_ch, 2),
                    D=T['slot_D'][ss].reshape(n_ch, 2), T=T['slot_T'][ss].reshape(n_ch, 2),
                    refl_angle=T['slot_refl_angle'][ss].reshape(n_ch, 2),
                    launch=T['slot_launch'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3),
                    receive=T['slot_receive'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3))
        o = so.simulate_event(v[ev], zen[ev], az[ev], en[ev], str(types[ev]), float(kL[ev]), ost, ice, vrms, vrms_e, model=model,
                              rays=rays, arz=(oarz, int(iN[ev])) if oarz else None, birefringence=bire, **okw)
        r0 = T['ev_ray_begin'][ev]
        sel = np.arange(r0, r0 + T['ev_n_rays'][ev])
        assert [(q['channel'], q['iS']) for q in o['rays']] == list(zip(T['ray_channel'][sel], T['ray_solution'][sel]))
        for q, k in zip(o['rays'], sel):
            scale = max(np.max(np.abs(q['spec'][1:])), 1e-300)
            assert np.max(np.abs(spectra[k] - q['spec'][1:])) <= tol * scale, (ev, k)
            assert abs(q['max_efield'] - T['ray_max_efield'][k]) <= tol * max(q['max_efield'], 1e-300)
            n_rays += 1
        assert bool(T['ev_candidate'][ev]) == o['candidate'] and bool(trig[ev]) == o['triggered'], ev
        if o['candidate']:
            n_cand += 1
            n_trig += o['triggered']
            assert T['ev_L'][ev] == o['L']
            i = int(np.where(item_event == ev)[0][0])
            scale = np.max(np.abs(o['V']))
            for ch in range(n_ch):
                tr = trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]]
                assert np.max(np.abs(tr - o['V'][ch])) <= tol * scale, (ev, ch)
    assert n_rays > 100 and n_cand >= 8 and n_trig >= 2
    assert stats['n_candidate_events'] == n_cand
    if N <= 4096:
        # speedup.amp_per_ray_solution (the reference's default) on the general path: per-efield voltages from the rays' spectra in HBM
        trig_a, _ = st.simulate_events(v, zen, az, en, types, kL, askaryan_model=model, amp_per_ray=True, **kw)
        env, tsig, rev = st.fetch('ray_max_amp_envelope'), st.fetch('ray_signal_time'), st.fetch('ray_event')
        assert np.array_equal(trig_a, trig)
        n_env = 0
        for ev in range(n):
            if not T['ev_candidate'][ev]:
                continue
            ps = slice(ev * n_ch, (ev + 1) * n_ch)
            ss = slice(ev * n_ch * 2, (ev + 1) * n_ch * 2)
            rays = dict(n_sol=T['pair_n_sol'][ps], type=T['slot_type'][ss].reshape(n_ch, 2), C0=T['slot_C0'][ss].reshape(n_ch, 2),
                        D=T['slot_D'][ss].reshape(n_ch, 2), T=T['slot_T'][ss].reshape(n_ch, 2),
                        refl_angle=T['slot_refl_angle'][ss].reshape(n_ch, 2),
                        launch=T['slot_launch'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3),
                        receive=T['slot_receive'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3))
            o = so.simulate_event(v[ev], zen[ev], az[ev], en[ev], str(types[ev]), float(kL[ev]), ost, ice, vrms, vrms_e, model=model,
                                  rays=rays, arz=(oarz, int(iN[ev])) if oarz else None, birefringence=bire, **okw)
            for k, q in zip(np.flatnonzero(rev == ev), o['rays']):
                assert abs(env[k] - q['max_amp_ray']) <= tol * q['max_amp_ray'] and abs(tsig[k] - q['signal_time']) < 0.5 / fs + 1e-9, (ev, k)
                n_env += 1
        assert n_env >= 30
    # production mode (no trace dump) on a larger, weaker sample: with birefringence, events whose rays cannot reach the candidate
    # cut even with the largest possible gain of their paths skip the propagation; decisions are unchanged, the bounds bound
    m = 500
    r, ph = np.sqrt(rng.uniform(0, 3000. ** 2, m)), rng.uniform(0, 2 * np.pi, m)
    v2 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2500., -120., m)], axis=1)
    zen2, az2 = np.arccos(rng.uniform(-1, 1, m)), rng.uniform(0, 2 * np.pi, m)
    en2 = 10 ** rng.uniform(15.5, 18., m)
    ty2 = np.array(['HAD', 'EM'])[rng.integers(0, 2, m)]
    kw2 = dict(arz_iN=a.draw_profile_numbers(en2, list(ty2))) if 'arz' in mode else {}
    if 'focusing' in mode:
        kw2 = dict(kw2, focusing=True, focusing_limit=2.)
    trig_x, stats_x = st.simulate_events(v2, zen2, az2, en2, ty2, 10 ** 1.5, askaryan_model=model, no_pruning=True, **kw2)
    X = {k: st.fetch(k) for k in ('ray_max_efield', 'ev_candidate', 'ev_n_rays')}
    trig_p, stats_p = st.simulate_events(v2, zen2, az2, en2, ty2, 10 ** 1.5, askaryan_model=model, **kw2)
    assert np.array_equal(trig_p, trig_x) and stats_p['n_candidate_events'] == stats_x['n_candidate_events'] > 5
    assert np.array_equal(st.fetch('ev_candidate'), X['ev_candidate'])
    mp = st.fetch('ray_max_efield')
    ev_of = np.repeat(np.arange(m), X['ev_n_rays'])
    if bire:
        skipped = mp < 0
        # (round 4: a ray is propagated if its own bound exceeds the candidate cut, or -- in a candidate event -- if its channel can
        # reach the trigger threshold; the others, also those of candidate events, keep "at most <bound>", a bound below the cut)
        assert skipped.sum() > 50
        in_cand = X['ev_candidate'].astype(bool)[ev_of]
        print(mode, 'rays of candidate events never propagated: %d of %d' % ((skipped & in_cand).sum(), in_cand.sum()))
        assert np.all(-mp[skipped] * (1 + 1e-9) >= X['ray_max_efield'][skipped]) and np.all(-mp[skipped] <= 2.0 * st.vrms_efield)
        assert np.array_equal(mp[~skipped], X['ray_max_efield'][~skipped])
        assert np.all(st.fetch('ray_bound') * (1 + 1e-9) >= X['ray_max_efield'])
        print(mode, 'rays propagated: %d of %d' % ((~skipped).sum(), len(mp)))
    else:
        assert np.array_equal(mp, X['ray_max_efield'])


def test_general_path_errors_and_empty_inputs(gpu_ctx_factory):
    """error behaviour and degenerate sizes of the entry points added for reflections, ARZ and birefringence"""
    import bench
    from nuradiomc_amd import arz as arz_mod
    from test_oracle_golden import _arz_library
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = nuradiomc_amd.Station(ctx, bench.CHANNELS, n_samples=256, sampling_rate=2.0)
    v, z, a = bench.make_events(40, 3)
    en = np.full(40, 1e18)
    with pytest.raises(ValueError, match='shower library'):
        st.simulate_events(v, z, a, en, 'HAD', askaryan_model='ARZ2020', arz_iN=np.zeros(40, int))
    lib = _arz_library(golden('ref_arz.npz'))
    arz = arz_mod.ARZ(seed=1, library=lib)
    st.set_arz(arz)
    with pytest.raises(ValueError, match='profile number'):
        st.simulate_events(v, z, a, en, 'HAD', askaryan_model='ARZ2020')
    with pytest.raises(KeyError):   # HAD 1e18 has three profiles in this library
        st.simulate_events(v, z, a, en, 'HAD', askaryan_model='ARZ2020', arz_iN=np.full(40, 7))
    # (focusing and amp_per_ray with the ARZ models: test_general_path_arz_birefringence)
    # high/low and coincidence triggers on the general path: decided on the dumped traces (trace_trigger_kernel)
    v2, z2, a2 = bench.make_events(300, 5)
    v2[:, :2] *= 0.25
    vr = st.vrms
    opts = dict(trigger='high_low', n_coincidences=2, coinc_window=30., threshold_high=2 * vr, threshold_low=-2 * vr, high_low_window=5.)
    trig, stats = st.simulate_events(v2, z2, a2, np.full(300, 3e18), 'HAD', askaryan_model='ARZ2020', arz_iN=np.zeros(300, int),
                                     dump_traces=True, **opts)
    _check_trace_triggers(st, trig, opts, 300)
    assert 1 <= trig.sum() < stats['n_candidate_events']
    # far away, off cone: no ray survives -> nothing to do, no trigger
    far = np.tile([30000., 0., -500.], (3, 1))
    trig, stats = st.simulate_events(far, np.full(3, 0.3), np.zeros(3), np.full(3, 1e18), 'EM', askaryan_model='ARZ2020',
                                     arz_iN=np.zeros(3, int))
    assert not trig.any() and stats['n_candidate_events'] == 0
    trig, stats = st.simulate_events(np.zeros((0, 3)), [], [], [], 'HAD')
    assert len(trig) == 0
    # the birefringence switch
    b = golden('ref_birefringence.npz')
    tck = [(b['tck_southpole_A_%d_t' % j], b['tck_southpole_A_%d_c' % j]) for j in range(3)]
    t0, _ = st.simulate_events(v, z, a, en, 'HAD')
    st.set_birefringence(tck)
    t1, s1 = st.simulate_events(v, z, a, en, 'HAD')
    assert len(st.fetch('ray_spectra')) > 0
    st.set_birefringence(None)
    t2, _ = st.simulate_events(v, z, a, en, 'HAD')
    assert np.array_equal(t0, t2)
    with pytest.raises(Exception):
        st.fetch('ray_spectra')
    with pytest.raises(Exception, match='8 knots'):
        st.set_birefringence([(np.arange(5.), np.arange(5.))] * 3)
    # batch entry points with nothing in them
    assert ctx.find_solutions_reflections_batch(np.zeros((0, 3)), np.zeros((0, 3)), 1, -500.)['C0'].shape == (0, 6)
    assert ctx.attenuation_reflections_batch(np.zeros((0, 3)), np.zeros((0, 3)), [], [], [], -500., [0.1, 0.2]).shape == (0, 2)
    assert ctx.birefringence_batch(np.zeros((0, 3)), np.zeros((0, 3)), [], [], np.zeros((0, 2, 129), complex), 2.0, tck).shape == (0, 2, 129)
    assert arz.get_time_trace_batch([], [], 256, 0.5, [], 1.78, [], []).shape == (0, 3, 256)


def test_phased_array_trigger(gpu_ctx_factory):
    """Phased-array trigger inside simulate_events (a 4-dipole string + other channels): beam rolls as the reference's
    calculate_time_delays, decisions and per-beam maximum window powers vs the oracle's core (pinned against the reference's
    phase_signals / power_sum) applied to the oracle's channel traces."""
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -96.], [0., 0., -97.], [0., 0., -98.], [0., 0., -99.], [0., 0., -60.], [20., 15., -95.]])
    cable = np.array([1.2, 0., 2.6, 0.7, 0., 3.])
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, cable_delay=cable, n_samples=512, sampling_rate=2.0)
    ost = so.Station(pos, cable_delay=cable, n_samples=512, fs=2.0)
    vrms, vrms_e = so.vrms_from_filters(2.0)
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 11))
    window, step = 32, 16
    rolls = st.set_phased_array([0, 1, 2, 3], angles, ref_index=1.75, window=window, step=step)
    assert np.array_equal(rolls, so.phased_array_rolls(pos[:4, 2], cable[:4], angles, 2.0, 1.75))
    threshold = 2.5 * (2 * vrms) ** 2     # power of the 4-channel coherent sum
    rng = np.random.default_rng(14)
    n = 120
    r, ph = np.sqrt(rng.uniform(0, 1500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500., -10., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(16.8, 18.2, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold, dump_traces=True)
    cand = st.fetch('ev_candidate').astype(bool)
    item_event = st.fetch('item_event')
    pa_max = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
    # production mode: events whose channel bounds cannot add up to the power threshold are not transformed (their powers read 0)
    trig_p, stats_p = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold)
    pa_max_p = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
    assert np.array_equal(trig_p, trig) and stats_p['n_channel_transforms'] < stats['n_channel_transforms']
    skipped = np.all(pa_max_p == 0, axis=1)
    assert skipped.sum() >= 3 and np.all(pa_max[skipped] < threshold) and np.array_equal(pa_max_p[~skipped], pa_max[~skipped])
    n_cand = n_trig = 0
    for e in range(n):
        o = so.simulate_event(v[e], zen[e], az[e], en[e], 'HAD', None, ost, ice, vrms, vrms_e)
        assert o['candidate'] == bool(cand[e])
        if not o['candidate']:
            assert not trig[e]
            continue
        n_cand += 1
        t, mx = so.phased_array_trigger(o['V'][:4], rolls, window, step, threshold)
        i = int(np.where(item_event == e)[0][0])
        assert np.max(np.abs(pa_max[i] - mx)) <= 1e-6 * np.max(mx), e
        if np.min(np.abs(mx - threshold)) > 1e-6 * threshold:
            assert t == bool(trig[e]), e
        n_trig += t
    assert n_cand >= 15 and 3 <= n_trig < n_cand
    simple, _ = st.simulate_events(v, zen, az, en, 'HAD')      # the per-channel threshold trigger is untouched
    assert simple.sum() != trig.sum() or not np.array_equal(simple, trig)
    st.set_phased_array(None, angles)
    with pytest.raises(Exception, match='phased-array'):
        st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold)
    with pytest.raises(NotImplementedError):
        st.set_phased_array([0, 5], angles)


@pytest.mark.parametrize('cable_ns,l_lo,l_hi', [(4500., 12290, 16382), (6500., 16384, 21000), (9000., 21000, 32766)])
def test_common_traces_beyond_the_single_block_of_the_transforms(gpu_ctx_factory, cable_ns, l_lo, l_hi):
    """Common traces of 12 290 ... 32 766 samples at N = 4096 (round 3): the forward chirp-z of channel_kernel takes its L / 2 output bins
    in blocks when N / 2 + L / 2 - 1 exceeds the 8192-point transform, the inverse one cuts its L / 2 + 1 input bins into chunks
    when fewer than 1024 outputs per block would be left (L > 14 336).  A station whose last channel sits behind 4.5 / 6.5 / 9 us
    of cable stretches the read-out window; GPU vs the oracle (numpy FFTs of length L) on the same rays: L, t_min, traces 1e-6,
    decisions.  Beyond 32 766 samples the per-length tables end: refused."""
    import bench
    ice = bench.ICE
    pos = np.array(bench.CHANNELS, float)
    cable = np.array([0., 0., 0., 0., cable_ns])
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, cable_delay=cable, n_samples=4096, sampling_rate=2.0)
    ost = so.Station(pos, cable_delay=cable, n_samples=4096, fs=2.0)
    vrms, vrms_e = so.vrms_from_filters(2.0)
    rng = np.random.default_rng(8)
    n = 60
    r, ph = np.sqrt(rng.uniform(0, 700. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-900., -150., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.5, 18.5, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', dump_traces=True)
    cand, L, t_min = st.fetch('ev_candidate').astype(bool), st.fetch('ev_L'), st.fetch('ev_t_min')
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos_of = {int(e): i for i, e in enumerate(item_event)}
    n_long = n_trig = 0
    for e in range(n):
        o = so.simulate_event(v[e], zen[e], az[e], en[e], 'HAD', None, ost, ice, vrms, vrms_e)
        assert o['candidate'] == bool(cand[e]) and o['triggered'] == bool(trig[e]), e
        if not o['candidate']:
            continue
        assert o['L'] == L[e] and abs(o['t_min'] - t_min[e]) < 1e-9
        n_long += (l_lo < L[e] <= l_hi)
        n_trig += o['triggered']
        scale = np.max(np.abs(o['V']))
        for ch in range(5):
            it = pos_of[e] * 5 + ch
            assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (e, ch, L[e])
    assert n_long >= 8 and n_trig >= 3 and stats['max_length'] <= l_hi
    trig_p, _ = st.simulate_events(v, zen, az, en, 'HAD')
    assert np.array_equal(trig_p, trig)
    if cable_ns > 8000.:
        st2 = nuradiomc_amd.Station(ctx, pos, cable_delay=np.array([0., 0., 0., 0., 14500.]), n_samples=4096, sampling_rate=2.0)
        with pytest.raises(Exception, match='32766'):
            st2.simulate_events(v, zen, az, en, 'HAD', dump_traces=True)
        opts = dict(trigger='high_low', n_coincidences=2, coinc_window=30., threshold_high=2 * vrms, threshold_low=-2 * vrms, high_low_window=5.)
        with pytest.raises(Exception, match='16384'):
            st.simulate_events(v, zen, az, en, 'HAD', **opts)


@pytest.mark.parametrize('N', [8192, 6400, 4098, 10240, 14336])
def test_traces_of_more_than_4096_samples(gpu_ctx_factory, N):
    """N = 8192 and trace lengths between 4096 and 8192 that are no power of two in the batched path (round 3): the ray stages hold
    the N / 2-point transform (Bluestein on 8192 points for 6400 and 4098: the whole LDS of those kernels, their amplitude tables in
    HBM scratch), the channel stage is the chirp-z kernel (forward transform in output blocks, the amplitude table in HBM scratch
    because 128 + 32 KB of LDS do not exist).  N = 10 240 and 14 336 (N / 2 = 5 and 7 times 1024): one odd-radix pass + radix-2
    transforms in LDS (NPlan.radix).  GPU vs the oracle: rays, candidate flags, L, traces 1e-6, decisions."""
    import bench
    ice = bench.ICE
    pos = np.array(bench.CHANNELS, float)
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, n_samples=N, sampling_rate=2.0)
    ost = so.Station(pos, n_samples=N, fs=2.0)
    vrms, vrms_e = so.vrms_from_filters(2.0)
    assert abs(st.vrms - vrms) <= 1e-12 * vrms
    rng = np.random.default_rng(18)
    n = 50
    r, ph = np.sqrt(rng.uniform(0, 800. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1000., -150., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.3, 18.5, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', dump_traces=True)
    cand, L, n_rays, mx = st.fetch('ev_candidate').astype(bool), st.fetch('ev_L'), st.fetch('ev_n_rays'), st.fetch('ray_max_efield')
    rev = st.fetch('ray_event')
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos_of = {int(e): i for i, e in enumerate(item_event)}
    n_cand = n_trig = 0
    for e in range(n):
        o = so.simulate_event(v[e], zen[e], az[e], en[e], 'HAD', None, ost, ice, vrms, vrms_e)
        mine = np.flatnonzero(rev == e)
        assert len(mine) == len(o['rays']) == n_rays[e] and o['candidate'] == bool(cand[e]) and o['triggered'] == bool(trig[e]), e
        for k, q in zip(mine, o['rays']):   # (a negative entry: "not evaluated, at most this" -- the event cannot be a candidate)
            assert abs(mx[k] - q['max_efield']) <= 1e-6 * q['max_efield'] if mx[k] > 0 else -mx[k] >= q['max_efield'] * (1 - 1e-9)
        if not o['candidate']:
            continue
        n_cand += 1
        n_trig += o['triggered']
        assert o['L'] == L[e] > N + 1200
        scale = np.max(np.abs(o['V']))
        for ch in range(5):
            it = pos_of[e] * 5 + ch
            assert np.max(np.abs(tr[off[it]:off[it + 1]] - o['V'][ch])) <= 1e-6 * scale, (e, ch)
    assert n_cand >= 10 and n_trig >= 3
    trig_p, _ = st.simulate_events(v, zen, az, en, 'HAD', amp_per_ray=True)
    assert np.array_equal(trig_p, trig)
    # per-efield voltages (amp_per_ray) on these grids: the N-point analytic signal (two N / 2-point transforms where N / 2 is no
    # power of two), the kernel's amplitude table in HBM scratch
    env, tsig = st.fetch('ray_max_amp_envelope'), st.fetch('ray_signal_time')
    n_env = 0
    for e in np.flatnonzero(cand):
        o = so.simulate_event(v[e], zen[e], az[e], en[e], 'HAD', None, ost, ice, vrms, vrms_e)
        for k, q in zip(np.flatnonzero(rev == e), o['rays']):
            assert abs(env[k] - q['max_amp_ray']) <= 1e-6 * q['max_amp_ray'] and abs(tsig[k] - q['signal_time']) < 1e-6, (e, k)
            n_env += 1
    assert n_env >= 20


@pytest.mark.parametrize('mode', ['arz', 'phased_array'])
def test_split_event_time_diff_on_the_general_path_and_with_the_phased_array(gpu_ctx_factory, mode):
    """split_event_time_diff together with the time-domain emission model (the general path re-orders its per-ray tables with the
    sub-events) and with the phased-array trigger (decisions per sub-event, OR-ed into the group).  The multi-shower groups of the
    reference's split fixture: 'arz' -- GPU vs the oracle's simulate_event_group(arz=, split_event_time_diff=) on the same rays:
    membership, L, t_min, traces, decisions; 'phased_array' -- the oracle's beam former on the traces of every sub-event the GPU
    dumped (those are pinned by test_split_event_time_diff), decisions per sub-event and per group."""
    g = golden('chain_split_N256.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(ost.fs)
    split = float(g['split_event_time_diff'])
    kL = np.where(np.isnan(g['k_L']), 50.0, g['k_L'])
    n_groups = len(g['ev_candidate'])
    n_ch = len(g['det_pos'])
    args = (g['vertex'], g['zenith'], g['azimuth'], g['energy'], g['shower_type'], kL)
    kw = dict(vertex_time=g['vertex_time'], group_id=g['group'], split_event_time_diff=split)
    if mode == 'arz':
        from nuradiomc_amd import arz as arz_mod
        from oracle import arz_oracle
        from test_oracle_golden import _arz_library
        lib = _arz_library(golden('ref_arz.npz'))
        a = arz_mod.ARZ(seed=3, library=lib)
        st.set_arz(a)
        types = [str(t) for t in g['shower_type']]
        iN = a.draw_profile_numbers(g['energy'], types)
        oarz = arz_oracle.ARZ(lib, seed=3)
        trig, stats = st.simulate_events(*args, askaryan_model='ARZ2020', arz_iN=iN, dump_traces=True, **kw)
    else:
        angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 7))
        pa_ch = [0, 1, 2, 3]
        window, step = 16, 8
        rolls = st.set_phased_array(pa_ch, angles, ref_index=1.75, window=window, step=step)
        threshold = 2.0 * (2 * vrms) ** 2
        trig, stats = st.simulate_events(*args, trigger='phased_array', trigger_threshold=threshold, dump_traces=True, **kw)
    ev_group, ev_sub = st.fetch('ev_group'), st.fetch('ev_sub_event')
    cand, L, t_min, n_rays, ev_trig = (st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ev_t_min', 'ev_n_rays', 'ev_triggered'))
    assert stats['n_sub_events'] == len(ev_group) > n_groups and np.all(np.diff(ev_group) >= 0)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    pos = {int(e): i for i, e in enumerate(item_event)}
    n_split = n_sub_trig = 0
    if mode == 'arz':
        for gi in range(n_groups):
            idx = np.flatnonzero(g['group'] == gi)
            showers = [dict(vertex=g['vertex'][i], zenith=float(g['zenith'][i]), azimuth=float(g['azimuth'][i]),
                            energy=float(g['energy'][i]), shower_type=str(g['shower_type'][i]), k_L=float(kL[i]),
                            vertex_time=float(g['vertex_time'][i]), iN=int(iN[i])) for i in idx]
            o = so.simulate_event_group(showers, ost, g['ice'], vrms, vrms_e, model='ARZ2020', arz=oarz, split_event_time_diff=split)
            mine = np.flatnonzero(ev_group == gi)
            assert list(ev_sub[mine]) == list(range(len(mine)))
            assert n_rays[mine].sum() == len(o['rays']) and o['triggered'] == bool(trig[gi]), gi
            if not o['candidate']:
                assert not cand[mine].any()
                continue
            assert len(mine) == len(o['sub']), gi
            for e, q in zip(mine, o['sub']):
                assert cand[e] and n_rays[e] == len(q['rays']) and q['L'] == L[e] and abs(q['t_min'] - t_min[e]) < 1e-9
                assert q['triggered'] == bool(ev_trig[e])
                n_sub_trig += q['triggered']
                scale = np.max(np.abs(q['V']))
                for ch in range(n_ch):
                    it = pos[int(e)] * n_ch + ch
                    assert np.max(np.abs(tr[off[it]:off[it + 1]] - q['V'][ch])) <= 2e-6 * scale, (gi, e, ch)
            n_split += len(mine) > 1
        assert n_split >= 20 and n_sub_trig >= 8 and trig.sum() >= 5
        trig_p, _ = st.simulate_events(*args, askaryan_model='ARZ2020', arz_iN=iN, **kw)
        assert np.array_equal(trig_p, trig) and np.array_equal(st.fetch('ev_triggered'), ev_trig)
    else:
        pa_max = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
        want_group = np.zeros(n_groups, bool)
        for i, e in enumerate(item_event):
            V = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in pa_ch])
            t, mx = so.phased_array_trigger(V, rolls, window, step, threshold)
            assert np.max(np.abs(pa_max[i] - mx)) <= 1e-9 * max(np.max(mx), 1e-300), e
            if np.min(np.abs(mx - threshold)) > 1e-9 * threshold:
                assert t == bool(ev_trig[e]), e
            want_group[ev_group[e]] |= bool(ev_trig[e])
            n_sub_trig += t
        n_split = int(np.sum(np.bincount(ev_group[item_event], minlength=n_groups) > 1))
        assert np.array_equal(want_group, trig.astype(bool)) and not ev_trig[~cand.astype(bool)].any()
        assert n_split >= 20 and 5 <= n_sub_trig < len(item_event)
        trig_p, _ = st.simulate_events(*args, trigger='phased_array', trigger_threshold=threshold, **kw)
        assert np.array_equal(trig_p, trig)


@pytest.mark.parametrize('mode', ['noise', 'general', 'general+noise', 'general+noise+adc'])
def test_phased_array_with_noise_and_on_the_general_path(gpu_ctx_factory, mode):
    """What an RNO-G station triggers on (phasedArrayBase.py:370-496 after simulation.apply_det_response :594-606): the phased
    array on four deep dipoles WITH thermal noise, on the parametrised path and on the general path (ARZ2020 emission +
    birefringent propagation, BASELINE configs[3]).  Beam powers and decisions vs the oracle's array core applied to the
    oracle's (noisy) channel traces of the same rays; traces 1e-6 (3e-5 with birefringence) of the largest sample.  '+adc': the
    8-bit 472 MHz trigger ADC with 4 x FFT up-sampling in front of the beams (digitised traces, powers and decisions vs the
    oracle's ADC chain on the dumped traces: counts equal sample by sample)."""
    from nuradiomc_amd import arz as arz_mod
    from oracle import arz_oracle
    from test_oracle_golden import _arz_library
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -96.], [0., 0., -97.], [0., 0., -98.], [0., 0., -99.], [0., 0., -60.], [20., 15., -95.]])
    cable = np.array([1.2, 0., 2.6, 0.7, 0., 3.])
    N, fs = 512, 2.0
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, cable_delay=cable, n_samples=N, sampling_rate=fs)
    ost = so.Station(pos, cable_delay=cable, n_samples=N, fs=fs)
    vrms, vrms_e = so.vrms_from_filters(fs)
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 11))
    window, step = 32, 16
    adc = 'adc' in mode
    adc_fs, nbits, ncount, up = 0.472, 8, 5, 4
    if adc:
        window, step = 24, 8
        rolls = st.set_phased_array([0, 1, 2, 3], angles, ref_index=1.75, window=window, step=step, upsampling_factor=up,
                                    adc=dict(sampling_frequency=adc_fs, n_bits=nbits, noise_count=ncount, output='counts'))
        threshold = 2.5 * (2 * ncount) ** 2
    else:
        rolls = st.set_phased_array([0, 1, 2, 3], angles, ref_index=1.75, window=window, step=step)
        threshold = 2.5 * (2 * vrms) ** 2
    rng = np.random.default_rng(15)
    n = 70
    r, ph = np.sqrt(rng.uniform(0, 1100. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1300., -120., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.2, 18.5, n)
    types = np.array(['HAD'] * n)
    general, noisy = 'general' in mode, 'noise' in mode
    kw, model, bire, oarz, iN, tol = {}, 'Alvarez2009', None, None, np.zeros(n, int), 2e-6
    if general:
        b = golden('ref_birefringence.npz')
        tck = [(b['tck_southpole_A_%d_t' % j], b['tck_southpole_A_%d_c' % j]) for j in range(3)]
        st.set_birefringence(tck, angle_to_iceflow=25.)
        bire = (tck, 25.)
        lib = _arz_library(golden('ref_arz.npz'))
        a = arz_mod.ARZ(seed=3, library=lib)
        st.set_arz(a)
        iN = a.draw_profile_numbers(en, list(types))
        model, kw, tol = 'ARZ2020', dict(arz_iN=iN), 3e-5
        oarz = arz_oracle.ARZ(lib, seed=3)
    amp = None
    ids = 500 + 7 * np.arange(n)
    if noisy:
        amp = st.set_noise(300.)
        kw.update(noise=True, noise_seed=77, group_id=ids)
    trig, stats = st.simulate_events(v, zen, az, en, types, 1.0, askaryan_model=model, trigger='phased_array',
                                     trigger_threshold=threshold, dump_traces=True, **kw)
    T = {k: st.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                  'slot_refl_angle', 'ev_candidate', 'ev_L')}
    item_event = st.fetch('item_event')
    pa_max = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
    toff, trace = st.fetch('trace_offset'), st.fetch('trace')
    n_ch = len(pos)
    n_cand = n_trig = 0
    for ev in range(n):
        ps, ss = slice(ev * n_ch, (ev + 1) * n_ch), slice(ev * n_ch * 2, (ev + 1) * n_ch * 2)
        rays = dict(n_sol=T['pair_n_sol'][ps], type=T['slot_type'][ss].reshape(n_ch, 2), C0=T['slot_C0'][ss].reshape(n_ch, 2),
                    D=T['slot_D'][ss].reshape(n_ch, 2), T=T['slot_T'][ss].reshape(n_ch, 2),
                    refl_angle=T['slot_refl_angle'][ss].reshape(n_ch, 2),
                    launch=T['slot_launch'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3),
                    receive=T['slot_receive'][ev * n_ch * 6:(ev + 1) * n_ch * 6].reshape(n_ch, 2, 3))
        o = so.simulate_event(v[ev], zen[ev], az[ev], en[ev], 'HAD', 1.0, ost, ice, vrms, vrms_e, model=model, rays=rays,
                              arz=(oarz, int(iN[ev])) if oarz else None, birefringence=bire,
                              noise=(77, int(ids[ev]), 0, amp) if noisy else None)
        assert o['candidate'] == bool(T['ev_candidate'][ev]), ev
        if not o['candidate']:
            assert not trig[ev]
            continue
        n_cand += 1
        i = int(np.where(item_event == ev)[0][0])
        scale = np.max(np.abs(o['V']))
        for ch in range(4):
            tr = trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]]
            assert len(tr) == o['L'] and np.max(np.abs(tr - o['V'][ch])) <= tol * scale + (1e-9 * vrms if noisy else 0.), (ev, ch)
        if adc:   # the ADC chain on the traces the GPU dumped (a count flips with the last bit of a sample otherwise)
            Vg = np.array([trace[toff[i * n_ch + c]:toff[i * n_ch + c + 1]] for c in range(4)])
            U = np.array([so.digital_upsampling_fft(so.adc_digital_trace(x, fs, adc_fs, nbits, vrms, ncount, 'counts'), up) for x in Vg])
            p = so.phased_array_power_digital(U, rolls, window, step, 'counts')
            mx = p.max(axis=1)
            assert np.max(np.abs(pa_max[i] - mx)) <= 1e-9 * np.max(mx), ev
            t = bool(np.any(p > np.trunc(threshold)))
            assert t == bool(trig[ev]), ev
        else:
            t, mx = so.phased_array_trigger(o['V'][:4], rolls, window, step, threshold)
            assert np.max(np.abs(pa_max[i] - mx)) <= 10 * tol * np.max(mx), ev
            if np.min(np.abs(mx - threshold)) > 10 * tol * threshold:
                assert t == bool(trig[ev]), ev
        n_trig += t
    assert n_cand >= 8 and 2 <= n_trig
    # production mode (no dump): the same mask
    trig_p, _ = st.simulate_events(v, zen, az, en, types, 1.0, askaryan_model=model, trigger='phased_array',
                                   trigger_threshold=threshold, **kw)
    assert np.array_equal(trig_p, trig)


@pytest.mark.parametrize('kw', [dict(), dict(trigger='high_low', n_coincidences=2, coinc_window=80.)])
def test_trigger_channel_subset(gpu_ctx_factory, kw):
    """`triggered_channels` of the threshold triggers: decisions equal the oracle's trigger applied to those channels' traces
    only, in production mode (the other channels are not evaluated) and with dumped traces (they are, but cannot trigger)."""
    g = golden('chain_N256.npz')
    ice = g['ice']
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, g['det_pos'], n_samples=int(g['N']), sampling_rate=float(g['fs']))
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    n = 300
    sl = slice(0, n)
    args = (g['vertex'][sl], g['zenith'][sl], g['azimuth'][sl], g['energy'][sl] * 40., 'HAD')
    opts = dict(kw)
    if kw:
        opts.update(threshold_high=2.5 * st.vrms, threshold_low=-2.5 * st.vrms)
    all_ch, _ = st.simulate_events(*args, **opts)
    subset = [1, 3]
    st.set_trigger_channels(subset)
    prod, _ = st.simulate_events(*args, **opts)
    maxV = st.fetch('item_maxV').reshape(-1, 5)
    assert np.all(np.isnan(maxV[:, [0, 2, 4]]))
    dump, _ = st.simulate_events(*args, dump_traces=True, **opts)
    assert np.array_equal(prod, dump) and 3 <= prod.sum() <= all_ch.sum() and not np.any(prod & ~all_ch)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    for i, e in enumerate(item_event):
        V = np.array([tr[off[i * 5 + c]:off[i * 5 + c + 1]] for c in subset])
        if kw:
            t, _ = so.station_trigger(V, float(g['fs']), 'high_low', n_coincidences=2, threshold_high=2.5 * st.vrms,
                                      threshold_low=-2.5 * st.vrms, coinc_window=80.)
        else:
            t = so.threshold_trigger(V, 3.0 * st.vrms)
        assert t == bool(dump[e]), e
    st.set_trigger_channels(None)
    again, _ = st.simulate_events(*args, **opts)
    assert np.array_equal(again, all_ch)


def test_dumped_traces_of_channels_without_rays_are_zero(gpu_ctx_factory):
    """a channel no ray reaches (shadow zone) has an all-zero trace in the dump, also when the workspace held other traces
    before (regression: the dump buffer was not cleared and such channels kept stale samples)"""
    import bench
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    pos = np.array([[0., 0., -100.], [0., 0., -101.], [0., 0., -102.], [0., 0., -2.], [0., 0., -1.]])
    st = nuradiomc_amd.Station(ctx, pos, n_samples=512, sampling_rate=2.0)
    v, z, a = bench.make_events(6000, 2)
    en = np.full(6000, 3e18)
    st.simulate_events(v, z, a, en, 'HAD', dump_traces=True)                      # fills the workspace
    far = np.linalg.norm(v[:, :2], axis=1) > 1200.
    trig, stats = st.simulate_events(v[far], z[far], a[far], en[far], 'HAD', dump_traces=True)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    r0, nr, rch = st.fetch('ev_ray_begin'), st.fetch('ev_n_rays'), st.fetch('ray_channel')
    n_empty = 0
    for i, e in enumerate(item_event):
        have = set(rch[r0[e]:r0[e] + nr[e]])
        for c in range(5):
            t = tr[off[i * 5 + c]:off[i * 5 + c + 1]]
            if c not in have:
                assert not np.any(t), (e, c)
                n_empty += 1
            else:
                assert np.any(t)
    assert n_empty > 20


@pytest.mark.parametrize('name', ['N256', 'groups_N256'])
def test_two_phase_random_draws_like_the_reference(gpu_ctx_factory, name):
    """k_L of the electromagnetic showers is NOT given: Station.simulate_events(seed=...) traces the rays first, walks the
    showers in the reference's loop order (event group -> channel -> shower, simulation.py:1454-1600 / :143-242) drawing from
    RandomState(seed) exactly as parametrizations.py:160-173 does, then finishes on the same ray tables.  The fixtures hold the
    values the reference itself drew with config['seed'] = 1235 from a fresh generator: they must come out again, and the
    masks must equal those of a run that is handed the reference's k_L.  Without a seed missing values are an error."""
    g = golden('chain_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    groups = 'group' in g
    ref_kL = g['k_L'] if groups else g['ev_k_L']
    em = g['shower_type'] == 'EM'
    kw = dict(vertex_time=g['vertex_time'], group_id=g['group']) if groups else {}
    args = (g['vertex'], g['zenith'], g['azimuth'], g['energy'], g['shower_type'])
    with pytest.raises(ValueError):
        st.simulate_events(*args, None, **kw)
    trig, stats = st.simulate_events(*args, None, seed=1235, **kw)
    drawn = stats['k_L']
    cand = st.fetch('ev_candidate').copy()
    has_ref = em & np.isfinite(ref_kL)
    assert has_ref.sum() >= 15
    assert np.array_equal(np.isfinite(drawn) & em, has_ref)      # the same showers were met
    assert np.array_equal(drawn[has_ref], ref_kL[has_ref])        # and got the same numbers, bit for bit
    trig_ref, _ = st.simulate_events(*args, np.where(np.isnan(ref_kL), 1.0, ref_kL), **kw)
    assert np.array_equal(trig, trig_ref) and np.array_equal(cand, st.fetch('ev_candidate'))
    # chunked calls continue ONE random stream
    trig_c, stats_c = st.simulate_events(*args, None, seed=1235, max_showers_per_call=37, **kw)
    assert np.array_equal(trig_c, trig) and np.array_equal(stats_c['k_L'][has_ref], ref_kL[has_ref])


def _mb_run(gpu_ctx_factory, g, n, **kw):
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    refl = dict(n_reflections=int(g['n_reflections']), z_reflection=float(g['z_reflection']),
                reflection_coefficient=float(g['reflection_coefficient']), reflection_phase_shift=float(g['reflection_phase_shift']))
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    trig, stats = st.simulate_events(g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], g['energy'][:n], g['shower_type'][:n], kL,
                                     **refl, **kw)
    return ctx, st, refl, kL, trig, stats


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'chain_N256_mb.npz')), reason='fixture not generated')
@pytest.mark.parametrize('fixture,n,mins', [('N256_mb', 200, (400, 100, 10, 40)), ('N4096_mb', 60, (100, 25, 5, 10))])
def test_bottom_reflections_in_the_batched_path(gpu_ctx_factory, fixture, n, mins):
    """(N4096_mb: the same shelf with the headline's traces -- 4096 samples at 2 GHz, common traces of up to ~ 19 000 samples: the
    chirp-z channel kernel with the forward transform in output blocks and the inverse one in input chunks.)
    Moore's Bay: a reflective layer at -576 m (mooresbay_simple), MB1 attenuation, propagation.n_reflections = 1 inside
    nrhip_simulate_event_groups -- ray tables with 2 + 4 n slots per pair, attenuation as the product over the path segments,
    one Fresnel factor per surface reflection, coefficient and phase shift per bottom reflection (analyticraytracing.py:2118-2130,
    :933-1089, :2966-3009).  GPU vs the oracle on identical ray tables (1e-6; decisions exact), and vs the reference's own
    outputs wherever it found the same rays: its Python path loses most roots of rays that start downwards
    (reflection_case 2: get_delta_y shifts the start point in place), the golden table of its C++ twin has them."""
    g = golden('chain_%s.npz' % fixture)
    ctx, st, refl, kL, trig, stats = _mb_run(gpu_ctx_factory, g, n, no_pruning=True, dump_traces=True)
    S = 2 + 4 * refl['n_reflections']
    n_ch = len(g['det_pos'])
    T = {k: st.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                  'slot_refl_angle', 'slot_reflection', 'slot_reflection_case', 'slot_surface_mask', 'ray_event',
                                  'ray_channel', 'ray_solution', 'ray_t0', 'ray_r_theta', 'ray_r_phi', 'ray_att',
                                  'ray_max_efield', 'ev_n_rays', 'ev_L', 'ev_candidate', 'ev_t_min', 'ev_ray_begin')}
    assert len(T['slot_C0']) == n * n_ch * S and T['slot_reflection'].max() == 1
    item_event = st.fetch('item_event')
    toff, trace = st.fetch('trace_offset'), st.fetch('trace')
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    orefl = (refl['n_reflections'], refl['z_reflection'], refl['reflection_coefficient'], refl['reflection_phase_shift'])
    att = T['ray_att'].reshape(-1, len(st.att_freq))
    n_rays = n_refl_rays = n_cand = 0
    for ev in range(n):
        sl = slice(ev * n_ch * S, (ev + 1) * n_ch * S)
        rays = dict(n_sol=T['pair_n_sol'][ev * n_ch:(ev + 1) * n_ch])
        for k, name in (('type', 'slot_type'), ('C0', 'slot_C0'), ('D', 'slot_D'), ('T', 'slot_T'), ('refl_angle', 'slot_refl_angle'),
                        ('reflection', 'slot_reflection'), ('reflection_case', 'slot_reflection_case'),
                        ('surface_mask', 'slot_surface_mask')):
            rays[k] = T[name][sl].reshape(n_ch, S)
        for k, name in (('launch', 'slot_launch'), ('receive', 'slot_receive')):
            rays[k] = T[name][3 * sl.start:3 * sl.stop].reshape(n_ch, S, 3)
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev], str(g['shower_type'][ev]),
                              float(kL[ev]), ost, g['ice'], st.vrms, st.vrms_efield, att_model=str(g['att_model']), rays=rays,
                              reflections=orefl)
        r0 = T['ev_ray_begin'][ev]
        sel = np.arange(r0, r0 + T['ev_n_rays'][ev])
        assert [(r['channel'], r['iS']) for r in o['rays']] == list(zip(T['ray_channel'][sel], T['ray_solution'][sel])), ev
        for r, k in zip(o['rays'], sel):
            assert abs(r['t0'] - T['ray_t0'][k]) < 1e-9
            assert abs(r['r_theta'] - T['ray_r_theta'][k]) < 1e-12 and abs(r['r_phi'] - T['ray_r_phi'][k]) < 1e-12
            a_ref = rto.attenuation_batch_refl(g['vertex'][ev][None], g['det_pos'][r['channel']][None], [r['C0']],
                                               [rays['reflection'][r['channel'], r['iS']]],
                                               [rays['reflection_case'][r['channel'], r['iS']]], g['ice'], refl['z_reflection'],
                                               str(g['att_model']), st.att_freq)[0]
            assert np.max(np.abs(att[k] - a_ref) / a_ref) < 1e-6
            assert abs(r['max_efield'] - T['ray_max_efield'][k]) <= 1e-6 * r['max_efield']
            n_rays += 1
            n_refl_rays += rays['reflection'][r['channel'], r['iS']] > 0
        assert bool(T['ev_candidate'][ev]) == o['candidate'] and bool(trig[ev]) == o['triggered'], ev
        if o['candidate']:
            n_cand += 1
            assert T['ev_L'][ev] == o['L'] and abs(T['ev_t_min'][ev] - o['t_min']) < 1e-9
            i = int(np.where(item_event == ev)[0][0])
            scale = np.max(np.abs(o['V']))
            for ch in range(n_ch):
                assert np.max(np.abs(trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]] - o['V'][ch])) <= 1e-6 * scale, (ev, ch)
    assert n_rays > mins[0] and n_refl_rays > mins[1] and n_cand >= mins[2]
    print('%s: longest common trace %d samples' % (fixture, int(T['ev_L'].max())))
    # the reference itself, where it found the same number of rays
    same = T['ev_n_rays'] == g['ev_n_rays'][:n]
    cand = T['ev_candidate'].astype(bool)
    assert same.sum() >= mins[3]
    assert np.array_equal(cand[same], g['ev_candidate'][:n][same]) and np.array_equal(trig[same], g['ev_triggered'][:n][same])
    both = same & cand
    assert np.array_equal(T['ev_L'][both], g['ev_L'][:n][both])
    maxV = st.fetch('item_maxV').reshape(len(item_event), -1)
    worst = 0.
    for i, ev in enumerate(item_event):
        if both[ev]:
            worst = max(worst, float(np.max(np.abs(maxV[i] - g['ev_maxV'][ev])) / np.max(g['ev_maxV'][ev])))
    print('Moore\'s Bay fixture, channel maxima vs reference: max |dV| / max V = %.2e' % worst)
    assert worst <= 1.5e-3
    # production mode: the same masks; without reflections the direct / refracted / surface-reflected rays only
    ctx, st, refl, kL, trig_p, stats_p = _mb_run(gpu_ctx_factory, g, n)
    assert np.array_equal(trig_p, trig) and np.array_equal(st.fetch('ev_candidate'), T['ev_candidate'])
    assert stats_p['n_rays'] == stats['n_rays'] and stats_p['n_active_rays'] <= stats['n_active_rays']


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'chain_N256_mb_focus.npz')), reason='fixture not generated')
def test_bottom_reflections_with_focusing(gpu_ctx_factory):
    """propagation.focusing on a shelf with a reflective bottom (n_reflections = 1): the second trace of get_focusing lists the
    bottom-reflected solutions as the first one does (the reference builds that tracer with the same n_reflections,
    analyticraytracing.py:2835-2840), and solution iS of the one is compared with solution iS of the other.  GPU vs the oracle on
    identical ray tables (per-ray maxima, candidate flags, traces 1e-6, decisions exact); the factor changes the amplitudes of the
    bottom-reflected rays too; and vs the reference's own outputs where it found the same rays (its Python path loses roots of rays
    starting downwards in BOTH traces, see test_bottom_reflections_in_the_batched_path: per-ray maxima where the solution lists of
    both its traces are complete, i.e. equal ours)."""
    g = golden('chain_N256_mb_focus.npz')
    n = 140
    assert bool(g['focusing'])
    ctx, st, refl, kL, trig, stats = _mb_run(gpu_ctx_factory, g, n, no_pruning=True, dump_traces=True, focusing=True, focusing_limit=2.)
    S = 2 + 4 * refl['n_reflections']
    n_ch = len(g['det_pos'])
    T = {k: st.fetch(k).copy() for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_D', 'slot_T', 'slot_launch', 'slot_receive',
                                         'slot_refl_angle', 'slot_reflection', 'slot_reflection_case', 'slot_surface_mask', 'ray_event',
                                         'ray_channel', 'ray_solution', 'ray_max_efield', 'ev_n_rays', 'ev_L', 'ev_candidate',
                                         'ev_ray_begin')}
    item_event = st.fetch('item_event').copy()
    toff, trace = st.fetch('trace_offset').copy(), st.fetch('trace').copy()
    maxV = st.fetch('item_maxV').reshape(len(item_event), -1).copy()
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    orefl = (refl['n_reflections'], refl['z_reflection'], refl['reflection_coefficient'], refl['reflection_phase_shift'])
    n_rays = n_refl_rays = n_cand = 0
    for ev in range(n):
        sl = slice(ev * n_ch * S, (ev + 1) * n_ch * S)
        rays = dict(n_sol=T['pair_n_sol'][ev * n_ch:(ev + 1) * n_ch])
        for k, name in (('type', 'slot_type'), ('C0', 'slot_C0'), ('D', 'slot_D'), ('T', 'slot_T'), ('refl_angle', 'slot_refl_angle'),
                        ('reflection', 'slot_reflection'), ('reflection_case', 'slot_reflection_case'),
                        ('surface_mask', 'slot_surface_mask')):
            rays[k] = T[name][sl].reshape(n_ch, S)
        for k, name in (('launch', 'slot_launch'), ('receive', 'slot_receive')):
            rays[k] = T[name][3 * sl.start:3 * sl.stop].reshape(n_ch, S, 3)
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev], str(g['shower_type'][ev]),
                              float(kL[ev]), ost, g['ice'], st.vrms, st.vrms_efield, att_model=str(g['att_model']), rays=rays,
                              reflections=orefl, focusing=True, focusing_limit=2.)
        r0 = T['ev_ray_begin'][ev]
        sel = np.arange(r0, r0 + T['ev_n_rays'][ev])
        assert [(r['channel'], r['iS']) for r in o['rays']] == list(zip(T['ray_channel'][sel], T['ray_solution'][sel])), ev
        for r, k in zip(o['rays'], sel):
            assert abs(r['max_efield'] - T['ray_max_efield'][k]) <= 1e-6 * r['max_efield'], (ev, r['channel'], r['iS'])
            n_rays += 1
            n_refl_rays += rays['reflection'][r['channel'], r['iS']] > 0
        assert bool(T['ev_candidate'][ev]) == o['candidate'] and bool(trig[ev]) == o['triggered'], ev
        if o['candidate']:
            n_cand += 1
            assert T['ev_L'][ev] == o['L']
            i = int(np.where(item_event == ev)[0][0])
            scale = np.max(np.abs(o['V']))
            for ch in range(n_ch):
                assert np.max(np.abs(trace[toff[i * n_ch + ch]:toff[i * n_ch + ch + 1]] - o['V'][ch])) <= 1e-6 * scale, (ev, ch)
    assert n_rays > 250 and n_refl_rays > 60 and n_cand >= 8
    # the factor is at work on direct and on bottom-reflected rays alike
    ctx, st0, _, _, trig0, _ = _mb_run(gpu_ctx_factory, g, n, no_pruning=True)
    mx0 = st0.fetch('ray_max_efield')
    assert len(mx0) == len(T['ray_max_efield'])
    ratio = T['ray_max_efield'] / mx0
    is_refl = T['slot_reflection'][(T['ray_event'] * n_ch + T['ray_channel']) * S + T['ray_solution']] > 0
    assert np.all(ratio <= 2.0 * 1.2)
    assert np.mean(np.abs(ratio[is_refl] - 1) > 1e-3) > 0.9 and np.mean(np.abs(ratio[~is_refl] - 1) > 1e-3) > 0.9
    # the reference, where it found the same rays in the event (then both its traces are most likely complete): 90 % of those rays
    # within 1e-3 (first-root noise of its finder at the 1e-2 level in the finite difference, as in test_focusing_chain_vs_reference)
    same = T['ev_n_rays'] == g['ev_n_rays'][:n]
    rel = []
    for ev in np.flatnonzero(same):
        ref = g['ray_max_efield'][g['ray_event'] == ev]
        mine = T['ray_max_efield'][T['ev_ray_begin'][ev]:T['ev_ray_begin'][ev] + T['ev_n_rays'][ev]]
        rel += list(np.abs(mine - ref) / ref)
    rel = np.array(rel)
    print('Moore\'s Bay + focusing vs reference: %d events with the same rays, %d rays, %.0f %% within 1e-3, max %.2e'
          % (same.sum(), len(rel), 100 * np.mean(rel < 1e-3), rel.max() if len(rel) else 0))
    assert same.sum() >= 25 and len(rel) > 40 and np.mean(rel < 1e-3) > 0.8
    # production mode: the same masks
    ctx, st, refl, kL, trig_p, stats_p = _mb_run(gpu_ctx_factory, g, n, focusing=True, focusing_limit=2.)
    assert np.array_equal(trig_p, trig)


def test_split_event_time_diff(gpu_ctx_factory):
    """nrhip_sim_config.split_event_time_diff = simulation.group_into_events (:906-947): groups whose signals are farther apart
    than the limit are cut into sub-events, each with its own readout window, channel sums and trigger; the candidate cut stays
    per group.  GPU vs the oracle on the same rays (sub-event membership exact, traces 1e-6) and vs the reference's own sub-events
    (tests/golden/chain_split_N256.npz, 43 of 54 candidate groups split) wherever its first-root noise kept the ray count."""
    g = golden('chain_split_N256.npz')
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(ost.fs)
    split = float(g['split_event_time_diff'])
    kL = np.where(np.isnan(g['k_L']), 50.0, g['k_L'])
    args = (g['vertex'], g['zenith'], g['azimuth'], g['energy'], g['shower_type'], kL)
    trig, stats = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'], dump_traces=True,
                                     split_event_time_diff=split)
    n_groups = len(g['ev_candidate'])
    assert trig.shape == (n_groups,)
    ev_group, ev_sub = st.fetch('ev_group'), st.fetch('ev_sub_event')
    cand, L, t_min, n_rays, ev_trig = (st.fetch(k) for k in ('ev_candidate', 'ev_L', 'ev_t_min', 'ev_n_rays', 'ev_triggered'))
    assert stats['n_sub_events'] == len(ev_group) > n_groups and np.all(np.diff(ev_group) >= 0)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    n_ch = len(g['det_pos'])
    pos = {int(e): i for i, e in enumerate(item_event)}
    n_split = n_ref = 0
    for gi in range(n_groups):
        idx = np.flatnonzero(g['group'] == gi)
        showers = [dict(vertex=g['vertex'][i], zenith=float(g['zenith'][i]), azimuth=float(g['azimuth'][i]),
                        energy=float(g['energy'][i]), shower_type=str(g['shower_type'][i]), k_L=float(kL[i]),
                        vertex_time=float(g['vertex_time'][i])) for i in idx]
        o = so.simulate_event_group(showers, ost, g['ice'], vrms, vrms_e, split_event_time_diff=split)
        mine = np.flatnonzero(ev_group == gi)
        assert list(ev_sub[mine]) == list(range(len(mine)))
        assert n_rays[mine].sum() == len(o['rays']) and o['triggered'] == bool(trig[gi]), gi
        if not o['candidate']:
            assert not cand[mine].any()
            continue
        assert len(mine) == len(o['sub']), gi
        for e, q in zip(mine, o['sub']):
            assert cand[e] and n_rays[e] == len(q['rays']) and q['L'] == L[e] and abs(q['t_min'] - t_min[e]) < 1e-9
            assert q['triggered'] == bool(ev_trig[e])
            scale = np.max(np.abs(q['V']))
            for ch in range(n_ch):
                it = pos[int(e)] * n_ch + ch
                assert np.max(np.abs(tr[off[it]:off[it + 1]] - q['V'][ch])) <= 1e-6 * scale, (gi, e, ch)
        n_split += len(mine) > 1
        if len(o['rays']) == g['ev_n_rays'][gi]:   # the reference itself
            rows = np.flatnonzero(g['sub_group'] == gi)
            assert len(rows) == len(mine) and bool(trig[gi]) == bool(g['ev_triggered'][gi])
            assert np.array_equal(L[mine], g['sub_L'][rows]) and np.array_equal(ev_trig[mine].astype(bool), g['sub_triggered'][rows])
            n_ref += 1
    assert n_split >= 30 and n_ref >= 40 and trig.sum() >= 15
    # production mode: same masks
    trig_p, stats_p = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'], split_event_time_diff=split)
    assert np.array_equal(trig_p, trig) and np.array_equal(st.fetch('ev_triggered'), ev_trig) and stats_p['n_sub_events'] == len(ev_group)
    # a limit nothing exceeds: one readout per group, identical to the call without it
    t1, s1 = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'], split_event_time_diff=1e6)
    L1 = st.fetch('ev_L')
    t0_, s0 = st.simulate_events(*args, vertex_time=g['vertex_time'], group_id=g['group'])
    assert s1['n_sub_events'] == n_groups and np.array_equal(t1, t0_) and np.array_equal(L1, st.fetch('ev_L'))


def _check_trace_triggers(st, trig, okw, n_events):
    """the trigger mask and first bins against the reference's trigger logic (oracle restatement) on the traces the kernels dumped"""
    item_event, tr, off, tbin = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset'), st.fetch('ev_trigger_bin')
    n_ch = len(st.position)
    expect = np.zeros(n_events, bool)
    Lmax = 0
    for i, e in enumerate(item_event):
        V = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in range(n_ch)])
        Lmax = max(Lmax, V.shape[1])
        t, bins = so.station_trigger(V, st.sampling_rate, **okw)
        expect[e] = t
        assert tbin[e] == (bins[0] if t else -1), e
    assert np.array_equal(trig, expect)
    return Lmax


@pytest.mark.parametrize('kw', [dict(trigger='high_low', n_coincidences=2, hi=2.0, lo=-2.0, high_low_window=5., coinc_window=30.),
                                dict(trigger='simple', n_coincidences=3, thr=2.0, coinc_window=40.)])
def test_coincidence_triggers_beyond_the_fused_kernel(gpu_ctx_factory, kw):
    """High/low and n-fold coincidence triggers where channel_conv_kernel does not run: common traces longer than 8192 samples
    (Moore's Bay with bottom reflections: up to 15 046) and tabulated antenna patterns -- the channel stage dumps the traces,
    trace_trigger_kernel decides on them.  Mask and first triggered bin = the reference's trigger logic on those traces;
    production mode gives the same mask."""
    def opts_for(st):
        vr = st.vrms
        opts = dict(trigger=kw['trigger'], n_coincidences=kw['n_coincidences'], coinc_window=kw['coinc_window'])
        okw = dict(opts)
        if kw['trigger'] == 'high_low':
            opts.update(threshold_high=kw['hi'] * vr, threshold_low=kw['lo'] * vr, high_low_window=kw['high_low_window'])
            okw.update(threshold_high=kw['hi'] * vr, threshold_low=kw['lo'] * vr, high_low_window=kw['high_low_window'])
        else:
            opts.update(trigger_threshold=kw['thr'] * vr)
            okw.update(threshold=kw['thr'] * vr)
        return opts, okw
    # long traces
    g = golden('chain_N256_mb.npz')
    n = 200
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    opts, okw = opts_for(st)
    refl = dict(n_reflections=int(g['n_reflections']), z_reflection=float(g['z_reflection']),
                reflection_coefficient=float(g['reflection_coefficient']), reflection_phase_shift=float(g['reflection_phase_shift']))
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    args = (g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], g['energy'][:n], g['shower_type'][:n], kL)
    trig, stats = st.simulate_events(*args, dump_traces=True, **refl, **opts)
    assert _check_trace_triggers(st, trig, okw, n) > 8192 and 3 <= trig.sum() < stats['n_candidate_events']
    trig_p, _ = st.simulate_events(*args, **refl, **opts)
    assert np.array_equal(trig_p, trig)
    # tabulated antenna pattern
    g = golden('chain_N256_tab.npz')
    n = len(g['vertex'])
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    opts, okw = opts_for(st)
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    args = (g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], 3 * g['energy'][:n], g['shower_type'][:n], kL)
    trig, stats = st.simulate_events(*args, dump_traces=True, **opts)
    _check_trace_triggers(st, trig, okw, n)
    assert 1 <= trig.sum() < stats['n_candidate_events']
    trig_p, _ = st.simulate_events(*args, **opts)
    assert np.array_equal(trig_p, trig)


def test_envelope_trigger(gpu_ctx_factory):
    """trigger='envelope' (envelopeTrigger.py): every channel trace through the trigger's Butterworth band pass, Hilbert envelope
    above the threshold, majority logic.  GPU vs the oracle on the traces the kernels dumped (envelopes 1e-9, masks and first bins
    exact), and vs the reference's own decisions (tests/golden/chain_envelope_N256.npz) where the ray counts agree; production mode
    gives the same mask."""
    g = golden('chain_envelope_N256.npz')
    n = len(g['vertex'])
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    args = (g['vertex'], g['zenith'], g['azimuth'], g['energy'], g['shower_type'], np.ones(n))
    for i in range(2):
        pb, order = g['s%d_passband' % i], int(g['s%d_order' % i])
        st.set_envelope_trigger(pb, order)
        opts = dict(trigger='envelope', trigger_threshold=float(g['s%d_threshold' % i]), n_coincidences=int(g['s%d_n_coincidences' % i]),
                    coinc_window=float(g['s%d_coinc_window' % i]))
        okw = dict(trigger='envelope', threshold=opts['trigger_threshold'], n_coincidences=opts['n_coincidences'],
                   coinc_window=opts['coinc_window'], passband=pb, order=order)
        trig, stats = st.simulate_events(*args, dump_traces=True, **opts)
        _check_trace_triggers(st, trig, okw, n)
        item_event, tr, env, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('envelope_trace'), st.fetch('trace_offset')
        for k in range(0, len(item_event) * 5, 7):
            ref = so.envelope_of_filtered(tr[off[k]:off[k + 1]], st.sampling_rate, pb, order)
            assert np.max(np.abs(env[off[k]:off[k + 1]] - ref)) <= 1e-9 * max(np.max(ref), 1e-30), k
        same = st.fetch('ev_n_rays') == g['ev_n_rays']
        assert same.mean() > 0.97 and np.array_equal(trig[same], g['s%d_triggered' % i][same]) and trig.sum() >= 25
        trig_p, _ = st.simulate_events(*args, **opts)
        assert np.array_equal(trig_p, trig)
    st.set_envelope_trigger(None)
    with pytest.raises(Exception, match='envelope'):
        st.simulate_events(*args, trigger='envelope')
    # with tabulated antenna patterns (the chirp-z channel kernel forms the envelopes of whatever spectrum it summed)
    g2 = golden('chain_N256_tab.npz')
    st2 = _station(gpu_ctx_factory(g2['ice'], str(g2['att_model'])), g2)
    n2 = 160
    kL2 = np.where(np.isnan(g2['ev_k_L'][:n2]), 1.0, g2['ev_k_L'][:n2])
    args2 = (g2['vertex'][:n2], g2['zenith'][:n2], g2['azimuth'][:n2], g2['energy'][:n2], g2['shower_type'][:n2], kL2)
    pb, order = g['s0_passband'], int(g['s0_order'])
    st2.set_envelope_trigger(pb, order)
    opts = dict(trigger='envelope', trigger_threshold=2.0 * st2.vrms, n_coincidences=2, coinc_window=float(g['s0_coinc_window']))
    okw = dict(trigger='envelope', threshold=opts['trigger_threshold'], n_coincidences=2, coinc_window=opts['coinc_window'], passband=pb,
               order=order)
    trig2, stats2 = st2.simulate_events(*args2, dump_traces=True, **opts)
    _check_trace_triggers(st2, trig2, okw, n2)
    item_event, tr, env, off = st2.fetch('item_event'), st2.fetch('trace'), st2.fetch('envelope_trace'), st2.fetch('trace_offset')
    n_ch2 = len(g2['det_pos'])
    for k in range(0, len(item_event) * n_ch2, 5):
        ref = so.envelope_of_filtered(tr[off[k]:off[k + 1]], st2.sampling_rate, pb, order)
        assert np.max(np.abs(env[off[k]:off[k + 1]] - ref)) <= 1e-9 * max(np.max(ref), 1e-30), k
    assert len(item_event) >= 10
    trig2_p, _ = st2.simulate_events(*args2, **opts)
    assert np.array_equal(trig2_p, trig2)


def test_thermal_noise(gpu_ctx_factory):
    """noise=True (channelGenericNoiseAdder as simulation.apply_det_response calls it): Rayleigh / uniform-phase noise on every
    channel spectrum of the candidate events before filters and trigger, from the counter-based generator of csrc/noise.h.  GPU
    traces = the oracle's with the same generator (1e-9 of the noise RMS), triggers exact; the noise of an event does not depend
    on how the list is cut into calls; its RMS after the filters is the station's Vrms; noiseless channels stay noiseless."""
    g = golden('chain_N256.npz')
    n = 240
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    st = _station(ctx, g)
    amp = st.set_noise(300., noiseless_channels=[4])
    ost = so.Station(g['det_pos'], n_samples=int(g['N']), fs=float(g['fs']))
    assert abs(amp[0] - so.noise_amplitude(ost.fs)) < 1e-12 * amp[0] and amp[4] == 0.
    kL = np.where(np.isnan(g['ev_k_L'][:n]), 1.0, g['ev_k_L'][:n])
    args = (g['vertex'][:n], g['zenith'][:n], g['azimuth'][:n], g['energy'][:n], g['shower_type'][:n], kL)
    ids = 1000 + 3 * np.arange(n)
    trig, stats = st.simulate_events(*args, group_id=ids, noise=True, noise_seed=2024, dump_traces=True)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    cand = st.fetch('ev_candidate').astype(bool)
    n_ch = len(g['det_pos'])
    rms = []
    for i, ev in enumerate(item_event):
        rays, sel = None, None
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev], str(g['shower_type'][ev]),
                              float(kL[ev]), ost, g['ice'], st.vrms, st.vrms_efield, noise=(2024, int(ids[ev]), 0, amp))
        assert o['candidate'] and o['triggered'] == bool(trig[ev]), ev
        for ch in range(n_ch):
            v = tr[off[i * n_ch + ch]:off[i * n_ch + ch + 1]]
            assert np.max(np.abs(v - o['V'][ch])) <= 1e-9 * st.vrms + 1e-6 * np.max(np.abs(o['V'][ch])), (ev, ch)
        rms.append(np.sqrt(np.mean(tr[off[i * n_ch + 3]:off[i * n_ch + 4]] ** 2)))
    assert len(item_event) >= 15 and not trig[~cand].any()
    # a channel that hardly sees the pulses: its RMS is the noise RMS the trigger thresholds are quoted in
    quiet = np.array(rms)[np.array(rms) < 2 * st.vrms]
    assert len(quiet) >= 5 and abs(np.median(quiet) / st.vrms - 1) < 0.1
    # chunked calls: the same noise, hence the same mask and traces
    trig_c, _ = st.simulate_events(*args, group_id=ids, noise=True, noise_seed=2024, max_showers_per_call=37)
    assert np.array_equal(trig_c, trig)
    st.simulate_events(*args, group_id=ids, noise=True, noise_seed=2025, dump_traces=True)      # another seed: other noise
    tr2 = st.fetch('trace')
    assert tr2.shape == tr.shape and abs(np.corrcoef(tr[off[3]:off[4]], tr2[off[3]:off[4]])[0, 1]) < 0.2
    # noise raises the trigger rate of the candidates (3 sigma on 5 channels x ~1500 samples)
    trig_q, _ = st.simulate_events(*args, group_id=ids)
    assert trig.sum() > trig_q.sum()
    st.set_noise(None)
    with pytest.raises(Exception, match='noise'):
        st.simulate_events(*args, noise=True)


@pytest.mark.parametrize('output,up,clk', [('counts', 4, 0), ('voltage', 2, 0), ('counts', 1, 0), ('counts', 4, 3), ('voltage', 2, 1),
                                           ('counts', 1, 12)])
def test_phased_array_with_trigger_adc(gpu_ctx_factory, output, up, clk):
    """phasedArrayTrigger with apply_digitization and FFT up-sampling inside simulate_events: per array channel of every candidate
    event the trigger ADC (5 GHz resampling, linear down-sampling to 472 MHz, floor comparator) and the up-sampling, then beams with
    saturation and rounded window powers.  GPU vs the oracle's restatement (pinned sample by sample on the reference's own functions,
    test_phased_array_adc_vs_reference) applied to the channel traces the GPU dumped: digitised traces equal (counts: every sample;
    volts: 1e-9 lsb), per-beam maximum powers and decisions equal.  clk: the trigger modules' clock_offset (the traces delayed by
    whole ADC clock cycles in front of the digitiser; oracle pinned by test_trigger_adc_clock_offset_vs_reference)."""
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -96.], [0., 0., -97.], [0., 0., -98.], [0., 0., -99.], [0., 0., -60.], [20., 15., -95.]])
    cable = np.array([1.2, 0., 2.6, 0.7, 0., 3.])
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, cable_delay=cable, n_samples=512, sampling_rate=2.0)
    vrms = st.vrms
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), 11))
    window, step, adc_fs, nbits, ncount = 24, 8, 0.472, 8, 5
    rolls = st.set_phased_array([0, 1, 2, 3], angles, ref_index=1.75, window=window, step=step, upsampling_factor=up,
                                adc=dict(sampling_frequency=adc_fs, n_bits=nbits, noise_count=ncount, output=output, clock_offset=clk))
    assert np.array_equal(rolls, so.phased_array_rolls(pos[:4, 2], cable[:4], angles, adc_fs * up, 1.75))
    lsb = vrms / ncount
    threshold = 2.5 * (2 * (vrms / lsb if output == 'counts' else vrms)) ** 2
    rng = np.random.default_rng(14)
    n = 120
    r, ph = np.sqrt(rng.uniform(0, 1500. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    v = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500., -10., n)], axis=1)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(16.8, 18.2, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold, dump_traces=True)
    item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
    dig, dlen = st.fetch('pa_digital_trace'), st.fetch('pa_digital_length').reshape(len(item_event), 4)
    stride = len(dig) // (len(item_event) * 4)
    dig = dig.reshape(len(item_event), 4, stride)
    pa_max = st.fetch('pa_max_power').reshape(len(item_event), len(angles))
    n_ch = len(pos)
    n_trig = 0
    for i, e in enumerate(item_event):
        V = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in range(4)])
        U = np.array([so.digital_upsampling_fft(so.adc_digital_trace(x, 2.0, adc_fs, nbits, vrms, ncount, output, clock_offset=clk), up)
                      for x in V])
        assert np.all(dlen[i] == U.shape[1])
        got = dig[i, :, :U.shape[1]]
        assert np.max(np.abs(got - U)) <= (0 if output == 'counts' else 1e-9 * lsb), e
        p = so.phased_array_power_digital(U, rolls, window, step, output)
        mx = p.max(axis=1)
        assert np.max(np.abs(pa_max[i] - mx)) <= 1e-9 * np.max(mx), e
        t = bool(np.any(p > (np.trunc(threshold) if output == 'counts' else threshold)))
        assert t == bool(trig[e]), e
        n_trig += t
    assert len(item_event) >= 15 and 2 <= n_trig < len(item_event)
    trig_p, _ = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold)
    assert np.array_equal(trig_p, trig)
    # the direct-sum digitiser (what traces beyond the chirp-z version's 7169 samples or without the 5 GHz step go through)
    import os
    if clk:   # (the clock offset is part of the chirp-z digitiser only: refused there, by name)
        os.environ['NRHIP_PA_DIRECT'] = '1'
        try:
            with pytest.raises(Exception, match='clock offset'):
                st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold)
        finally:
            del os.environ['NRHIP_PA_DIRECT']
        with pytest.raises(ValueError, match='integer number of clock cycles'):
            st.set_phased_array([0, 1, 2, 3], angles, adc=dict(sampling_frequency=adc_fs, n_bits=nbits, noise_count=ncount, clock_offset=1.5))
        with pytest.raises(Exception, match='negative'):
            st.set_phased_array([0, 1, 2, 3], angles, adc=dict(sampling_frequency=adc_fs, n_bits=nbits, noise_count=ncount, clock_offset=-2))
        return
    os.environ['NRHIP_PA_DIRECT'] = '1'
    try:
        trig_d, _ = st.simulate_events(v, zen, az, en, 'HAD', trigger='phased_array', trigger_threshold=threshold, dump_traces=True)
        dig_d = st.fetch('pa_digital_trace').reshape(len(item_event), 4, stride)
    finally:
        del os.environ['NRHIP_PA_DIRECT']
    assert np.array_equal(trig_d, trig)
    for i in range(len(item_event)):
        n_up = dlen[i, 0]
        assert np.max(np.abs(dig_d[i, :, :n_up] - dig[i, :, :n_up])) <= (0 if output == 'counts' else 1e-9 * lsb)


def test_extreme_geometries_through_the_whole_path(gpu_ctx_factory):
    """Vertices the surveys rarely draw -- within metres of the antennas, above them, just under the surface, exactly above /
    below a channel (the vertical ray: end points above each other go through the reference's procedure), 5 km away, very deep --
    through the whole batched path against the oracle's chain: ray counts, candidate flags, trigger decisions and trace lengths
    equal on every event."""
    ice = (1.78, 0.423, 77.)
    pos = np.array([[0., 0., -100. - 2. * i] for i in range(5)])
    ctx = gpu_ctx_factory(ice, 'SP1')
    st = nuradiomc_amd.Station(ctx, pos, n_samples=256, sampling_rate=2.0)
    ost = so.Station(pos, n_samples=256, fs=2.0)
    vrms, vrms_e = so.vrms_from_filters(2.0)
    rng = np.random.default_rng(41)
    blocks = []
    m = 60
    u = lambda a, b: rng.uniform(a, b, m)   # noqa: E731
    blocks.append(np.stack([u(-8, 8), u(-8, 8), u(-125, -90)], 1))            # inside / next to the string
    blocks.append(np.stack([u(-300, 300), u(-300, 300), u(-60, -0.2)], 1))    # above the antennas, up to the surface
    blocks.append(np.stack([np.zeros(m), np.zeros(m), u(-2700, -110)], 1))    # exactly below the string: vertical rays
    blocks.append(np.stack([np.zeros(m), np.zeros(m), u(-95, -0.5)], 1))      # exactly above it
    blocks.append(np.stack([u(4000, 5000), u(-500, 500), u(-2700, -5)], 1))   # far away
    blocks.append(np.stack([u(-1500, 1500), u(-1500, 1500), u(-2700, -2400)], 1))   # very deep
    v = np.concatenate(blocks)
    n = len(v)
    zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    en = 10 ** rng.uniform(17.5, 19.5, n)
    trig, stats = st.simulate_events(v, zen, az, en, 'HAD')
    n_rays, cand, L = st.fetch('ev_n_rays')[:n], st.fetch('ev_candidate')[:n].astype(bool), st.fetch('ev_L')[:n]
    n_c = n_t = 0
    for i in range(n):
        o = so.simulate_event(v[i], zen[i], az[i], en[i], 'HAD', None, ost, ice, vrms, vrms_e)
        assert len(o['rays']) == n_rays[i], (i, v[i])
        assert o['candidate'] == bool(cand[i]) and o['triggered'] == bool(trig[i]), (i, v[i])
        if o['candidate']:
            assert o['L'] == L[i], (i, v[i])
        n_c += o['candidate']
        n_t += o['triggered']
    assert n_c >= 40 and n_t >= 10 and n_rays[2 * m:4 * m].sum() > 100   # (the vertical pairs do have rays)
    print('extreme geometries: %d events, %d rays, %d candidates, %d triggers' % (n, n_rays.sum(), n_c, n_t))
