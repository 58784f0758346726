"""The host-side mirrors of the reference's plugin interfaces, written like the reference's own tests."""
import numpy as np
import pytest
from conftest import golden, max_rel
from oracle import raytrace_oracle as orc

pytestmark = pytest.mark.gpu


class _Ice:
    def __init__(self, n_ice, delta_n, z_0):
        self.n_ice, self.delta_n, self.z_0 = n_ice, delta_n, z_0
        self.reflection = None


def test_askaryan_unit_test_like_U01():
    """NuRadioMC/test/SignalGen/U01unit_test.py: get_time_trace for 2 models x 5 energies x {EM, HAD} x 10 angles
    against NuRadioMC/test/SignalGen/reference_v2.npy (assert_almost_equal, 7 decimals), including the EM k_L random
    stream (seed 1234, one RandomState per model)."""
    from nuradiomc_amd import askaryan
    askaryan._random_generators.clear()
    g = golden('ref_askaryan_v2.npz')
    for i in range(len(g['model'])):
        trace = askaryan.get_time_trace(float(g['energy'][i]), float(g['theta'][i]), int(g['N']), float(g['dt']),
                                        str(g['shower_type'][i]), float(g['n_index']), float(g['R']), str(g['model'][i]),
                                        seed=int(g['seed']))
        np.testing.assert_almost_equal(trace, g['trace'][i], decimal=7)
        ref = g['trace'][i]
        assert np.max(np.abs(trace - ref)) <= 1e-9 * max(np.max(np.abs(ref)), 1e-300)
    with pytest.raises(NotImplementedError):
        askaryan.get_frequency_spectrum(1e18, 1., 256, 0.5, 'HAD', 1.78, 1000., 'HCRB2017')
    with pytest.raises(FileNotFoundError):   # the ARZ models need their shower library (askaryan.arz_library)
        askaryan.get_frequency_spectrum(1e18, 1., 256, 0.5, 'HAD', 1.78, 1000., 'ARZ2020', seed=987)


def test_ray_tracing_class_like_T05():
    """NuRadioMC/test/SignalProp/T05unit_test_C0_SP.py through the drop-in class, first 150 vertices."""
    from nuradiomc_amd import propagation
    g = golden('ref_C0_SP.npz')
    ice = _Ice(*g['ice'])
    r = propagation.get_propagation_module('analytic')(ice)
    n = 150
    C0 = np.zeros((n, 2))
    for iX, x in enumerate(g['points'][:n]):
        r.set_start_and_end_point(x, g['x_receiver'])
        r.find_solutions()
        if r.has_solution():
            for iS in range(r.get_number_of_solutions()):
                C0[iX, iS] = r.get_results()[iS]['C0']
    np.testing.assert_allclose(C0, g['C0_ref'][:n], rtol=1e-6, atol=1e-8)


def test_ray_tracing_class_api_and_errors():
    from nuradiomc_amd import propagation
    g = golden('raytrace_A.npz')
    ice = _Ice(*g['ice'])
    cfg = {'propagation': {'attenuate_ice': True, 'focusing': False, 'focusing_limit': 2, 'birefringence': False,
                           'n_freq': 25, 'attenuation_model': 'SP1', 'n_reflections': 0}}
    r = propagation.ray_tracing(ice, config=cfg)
    i = int(np.where(g['n_sol'] == 2)[0][0])
    r.set_start_and_end_point(g['x1'][i], g['x2'][i])
    r.find_solutions()
    assert r.get_number_of_solutions() == 2 and r.get_number_of_raytracing_solutions() == 2
    for iS in range(2):
        assert r.get_solution_type(iS) == g['type'][i, iS]
        assert abs(r.get_path_length(iS) - g['D'][i, iS]) < 1e-6 * g['D'][i, iS]
        # get_path (plotting helper): from the lower end point to the other one, polyline length = path length
        p = r.get_path(iS, n_points=4000)
        lo, hi = (g['x1'][i], g['x2'][i]) if g['x2'][i][2] >= g['x1'][i][2] else (g['x2'][i], g['x1'][i])
        assert np.max(np.abs(p[0] - lo)) < 1e-6 and np.max(np.abs(p[-1] - hi)) < 1e-2
        assert abs(np.sum(np.linalg.norm(np.diff(p, axis=0), axis=1)) - g['D'][i, iS]) < 2e-3 * g['D'][i, iS]
        assert abs(r.get_travel_time(iS) - g['T'][i, iS]) < 1e-6 * g['T'][i, iS]
        assert np.max(np.abs(r.get_launch_vector(iS) - g['launch'][i, iS])) < 1e-6
        assert np.max(np.abs(r.get_receive_vector(iS) - g['receive'][i, iS])) < 1e-6
        ra = r.get_reflection_angle(iS)
        assert (ra is None) == bool(np.isnan(g['refl_angle'][i, iS]))
        out = r.get_raytracing_output(iS)
        assert out['ray_tracing_solution_type'] == g['type'][i, iS] and out['focusing_factor'] == 1
    if i < g['att'].shape[0]:
        ff = np.fft.rfftfreq(4096, 0.5)
        att = r.get_attenuation(0, ff, 1.0)
        ref = np.ones_like(ff)
        ref[1:] = np.interp(ff[1:], g['fcoarse'], g['att'][i, 0])
        assert att[0] == 1 and np.max(np.abs(att - ref) / ref) < 1e-6
    with pytest.raises(IndexError):
        r.get_launch_vector(2)
    with pytest.raises(TypeError):
        propagation.ray_tracing(object())
    with pytest.raises(NotImplementedError):
        propagation.get_propagation_module('radiopropa')
    # a pair in the shadow zone: no solution
    j = int(np.where(g['n_sol'] == 0)[0][0])
    r.set_start_and_end_point(g['x1'][j], g['x2'][j])
    r.find_solutions()
    assert not r.has_solution() and r.get_results() == []
    # a receiver in air: like the reference's Python path (0 solutions for any such pair), never an exception
    r.set_start_and_end_point([300., 100., -400.], [0., 0., 12.])
    r.find_solutions()
    assert not r.has_solution() and r.get_number_of_solutions() == 0


def test_apply_propagation_effects_like_reference():
    """apply_propagation_effects on a duck-typed ElectricField: attenuation * Fresnel (reflected ray)"""
    from nuradiomc_amd import propagation
    g = golden('raytrace_B.npz')  # shallow receiver -> reflected rays exist
    ice = _Ice(*g['ice'])
    cfg = {'propagation': {'attenuate_ice': True, 'focusing': False, 'focusing_limit': 2, 'birefringence': False,
                           'n_freq': 25, 'attenuation_model': 'SP1'}}
    r = propagation.ray_tracing(ice, config=cfg)
    i, s = [(a, b) for a, b in zip(*np.where(g['type'][:g['att'].shape[0]] == 3))][0]
    r.set_start_and_end_point(g['x1'][i], g['x2'][i])
    r.find_solutions()

    class EF:
        def __init__(self):
            self.fs = 2.0
            self.spec = np.ones((3, 2049), complex)
        def get_frequency_spectrum(self): return self.spec
        def get_frequencies(self): return np.fft.rfftfreq(4096, 0.5)
        def get_sampling_rate(self): return self.fs
        def set_frequency_spectrum(self, s, fs): self.spec = s
        def __setitem__(self, k, v): pass
    ef = r.apply_propagation_effects(EF(), int(s))
    ff = ef.get_frequencies()
    att = np.ones_like(ff)
    att[1:] = np.interp(ff[1:], g['fcoarse'], g['att'][i, s])
    from oracle import spectral_oracle as so
    n1 = ice.n_ice - ice.delta_n * np.exp(-0.01 / ice.z_0)
    rp, rs = so.fresnel_r_p(g['refl_angle'][i, s], 1., n1), so.fresnel_r_s(g['refl_angle'][i, s], 1., n1)
    assert np.max(np.abs(ef.spec[0] - att)) < 1e-6
    assert np.max(np.abs(ef.spec[1] - att * rp)) < 1e-6 and np.max(np.abs(ef.spec[2] - att * rs)) < 1e-6


class _FakeEfield:
    def __init__(self, ch_pos, trace, t0, fs, zen, az):
        self._p, self._tr, self._t0, self._fs, self._par = ch_pos, trace, t0, fs, {'zenith': zen, 'azimuth': az}
    def get_position(self): return self._p
    def get_trace(self): return self._tr
    def get_trace_start_time(self): return self._t0
    def get_sampling_rate(self): return self._fs
    def get_number_of_samples(self): return self._tr.shape[-1]
    def __getitem__(self, k): return self._par[getattr(k, 'name', k)]


class _FakeChannel:
    def __init__(self, cid): self.cid = cid
    def set_trace(self, tr, fs): self.trace, self.fs = np.array(tr), fs
    def set_trace_start_time(self, t): self.t0 = t
    def get_id(self): return self.cid


class _FakeSimStation:
    def __init__(self, efields): self._ef = efields
    def get_id(self): return 101
    def get_electric_fields(self): return [e for v in self._ef.values() for e in v]
    def get_electric_fields_for_channels(self, ids): return [e for c in ids for e in self._ef.get(c, [])]


class _FakeStation:
    def __init__(self, sim): self._sim, self.channels = sim, {}
    def get_id(self): return 101
    def get_sim_station(self): return self._sim
    def add_channel(self, ch): self.channels[ch.get_id()] = ch


class _FakeDet:
    def __init__(self, pos, antenna, cable, n, fs):
        self.pos, self.antenna, self.cable, self.n, self.fs = pos, antenna, cable, n, fs
    def get_channel_ids(self, sid): return list(range(len(self.pos)))
    def get_relative_position(self, sid, c): return self.pos[c]
    def get_cable_delay(self, sid, c): return self.cable[c]
    def get_antenna_model(self, sid, c, zen=None): return self.antenna
    def get_antenna_orientation(self, sid, c): return [0., 0., np.pi / 2, np.pi / 2]
    def get_number_of_samples(self, sid, c): return self.n
    def get_sampling_frequency(self, sid, c): return self.fs


@pytest.mark.parametrize('antenna,cable', [('analytic_VPol', [0.] * 5), ('analytic_HPol', [0., 3.3, 7.77, 12.2, 19.8]),
                                           ('analytic_LPDA', [0., 0., 4.4, 0., 1.1]),
                                           ('synthetic_table_v1', [0., 2.2, 0., 0., 5.5])])
@pytest.mark.parametrize('N', [256, 300])
def test_efieldToVoltageConverter_module(antenna, cable, N):
    _efield_to_voltage_module_case(antenna, cable, N, 40, 10)


@pytest.mark.parametrize('antenna,cable,N', [('analytic_VPol', [0., 1000.3, 5000.77, 7000.2, 9000.8], 8192),   # L ~ 26 400: blocks both ways
                                             ('analytic_LPDA', [0., 0., 2504.4, 0., 801.1], 6144),
                                             ('analytic_HPol', [0., 3.3, 7.77, 12.2, 19.8], 10240)])
def test_efieldToVoltageConverter_module_at_the_batched_limits(antenna, cable, N):
    """nrhip_efield_to_voltage takes what nrhip_simulate_events takes: station traces up to 8192 samples (and the radix-5 10 240),
    common traces up to 32 766 samples (forward and inverse chirp-z in blocks)."""
    _efield_to_voltage_module_case(antenna, cable, N, 20, 3)


def _efield_to_voltage_module_case(antenna, cable, N, n_events, n_min):
    """The module-level drop-in on arbitrary ElectricField-like objects vs the oracle's restatement of
    efieldToVoltageConverter.run (no filter), incl. the sub-sample Fourier shift (unequal cable delays)."""
    from nuradiomc_amd import modules
    from oracle import spectral_oracle as so
    fs = 2.0   # (N = 300: no power of two)
    pos = np.array([[0., 0., -100. - i] for i in range(5)])
    ice = (1.78, 0.423, 77.)
    import nuradiomc_amd
    models, o_antenna = None, antenna
    if antenna == 'synthetic_table_v1':  # the table of tests/golden/chain_N256_tab.npz (pinned against the reference there)
        g = golden('chain_N256_tab.npz')
        o_antenna = dict(freqs=g['tab_freqs'], thetas=g['tab_thetas'], phis=g['tab_phis'], H_theta=g['tab_H_theta'],
                         H_phi=g['tab_H_phi'], orientation=g['tab_orientation'])
        models = {antenna: nuradiomc_amd.TabulatedAntenna(g['tab_freqs'], g['tab_thetas'], g['tab_phis'], g['tab_H_theta'],
                                                          g['tab_H_phi'], g['tab_orientation'])}
    ost = so.Station(pos, antenna=o_antenna, cable_delay=cable, n_samples=N, fs=fs)
    det = _FakeDet(pos, antenna, cable, N, fs)
    conv = modules.efieldToVoltageConverter(channel_factory=_FakeChannel, antenna_models=models)
    conv.begin(caching=False)
    rng = np.random.default_rng(9)
    n_done = 0
    for ev in range(n_events):
        r, ph = np.sqrt(rng.uniform(0, 1500. ** 2)), rng.uniform(0, 2 * np.pi)
        vertex = np.array([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500, -50)])
        efs = so.sim_efields_for_event(vertex, np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi), 1e18, 'HAD', None,
                                       ost, ice, n_freq=25)
        if not efs:
            continue
        by_ch = {}
        for ef in efs:
            tr = so.freq2time(ef['spec'], fs)
            by_ch.setdefault(ef['channel'], []).append(_FakeEfield(pos[ef['channel']], tr, ef['t0'], fs, ef['zenith'],
                                                                   ef['azimuth']))
        station = _FakeStation(_FakeSimStation(by_ch))
        conv.run(None, station, det)
        V_ref, t_min, L = so.combined_voltage(efs, ost, filters=())
        assert sorted(station.channels) == list(range(5))
        scale = np.max(np.abs(V_ref))
        for c in range(5):
            ch = station.channels[c]
            assert ch.t0 == t_min and len(ch.trace) == L and ch.fs == fs
            assert np.max(np.abs(ch.trace - V_ref[c])) <= 1e-6 * scale, (ev, c)
        n_done += 1
    assert n_done >= n_min
    with pytest.raises(LookupError):
        conv.run(None, _FakeStation(_FakeSimStation({})), det)


def test_efieldToVoltageConverter_uncertainties():
    """begin(uncertainty=...) (efieldToVoltageConverter.py:42-90, :320-324): the systematic draws in begin (positions, then one gain
    per channel), the statistical gain per (channel, electric field) in run, all from numpy's global generator in the reference's
    order -- so with the same seed the factors can be re-drawn here, and the channel traces are the per-field voltages times them."""
    from nuradiomc_amd import modules
    from oracle import spectral_oracle as so
    N, fs, antenna, cable = 512, 2.0, 'analytic_VPol', [0., 1.3, 0., 2.6, 0.]
    pos = np.array([[0., 0., -100. - i] for i in range(5)])
    ice = (1.78, 0.423, 77.)
    ost = so.Station(pos, antenna=antenna, cable_delay=cable, n_samples=N, fs=fs)
    det = _FakeDet(pos, antenna, cable, N, fs)
    rng = np.random.default_rng(4)
    efs = []
    while not efs:
        r, ph = np.sqrt(rng.uniform(0, 800. ** 2)), rng.uniform(0, 2 * np.pi)
        vertex = np.array([r * np.cos(ph), r * np.sin(ph), rng.uniform(-900, -200)])
        efs = so.sim_efields_for_event(vertex, 1.1, 0.4, 1e18, 'HAD', None, ost, ice, n_freq=25)
    by_ch = {}
    for ef in efs:
        by_ch.setdefault(ef['channel'], []).append(_FakeEfield(pos[ef['channel']], so.freq2time(ef['spec'], fs), ef['t0'], fs,
                                                               ef['zenith'], ef['azimuth']))
    unc = dict(sys_dx=0.1, sys_dz=0.2, sys_amp={c: 0.05 for c in range(5)}, amp={c: 0.1 + 0.01 * c for c in range(5)})
    conv = modules.efieldToVoltageConverter(channel_factory=_FakeChannel)
    np.random.seed(1234)
    conv.begin(uncertainty={k: (dict(v) if isinstance(v, dict) else v) for k, v in unc.items()})
    station = _FakeStation(_FakeSimStation(by_ch))
    conv.run(None, station, det)
    # the same draws, in the reference's order
    np.random.seed(1234)
    np.random.normal(0, unc['sys_dx']); np.random.normal(0, unc['sys_dz'])
    sys_amp = {c: np.random.normal(1, unc['sys_amp'][c]) for c in range(5)}
    scaled = []
    for c in range(5):
        for k, ef in enumerate([e for e in efs if e['channel'] == c]):
            g = np.random.normal(1, unc['amp'][c]) * sys_amp[c]
            scaled.append(dict(ef, spec=ef['spec'] * g))
    assert len(scaled) == len(efs)
    V_ref, t_min, L = so.combined_voltage(scaled, ost, filters=())
    V_plain, _, _ = so.combined_voltage(efs, ost, filters=())
    scale = np.max(np.abs(V_ref))
    assert np.max(np.abs(V_ref - V_plain)) > 1e-3 * scale       # the factors do something
    for c in range(5):
        assert np.max(np.abs(station.channels[c].trace - V_ref[c])) <= 1e-6 * scale, c


class _FakeSimChannel(_FakeChannel):
    def __init__(self, cid, ef):
        super().__init__(cid)
        self.ef = ef


class _FakeSimStation2(_FakeSimStation):
    def __init__(self, efields):
        super().__init__(efields)
        self.sim_channels = []
    def add_channel(self, ch): self.sim_channels.append(ch)


@pytest.mark.parametrize('antenna,cable', [('analytic_VPol', [0.] * 5), ('analytic_LPDA', [0., 0., 4.4, 0., 1.1])])
def test_efieldToVoltageConverterPerEfield_module(antenna, cable):
    """The per-efield converter (one SimChannel per electric field, its own N-sample grid, no cable delay) vs the oracle's
    restatement of efieldToVoltageConverterPerEfield.run."""
    from nuradiomc_amd import modules
    from oracle import spectral_oracle as so
    N, fs = 256, 2.0
    pos = np.array([[0., 0., -100. - i] for i in range(5)])
    ice = (1.78, 0.423, 77.)
    ost = so.Station(pos, antenna=antenna, cable_delay=cable, n_samples=N, fs=fs)
    det = _FakeDet(pos, antenna, cable, N, fs)
    conv = modules.efieldToVoltageConverterPerEfield(sim_channel_factory=_FakeSimChannel)
    rng = np.random.default_rng(19)
    n_done = 0
    for ev in range(25):
        r, ph = np.sqrt(rng.uniform(0, 1500. ** 2)), rng.uniform(0, 2 * np.pi)
        vertex = np.array([r * np.cos(ph), r * np.sin(ph), rng.uniform(-1500, -50)])
        efs = so.sim_efields_for_event(vertex, np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi), 1e18, 'HAD', None,
                                       ost, ice, n_freq=25)
        if not efs:
            continue
        by_ch = {}
        for ef in efs:
            fe = _FakeEfield(pos[ef['channel']], so.freq2time(ef['spec'], fs), ef['t0'], fs, ef['zenith'], ef['azimuth'])
            fe.ref = ef
            by_ch.setdefault(ef['channel'], []).append(fe)
        sim = _FakeSimStation2(by_ch)
        conv.run(None, sim, det)
        assert len(sim.sim_channels) == len(efs)
        for sc in sim.sim_channels:
            v, _ = so.per_efield_voltage(sc.ef.ref, ost, filters=())
            ref = so.freq2time(v, fs)
            assert sc.t0 == sc.ef.ref['t0'] and sc.cid == sc.ef.ref['channel'] and len(sc.trace) == N
            assert np.max(np.abs(sc.trace - ref)) <= 1e-6 * max(np.max(np.abs(ref)), 1e-300)
            n_done += 1
    assert n_done > 40
    with pytest.raises(LookupError):
        conv.run(None, _FakeSimStation2({}), det)


def test_get_focusing_and_raytracing_output(gpu_ctx_factory):
    """ray_tracing.get_focusing / get_raytracing_output / apply_propagation_effects with config focusing: the drop-in's
    values equal the oracle's (same bits in the two ray tables -> same finite difference), which is pinned against the
    reference's get_focusing by tests/test_oracle_golden.py::test_focusing_vs_reference."""
    from nuradiomc_amd import propagation
    from oracle import raytrace_oracle as rto
    g = golden('ref_focusing.npz')

    class Ice:
        n_ice, delta_n, z_0 = [float(v) for v in g['ice']]
    cfg = {'propagation': {'attenuate_ice': False, 'focusing_limit': 2, 'focusing': True, 'birefringence': False}}
    rt = propagation.ray_tracing(Ice(), attenuation_model='SP1', config=cfg)
    ref = rto.focusing(g['x1'][:60], g['x2'][:60], g['ice'], -0.01, 2.)
    n = 0
    for i in range(60):
        rt.set_start_and_end_point(g['x1'][i], g['x2'][i])
        rt.find_solutions()
        for iS in range(rt.get_number_of_solutions()):
            f = rt.get_focusing(iS)
            assert abs(f - ref[i, iS]) <= 1e-12 * ref[i, iS]
            assert rt.get_raytracing_output(iS)['focusing_factor'] == f
            n += 1
    assert n > 60


def test_context_close_before_station():
    """closing the context first must not leave the station with a dangling handle (a SIGBUS at interpreter exit once)"""
    import nuradiomc_amd
    ctx = nuradiomc_amd.Context((1.78, 0.423, 77.), 'SP1')
    st = nuradiomc_amd.Station(ctx, np.array([[0., 0., -100.]]), n_samples=256, sampling_rate=2.0)
    trig, _ = st.simulate_events(np.array([[200., 0., -300.]]), 1.0, 0.3, 1e18, 'HAD')
    ctx.close()
    st.close()
    del st, ctx


def test_set_solution_roundtrip(gpu_ctx_factory):
    """ray_tracing.set_solution (analyticraytracing.py:2092): the tables rebuilt from stored launch parameters equal the
    ones find_solutions produced, bit for bit (the order of the stored solutions is kept)."""
    from nuradiomc_amd import propagation

    class Ice:
        n_ice, delta_n, z_0 = 1.78, 0.423, 77.
    rt = propagation.ray_tracing(Ice(), attenuation_model='SP1')
    rng = np.random.default_rng(4)
    n = 0
    for _ in range(60):
        x1 = np.array([rng.uniform(-1500, 1500), rng.uniform(-1500, 1500), rng.uniform(-2000, -20)])
        x2 = np.array([0., 0., rng.choice([-5., -100., -300.])])
        rt.set_start_and_end_point(x1, x2)
        rt.find_solutions()
        ns = rt.get_number_of_solutions()
        if ns == 0:
            continue
        ref = [(rt.get_solution_type(i), rt.get_launch_vector(i), rt.get_receive_vector(i), rt.get_path_length(i),
                rt.get_travel_time(i), rt.get_reflection_angle(i)) for i in range(ns)]
        stored = {k: np.array([rt.get_raytracing_output(i)[k] for i in range(ns)] + [np.nan] * (2 - ns)) for k in
                  ('ray_tracing_C0', 'ray_tracing_C1', 'ray_tracing_reflection', 'ray_tracing_reflection_case',
                   'ray_tracing_solution_type')}
        rt.set_start_and_end_point(x1, x2)
        rt.set_solution(stored)
        assert rt.get_number_of_solutions() == ns
        for i in range(ns):
            got = (rt.get_solution_type(i), rt.get_launch_vector(i), rt.get_receive_vector(i), rt.get_path_length(i),
                   rt.get_travel_time(i), rt.get_reflection_angle(i))
            assert got[0] == ref[i][0] and np.array_equal(got[1], ref[i][1]) and np.array_equal(got[2], ref[i][2])
            assert got[3] == ref[i][3] and got[4] == ref[i][4] and got[5] == ref[i][5]
            n += 1
    assert n > 40


class _EF:
    """the slice of NuRadioReco's ElectricField that apply_propagation_effects touches"""
    def __init__(self, n_samples=256, fs=2.0):
        self.fs, self.n = fs, n_samples
        self.spec = np.ones((3, n_samples // 2 + 1), complex)
    def get_frequency_spectrum(self): return self.spec
    def get_frequencies(self): return np.fft.rfftfreq(self.n, 1. / self.fs)
    def get_sampling_rate(self): return self.fs
    def set_frequency_spectrum(self, s, fs): self.spec = s
    def __setitem__(self, k, v): pass


def test_ray_tracing_class_like_T06_mooresbay():
    """NuRadioMC/test/SignalProp/T06unit_test_C0_mooresbay.py through the drop-in class (ice shelf with a reflective bottom,
    n_reflections = 2): first 120 vertices of the reference's golden table, then the per-solution getters and
    apply_propagation_effects (attenuation per path segment, Fresnel factors per surface reflection, coefficient and phase
    per bottom reflection) against the reference's outputs."""
    from nuradiomc_amd import propagation
    g = golden('ref_mooresbay.npz')
    ice = _Ice(*g['ice'])
    ice.reflection = float(g['z_reflection'])
    ice.reflection_coefficient = float(g['reflection_coefficient'])
    ice.reflection_phase_shift = float(g['reflection_phase_shift'])
    r = propagation.ray_tracing(ice, attenuation_model='MB1', n_reflections=2, n_frequencies_integration=25)
    assert r.get_number_of_raytracing_solutions() == 10
    n = 120
    C0 = np.zeros((n, 10))
    n_checked = n_refl_checked = 0
    for iX, x in enumerate(g['points'][:n]):
        r.set_start_and_end_point(x, g['x_receiver'])
        r.find_solutions()
        assert r.get_number_of_solutions() == g['n_sol'][iX]
        for iS in range(r.get_number_of_solutions()):
            res = r.get_results()[iS]
            C0[iX, iS] = res['C0']
            assert (res['reflection'], res['reflection_case'], res['type']) == \
                (g['reflection'][iX, iS], g['reflection_case'][iX, iS], g['type'][iX, iS])
            assert r.get_raytracing_output(iS)['ray_tracing_reflection'] == res['reflection']
            assert abs(r.get_path_length(iS) - g['D'][iX, iS]) < 1e-5 * g['D'][iX, iS]     # C0 differ by <= 1e-6
            assert abs(r.get_travel_time(iS) - g['T'][iX, iS]) < 1e-5 * g['T'][iX, iS]
            assert np.max(np.abs(r.get_launch_vector(iS) - g['launch'][iX, iS])) < 1e-5
            assert np.max(np.abs(r.get_receive_vector(iS) - g['receive'][iX, iS])) < 1e-5
            ra = np.atleast_1d(r.get_reflection_angle(iS))
            ref_ra = g['refl_angle'][iX, iS]
            assert [a is None for a in ra] == list(np.isnan(ref_ra[:len(ra)])) and np.all(np.isnan(ref_ra[len(ra):]))
            n_checked += 1
    np.testing.assert_allclose(C0, g['ref_C0'][:n], rtol=1.e-6)     # T06unit_test_C0_mooresbay.py:47
    # apply_propagation_effects from the reference's own solution records (set_solution)
    for iX in range(int(g['n_prop'])):
        m = int(g['n_sol'][iX])
        if not m:
            continue
        r.set_start_and_end_point(g['points'][iX], g['x_receiver'])
        r.set_solution({'ray_tracing_C0': g['C0'][iX, :m], 'ray_tracing_C1': g['C1'][iX, :m],
                        'ray_tracing_solution_type': g['type'][iX, :m], 'ray_tracing_reflection': g['reflection'][iX, :m],
                        'ray_tracing_reflection_case': g['reflection_case'][iX, :m]})
        for iS in range(m):
            out = r.apply_propagation_effects(_EF(), iS).spec
            ref = g['prop_spec'][iX, iS]
            scale = np.max(np.abs(ref))
            assert np.max(np.abs(out[1] - ref[0])) <= 1e-6 * scale and np.max(np.abs(out[2] - ref[1])) <= 1e-6 * scale, (iX, iS)
            n_refl_checked += g['reflection'][iX, iS] > 0
    assert n_checked > 400 and n_refl_checked > 100
    with pytest.raises(IndexError):
        r.get_launch_vector(10)
    with pytest.raises(AttributeError):   # propagation_base_class.py:156-161
        r.set_start_and_end_point([100., 0., -600.], g['x_receiver'])
    # a medium without a reflective layer: the request is dropped with a warning (propagation_base_class.py:128-134)
    assert propagation.ray_tracing(_Ice(1.78, 0.423, 77.), n_reflections=2).get_number_of_raytracing_solutions() == 2


def test_arz_like_the_reference():
    """The ARZ time-domain model through the drop-in ARZ class and through askaryan.get_time_trace /
    get_frequency_spectrum(model='ARZ2020') against the reference's outputs (tests/golden/gen/gen_arz.py: the
    charge-excess profile the reference ships and Gaisser-Hillas shaped ones in a library of the reference's layout),
    and the batched form against the oracle."""
    from nuradiomc_amd import arz, askaryan
    from oracle import arz_oracle
    from test_oracle_golden import _arz_library
    g = golden('ref_arz.npz')
    lib = _arz_library(g)
    a = arz.ARZ(seed=1234, library=lib)
    for i, c in enumerate(g['vp_cases']):
        typ = 'HAD' if c[0] else 'EM'
        E, th, N, dt, R, f1, f2, shift, emf = c[1:]
        a.set_interpolation_factor(f1)
        a.set_interpolation_factor2(f2)
        prof = g['lib_HAD_1e18'][0] if c[0] else g['lib_EM_1e18'][0]
        vp = a.get_vector_potential(E, th, int(N), dt, g['lib_depth'], prof, typ, 1.78, R, bool(shift), emf)
        ref = g['vp_%d' % i]
        assert vp.shape == ref.shape and np.max(np.abs(vp - ref)) <= 1e-9 * np.max(np.abs(ref)), i
    a.set_interpolation_factor(1)
    a.set_interpolation_factor2(100)
    a.set_seed(int(g['tr_seed']))
    for k, c in enumerate(g['tr_cases']):
        typ = 'HAD' if c[0] else 'EM'
        tr = a.get_time_trace(c[1], c[2], 256, 0.5, typ, 1.78, c[3], same_shower=bool(c[4]), iN=None if c[5] < 0 else c[5])
        assert a.get_last_shower_profile_id()[typ] == int(c[6]), k
        ref = g['tr'][k]
        assert tr.shape == ref.shape and np.max(np.abs(tr - ref)) <= 1e-9 * max(np.max(np.abs(ref)), 1e-300), k
    askaryan.arz_library = lib
    askaryan._arz.clear()
    for k, c in enumerate(g['ask_cases']):
        typ = 'had' if c[0] else 'em'    # the wrapper upper-cases the shower type
        kw = {} if c[3] < 0 else {'iN': int(c[3])}
        tr, add = askaryan.get_time_trace(c[1], c[2], 256, 0.5, typ, 1.78, 1500., 'ARZ2020', full_output=True,
                                          seed=int(g['ask_seed']), **kw)
        assert add['iN'] == int(c[4])
        spec = askaryan.get_frequency_spectrum(c[1], c[2], 256, 0.5, typ, 1.78, 1500., 'ARZ2020', iN=add['iN'],
                                               seed=int(g['ask_seed']))
        assert np.max(np.abs(tr - g['ask_tr'][k])) <= 1e-9 * np.max(np.abs(g['ask_tr'][k])), k
        assert np.max(np.abs(spec - g['ask_spec'][k])) <= 1e-9 * np.max(np.abs(g['ask_spec'][k])), k
    # batched: 300 (shower, ray) pairs, N = 512 at 5 GHz, vs the oracle one by one
    rng = np.random.default_rng(3)
    n = 300
    types = [['HAD', 'EM'][i % 2] for i in range(n)]
    E = 10 ** rng.uniform(16., 19., n)
    th = np.arccos(1 / 1.78) + rng.uniform(-22, 22, n) * np.pi / 180
    R = 10 ** rng.uniform(2., 3.7, n)
    b = arz.ARZ(seed=5, library=lib)
    iN = b.draw_profile_numbers(E, types)
    tr = b.get_time_trace_batch(E, th, 512, 0.2, types, 1.78, R, iN)
    o = arz_oracle.ARZ(lib, seed=5)
    n_zero = 0
    for i in range(n):
        ref = o.get_time_trace(E[i], th[i], 512, 0.2, types[i], 1.78, R[i], iN=int(iN[i]))
        n_zero += not np.any(ref)
        assert np.max(np.abs(tr[i] - ref)) <= 1e-9 * max(np.max(np.abs(ref)), 1e-300), i
    assert 5 < n_zero < 60
    with pytest.raises(NotImplementedError):
        a.get_time_trace(1e18, 1., 256, 0.5, 'TAU', 1.78, 1000.)
    # the finest profile the entry points admit (2048 depth bins: 80 KB of LDS per block) and a coarse one, vs the oracle
    for nd in (2048, 97):
        depth = np.linspace(g['lib_depth'][0], g['lib_depth'][-1], nd)
        prof = np.interp(depth, g['lib_depth'], g['lib_HAD_1e18'][0])
        for th_off, R in ((1.5, 900.), (-9., 2500.)):
            th = np.arccos(1 / 1.78) + th_off * np.pi / 180
            vp = a.get_vector_potential(3e18, th, 512, 0.25, depth, prof, 'HAD', 1.78, R, False, 0.9)
            ref = arz_oracle.vector_potential(3e18, th, 512, 0.25, depth, prof, arz_oracle.MODEL_PARAMETERS['ARZ2020']['HAD'], 'HAD', 1.78,
                                              R, 1., 100., False, 0.9)
            assert np.max(np.abs(vp - ref)) <= 1e-9 * np.max(np.abs(ref)), (nd, th_off)


def test_birefringence_like_T07():
    """NuRadioMC/test/SignalProp/T07test_birefringence.py through the drop-in class (apply_propagation_effects with config
    birefringence, model southpole_A) against the reference's golden file reference_BF.npy at T07's tolerance and 100x
    tighter; the step records and spectra of the batched call against the oracle and the reference's own outputs, incl. a
    Greenland set with an ice-flow angle."""
    from nuradiomc_amd import propagation
    from oracle import birefringence_oracle as bo
    from test_oracle_golden import _bire_case, _bire_input
    g = golden('ref_birefringence.npz')
    fs = float(g['sampling_rate'])
    size = len(g['input_trace'])
    for tag, ice_par in (('sp', g['sp_ice']), ('gl', g['gl_ice'])):
        tck, angle, pts, rec = _bire_case(g, tag)
        model = str(g[tag + '_model'])
        propagation.birefringence_models[model] = [(t, c, 3) for t, c in tck]
        config = {'propagation': dict(attenuate_ice=False, focusing_limit=2, focusing=False, birefringence=True,
                                      birefringence_model=model, birefringence_propagation='analytical')}
        if angle is not None:
            config['propagation']['angle_to_iceflow'] = angle
        r = propagation.ray_tracing(_Ice(*ice_par))
        th, ph = [g['input_trace']], [g['input_trace']]
        k = 0
        for iX, x in enumerate(pts):
            r.set_start_and_end_point(x, rec)
            r.find_solutions()
            r.set_config(config)
            for iS in range(r.get_number_of_solutions()):
                ef = _EF(size, fs)
                spec_in = np.fft.rfft(g['input_trace']) / fs * 2 ** 0.5
                ef.spec = np.array([np.zeros_like(spec_in), spec_in, spec_in])
                out = r.apply_propagation_effects(ef, iS).spec
                ref = g['%s_spec_%d' % (tag, k)]
                # the ray parameters differ from the reference's by its first-root noise (1e-7 in C0): the phases move
                assert np.max(np.abs(out[1:] - ref)) <= 2e-4 * np.max(np.abs(ref)), (tag, k)
                th.append(np.fft.irfft(out[1], n=size) * fs / 2 ** 0.5)
                ph.append(np.fft.irfft(out[2], n=size) * fs / 2 ** 0.5)
                k += 1
        assert k == len(g[tag + '_rays'])
        got = np.vstack((np.array(th), np.array(ph)))
        if tag == 'sp':
            np.testing.assert_allclose(got, g['ref_BF'], atol=2e-4, rtol=1e-7)    # T07test_birefringence.py:98
            assert np.max(np.abs(got - g['ref_BF'])) < 1e-4
        # batched, from the reference's own ray parameters: step records and spectra
        rays = g[tag + '_rays']
        ctx = r._ctx
        X1 = np.array([pts[int(q[0])] for q in rays])
        X2 = np.tile(rec, (len(rays), 1))
        spec_in = np.array([_bire_input(g, ice_par, X1[q], rec, int(rays[q][1])) for q in range(len(rays))])
        spec, steps = ctx.birefringence_batch(X1, X2, rays[:, 2], rays[:, 3], spec_in, fs, tck, angle_to_iceflow=angle,
                                              return_steps=True)
        o = 0
        for q, (iX, iS, C0, D, n_steps) in enumerate(rays):
            ref = g['%s_spec_%d' % (tag, q)]
            assert np.max(np.abs(spec[q] - ref)) <= 1e-6 * np.max(np.abs(ref)), (tag, q)
            st = steps[o:o + int(n_steps)]
            o += int(n_steps)
            if q < 3:
                os_ = bo.path_steps(X1[q], rec, C0, D, ice_par, tck, angle)
                assert np.max(np.abs(st[:, 0:2] - os_['P1'][:, 1:])) < 1e-7 and np.max(np.abs(st[:, 2:4] - os_['P2'][:, 1:])) < 1e-7
                assert np.max(np.abs(st[:, 4] - (os_['T2'] - os_['T1']))) < 1e-10   # difference of two ~6 ns delays
                e = bo.propagate(spec_in[q][0], spec_in[q][1], fs, os_)
                assert np.max(np.abs(spec[q] - e)) <= 1e-6 * np.max(np.abs(e))
        assert o == len(steps)
    cfg = {'propagation': dict(attenuate_ice=False, focusing_limit=2, focusing=False, birefringence=True,
                               birefringence_model='southpole_A', birefringence_propagation='numerical')}
    r.set_config(cfg)
    with pytest.raises(NotImplementedError):
        r.apply_propagation_effects(_EF(size, fs), 0)
