"""Station = flat-array detector description + filter chain handed to libnrhip (nrhip_station_desc)."""
import ctypes
import numpy as np
from . import _lib as L
from . import filters as flt

ANTENNA_TO_INT = {'analytic_VPol': 0, 'analytic_HPol': 1, 'analytic_LPDA': 2}
ANTENNA_TABLE = 3   # NRHIP_ANT_TABLE
ASKARYAN_TO_INT = {'Alvarez2009': 0, 'Alvarez2000': 1, 'ZHS1992': 2, 'ARZ2019': 3, 'ARZ2020': 4}
SHOWER_TO_INT = {'HAD': 0, 'EM': 1}


class StationDesc(ctypes.Structure):
    _fields_ = [('n_channels', ctypes.c_int32), ('position', L.c_double_p), ('cable_delay', L.c_double_p),
                ('antenna_model', L.c_int32_p), ('orientation', L.c_double_p), ('n_samples', ctypes.c_int32),
                ('sampling_rate', ctypes.c_double), ('readout_length', ctypes.c_double),
                ('pre_pulse_time', ctypes.c_double), ('post_pulse_time', ctypes.c_double),
                ('n_att_freq', ctypes.c_int32), ('att_freq', L.c_double_p), ('att_bound_inv_length', L.c_double_p),
                ('att_bound_depth', ctypes.c_double), ('n_filters', ctypes.c_int32),
                ('filter_nb', L.c_int32_p), ('filter_na', L.c_int32_p), ('filter_b', L.c_double_p),
                ('filter_a', L.c_double_p), ('att_bound_n_bins', ctypes.c_int32), ('att_bound_bin_width', ctypes.c_double),
                ('att_bound_bin_inv_length', L.c_double_p), ('filter_kind', L.c_int32_p),
                ('n_antenna_tables', ctypes.c_int32), ('antenna_tables', ctypes.c_void_p),
                ('antenna_table_index', L.c_int32_p), ('n_filter_sets', ctypes.c_int32), ('channel_filter_set', L.c_int32_p),
                ('set_n_filters', L.c_int32_p), ('n_filter_table_points', ctypes.c_int32), ('filter_table', L.c_double_p)]


class AntennaTable(ctypes.Structure):   # nrhip_antenna_table
    _fields_ = [('n_freq', ctypes.c_int32), ('n_theta', ctypes.c_int32), ('n_phi', ctypes.c_int32),
                ('freqs', L.c_double_p), ('thetas', L.c_double_p), ('phis', L.c_double_p),
                ('vel_theta', L.c_double_p), ('vel_phi', L.c_double_p), ('orientation', ctypes.c_double * 4)]


class TabulatedAntenna:
    """A tabulated antenna pattern (NuRadioReco/detector/antennapattern.py:1338-1424, AntennaPattern): complex vector
    effective length H_theta / H_phi on a regular (frequency, theta, phi) grid, flat index iF * nT * nP + iP * nT + iT,
    frequencies in GHz, angles in rad, orientation = (theta, phi, rotation theta, rotation phi) of the frame the
    pattern was simulated in.  `from_pickle` reads the reference's antenna-model pickle files."""

    def __init__(self, freqs, thetas, phis, H_theta, H_phi, orientation, name='tabulated'):
        self.name = name
        self.freqs = np.ascontiguousarray(freqs, float)
        self.thetas = np.ascontiguousarray(thetas, float)
        self.phis = np.ascontiguousarray(phis, float)
        n = len(self.freqs) * len(self.thetas) * len(self.phis)
        self.H_theta = np.ascontiguousarray(H_theta, complex).reshape(-1)
        self.H_phi = np.ascontiguousarray(H_phi, complex).reshape(-1)
        if len(self.H_theta) != n or len(self.H_phi) != n:
            raise ValueError("antenna table: {} values expected, H_theta has {}, H_phi {}".format(
                n, len(self.H_theta), len(self.H_phi)))
        self.orientation = np.ascontiguousarray(orientation, float).reshape(4)

    @classmethod
    def from_pickle(cls, path, name=None):
        """antennapattern.py:1362-1365: [orientation_theta, orientation_phi, rotation_theta, rotation_phi, ff, thetas,
        phis, H_phi, H_theta] with the angles repeated per table entry."""
        import pickle
        with open(path, 'rb') as fin:
            res = pickle.load(fin, encoding='latin1')
        ot, op, rt, rp, ff, thetas, phis, H_phi, H_theta = res
        return cls(np.unique(ff), np.unique(thetas), np.unique(phis), H_theta, H_phi, (ot, op, rt, rp),
                   name=name or str(path))

    def _ctypes(self):
        t = AntennaTable(len(self.freqs), len(self.thetas), len(self.phis), L.dptr(self.freqs), L.dptr(self.thetas),
                         L.dptr(self.phis), self.H_theta.view(float).ctypes.data_as(L.c_double_p),
                         self.H_phi.view(float).ctypes.data_as(L.c_double_p))
        for i in range(4):
            t.orientation[i] = self.orientation[i]
        return t


class SimConfig(ctypes.Structure):
    _fields_ = [('askaryan_model', ctypes.c_int32), ('delta_C_cut', ctypes.c_double),
                ('min_efield_amplitude', ctypes.c_double), ('trigger_threshold', ctypes.c_double),
                ('dump_traces', ctypes.c_int32), ('no_pruning', ctypes.c_int32), ('trigger_type', ctypes.c_int32),
                ('n_coincidences', ctypes.c_int32), ('threshold_high', ctypes.c_double), ('threshold_low', ctypes.c_double),
                ('high_low_window', ctypes.c_double), ('coinc_window', ctypes.c_double), ('amp_per_ray', ctypes.c_int32),
                ('focusing', ctypes.c_int32), ('focusing_limit', ctypes.c_double), ('select_only', ctypes.c_int32),
                ('reuse_ray_tables', ctypes.c_int32), ('accumulate_triggered', ctypes.c_int32), ('n_reflections', ctypes.c_int32),
                ('z_reflection', ctypes.c_double), ('reflection_coefficient', ctypes.c_double),
                ('reflection_phase_shift', ctypes.c_double), ('split_event_time_diff', ctypes.c_double),
                ('noise', ctypes.c_int32), ('noise_seed', ctypes.c_uint64), ('noise_group_offset', ctypes.c_int64),
                ('noise_group_id', ctypes.c_void_p), ('custom_polarization', ctypes.c_int32),
                ('polarization_ephi', ctypes.c_double), ('emit_triggered_traces', ctypes.c_int32),
                ('emit_capacity_samples', ctypes.c_int64), ('given_C0', ctypes.c_void_p), ('given_D', ctypes.c_void_p),
                ('given_T', ctypes.c_void_p)]


class SimStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in ('n_events', 'n_pairs', 'n_rays', 'n_candidate_events', 'n_triggered',
                                              'n_channel_items', 'n_distinct_lengths', 'n_candidate_rays', 'n_active_rays',
                                              'n_integrand_evals', 'n_channel_transforms', 'n_ray_transforms',
                                              'n_efield_transforms')] + \
               [('max_length', ctypes.c_int32), ('n_sub_events', ctypes.c_int32), ('stage_ms', ctypes.c_double * 9)] + \
               [(k, ctypes.c_int64) for k in ('n_emitted_events', 'n_emit_overflow', 'n_emitted_samples', 'n_objective_evals', 'n_adc_convolution_flops',
                                              'n_arz_evals', 'n_bire_steps', 'n_bire_step_bins', 'n_efield_sampled')]

    STAGES = ('raytrace', 'ray_setup', 'amp_bound', 'attenuation', 'efield_max', 'event_grid', 'length_tables', 'channel',
              'total')

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_[:15]}
        d.update({k: int(getattr(self, k)) for k in ('n_emitted_events', 'n_emit_overflow', 'n_emitted_samples', 'n_objective_evals', 'n_adc_convolution_flops',
                                                     'n_arz_evals', 'n_bire_steps', 'n_bire_step_bins', 'n_efield_sampled')})
        d['stage_ms'] = {n: float(self.stage_ms[i]) for i, n in enumerate(self.STAGES)}
        return d


L._OPTIONAL.update({
    'nrhip_station_create': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StationDesc), L.c_void_pp]),
    'nrhip_station_destroy': (None, [ctypes.c_void_p]),
    'nrhip_station_release_workspace': (ctypes.c_int64, [ctypes.c_void_p]),
    'nrhip_station_set_positions': (ctypes.c_int, [ctypes.c_void_p, L.c_double_p]),
    'nrhip_station_set_trigger_channels': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, L.c_int32_p]),
    'nrhip_station_set_envelope_trigger': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, L.c_double_p, L.c_double_p]),
    'nrhip_station_set_noise': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, L.c_double_p]),
    'nrhip_station_set_phased_array_adc': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                          L.c_int32_p]),
    'nrhip_station_set_phased_array_clock_offset': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    'nrhip_station_set_phased_array_processing': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, L.c_double_p,
                                                                 ctypes.c_int32, ctypes.c_int32, L.c_double_p]),
    'nrhip_station_set_phased_array': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, L.c_int32_p, ctypes.c_int32, L.c_int32_p,
                                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'nrhip_station_set_arz': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, L.c_double_p, L.c_double_p,
                                            L.c_double_p, ctypes.c_double, ctypes.c_int32]),
    'nrhip_station_set_shower_profiles': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, L.c_int32_p, L.c_double_p]),
    'nrhip_station_set_birefringence': (ctypes.c_int, [ctypes.c_void_p, L.c_int32_p, L.c_double_p, L.c_double_p,
                                                      ctypes.c_double, ctypes.c_double]),
    'nrhip_simulate_events': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(SimConfig), ctypes.c_int64]
                              + [ctypes.c_void_p] * 7 + [ctypes.POINTER(SimStats)]),
    'nrhip_simulate_event_groups': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(SimConfig),
                                                   ctypes.c_int64] + [ctypes.c_void_p] * 8 + [ctypes.c_int64]
                                    + [ctypes.c_void_p] * 2 + [ctypes.POINTER(SimStats)]),
    'nrhip_sim_fetch': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64]),
    'nrhip_readout_windows': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int64,
                                            L.c_int32_p, L.c_double_p, L.c_double_p]),
    'nrhip_askaryan_spectrum_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, L.c_double_p, L.c_double_p,
                                                     L.c_int32_p, L.c_double_p, L.c_double_p, L.c_double_p, ctypes.c_int32,
                                                     ctypes.c_int32, ctypes.c_double, L.c_double_p]),
    'nrhip_efield_to_voltage': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, L.c_double_p, L.c_double_p,
                                               L.c_double_p, L.c_double_p, L.c_int32_p, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_double, L.c_double_p]),
    'nrhip_debug_czt': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_double, L.c_double_p, L.c_double_p]),
    'nrhip_debug_wave_sums': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, L.c_double_p, L.c_double_p]),
})


def attenuation_frequencies(frequency, n_freq, max_detector_freq=None):
    """Sparse frequency grid of the attenuation calculation (NuRadioMC/SignalProp/analyticraytracing.py:885-931)."""
    frequency = np.asarray(frequency, float)
    non_null = frequency > 0
    n = min(n_freq, int(np.sum(non_null)))
    freqs = np.linspace(frequency[non_null].min(), frequency[non_null].max(), n)
    if n < np.sum(non_null) and max_detector_freq is not None:
        det_mask = frequency <= max_detector_freq
        total = det_mask & non_null
        n = min(n_freq, int(np.sum(total)))
        freqs = np.linspace(frequency[total].min(), frequency[total].max(), n)
        if np.sum(~det_mask) > 1:
            freqs = np.append(freqs, np.linspace(frequency[~det_mask].min(), frequency[~det_mask].max(), n // 2))
    return freqs


DEFAULT_FILTERS = ((2, (0.08, 1000.)), (10, (0., 0.5)))  # NuRadioMC/examples/01_Veff_simulation/T02RunSimulation.py:18-22


def distance_cut(vertex, energy, group_begin, coefficients, sum_length=10.):
    """speedup.distance_cut (simulation.py:155-163, :1398-1409): per shower the largest vertex-antenna distance [m] that is
    still simulated, max(100 m, 10 ** polynomial(log10(E_sum))), E_sum = energy of the showers of the same event group whose
    distance to the group's first vertex differs by less than sum_length.  Host logic (one pass over the shower list)."""
    vertex = np.asarray(vertex, float).reshape(-1, 3)
    energy = np.asarray(energy, float)
    n = len(vertex)
    poly = np.polynomial.polynomial.Polynomial(coefficients)
    e_sum = energy.copy()
    if group_begin is not None:
        for g in range(len(group_begin) - 1):
            a, b = int(group_begin[g]), int(group_begin[g + 1])
            if b - a > 1:
                d = np.linalg.norm(vertex[a:b] - vertex[a], axis=1)
                mask = np.abs(d[:, None] - d[None, :]) < sum_length
                e_sum[a:b] = (mask * energy[a:b][None, :]).sum(axis=1)
    out = np.full(n, 100.)
    pos = e_sum > 0
    out[pos] = np.maximum(100., 10 ** poly(np.log10(e_sum[pos])))
    return np.ascontiguousarray(out)


def _same_chain(a, b):
    """equal filter chains (sequences of stage specs)?"""
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if isinstance(x, dict) != isinstance(y, dict):
            return False
        if isinstance(x, dict):
            if x.keys() != y.keys() or any(not np.array_equal(np.asarray(x[k]), np.asarray(y[k])) for k in x):
                return False
        elif not (x[0] == y[0] and tuple(x[1]) == tuple(y[1])):
            return False
    return True


def _shower_type_codes(shower_type, n):
    """'HAD' / 'EM' (any case; one value or one per shower) or the integer codes -> int32 [n]"""
    code = lambda v: int(v) if isinstance(v, (int, np.integer)) else SHOWER_TO_INT[str(v).upper()]
    if isinstance(shower_type, (str, int, np.integer)):
        return np.full(n, code(shower_type), np.int32)
    a = np.asarray(shower_type).reshape(-1)
    if a.dtype.kind in 'iu':
        return np.ascontiguousarray(np.broadcast_to(a, (n,)), dtype=np.int32)
    values, inverse = np.unique(a, return_inverse=True)
    return np.ascontiguousarray(np.broadcast_to(np.array([code(v) for v in values], np.int32)[inverse], (n,)))


class Station:
    """One station on one Context.

    position [n_ch, 3] (m), antenna in {'analytic_VPol', 'analytic_HPol'} (one name or a list),
    orientation (theta, phi, rotation theta, rotation phi) in rad (one tuple or [n_ch, 4]), cable_delay (ns),
    n_samples / sampling_rate: the simulated trace grid; detector_sampling_rate: the detector's own ADC rate
    (sets the attenuation grid's max_detector_freq, propagation_base_class.py:66-80);
    filters: sequence of (order, (f_lo, f_hi)) Butterworth stages (f_lo = 0: low-pass) or dicts {'type': 'butter' |
    'butterabs' | 'cheby1' | 'rectangular' | 'gaussian_tapered', 'passband', 'order', 'rp', 'roll_width'}
    (signal_processing.get_filter_response) or filters.hardware_response(...) stages (measured amplifier responses), applied
    in order; channel_filters: one such chain per channel (at most 4 distinct chains per station).
    """

    def __init__(self, ctx, position, antenna='analytic_VPol', orientation=(0., 0., np.pi / 2, np.pi / 2),
                 cable_delay=0., n_samples=4096, sampling_rate=2.0, detector_sampling_rate=None, n_freq=25,
                 filters=DEFAULT_FILTERS, pre_pulse_time=200., post_pulse_time=400., readout_length=None,
                 att_bound_depth=3000., channel_filters=None):
        self.ctx = ctx
        self._lib = L.load()
        pos = L.f64(position).reshape(-1, 3)
        n = len(pos)
        names = [antenna] * n if isinstance(antenna, (str, TabulatedAntenna)) else list(antenna)
        tables, tab_index = [], np.zeros(n, np.int32)
        for c, a in enumerate(names):
            if isinstance(a, TabulatedAntenna):
                if not any(a is t for t in tables):
                    tables.append(a)
                tab_index[c] = [a is t for t in tables].index(True)
            elif a not in ANTENNA_TO_INT:
                raise NotImplementedError("antenna model {} is not available (analytic_VPol, analytic_HPol, analytic_LPDA, "
                                          "or a TabulatedAntenna)".format(a))
        model = np.array([ANTENNA_TABLE if isinstance(a, TabulatedAntenna) else ANTENNA_TO_INT[a] for a in names], np.int32)
        self._tabulated = bool(np.any(model == ANTENNA_TABLE))   # (amp_per_ray is not provided with tabulated patterns)
        ctabs = (AntennaTable * max(len(tables), 1))(*[t._ctypes() for t in tables])
        ori = np.ascontiguousarray(np.broadcast_to(L.f64(orientation), (n, 4)))
        cab = np.ascontiguousarray(np.broadcast_to(L.f64(cable_delay), (n,)))
        self.position, self.antenna, self.orientation, self.cable_delay = pos, names, ori, cab
        self.n_samples, self.sampling_rate = int(n_samples), float(sampling_rate)
        self.pre_pulse_time, self.post_pulse_time = float(pre_pulse_time), float(post_pulse_time)
        self.readout_length = float(readout_length if readout_length is not None else self.n_samples / self.sampling_rate)
        det_fs = float(detector_sampling_rate or sampling_rate)
        ff = np.fft.rfftfreq(self.n_samples, 1. / self.sampling_rate)
        self.att_freq = np.ascontiguousarray(attenuation_frequencies(ff, n_freq, 0.5 * det_fs))
        # filter chains: `filters` for every channel, or `channel_filters` = one chain per channel (equal chains share a set)
        if channel_filters is not None:
            if len(channel_filters) != n:
                raise ValueError("channel_filters needs one chain per channel")
            chains, self.channel_filter_set = [], np.zeros(n, np.int32)
            for c, ch in enumerate(channel_filters):
                k = next((i for i, other in enumerate(chains) if other is ch or _same_chain(other, ch)), None)
                if k is None:
                    chains.append(ch)
                    k = len(chains) - 1
                self.channel_filter_set[c] = k
        else:
            chains, self.channel_filter_set = [filters], np.zeros(n, np.int32)
        if len(chains) > 4:
            raise ValueError("at most 4 distinct filter chains per station")
        self.filter_sets = [[flt.design(spec) for spec in chain] for chain in chains]   # (kind, b, a) per stage
        self.filters = self.filter_sets[0]
        stages = [stg for chain in self.filter_sets for stg in chain]
        set_n = np.array([len(chain) for chain in self.filter_sets], np.int32)
        fkind = np.array([k for k, _, _ in stages] or [0], np.int32)
        nb = np.zeros(max(len(stages), 1), np.int32)
        na = np.zeros(max(len(stages), 1), np.int32)
        fb = np.zeros((max(len(stages), 1), flt.MAX_POLY))
        fa = np.zeros((max(len(stages), 1), flt.MAX_POLY))
        pool = []
        for i, (kind, b, a) in enumerate(stages):
            if kind == flt.KIND_TABULATED:   # nb = rows of the table, na = its first row in the pool, b = (c0, c1)
                nb[i], na[i] = len(a), sum(len(t) for t in pool)
                fb[i, :2] = b
                pool.append(a)
                continue
            nb[i], na[i] = len(b), len(a)
            fb[i, :len(b)] = b
            fa[i, :len(a)] = a
        pool = np.ascontiguousarray(np.concatenate(pool)) if pool else np.zeros((0, 3))
        # largest attenuation length between the surface and att_bound_depth per coarse frequency (1 m grid): lets the
        # library bound a ray's attenuation factor by exp(-0.95 D / L_max) before paying for the path integral
        self.att_bound_depth = float(att_bound_depth)
        cache = ctx.__dict__.setdefault('_att_bound_cache', {})   # the tables depend on (ice, model, frequencies, depth) only
        key = (self.att_freq.tobytes(), self.att_bound_depth)
        if key not in cache:
            cache[key] = self._attenuation_bound_tables(ctx)
        self.att_bound_inv_length, self.att_bound_bin_width, n_bins, self.att_bound_bin_inv_length = cache[key]
        self._keep = (pos, cab, model, ori, self.att_freq, nb, na, fb, fa, self.att_bound_inv_length,
                      self.att_bound_bin_inv_length, fkind, tables, tab_index, ctabs, set_n, pool, self.channel_filter_set)
        d = StationDesc(n, L.dptr(pos), L.dptr(cab), L.iptr(model), L.dptr(ori), self.n_samples, self.sampling_rate,
                        float(readout_length if readout_length is not None else self.n_samples / self.sampling_rate),
                        float(pre_pulse_time), float(post_pulse_time), len(self.att_freq), L.dptr(self.att_freq),
                        L.dptr(self.att_bound_inv_length), self.att_bound_depth, len(self.filters), L.iptr(nb), L.iptr(na), L.dptr(fb), L.dptr(fa),
                        n_bins, self.att_bound_bin_width, L.dptr(self.att_bound_bin_inv_length), L.iptr(fkind),
                        len(tables), ctypes.cast(ctabs, ctypes.c_void_p), L.iptr(tab_index), len(self.filter_sets),
                        L.iptr(self.channel_filter_set), L.iptr(set_n), len(pool), L.dptr(pool))
        h = ctypes.c_void_p()
        L.check(self._lib.nrhip_station_create(ctx._h, ctypes.byref(d), ctypes.byref(h)))
        self._h = h
        ctx._register_station(self)
        # Vrms per filter chain (simulation.py:1301-1376 computes it per channel); .vrms / .vrms_efield: chain of channel 0
        self.vrms_per_set = [flt.vrms_from_filters(self.sampling_rate, chain) for chain in self.filter_sets]
        self.vrms, self.vrms_efield = self.vrms_per_set[self.channel_filter_set[0]]

    def _attenuation_bound_tables(self, ctx):
        """largest attenuation length between the surface and att_bound_depth per coarse frequency (1 m grid), and per 50 m
        depth bin: lets the library bound a ray's attenuation factor before paying for the path integral"""
        zz = np.linspace(-self.att_bound_depth, 0., int(self.att_bound_depth) + 1)
        # depths where the model has no (finite, positive) length -- e.g. below the shelf of the Moore's Bay fit -- bound nothing
        def inverse(z, f):
            with np.errstate(all='ignore'):
                length = ctx.attenuation_length(z, f)
                return np.where(np.isfinite(length) & (length > 0), 1.0 / length, 0.)
        att_bound_inv_length = np.ascontiguousarray([np.min(inverse(zz, f)) / (1 + 1e-3) for f in self.att_freq])
        # the same per 50 m depth bin (0.25 m grid, 1e-3 for what happens between grid points): the library sums
        # (path length inside the bin) / L_max(bin) along the ray, a much tighter bound on the path integral
        bin_width = 50.
        n_bins = min(63, int(np.ceil(self.att_bound_depth / bin_width)))
        per = int(round(bin_width / 0.25))
        zz = -np.arange(n_bins * per + 1) * 0.25
        inv = np.array([inverse(zz, f) for f in self.att_freq])            # [n_fc][n_z]
        idx = np.arange(n_bins)[:, None] * per + np.arange(per + 1)[None, :]
        bin_inv = np.ascontiguousarray(inv[:, idx].min(axis=2).T * (1 - 1e-3))  # [n_bins][n_fc]
        return att_bound_inv_length, bin_width, n_bins, bin_inv

    # ---- general emission / propagation (BASELINE config 4: ARZ, birefringence) -----------------------------------
    def set_arz(self, arz):
        """Attach an ARZ shower library (nuradiomc_amd.arz.ARZ object: library, model version, interpolation factors) for
        simulate_events(askaryan_model='ARZ2019' | 'ARZ2020', arz_iN=...)."""
        from . import arz as arz_mod
        rows, index, depth = [], {}, None
        for t in ('HAD', 'EM'):
            for E, prof in arz._library.get(t, {}).items():
                d = np.asarray(prof['depth'], float)
                if depth is None:
                    depth = d
                elif len(d) != len(depth) or np.any(d != depth):
                    raise ValueError("the profiles of the library do not share one depth grid")
                for i, ce in enumerate(prof['charge_excess']):
                    index[(t, float(E), i)] = len(rows)
                    rows.append(np.asarray(ce, float))
        rows = np.ascontiguousarray(rows)
        if arz._interp_factor != 1:   # ARZ.py:108-113
            dense = np.linspace(min(depth), max(depth), int(arz._interp_factor * len(depth)))
            rows = np.ascontiguousarray([np.interp(dense, depth, c) for c in rows])
            depth = dense
        depth = np.ascontiguousarray(depth)
        par = np.ascontiguousarray([arz_mod._PARAMETERS[arz._arz_version]['HAD'], arz_mod._PARAMETERS[arz._arz_version]['EM']], float)
        L.check(self._lib.nrhip_station_set_arz(self._h, len(rows), len(depth), L.dptr(depth), L.dptr(rows), L.dptr(par),
                                                float(arz._interp_factor2), int(arz_mod._PARAMETERS[arz._arz_version]['em'])))
        self._arz, self._arz_index = arz, index

    def _arz_shower_profiles(self, energy, shower_type_codes, iN):
        """library row and amplitude factor E / E_library per shower (ARZ.get_time_trace, ARZ.py:561-591)"""
        lib = self._arz._library
        rows, resc = np.zeros(len(energy), np.int32), np.ones(len(energy))
        energies = {t: np.array([*lib[t]]) for t in lib}
        for i, (E, c) in enumerate(zip(energy, shower_type_codes)):
            t = 'HAD' if c == 0 else 'EM'
            E_lib = energies[t][np.argmin(np.abs(energies[t] - E))]
            rows[i] = self._arz_index[(t, float(E_lib), int(iN[i]))]
            resc[i] = E / E_lib
        return rows, resc

    def set_birefringence(self, tck, angle_to_iceflow=0., n_ref=1.78):
        """Birefringent propagation inside simulate_events: tck = the three depth splines (knots, coefficients[, 3]) of a
        birefringence ice model (propagation.birefringence_model(name)); None switches it off.  angle_to_iceflow [deg] as
        config['propagation']['angle_to_iceflow'] (0 = none)."""
        self._birefringence_on = tck is not None
        if tck is None:
            L.check(self._lib.nrhip_station_set_birefringence(self._h, None, None, None, 1.78, 0.))
            return
        knots = np.ascontiguousarray(np.concatenate([np.asarray(t[0], float) for t in tck]))
        coeffs = np.ascontiguousarray(np.concatenate([np.asarray(t[1], float) for t in tck]))
        nk = np.array([len(t[0]) for t in tck], np.int32)
        L.check(self._lib.nrhip_station_set_birefringence(self._h, L.iptr(nk), L.dptr(knots), L.dptr(coeffs), float(n_ref),
                                                          float('nan') if angle_to_iceflow is None else float(angle_to_iceflow)))

    def set_trigger_channels(self, channels=None):
        """`triggered_channels` of the reference's simple / high-low threshold triggers: only these channels can trigger or
        count towards a coincidence (None: all).  The candidate cut on the electric fields still looks at every channel."""
        if channels is None:
            L.check(self._lib.nrhip_station_set_trigger_channels(self._h, 0, None))
        else:
            ch = np.ascontiguousarray(channels, np.int32)
            L.check(self._lib.nrhip_station_set_trigger_channels(self._h, len(ch), L.iptr(ch)))

    def set_envelope_trigger(self, passband=None, order=None):
        """The band pass of the envelope trigger (envelopeTrigger.triggerSimulator.run(passband=, order=): every channel is filtered
        with a Butterworth of this pass band [GHz] and order before its Hilbert envelope is compared with the threshold); then
        simulate_events(..., trigger='envelope', trigger_threshold=, n_coincidences=, coinc_window=).  passband=None: off."""
        if passband is None:
            L.check(self._lib.nrhip_station_set_envelope_trigger(self._h, 0, 0, None, None))
            return
        _, b, a = flt.design(dict(type='butter', passband=tuple(passband), order=int(order)))
        b, a = np.ascontiguousarray(b, float), np.ascontiguousarray(a, float)
        L.check(self._lib.nrhip_station_set_envelope_trigger(self._h, len(b), len(a), L.dptr(b), L.dptr(a)))

    def set_noise(self, noise_temperature=300., noiseless_channels=(), amplitude=None):
        """Thermal noise for simulate_events(..., noise=True, noise_seed=): per channel the `amplitude` simulation.apply_det_response
        hands to channelGenericNoiseAdder (simulation.py:594-606): Vrms / sqrt(norm / max_freq) with the channel's filter chain,
        norm = int |H|^2 df, max_freq = sampling rate / 2 -- or given directly (`amplitude` [n_channels]).  noise_temperature=None
        removes it.  Returns the amplitudes."""
        if noise_temperature is None and amplitude is None:
            L.check(self._lib.nrhip_station_set_noise(self._h, 0, None))
            return None
        if amplitude is None:
            amp = np.zeros(len(self.position))
            for c in range(len(self.position)):
                chain = self.filter_sets[self.channel_filter_set[c]]
                ff = np.linspace(0, 0.5 * self.sampling_rate, 10000)
                H = np.abs(flt.response(ff, chain))
                norm = np.sum(0.5 * (H[1:] ** 2 + H[:-1] ** 2) * np.diff(ff))
                amp[c] = flt.vrms_from_filters(self.sampling_rate, chain, noise_temperature)[0] / np.sqrt(norm / (0.5 * self.sampling_rate))
        else:
            amp = np.array(np.broadcast_to(L.f64(amplitude), (len(self.position),)), float)
        amp[list(noiseless_channels)] = 0.
        amp = np.ascontiguousarray(amp)
        L.check(self._lib.nrhip_station_set_noise(self._h, len(amp), L.dptr(amp)))
        return amp

    def set_phased_array(self, channels, phasing_angles, ref_index=1.75, window=32, step=16, averaging_divisor=None, adc=None,
                         upsampling_factor=1, saturation_bits=8, upsampling_method='fft', coeff_gain=1, filter_taps=45,
                         mode='power_sum', hilbert_transformer_kwargs=None):
        """Phased-array trigger on the given channels (a vertical string): beams towards `phasing_angles` [rad], whole-sample
        channel shifts as PhasedArrayBase.calculate_time_delays (phasedArrayBase.py:58-124: antenna depths, ref_index, cable
        delays; no group delays), mean power in windows of `window` samples every `step` (power_sum :217-271).
        adc = dict(sampling_frequency [GHz], n_bits, noise_count, vrms=<station's>, output='voltage' | 'counts', clock_offset=0 [whole
        ADC clock cycles the trace is delayed by in front of the digitiser, phasedArrayTrigger.py:32,124]): the trigger ADC of
        phasedArrayTrigger.run(apply_digitization=True) (trigger_adc_sampling_frequency / trigger_adc_nbits / trigger_adc_noise_count
        of the detector description, Vrms of adc_kwargs) followed by FFT up-sampling by `upsampling_factor`; beams, windows and
        steps then count samples of the up-sampled ADC trace, count sums saturate at `saturation_bits`.  Without `adc` everything
        runs on the analog traces at the simulation's sampling rate (times upsampling_factor).  channels=None switches the trigger off.
        upsampling_method 'fft' | 'lin' | 'fir' (coeff_gain, filter_taps: upsampling_kwargs of phased_trigger;
        signal_processing.digital_upsampling :111-190, upsampling_fir :192-234) and mode 'power_sum' | 'hilbert_env'
        (hilbert_transformer_kwargs = dict(hilbert_n_taps=31, hilbert_coeff_gain=128): PhasedArrayBase.hilbert_envelope :337-367
        with the FIR transformer, or ideal_transformer=True: scipy.signal.hilbert's imaginary part and the exact magnitude) -- the
        threshold then compares with the envelope."""
        if upsampling_method not in ('fft', 'lin', 'fir'):
            raise NotImplementedError('Interpolation method must be lin, fft, or fir')
        if mode not in ('power_sum', 'hilbert_env'):
            raise ValueError("mode must be either 'power_sum' or 'hilbert_env'")
        if channels is None:
            L.check(self._lib.nrhip_station_set_phased_array(self._h, 0, None, 0, None, 0, 0, 0))
            return None
        ch = np.ascontiguousarray(channels, np.int32)
        x, y, z = self.position[ch].T
        if np.sum(np.abs(x - x[0])) > 1e-3 or np.sum(np.abs(y - y[0])) > 1e-3:   # check_vertical_string :158-181
            raise NotImplementedError('The phased triggering array should lie on a vertical line')
        rolls = []
        for angle in np.atleast_1d(phasing_angles):
            delays = (z - np.max(z)) / 0.299792458 * ref_index * np.sin(angle) - self.cable_delay[ch]
            delays -= np.min(delays)
            rolls.append(np.round(delays * self.sampling_rate).astype(int))
        rolls = np.ascontiguousarray(rolls, np.int32)
        L.check(self._lib.nrhip_station_set_phased_array(self._h, len(ch), L.iptr(ch), len(rolls), L.iptr(rolls), int(window),
                                                         int(step), int(averaging_divisor or 0)))
        analog = adc is None and (mode != 'power_sum' or int(upsampling_factor) >= 2)
        if analog:   # apply_digitization=False with up-sampling / envelope (get_traces :312-321): no comparator, the simulation's rate
            adc = dict(sampling_frequency=self.sampling_rate, n_bits=0, noise_count=1., output='voltage')
        if adc is not None:
            import fractions
            import decimal
            f_adc, n_bits = float(adc['sampling_frequency']), int(adc['n_bits'])
            vrms = float(adc.get('vrms') or self.vrms)
            half = vrms * (2 ** n_bits - 1) / float(adc['noise_count']) / 2 if n_bits else 0.5   # _get_adc_parameters :236-240
            fr = fractions.Fraction(decimal.Decimal(5.0 / self.sampling_rate)).limit_denominator(5000)   # signal_processing.resample :86
            up = max(int(upsampling_factor), 1)
            rolls = []
            for angle in np.atleast_1d(phasing_angles):
                delays = (z - np.max(z)) / 0.299792458 * ref_index * np.sin(angle) - self.cable_delay[ch]
                delays -= np.min(delays)
                rolls.append(np.round(delays * (f_adc * up)).astype(int))
            rolls = np.ascontiguousarray(rolls, np.int32)
            L.check(self._lib.nrhip_station_set_phased_array_adc(self._h, f_adc, n_bits, -half, half,
                                                                 int(adc.get('output', 'voltage') == 'counts'), up, int(saturation_bits),
                                                                 fr.numerator, fr.denominator, L.iptr(rolls)))
            clk = adc.get('clock_offset', 0)
            if clk:
                if clk - int(clk) != 0:   # analogToDigitalConverter.py:328-329
                    raise ValueError("The clock offset must be an integer number of clock cycles")
                L.check(self._lib.nrhip_station_set_phased_array_clock_offset(self._h, int(clk)))
            up_taps = hil_taps = None
            ideal = False
            if upsampling_method == 'fir' and up >= 2:    # upsampling_fir :224-230
                from . import filters
                up_taps = filters.firwin(int(filter_taps), f_adc * 0.5, True, f_adc * up)
                if coeff_gain != 1:
                    up_taps = np.trim_zeros(np.round(up_taps * coeff_gain) / coeff_gain)
                up_taps = np.ascontiguousarray(up_taps, np.float64)
            if mode == 'hilbert_env':                     # hilbert_envelope :348-355
                from . import filters
                hk = dict(ideal_transformer=False, hilbert_n_taps=31, hilbert_coeff_gain=128)
                hk.update(hilbert_transformer_kwargs or {})
                ideal = bool(hk['ideal_transformer'])
                if not ideal:
                    nt = int(hk['hilbert_n_taps'])
                    assert nt % 2 != 0, "Num taps MUST be odd for a hilbert transformer"
                    sin_factor = np.sin(np.linspace(-(nt - 1) / 2, (nt - 1) / 2, nt))
                    hil_taps = 2 * sin_factor * (-1 * filters.firwin(nt, 0.25, False, 1))
                    if hk['hilbert_coeff_gain'] != 1:
                        hil_taps = np.round(hil_taps * hk['hilbert_coeff_gain']) / hk['hilbert_coeff_gain']
                    hil_taps = np.ascontiguousarray(hil_taps, np.float64)
            method = {'fft': 0, 'lin': 1, 'fir': 2}[upsampling_method] if up >= 2 else 0
            env_mode = 0 if mode != 'hilbert_env' else (2 if ideal else 1)
            if method or env_mode:
                L.check(self._lib.nrhip_station_set_phased_array_processing(
                    self._h, method, 0 if up_taps is None else len(up_taps), None if up_taps is None else L.dptr(up_taps),
                    env_mode, 0 if hil_taps is None else len(hil_taps), None if hil_taps is None else L.dptr(hil_taps)))
        return rolls

    @staticmethod
    def empty_stats():
        """the stats dict of a call that had nothing to do"""
        return SimStats().as_dict()

    def move_to(self, position):
        """Use this object for another station of an array of identical stations: new antenna positions [n_ch, 3], everything
        else (antennas, orientations, cable delays, filters, device tables, workspace) stays."""
        pos = L.f64(position).reshape(-1, 3)
        if pos.shape != self.position.shape:
            raise ValueError("move_to: %d channels expected" % len(self.position))
        L.check(self._lib.nrhip_station_set_positions(self._h, L.dptr(pos)))
        self.position = pos

    def release_workspace(self):
        """Give the tables of the last call back to the GPU (they stay resident for `fetch` and for reuse by the next call);
        returns the number of bytes freed.  For arrays simulated station by station on one GPU."""
        return int(self._lib.nrhip_station_release_workspace(self._h)) if getattr(self, '_h', None) else 0

    def close(self):
        sc = getattr(self, '_pass2', None)
        if sc is not None and getattr(self.ctx, '_h', None):
            for q in sc['p']:
                self.ctx.free(q)
        self._pass2 = None
        if getattr(self, '_h', None):
            self._lib.nrhip_station_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path --------------------------------------------------------------------------------------
    def simulate_events_dev(self, n_events, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL, d_triggered,
                            askaryan_model='Alvarez2009', delta_C_cut=0.698, min_efield_amplitude=None,
                            trigger_threshold=None, dump_traces=False, no_pruning=False, want_stats=True,
                            d_vertex_time=None, n_groups=None, d_group_begin=None, trigger='simple', n_coincidences=1,
                            threshold_high=None, threshold_low=None, high_low_window=5., coinc_window=200., amp_per_ray=False,
                            d_max_distance=None, focusing=False, focusing_limit=2., select_only=False, reuse_ray_tables=False,
                            accumulate_triggered=False, n_reflections=0, z_reflection=0., reflection_coefficient=1.,
                            reflection_phase_shift=0., split_event_time_diff=0., noise=False, noise_seed=0, noise_group_offset=0,
                            polarization='auto', ePhi=0.,
                            d_noise_group_id=None, emit_traces=False, emit_capacity_samples=0, d_given_C0=None, d_given_D=None,
                            d_given_T=None):
        """Device-pointer form (ints): everything stays in HBM.  Returns the stats dict (or None).
        Event groups of several showers: d_group_begin = device int32 [n_groups + 1] (first shower of every group),
        d_triggered then has n_groups entries; d_vertex_time = device f64 [n_events] or None.
        trigger: 'simple' (|V| >= trigger_threshold, simpleThreshold.py) or 'high_low' (highLowThreshold.py: threshold_high /
        threshold_low inside high_low_window), both followed by the majority logic over coinc_window with n_coincidences;
        'phased_array' (set_phased_array first): trigger_threshold is the threshold on the beams' mean window power;
        'envelope' (set_envelope_trigger first): Hilbert envelope of the band-passed trace > trigger_threshold, then the majority logic.
        select_only: stop after ray tracing and the delta_C cut (then fetch('shower_first_channel')); reuse_ray_tables:
        continue from the tables of such a call on the same device arrays; accumulate_triggered: OR into d_triggered
        instead of overwriting it (nrhip_sim_config).  n_reflections > 0: rays reflected off the bottom of an ice shelf at depth
        z_reflection (< 0) with the layer's reflection_coefficient and reflection_phase_shift [rad] (medium.reflection...).
        polarization='custom', ePhi=: config signal.polarization / signal.ePhi (simulation.calculate_polarization_vector).
        split_event_time_diff > 0 [ns]: simulation.group_into_events -- a group's signals at this station are cut into sub-events
        where consecutive start times are farther apart; the ev_* / item_* tables are then per sub-event (fetch('ev_group'),
        fetch('ev_sub_event')), the mask stays per group.
        noise (set_noise first): thermal noise on every channel of the candidate events before filters and trigger; counter-based
        draws keyed by noise_seed and the group ids (d_noise_group_id: device int64 [n_groups], or noise_group_offset + index).
        emit_traces (production mode, trigger 'simple' without coincidences): the convolution kernel writes the traces of all
        channels of every event the moment it triggers (nrhip_sim_config.emit_triggered_traces) -- triggered_traces() returns
        them; stats['n_emitted_events'] == stats['n_triggered'] and stats['n_emit_overflow'] == 0 say that every triggered event
        got its block (otherwise run those through dump_traces / triggered_pass_dev).
        d_given_C0: device f64 [n_events * n_channels][2] (NaN = none) -- the rays' launch parameters are GIVEN (set_solution,
        analyticraytracing.py:2092), the root search does not run (nrhip_sim_config.given_C0); d_given_D / d_given_T (same layout):
        their path lengths / travel times taken as given, too."""
        if polarization not in ('auto', 'custom'):   # simulation.py:827-829
            raise ValueError("{} for config.signal.polarization is not a valid option".format(polarization))
        if trigger not in ('simple', 'high_low', 'phased_array', 'envelope'):
            raise NotImplementedError("trigger {} is not provided (simple, high_low, phased_array, envelope)".format(trigger))
        if (trigger == 'high_low' or int(n_coincidences) > 1) and len(self.position) > 255:
            # (the convolution kernel counts the coinciding channels of a sample in one byte)
            raise NotImplementedError("coincidence triggers take stations of at most 255 channels")
        cfg = SimConfig(ASKARYAN_TO_INT[askaryan_model], float(delta_C_cut),
                        float(2.0 * self.vrms_efield if min_efield_amplitude is None else min_efield_amplitude),
                        float(3.0 * self.vrms if trigger_threshold is None else trigger_threshold), int(bool(dump_traces)),
                        int(bool(no_pruning)), {'simple': 0, 'high_low': 1, 'phased_array': 2, 'envelope': 3}[trigger], int(n_coincidences),
                        float(3.0 * self.vrms if threshold_high is None else threshold_high),
                        float(-3.0 * self.vrms if threshold_low is None else threshold_low), float(high_low_window),
                        float(coinc_window), int(bool(amp_per_ray)), int(bool(focusing)), float(focusing_limit),
                        int(bool(select_only)), int(bool(reuse_ray_tables)), int(bool(accumulate_triggered)), int(n_reflections),
                        float(z_reflection), float(reflection_coefficient), float(reflection_phase_shift),
                        float(split_event_time_diff or 0.), int(bool(noise)), int(noise_seed) & 0xffffffffffffffff,
                        int(noise_group_offset), d_noise_group_id, int(polarization == 'custom'), float(ePhi),
                        int(bool(emit_traces)), int(emit_capacity_samples), d_given_C0, d_given_D, d_given_T)
        stats = SimStats()
        L.check(self._lib.nrhip_simulate_event_groups(
            self.ctx._h, self._h, ctypes.byref(cfg), int(n_events), d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL,
            d_vertex_time, d_max_distance, int(n_events if n_groups is None else n_groups), d_group_begin, d_triggered,
            ctypes.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def triggered_pass_dev(self, n_events, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL, d_triggered, n_groups=None,
                           d_group_begin=None, d_vertex_time=None, d_max_distance=None, amp_per_ray=False, **kw):
        """Pass 2 of a survey without a host round trip of the list: the showers of the event groups flagged in the DEVICE mask
        d_triggered (what simulate_events_dev left there) are gathered into a compact list in HBM (nrhip_select_groups /
        nrhip_gather_groups) and run again with dump_traces -- every channel of every candidate event is evaluated and its trace
        kept, which is what the reference stores for triggered events (output_writer_hdf5.py:215-320; amp_per_ray adds the
        per-ray envelope maxima).  Afterwards fetch('trace'), fetch('trace_offset'), fetch('item_event'), fetch('ev_trigger_bin'),
        ... describe the compact list; returns (stats of the pass, d_keep_index: DEVICE int32 [n_selected] original group of every
        compact group, n_selected).  The compact copies live in a scratch owned by the station (grown on demand)."""
        from . import comm as _comm   # (registers the signatures of the list helpers)
        ctx, lib = self.ctx, self._lib
        n_groups = int(n_events if n_groups is None else n_groups)
        sc = getattr(self, '_pass2', None)
        if sc is None or sc['n'] < n_events or sc['g'] < n_groups:
            if sc is not None:
                for q in sc['p']:
                    ctx.free(q)
            cap_n, cap_g = int(n_events), int(n_groups)
            vp = ctx.malloc
            ptrs = [vp(24 * cap_n), vp(8 * cap_n), vp(8 * cap_n), vp(8 * cap_n), vp(4 * cap_n), vp(8 * cap_n), vp(8 * cap_n),
                    vp(8 * cap_n), vp(4 * cap_g), vp(4 * (cap_g + 1)), vp(max(cap_g, 1)), vp(8 * max(cap_g, 1))]
            sc = self._pass2 = dict(n=cap_n, g=cap_g, p=ptrs)
        (s_vertex, s_zen, s_az, s_en, s_type, s_kL, s_md, s_vt, s_keep, s_gb, s_trig, s_gid64) = sc['p']
        nk, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        L.check(lib.nrhip_select_groups(ctx._h, int(n_events), n_groups, d_group_begin, ctypes.c_void_p(d_triggered), s_keep, s_gb,
                                        ctypes.byref(nk), ctypes.byref(ns)))
        nk, ns = int(nk.value), int(ns.value)
        if nk == 0:
            return None, s_keep, 0
        L.check(lib.nrhip_gather_groups(ctx._h, nk, s_keep, d_group_begin, s_gb, d_vertex, d_zenith, d_azimuth, d_energy, d_type,
                                        d_kL, d_vertex_time, d_max_distance, s_vertex, s_zen, s_az, s_en, s_type, s_kL,
                                        s_vt if d_vertex_time is not None else None, s_md if d_max_distance is not None else None,
                                        None))
        if kw.get('noise'):
            # the noise of an event group is keyed by its ORIGINAL id (the pass that decided the trigger drew it under that id):
            # the compact list carries the ids of the groups it was gathered from, not its own running index
            kw = dict(kw)
            d_ids = kw.pop('d_noise_group_id', None)
            off = int(kw.pop('noise_group_offset', 0))
            if d_ids is not None:
                L.check(lib.nrhip_gather_i64(ctx._h, nk, s_keep, ctypes.c_void_p(d_ids), s_gid64))
            else:
                L.check(lib.nrhip_index_to_i64(ctx._h, nk, s_keep, off, s_gid64))
            kw['d_noise_group_id'] = s_gid64
        stats = self.simulate_events_dev(ns, s_vertex, s_zen, s_az, s_en, s_type, s_kL, s_trig, dump_traces=True,
                                         amp_per_ray=amp_per_ray, n_groups=nk,
                                         d_group_begin=s_gb if d_group_begin is not None else None,
                                         d_vertex_time=s_vt if d_vertex_time is not None else None,
                                         d_max_distance=s_md if d_max_distance is not None else None, **kw)
        return stats, s_keep, nk

    def simulate_events(self, vertex, zenith, azimuth, energy, shower_type, k_L=None, vertex_time=None, group_id=None,
                        distance_cut_coefficients=None, distance_cut_sum_length=10., arz_iN=None, max_showers_per_call=None,
                        seed=None, rng=None, given_C0=None, given_D=None, given_T=None, **kw):
        """Host-array convenience form: uploads the shower list, runs the hot path, returns (triggered mask, stats).
        Long lists are cut into calls of at most `max_showers_per_call` showers at event-group boundaries (default: what
        keeps the per-call tables near 40 GB: ~1.2e7 (shower, channel) pairs, 2.5e5 with ARZ / birefringence, whose rays carry
        spectra and traces); the mask is concatenated, the counters of `stats` are summed (the fetchable tables are those of
        the last call).
        `group_id` [n] (equal ids consecutive, like the event_group_ids of the reference's input files) makes showers of
        one id a single event group: their signals add up in the channels (simulation.py:143) and the mask has one entry
        per group, in order of first appearance.  `vertex_time` [n] shifts a shower's signals (simulation.py:259-268).
        Random shower parameters -- k_L of Alvarez2009 EM showers (NaN / None entries), the ARZ profile numbers (arz_iN
        None) -- are drawn as the reference draws them when `seed` (or `rng`, a np.random.RandomState) is given: a first pass
        traces the rays, the host walks the showers in the reference's order (sequencing.reference_draw_order) drawing from
        RandomState(seed), the second pass reuses the ray tables; stats['k_L'] / stats['arz_iN'] return what was used.  Without
        a seed missing values are an error (the reference never runs an EM shower with k_L = 1).
        given_C0 [n, n_channels, 2] (NaN = none): the launch parameters of the rays are given (ray_tracing.set_solution), the
        root search does not run -- e.g. the reference's own rays; given_D / given_T (same shape, NaN = computed): their path
        lengths and travel times as given (nrhip_sim_config.given_D)."""
        ctx = self.ctx
        vertex = L.f64(vertex).reshape(-1, 3)
        n = len(vertex)
        if max_showers_per_call is None:
            general = kw.get('askaryan_model') in ('ARZ2019', 'ARZ2020') or getattr(self, '_birefringence_on', False)
            max_showers_per_call = max(1, int((2.5e5 if general else 1.2e7) / len(self.position)))
        if rng is None and seed is not None:
            rng = np.random.RandomState(seed)
        if given_C0 is not None:
            given_C0 = np.ascontiguousarray(L.f64(given_C0).reshape(n, len(self.position), 2))
            given_D, given_T = (None if a is None else np.ascontiguousarray(L.f64(a).reshape(n, len(self.position), 2)) for a in (given_D, given_T))
            max_showers_per_call = max(max_showers_per_call, n)   # (one call: the table is indexed by the shower)
        if n > max_showers_per_call:
            return self._simulate_in_chunks(int(max_showers_per_call), vertex, zenith, azimuth, energy, shower_type, k_L,
                                            vertex_time, group_id, distance_cut_coefficients, distance_cut_sum_length, arz_iN,
                                            dict(kw, rng=rng))
        n_groups, gb, vt = n, None, None
        if group_id is not None and n:
            gid = np.asarray(group_id).reshape(-1)
            if len(gid) != n:
                raise ValueError("group_id must have one entry per shower")
            first = np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
            if len(np.unique(gid)) != len(first):
                raise ValueError("showers of one event group must be consecutive")
            gb = np.ascontiguousarray(np.concatenate([first, [n]]), dtype=np.int32)
            n_groups = len(first)
        if vertex_time is not None:
            vt = np.ascontiguousarray(np.broadcast_to(L.f64(vertex_time), (n,)))
        md = None
        if distance_cut_coefficients is not None and n:
            md = distance_cut(vertex, np.broadcast_to(L.f64(energy), (n,)), gb, distance_cut_coefficients,
                              distance_cut_sum_length)
        st = _shower_type_codes(shower_type, n)
        model = kw.get('askaryan_model', 'Alvarez2009')
        is_arz = model in ('ARZ2019', 'ARZ2020')
        en = np.ascontiguousarray(np.broadcast_to(L.f64(energy), (n,)))
        if is_arz and getattr(self, '_arz', None) is None:
            raise ValueError("the ARZ models need a shower library: Station.set_arz(nuradiomc_amd.arz.ARZ(library=...))")
        kL = np.array(np.broadcast_to(np.nan if k_L is None else L.f64(k_L), (n,)), dtype=np.float64)
        need_kL = model == 'Alvarez2009' and bool(np.any(np.isnan(kL) & (st == SHOWER_TO_INT['EM'])))
        need_iN = is_arz and arz_iN is None
        if (need_kL or need_iN) and rng is None:
            raise ValueError("the ARZ models need the profile number of every shower (arz_iN), or a seed to draw them"
                             if need_iN else
                             "Alvarez2009 needs k_L for every electromagnetic shower (parametrizations.py:160-173), or a seed "
                             "to draw them in the reference's order")
        two_phase = (need_kL or need_iN) and n > 0
        arrs = [vertex, np.ascontiguousarray(np.broadcast_to(L.f64(zenith), (n,))),
                np.ascontiguousarray(np.broadcast_to(L.f64(azimuth), (n,))), en, st,
                np.ascontiguousarray(np.where(np.isnan(kL), 1.0, kL))]
        dptrs = [ctx.to_device(a) for a in arrs]
        dtrig = ctx.malloc(max(n_groups, 1))
        extra = [ctx.to_device(a) if a is not None else None for a in (vt, gb, md, given_C0, given_D, given_T)]
        dev_kw = dict(d_vertex_time=extra[0], n_groups=n_groups, d_group_begin=extra[1], d_max_distance=extra[2],
                      d_given_C0=extra[3], d_given_D=extra[4], d_given_T=extra[5])
        if kw.get('noise'):   # the noise of an event group is keyed by its id: the caller's group ids, else the running index
            off = int(kw.pop('noise_group_offset', 0))
            ids = gid[first] if group_id is not None and n else np.arange(n_groups) + off
            extra.append(ctx.to_device(np.ascontiguousarray(ids, dtype=np.int64)))
            dev_kw['d_noise_group_id'] = extra[-1]
        try:
            if two_phase:
                from . import sequencing
                self.simulate_events_dev(n, *dptrs, dtrig, select_only=True, **dev_kw, **kw)
                order = sequencing.reference_draw_order(self.fetch('shower_first_channel'), gb)
                if need_kL:
                    kL = sequencing.draw_k_L(kL, en, st, order, rng)
                    host = np.ascontiguousarray(np.where(np.isnan(kL), 1.0, kL))   # showers without a ray: never read
                    ctx.copy_to_device(dptrs[5], host)
                if need_iN:
                    arz_iN = np.zeros(n, np.int64)
                    arz_iN[order] = self._arz.draw_profile_numbers(en[order], ['HAD' if c == 0 else 'EM' for c in st[order]])
            if is_arz:   # the profile number of every shower -> library row and E / E_library
                rows, resc = self._arz_shower_profiles(en, st, np.broadcast_to(arz_iN, (n,)))
                L.check(self._lib.nrhip_station_set_shower_profiles(self._h, n, L.iptr(rows), L.dptr(resc)))
            stats = self.simulate_events_dev(n, *dptrs, dtrig, reuse_ray_tables=two_phase, **dev_kw, **kw)
            trig = np.zeros(n_groups, np.uint8)
            ctx.to_host(trig, dtrig)
        finally:
            for p in dptrs + [dtrig] + [e for e in extra if e is not None]:
                ctx.free(p)
        if stats is not None and rng is not None:   # what the showers were simulated with (NaN k_L: shower without any ray)
            if model == 'Alvarez2009':
                stats['k_L'] = kL
            if is_arz:
                stats['arz_iN'] = np.array(np.broadcast_to(arz_iN, (n,)), np.int64)
        return trig.astype(bool), stats

    def _simulate_in_chunks(self, max_showers, vertex, zenith, azimuth, energy, shower_type, k_L, vertex_time, group_id,
                            dcc, dcs, arz_iN, kw):
        n = len(vertex)
        per = lambda a: None if a is None else np.broadcast_to(np.asarray(a), (n,))
        zenith, azimuth, energy, k_L, vertex_time, arz_iN = (per(a) for a in (zenith, azimuth, energy, k_L, vertex_time, arz_iN))
        types = _shower_type_codes(shower_type, n)
        gid = None if group_id is None else np.asarray(group_id).reshape(-1)
        starts = np.arange(n) if gid is None else np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
        cuts, a = [0], 0
        while a < n:   # the last group boundary at most max_showers further on (a group longer than that goes alone)
            k = np.searchsorted(starts, a + max_showers, side='right') - 1
            b = int(starts[k]) if k < len(starts) and starts[k] > a else (int(starts[k + 1]) if k + 1 < len(starts) else n)
            if a + max_showers >= n:
                b = n
            cuts.append(b)
            a = b
        trig, total = [], None
        for a, b in zip(cuts[:-1], cuts[1:]):
            sl = slice(a, b)
            kw_c = dict(kw, noise_group_offset=int(kw.get('noise_group_offset', 0)) + a) if (kw.get('noise') and gid is None) else kw
            t, s_ = self.simulate_events(vertex[sl], zenith[sl], azimuth[sl], energy[sl], types[sl],
                                         None if k_L is None else k_L[sl], None if vertex_time is None else vertex_time[sl],
                                         None if gid is None else gid[sl], dcc, dcs, None if arz_iN is None else arz_iN[sl],
                                         max_showers_per_call=max(b - a, 1), **kw_c)
            trig.append(t)
            if total is None:
                total = s_
            elif s_ is not None:
                for k_, v_ in s_.items():
                    if k_ in ('k_L', 'arz_iN'):
                        total[k_] = np.concatenate([total[k_], v_])
                    elif k_ == 'stage_ms':
                        total[k_] = {q: total[k_][q] + v_[q] for q in v_}
                    elif k_ in ('max_length', 'n_distinct_lengths'):
                        total[k_] = max(total[k_], v_)
                    else:
                        total[k_] += v_
        return np.concatenate(trig), total

    def common_time_grid(self, t0, channel):
        """t_min and L of efieldToVoltageConverter.run (efieldToVoltageConverter.py:120-169)"""
        t0 = np.asarray(t0, float) + self.cable_delay[np.asarray(channel, int)]
        times_min = np.min(t0)
        times_max = np.max(t0 + self.n_samples / self.sampling_rate)
        max_len = self.readout_length
        times_min -= self.pre_pulse_time
        times_max += self.post_pulse_time
        while times_max - times_min < max_len:
            times_max += self.post_pulse_time
        res = 1. / self.sampling_rate
        n = int(round((times_max - times_min) / res))
        if n % 2 != 0:
            n += 1
        return times_min, n

    def efield_to_voltage(self, traces, t0, zenith, azimuth, channel, apply_filters=False, grid=None):
        """Channel voltages of ONE station event from arbitrary efield traces [n, 2 (eTheta, ePhi), n_samples];
        returns (V [n_channels, L], t_min).  grid = (t_min, L): that time grid instead of the common one of
        efieldToVoltageConverter (used by the per-efield converter: the efield's own N-sample grid)."""
        traces = L.f64(traces)
        n = traces.shape[0]
        if n == 0:
            raise LookupError("station has no efields")
        if traces.shape[1:] != (2, self.n_samples):
            raise ValueError("traces must be [n, 2, %d]" % self.n_samples)
        t0, zenith, azimuth = (np.ascontiguousarray(np.broadcast_to(L.f64(a), (n,))) for a in (t0, zenith, azimuth))
        ch = np.ascontiguousarray(np.broadcast_to(channel, (n,)), dtype=np.int32)
        t_min, Lc = self.common_time_grid(t0, ch) if grid is None else (float(grid[0]), int(grid[1]))
        V = np.zeros((len(self.position), Lc))
        L.check(self._lib.nrhip_efield_to_voltage(self.ctx._h, self._h, n, L.dptr(traces), L.dptr(t0), L.dptr(zenith),
                                                  L.dptr(azimuth), L.iptr(ch), int(bool(apply_filters)), Lc, float(t_min),
                                                  L.dptr(V)))
        return V, t_min

    _FETCH_DTYPES = {'ray_event': np.int32, 'ray_channel': np.int32, 'ray_solution': np.int32, 'ev_n_rays': np.int32,
                     'ev_L': np.int32, 'ev_candidate': np.uint8, 'ev_trigger_bin': np.int32, 'item_event': np.int32, 'trace_offset': np.int64,
                     'ray_r_theta': np.complex128, 'ray_r_phi': np.complex128, 'lengths': np.int32,
                     'pair_n_sol': np.int32, 'slot_type': np.int32, 'ev_ray_begin': np.int32, 'ray_active': np.int32,
                     'ray_active_list': np.int32, 'ray_slot': np.int32, 'slot_keep': np.int32, 'slot_offset': np.int32,
                     'shower_first_channel': np.int32, 'slot_reflection': np.int32, 'slot_reflection_case': np.int32,
                     'slot_n_segments': np.int32, 'slot_surface_mask': np.int32, 'ev_group': np.int32, 'ev_sub_event': np.int32,
                     'ev_triggered': np.uint8, 'ray_sub_event': np.int32, 'group_n_sub_events': np.int32,
                     'pa_digital_length': np.int32, 'ray_propagated': np.int32, 'gen_n_steps': np.int32, 'emit_offset': np.int64}

    def triggered_traces(self):
        """after simulate_events_dev(..., emit_traces=True): {event index: array [n_channels, L]} of the events whose traces the
        convolution kernel wrote when they triggered"""
        off = self.fetch('emit_offset')
        L_ = self.fetch('ev_L')
        tr = self.fetch('emit_trace')
        n_ch = len(self.position)
        return {int(e): tr[off[e]:off[e] + n_ch * L_[e]].reshape(n_ch, L_[e]) for e in np.flatnonzero(off >= 0)}

    def readout_windows(self, n_window, pre_bins, threshold):
        """after a call with dump_traces: per candidate event (fetch('item_event') order) the trigger bin of the simple threshold on any
        channel, and per channel max |V| and the maximum Hilbert envelope of the read-out window of n_window samples that starts
        pre_bins before the trigger -- computed on the device (nrhip_readout_windows), the traces stay there.  -> (trigger_bin [n],
        max_amp [n, n_channels], max_env [n, n_channels]); NaN rows: no trigger, or a common trace shorter than the window."""
        n = self.fetch_bytes('item_event') // 4
        n_ch = len(self.position)
        tb, amp, env = np.zeros(n, np.int32), np.zeros((n, n_ch)), np.zeros((n, n_ch))
        L.check(self._lib.nrhip_readout_windows(self.ctx._h, self._h, int(n_window), int(pre_bins), float(threshold), n, L.iptr(tb),
                                                L.dptr(amp), L.dptr(env)))
        return tb, amp, env

    def fetch_bytes(self, name):
        """size in bytes of a table of the last simulated batch (nothing is copied)"""
        n = self._lib.nrhip_sim_fetch(self._h, name.encode(), None, 0)
        if n < 0:
            raise L.NrhipError(self._lib.nrhip_last_error().decode())
        return int(n)

    def fetch(self, name):
        """One table of the last simulated batch as a numpy array (see include/nrhip.h: nrhip_sim_fetch)."""
        n = self._lib.nrhip_sim_fetch(self._h, name.encode(), None, 0)
        if n < 0:
            raise L.NrhipError(self._lib.nrhip_last_error().decode())
        dt = np.dtype(self._FETCH_DTYPES.get(name, np.float64))
        raw = np.empty(n, np.uint8)   # (filled by the copy below: no zeroing pass over what may be gigabytes of traces)
        if n:
            got = self._lib.nrhip_sim_fetch(self._h, name.encode(), raw.ctypes.data_as(ctypes.c_void_p), n)
            if got != n:   # a failed copy (or a table replaced in between) must not hand out uninitialised memory
                raise L.NrhipError(self._lib.nrhip_last_error().decode() if got < 0 else
                                   "nrhip_sim_fetch(%s): %d bytes copied, %d expected" % (name, got, n))
        return raw[:(n // dt.itemsize) * dt.itemsize].view(dt)
