"""Event-list input and HDF5-layout output of a simulation -- the file formats either side of the hot path (SURVEY.md section 8,
rows f2 / f3).

* `EventList`: the datasets and attributes of the reference's input files (NuRadioMC/EvtGen/generator.py:1023-1414 writes
  them, simulation.read_input_hdf5 :1019-1057 and build_NuRadioEvents_from_hdf5 :659-762 read them): one row per shower,
  `event_group_ids` tying the showers of one neutrino together, the first row of a group being the primary interaction.
* `simulate_to_output`: the sequence of simulation.run() (:1454-1728) around the device path -- particle weights
  (calculate_particle_weight :852-903), the hot path over all stations, then for the triggered (station, event group) pairs
  what the reference stores: channelReadoutWindowCutter (window of the detector's samples starting pre_trigger_time before
  the trigger), channelSignalReconstructor (maximum amplitude and Hilbert-envelope maximum of the windowed traces) and the
  per (shower, channel, ray solution) tables of output_writer_hdf5.add_event_group (:95-430), in the dataset names, shapes,
  dtypes, NaN padding and shower-id ordering of write_output_file (:449-527).
* `OutputFile`: those datasets / attributes as a dictionary; `save_hdf5` needs h5py, `save_npz` does not -- tools/npz_to_hdf5.py
  turns the latter into the former 1:1 on any machine that has h5py (this image's default interpreter has none).
"""
import json
import os
import numpy as np
from . import _lib as L
from .station import Station
from .array import StationArray

_SHOWER_KEYS = ('event_group_ids', 'shower_ids', 'xx', 'yy', 'zz', 'zeniths', 'azimuths', 'energies', 'shower_energies',
                'shower_type', 'flavors', 'n_interaction', 'interaction_type', 'inelasticity', 'vertex_times')


def _str(a):
    a = np.asarray(a)
    return a.astype(str) if a.dtype.kind in 'SO' else a


class EventList:
    """The reference's input event list: `data` (one entry per shower) and `attrs` (generator attributes)."""

    def __init__(self, data, attrs=None):
        self.data = {k: (_str(v) if np.asarray(v).dtype.kind in 'SUO' else np.asarray(v)) for k, v in data.items()}
        self.attrs = dict(attrs or {})
        for k in ('event_group_ids', 'xx', 'yy', 'zz', 'zeniths', 'azimuths', 'shower_energies', 'shower_type'):
            if k not in self.data:
                raise KeyError("event list without '{}'".format(k))
        n = len(self.data['event_group_ids'])
        if 'shower_ids' not in self.data:
            self.data['shower_ids'] = np.arange(n)
        if 'vertex_times' not in self.data:   # simulation.py:712-716: "setting vertex time to zero"
            self.data['vertex_times'] = np.zeros(n)
        gid = self.data['event_group_ids']
        first = np.concatenate([[True], gid[1:] != gid[:-1]])
        if len(np.unique(gid)) != first.sum():
            raise ValueError("showers of one event group must be consecutive")

    @classmethod
    def from_hdf5(cls, path):
        """simulation.read_input_hdf5 (:1019-1057): every dataset and attribute of the file"""
        import h5py
        with h5py.File(path, 'r') as f:
            data = {k: np.array(v) for k, v in f.items() if isinstance(v, h5py.Dataset)}
            attrs = {k: v for k, v in f.attrs.items()}
        return cls(data, attrs)

    @classmethod
    def from_npz(cls, path):
        """the h5py-free twin: datasets under their names, attributes as 'attr/<name>'"""
        g = np.load(path)
        return cls({k: g[k] for k in g.files if not k.startswith('attr/')},
                   {k[5:]: g[k][()] for k in g.files if k.startswith('attr/')})

    def __len__(self):
        return len(self.data['event_group_ids'])

    @property
    def vertex(self):
        return np.stack([self.data['xx'], self.data['yy'], self.data['zz']], axis=1).astype(float)

    def shower_type_codes(self):
        a = np.asarray(self.data['shower_type'])
        if a.dtype.kind != 'U':
            a = a.astype(str)
        # the exact spellings first (a C loop each); whatever is left goes through np.char.lower, which on 1e6 strings would cost
        # more than the hot path
        codes = np.full(len(a), -1, np.int32)
        codes[(a == 'had') | (a == 'HAD')] = 0
        codes[(a == 'em') | (a == 'EM')] = 1
        rest = np.flatnonzero(codes < 0)
        if len(rest):
            t = np.char.lower(a[rest])
            codes[rest] = np.where(t == 'had', 0, np.where(t == 'em', 1, -1))
            if np.any(codes[rest] < 0):
                raise KeyError(str(t[codes[rest] < 0][0]))
        return codes

    def k_L(self):
        """stored shower realisations (input files of re-simulations) or NaN"""
        return np.asarray(self.data.get('shower_realization_Alvarez2009', np.full(len(self), np.nan)), float)


class OutputFile:
    def __init__(self):
        self.datasets, self.attrs = {}, {}     # 'name' / 'station_<id>/name' -> array; ('', name) / ('station_<id>', name) -> value

    def save_npz(self, path):
        out = dict(self.datasets)
        for (grp, name), v in self.attrs.items():
            out['attr/%s@%s' % (grp, name)] = np.asarray(v.encode() if isinstance(v, str) else v)
        for k, v in list(out.items()):
            if np.asarray(v).dtype.kind == 'U':
                out[k] = np.asarray(v).astype('S')
        np.savez_compressed(path, **out)

    def save_hdf5(self, path):
        import h5py
        write_hdf5(h5py, path, self.datasets, {('%s@%s' % k): v for k, v in self.attrs.items()})


def write_hdf5(h5py, path, datasets, attrs):
    """datasets: 'group/name' -> array; attrs: 'group@name' -> value ('' = file level); strings as variable-length UTF-8 like the
    reference (output_writer_hdf5.py:470-473)"""
    with h5py.File(path, 'w') as f:
        for k, v in datasets.items():
            v = np.asarray(v)
            if v.dtype.kind in 'SU':
                f[k] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v], dtype=h5py.string_dtype(encoding='utf-8'))
            else:
                f[k] = v
        for k, v in attrs.items():
            grp, name = k.split('@', 1)
            obj = f if grp == '' else f.require_group(grp)
            v = np.asarray(v)
            if v.dtype.kind == 'S':
                v = v.astype(str)
            obj.attrs[name] = v.tolist() if (v.dtype.kind == 'U' and v.ndim) else (str(v[()]) if v.dtype.kind == 'U' else v)


def _hilbert_envelope(x, workers=None):
    """|scipy.signal.hilbert(x)| along the last axis (trace_utilities.get_hilbert_envelope).  The analytic signal of a real trace is
    x + i y with y = irfft(-i X_k, 0 < k < n / 2) -- two real transforms instead of the two complex ones of scipy.signal.hilbert, and
    real arrays in between."""
    n = x.shape[-1]
    try:
        from scipy import fft as _fft
        kw = dict(workers=workers or 1)
    except ImportError:   # numpy only
        _fft, kw = np.fft, {}
    X = _fft.rfft(x, axis=-1, **kw)
    X *= -1j
    X[..., 0] = 0.
    if n % 2 == 0:
        X[..., n // 2] = 0.
    y = _fft.irfft(X, n, axis=-1, **kw)
    y *= y
    y += x * x
    return np.sqrt(y, out=y)


def _window_maxima(windows, step=64):
    """maximum |V| and maximum Hilbert envelope per (event, channel) of a list of equally shaped read-out windows [n_channels,
    n_samples]: chunks of `step` events on a pool of threads (numpy's element-wise passes and pocketfft release the GIL; one thread
    after the other, these passes over ~1.5 GB of windows were 40 % of the end-to-end time of a 1e6-event list)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(windows)
    amp = np.empty((n,) + windows[0].shape[:-1])
    env = np.empty_like(amp)

    def work(a0):
        Wb = np.array(windows[a0:a0 + step])
        np.max(np.abs(Wb), axis=-1, out=amp[a0:a0 + step])
        np.max(_hilbert_envelope(Wb), axis=-1, out=env[a0:a0 + step])
    starts = list(range(0, n, step))
    n_thr = max(1, min(16, len(os.sched_getaffinity(0)), len(starts)))
    if n_thr == 1:
        for a0 in starts:
            work(a0)
    else:
        with ThreadPoolExecutor(n_thr) as pool:
            list(pool.map(work, starts))
    return amp, env


def _host_windows(trace, trace_offset, ev_L, item_event, n_ch, threshold, n_window, pre_bins):
    """the same three tables as Station.readout_windows from traces fetched to the host (read-out windows that are no power of two,
    common traces shorter than the window): trigger bin, max |V| and Hilbert-envelope maximum per (item, channel)"""
    n = len(item_event)
    tb = np.full(n, -1, np.int32)
    wins, idx = [], []
    amp, env = np.full((n, n_ch), np.nan), np.full((n, n_ch), np.nan)
    for it in range(n):
        L_ = int(ev_L[item_event[it]])
        V = np.array([trace[trace_offset[it * n_ch + c]:trace_offset[it * n_ch + c + 1]] for c in range(n_ch)])
        hit = np.any(np.abs(V[:, :L_ - 1]) >= threshold, axis=0)   # get_majority_logic drops the last sample
        if not hit.any():
            continue
        tb[it] = int(np.argmax(hit))
        wins.append(_readout_window(V, int(tb[it]), n_window, pre_bins))
        idx.append(it)
    by_len = {}
    for k_, W_ in enumerate(wins):   # (a common trace shorter than the read-out window keeps its own length)
        by_len.setdefault(W_.shape, []).append(k_)
    for ks in by_len.values():
        a_, e_ = _window_maxima([wins[k_] for k_ in ks])
        for q_, k_ in enumerate(ks):
            amp[idx[k_]], env[idx[k_]] = a_[q_], e_[q_]
    return tb, amp, env


def _readout_window(V, trigger_bin, n_window, pre_bins):
    """channelReadoutWindowCutter.run (:28-137): the trace rolled so that the window starts pre_trigger_time before the trigger
    (whole samples here: roll; apply_time_shift on a whole number of samples is np.roll, base_trace.py:262-266), first n_window
    samples -- np.roll(V, -(trigger_bin - pre_bins), axis=-1)[..., :n_window] without the rolled copy of the whole trace"""
    L_ = V.shape[-1]
    n = min(n_window, L_)
    s0 = (trigger_bin - pre_bins) % L_
    if s0 + n <= L_:
        return V[..., s0:s0 + n]
    return np.concatenate([V[..., s0:], V[..., :s0 + n - L_]], axis=-1)


def simulate_to_output(det, events, config=None, station_ids=None, trigger_name='simple_threshold', seed=None,
                       detector_n_samples=None, detector_sampling_rate=None, pre_trigger_time=55., weight_mode='core_mantle_crust',
                       cross_section_type='ctw', minimum_weight_cut=None, noise_temperature=300., **sim_kw):
    """Run the event list through `det` (a Station or a StationArray) and return the OutputFile the reference would write (simple
    threshold trigger; amp_per_ray tables included unless the station cannot provide them)."""
    import time as _time
    _t = [_time.perf_counter()]
    timing = {}

    def _lap(name):
        now = _time.perf_counter()
        timing[name] = timing.get(name, 0.) + now - _t[0]
        _t[0] = now
    arr = det if isinstance(det, StationArray) else StationArray(det, np.zeros((1, 3)), relative_position=det.position, cull=False)
    st, ctx = arr.station, arr.station.ctx
    n_st, n_ch, nS = len(arr), len(st.position), 2
    station_ids = list(arr.station_ids if station_ids is None else station_ids)
    d = events.data
    n = len(events)
    gid = d['event_group_ids']
    first = np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
    gb = np.concatenate([first, [n]])
    n_groups = len(first)
    vertex, types = events.vertex, events.shower_type_codes()
    # ---- particle weights of the primaries (simulation.calculate_particle_weight :852-903)
    weights = np.ones(n_groups)
    if 'weights' in d and weight_mode == 'existing':
        weights = np.asarray(d['weights'], float)[first]
    elif weight_mode is not None:
        from . import earth_attenuation
        weights = earth_attenuation.get_weight(d['zeniths'][first], d['energies'][first], d['flavors'][first], mode=weight_mode,
                                               cross_section_type=cross_section_type, vertex_position=vertex[first],
                                               phi_nu=d['azimuths'][first], ctx=ctx)
    sim_groups = np.ones(n_groups, bool) if minimum_weight_cut is None else (weights >= minimum_weight_cut)   # run(): :1489-1492
    if sim_groups.all():   # (no weight cut: the list as it is -- no gathered copies of 1e6-row arrays)
        rows = np.arange(n)
        take = lambda a: a   # noqa: E731
    else:
        rows = np.flatnonzero(np.repeat(sim_groups, np.diff(gb)))
        take = lambda a: a[rows]   # noqa: E731
    args = (take(vertex), take(d['zeniths']), take(d['azimuths']), take(d['shower_energies']), take(types))
    kL_in = take(events.k_L())
    kw = dict(vertex_time=take(d['vertex_times']), group_id=take(gid), **sim_kw)
    _lap('weights_and_selection')
    # ---- pass 1: which (station, group) trigger
    trig, stats = arr.simulate_events(*args, kL_in, seed=seed, per_station=True, **kw)
    _lap('pass1_upload_and_hot_path')
    st_trig = stats['station_triggered']
    kL = stats.get('k_L', kL_in)
    g_of_row = np.repeat(np.arange(n_groups), np.diff(gb))[rows]      # original group index of every simulated shower
    sim_group_ids = np.flatnonzero(sim_groups)
    # ---- pass 2: everything the writer stores, for the triggered groups only
    fs = st.sampling_rate
    dt = 1. / fs
    det_fs = float(detector_sampling_rate or fs)
    n_det = int(detector_n_samples or st.n_samples)
    n_window = int(2 * np.ceil(n_det / 2 * fs / det_fs))          # channelReadoutWindowCutter._get_number_of_samples
    pre_bins = int(round(pre_trigger_time * fs))
    threshold = sim_kw.get('trigger_threshold', 3.0 * st.vrms)
    on_device = 16 <= n_window <= 8192 and (n_window & (n_window - 1)) == 0 and not os.environ.get('NRHIP_OUTPUT_HOST_WINDOWS')   # (nrhip_readout_windows: a power of two)
    sel_g = np.flatnonzero(trig)
    out = OutputFile()
    tables = {i: [] for i in range(n_st)}
    if len(sel_g):
        sub = np.flatnonzero(np.isin(g_of_row, sim_group_ids[sel_g]))
        sub_gid = gid[rows][sub]
        sub_first = np.flatnonzero(np.concatenate([[True], sub_gid[1:] != sub_gid[:-1]]))
        sub_gb = np.concatenate([sub_first, [len(sub)]])

        def collect(i, sl, s_, keep):
            idx = np.arange(len(sub_first)) if keep is None else np.asarray(keep)
            T = {k: s_.fetch(k) for k in ('pair_n_sol', 'slot_type', 'slot_C0', 'slot_C1', 'slot_D', 'slot_T', 'slot_launch',
                                                 'slot_keep', 'ray_slot', 'ray_zenith', 'ray_azimuth', 'ray_pol_theta',
                                                 'ray_pol_phi', 'ev_candidate', 'ev_L', 'ev_t_min', 'ev_n_rays')}
            if T['ev_candidate'].any():
                T['item_event'] = s_.fetch('item_event')
                # trigger bin, maximum and Hilbert-envelope maximum of every read-out window: on the device, where the traces are
                # (2 GB for the 9053 triggered showers of a 1e6-event survey) -- only windows that are no power of two come to the host
                if on_device:
                    T['win_bin'], T['win_amp'], T['win_env'] = s_.readout_windows(n_window, pre_bins, threshold)
                if not on_device or np.any(np.isnan(T['win_amp'][T['win_bin'] >= 0])):
                    T['win_bin'], T['win_amp'], T['win_env'] = _host_windows(s_.fetch('trace'), s_.fetch('trace_offset'), T['ev_L'],
                                                                             T['item_event'], n_ch, threshold, n_window, pre_bins)
                for k in ('ray_max_amp_envelope', 'ray_signal_time'):
                    try:
                        T[k] = s_.fetch(k)
                    except L.NrhipError:
                        pass
            T['groups'] = idx
            tables[i].append(T)
        kw2 = dict(vertex_time=d['vertex_times'][rows][sub], group_id=sub_gid, **sim_kw)
        can_amp = True   # (every path provides the per-ray envelopes since round 3)
        arr.simulate_events(vertex[rows][sub], d['zeniths'][rows][sub], d['azimuths'][rows][sub], d['shower_energies'][rows][sub],
                            types[rows][sub], np.where(np.isnan(kL[sub]), 1.0, kL[sub]), dump_traces=True, amp_per_ray=can_amp,
                            on_station=collect, max_showers_per_call=len(sub) + 1, **kw2)
    _lap('pass2_traces_and_tables_of_triggered_groups')
    # ---- assemble the reference's tables (array operations over all stored events / showers of a station call at once)
    sh_trig = np.zeros(n, bool)          # per original shower row: part of a triggered station event
    sh_t = np.full(n, np.nan)            #   its earliest trigger time
    sh_primary = np.zeros(n, bool)       #   stored as the primary of a triggered group (:392-430)
    for i in range(n_st):
        sname = 'station_%d' % station_ids[i]
        ev_blocks, sh_blocks = [], []
        for T in tables[i]:
            groups = np.asarray(T['groups'], int)
            if len(groups) == 0 or 'item_event' not in T:
                continue
            # showers of the (possibly culled) list the station ran on
            cnt = (sub_gb[groups + 1] - sub_gb[groups]).astype(int)
            local_gb = np.concatenate([[0], np.cumsum(cnt)]).astype(int)
            lst = np.repeat(sub_gb[groups], cnt) + (np.arange(local_gb[-1]) - np.repeat(local_gb[:-1], cnt))
            keep = T['slot_keep'][:len(lst) * n_ch * nS].reshape(len(lst), n_ch, nS).astype(bool)
            keep_any = keep.reshape(len(lst), -1).any(axis=1)
            ray_of_slot = np.full(len(lst) * n_ch * nS, -1)
            ray_of_slot[T['ray_slot']] = np.arange(len(T['ray_slot']))
            ray_of_slot = ray_of_slot.reshape(len(lst), n_ch, nS)
            # the station events that triggered: their item (candidate-event row), trigger bin and time
            ks = np.flatnonzero(st_trig[i, sel_g[groups]])
            if len(ks) == 0:
                continue
            item_of = np.full(len(groups), -1)
            item_of[T['item_event']] = np.arange(len(T['item_event']))
            its = item_of[ks]
            tbin = T['win_bin'][its]
            if np.any(its < 0) or np.any(tbin < 0):
                bad = ks[(its < 0) | (tbin < 0)][0]
                raise RuntimeError("station %d, event group %d: flagged as triggered in pass 1, but no sample of the pass-2 traces reaches "
                                   "the threshold" % (station_ids[i], int(gid[first[sim_group_ids[sel_g][groups[bad]]]])))
            t_trig = tbin * dt + T['ev_t_min'][ks]
            g_orig = sim_group_ids[sel_g][groups[ks]]
            ev_blocks.append(dict(event_group_ids=gid[first[g_orig]], event_ids=np.zeros(len(ks), int),
                                  maximum_amplitudes=T['win_amp'][its], maximum_amplitudes_envelope=T['win_env'][its],
                                  multiple_triggers_per_event=np.ones((len(ks), 1), bool), trigger_times_per_event=t_trig[:, None],
                                  triggered_per_event=np.ones(len(ks), bool)))
            sh_primary[first[g_orig]] = True   # the primary of a triggered group is stored even without a signal of its own
            c_k = cnt[ks]
            J = np.repeat(local_gb[ks], c_k) + (np.arange(c_k.sum()) - np.repeat(np.concatenate([[0], np.cumsum(c_k)[:-1]]), c_k))
            tt = np.repeat(t_trig, c_k)
            m = keep_any[J]   # a shower without any efield on this station is not part of the station's event (:983-1001)
            J, tt = J[m], tt[m]
            if len(J) == 0:
                continue
            row = rows[sub[lst[J]]]
            sh_trig[row] = True
            sh_t[row] = np.fmin(sh_t[row], tt)
            # the (channel, solution) tables of all stored showers of this call at once (NaN where no ray was kept)
            kj = keep[J]
            q = (J * (n_ch * nS))[:, None, None] + np.arange(n_ch * nS).reshape(n_ch, nS)
            ir = np.where(kj, ray_of_slot[J], 0)
            k3 = kj[:, :, :, None]

            def tab2(values):
                return np.where(kj, values, np.nan)
            blk = dict(shower_id=d['shower_ids'][row], event_group_id_per_shower=gid[row], event_id_per_shower=d['shower_ids'][row],
                       triggered=np.ones(len(J), bool), multiple_triggers=np.ones((len(J), 1), bool), trigger_times=tt[:, None])
            blk['travel_times'], blk['travel_distances'] = tab2(T['slot_T'][q]), tab2(T['slot_D'][q])
            blk['ray_tracing_C0'], blk['ray_tracing_C1'] = tab2(T['slot_C0'][q]), tab2(T['slot_C1'][q])
            blk['ray_tracing_solution_type'] = tab2(T['slot_type'][q].astype(float))
            blk['ray_tracing_reflection'], blk['ray_tracing_reflection_case'] = tab2(0.), tab2(1.)
            blk['focusing_factor'] = tab2(1.)
            blk['launch_vectors'] = np.where(k3, T['slot_launch'][(3 * q)[:, :, :, None] + np.arange(3)], np.nan)
            zen, az = T['ray_zenith'][ir], T['ray_azimuth'][ir]
            ct, st_, cp, sp = np.cos(zen), np.sin(zen), np.cos(az), np.sin(az)
            blk['receive_vectors'] = np.where(k3, np.stack([st_ * cp, st_ * sp, ct], axis=-1), np.nan)
            # polarisation angle on sky -> unit vector in the ground frame (output_writer_hdf5.py:289-297)
            a = np.arctan2(T['ray_pol_phi'][ir], T['ray_pol_theta'][ir])
            e_t, e_p = np.stack([ct * cp, ct * sp, -st_], axis=-1), np.stack([-sp, cp, np.zeros_like(sp)], axis=-1)
            blk['polarization'] = np.where(k3, np.cos(a)[..., None] * e_t + np.sin(a)[..., None] * e_p, np.nan)
            if 'ray_max_amp_envelope' in T:
                blk['max_amp_shower_and_ray'] = tab2(T['ray_max_amp_envelope'][ir])
                blk['time_shower_and_ray'] = tab2(T['ray_signal_time'][ir])
            else:
                blk['max_amp_shower_and_ray'], blk['time_shower_and_ray'] = tab2(np.nan), tab2(np.nan)
            sh_blocks.append(blk)
        if sh_blocks:
            order = np.argsort(np.concatenate([b_['shower_id'] for b_ in sh_blocks]), kind='stable')
            for key in sh_blocks[0]:
                out.datasets['%s/%s' % (sname, key)] = np.concatenate([b_[key] for b_ in sh_blocks])[order]
            for key in ev_blocks[0]:
                out.datasets['%s/%s' % (sname, key)] = np.concatenate([b_[key] for b_ in ev_blocks])
        vr = np.array([st.vrms_per_set[st.channel_filter_set[c]][0] for c in range(n_ch)])
        out.attrs[(sname, 'antenna_positions')] = arr.relative_position + arr.centres[i]
        if sh_blocks:
            out.attrs[(sname, 'Vrms')] = vr
            out.attrs[(sname, 'bandwidth')] = np.array([_bandwidth(st, c) for c in range(n_ch)])
            out.attrs[(sname, 'Vrms_trigger')] = np.zeros(0)
    stored = np.flatnonzero(sh_trig | sh_primary)
    srows = stored[np.argsort(d['shower_ids'][stored], kind='stable')]
    if len(srows):
        D = out.datasets
        for key in ('shower_ids', 'event_group_ids', 'xx', 'yy', 'zz', 'vertex_times', 'azimuths', 'zeniths', 'energies', 'flavors',
                    'n_interaction', 'interaction_type', 'inelasticity'):
            if key in d:
                D[key] = d[key][srows]
        only = sh_primary[srows] & ~sh_trig[srows]
        D['shower_energies'] = np.where(only, np.nan, d['shower_energies'][srows])
        D['shower_type'] = np.where(only, '', d['shower_type'][srows])
        g_idx = np.searchsorted(first, srows, side='right') - 1
        D['weights'] = weights[g_idx]
        k_all = np.full(n, np.nan)
        k_all[rows] = kL
        if sim_kw.get('askaryan_model', 'Alvarez2009') == 'Alvarez2009':
            D['shower_realization_Alvarez2009'] = np.where(only, np.nan, k_all[srows])
        D['triggered'] = sh_trig[srows]
        D['multiple_triggers'] = D['triggered'][:, None].copy()
        D['trigger_times'] = sh_t[srows][:, None]
    for k, v in events.attrs.items():
        out.attrs[('', k)] = v
    out.attrs[('', 'trigger_names')] = np.array([trigger_name])
    out.attrs[('', 'Vrms')] = st.vrms
    out.attrs[('', 'dt')] = dt
    out.attrs[('', 'Tnoise')] = float(noise_temperature)
    out.attrs[('', 'bandwidth')] = _bandwidth(st, 0)
    if config is not None:
        out.attrs[('', 'config')] = config if isinstance(config, str) else json.dumps(config)
    out.stats = stats
    _lap('assemble_output_tables_on_host')
    out.timing = timing   # seconds per phase of this call (bench.py --end-to-end)
    return out


def _bandwidth(st, channel):
    """integrated channel response int |H|^2 df on the reference's 10000-point grid (simulation.py:1301-1376)"""
    from . import filters as flt
    ff = np.linspace(0, 0.5 * st.sampling_rate, 10000)
    H = np.abs(flt.response(ff, st.filter_sets[st.channel_filter_set[channel]]))
    return float(np.sum(0.5 * (H[1:] ** 2 + H[:-1] ** 2) * np.diff(ff)))
