"""StationArray -- the station loop of simulation.run() (NuRadioMC/simulation/simulation.py:1454-1600) for an array of
identical stations (BASELINE configs 3-5: 35 RNO-G-like stations, 200 Gen2-like stations).

"each station is treated independently" (:1500): every event group is offered to every station -- ray tracing for all
its channels with the shower-energy dependent distance cut (:155-163), candidate flag, common time grid, channel voltages
and trigger per station -- and the group is kept when any station triggered.  The shower list of a chunk goes to HBM once
and serves every station; ONE Station object (tables, workspace) is moved through the array (Station.move_to).

Station-level selection: with the distance cut a group none of whose showers lies within (cut + station radius) of the station
centre cannot produce a ray on any channel (triangle inequality; the reference's own quick cut, :1503-1509).  Per station the
groups in range are gathered into a compact shower list on the device (nrhip_cull_groups / nrhip_gather_groups), the hot path
runs on that list and the triggered flags are OR-ed back (nrhip_mask_scatter_or): a shower meets a few of 200 stations, not all.

The stateful random shower parameters (k_L of Alvarez2009 EM showers, ARZ profile numbers) are drawn in the order in which
the reference's loops (group -> station -> channel -> shower) meet the showers: a first pass over all stations only traces
the rays (nrhip_sim_config.select_only), sequencing.reference_draw_order sorts, the host draws, the second pass simulates.
"""
import ctypes
import numpy as np
from . import _lib as L
from . import sequencing
from .station import Station, _shower_type_codes, distance_cut, SHOWER_TO_INT


class _Scratch:
    """compact per-station copies of a shower list in HBM (allocated once per call of the array)"""

    def __init__(self, ctx, n, n_groups, with_time):
        self.ctx = ctx
        vp = ctx.malloc
        self.vertex, self.zenith, self.azimuth, self.energy = vp(24 * n), vp(8 * n), vp(8 * n), vp(8 * n)
        self.type, self.kL, self.md = vp(4 * n), vp(8 * n), vp(8 * n)
        self.vt = vp(8 * n) if with_time else None
        self.keep, self.gb, self.sidx, self.trig = vp(4 * n_groups), vp(4 * (n_groups + 1)), vp(4 * n), vp(max(n_groups, 1))
        self.gid64 = vp(8 * max(n_groups, 1))    # original group ids of the compact list (keys of the thermal noise)

    def free(self):
        for p in (self.vertex, self.zenith, self.azimuth, self.energy, self.type, self.kL, self.md, self.vt, self.keep, self.gb,
                  self.sidx, self.trig, self.gid64):
            if p is not None:
                self.ctx.free(p)


class StationArray:
    def __init__(self, station, centres, relative_position=None, station_ids=None, cull=True):
        """station: the Station object that is moved through the array (antennas, orientations, cable delays, filters, tables);
        centres [n_st, 3]: absolute station positions (det.get_absolute_position); relative_position [n_ch, 3]: the channel
        positions inside a station (det.get_relative_position) -- channel c of station i sits at relative_position[c] +
        centres[i], the sum the reference forms (simulation.py:138).  Pass it explicitly: recovering it as station.position -
        centres[0] (the default) is off by an ulp of the station coordinate, and the reference's first ray root is sensitive to
        the last bit of its inputs.  cull: station-level selection of the groups in range (needs the
        distance cut; result-neutral)."""
        if not isinstance(station, Station):
            raise TypeError("StationArray needs a nuradiomc_amd.Station")
        self.station = station
        self.centres = L.f64(centres).reshape(-1, 3)
        self.relative_position = None if relative_position is None else L.f64(relative_position).reshape(-1, 3).copy()
        if self.relative_position is None:
            self.relative_position = station.position - self.centres[0]
        if self.relative_position.shape != station.position.shape:
            raise ValueError("relative_position must be [n_channels, 3] of the station")
        self.station_ids = list(range(len(self.centres))) if station_ids is None else list(station_ids)
        if len(self.station_ids) != len(self.centres):
            raise ValueError("one station id per centre")
        self.cull = bool(cull)
        self.lanes = []    # further Station objects (same configuration, each on its own Context = stream): add_lane
        mid = self.relative_position.mean(axis=0)
        self._mid = mid
        self._radius = float(np.max(np.linalg.norm(self.relative_position - mid, axis=1)))
        self.last_keep_index = None   # event groups (of the chunk) the last station call ran on; None = all

    def __len__(self):
        return len(self.centres)

    def add_lane(self, station):
        """A second (third ...) Station object with the same configuration as the first, created on its OWN Context of the same
        device (its own stream, tables and workspace).  simulate_events_dev then walks the stations on all lanes side by side, one
        host thread per lane: the tail of one station call's kernels and its host round trips run under the other lane's kernels
        (two processes on one GPU measured 9 % / 5 % on BASELINE configs 3 / 5).  Masks and counters are the same as
        with one lane: every station call is independent, and the OR into the common mask only ever stores ones."""
        if not isinstance(station, Station):
            raise TypeError("add_lane needs a nuradiomc_amd.Station")
        if station.ctx is self.station.ctx or any(station.ctx is q.ctx for q in self.lanes):
            raise ValueError("every lane needs its own Context (stream)")
        if station.position.shape != self.station.position.shape:
            raise ValueError("a lane's station must have the channels of the first")
        self.lanes.append(station)

    def _move(self, i, st=None):
        (st or self.station).move_to(self.relative_position + self.centres[i])

    # ---- one station on a (possibly culled) shower list ----------------------------------------------------------------
    def _station_call(self, i, n, d_in, d_trig, dev_kw, kw, scratch, want_index=False, arz_rows=None, st=None):
        """Runs station i.  With a scratch and a distance cut: on the compact list of the groups in range; the flags of the
        call are OR-ed into d_trig (which the caller has zeroed).  Returns (stats or None, keep_index or None, shower_index or
        None); the index arrays are fetched to the host only when want_index."""
        primary = st is None or st is self.station
        st = self.station if st is None else st
        ctx, lib = st.ctx, st._lib
        self._move(i, st)
        n_groups = int(dev_kw.get('n_groups') or n)
        d_md, d_gb, d_vt = dev_kw.get('d_max_distance'), dev_kw.get('d_group_begin'), dev_kw.get('d_vertex_time')
        if primary:
            self.last_keep_index = None
        if kw.get('noise'):   # every station its own noise stream; keyed by the ORIGINAL group ids whatever list the group travels in
            kw = dict(kw, noise_seed=(int(kw.get('noise_seed', 0)) + 0x9E3779B97F4A7C15 * (i + 1)) & 0xffffffffffffffff)
        if scratch is None or d_md is None:
            if arz_rows is not None:
                L.check(lib.nrhip_station_set_shower_profiles(st._h, n, L.iptr(arz_rows[0]), L.dptr(arz_rows[1])))
            s_ = st.simulate_events_dev(n, *d_in, scratch.trig if scratch is not None else d_trig, **dev_kw, **kw)
            if scratch is not None:
                L.check(lib.nrhip_mask_or(ctx._h, n_groups, ctypes.c_void_p(d_trig), ctypes.c_void_p(scratch.trig), 0))
            return s_, None, None
        centre = np.ascontiguousarray(self.centres[i] + self._mid)
        nk, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        L.check(lib.nrhip_cull_groups(ctx._h, n, n_groups, d_gb, d_in[0], d_md, L.dptr(centre), self._radius, scratch.keep,
                                      scratch.gb, ctypes.byref(nk), ctypes.byref(ns)))
        nk, ns = int(nk.value), int(ns.value)
        keep = sidx = None
        if want_index or arz_rows is not None:
            keep = np.zeros(nk, np.int32)
            ctx.to_host(keep, scratch.keep)
            if primary:
                self.last_keep_index = keep
        if nk == 0:
            return None, keep, np.zeros(0, np.int32)
        L.check(lib.nrhip_gather_groups(ctx._h, nk, scratch.keep, d_gb, scratch.gb, d_in[0], d_in[1], d_in[2], d_in[3], d_in[4],
                                        d_in[5], d_vt, d_md, scratch.vertex, scratch.zenith, scratch.azimuth, scratch.energy,
                                        scratch.type, scratch.kL, scratch.vt if d_vt is not None else None, scratch.md,
                                        scratch.sidx))
        if want_index or arz_rows is not None:
            sidx = np.zeros(ns, np.int32)
            ctx.to_host(sidx, scratch.sidx)
        if arz_rows is not None:
            rows, resc = np.ascontiguousarray(arz_rows[0][sidx]), np.ascontiguousarray(arz_rows[1][sidx])
            L.check(lib.nrhip_station_set_shower_profiles(st._h, ns, L.iptr(rows), L.dptr(resc)))
        sub_kw = dict(dev_kw, d_max_distance=scratch.md, n_groups=nk, d_group_begin=scratch.gb if d_gb is not None else None,
                      d_vertex_time=scratch.vt if d_vt is not None else None)
        if kw.get('noise') and kw.get('d_noise_group_id') is None:
            L.check(lib.nrhip_index_to_i64(ctx._h, nk, scratch.keep, int(kw.get('noise_group_offset', 0)), scratch.gid64))
            kw = dict(kw, d_noise_group_id=scratch.gid64)
        s_ = st.simulate_events_dev(ns, scratch.vertex, scratch.zenith, scratch.azimuth, scratch.energy, scratch.type, scratch.kL,
                                    scratch.trig, **sub_kw, **kw)
        if not kw.get('select_only'):
            L.check(lib.nrhip_mask_scatter_or(ctx._h, nk, scratch.keep, scratch.trig, ctypes.c_void_p(d_trig)))
        return s_, keep, sidx

    def simulate_events_dev(self, n, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL, d_triggered, stations=None,
                            d_station_triggered=None, arz_rows=None, **kw):
        """Device-pointer form: the shower list stays in HBM, every station is run on it.  d_triggered (uint8 [n_groups]) receives
        the OR over the stations; d_station_triggered (uint8 [n_st][n_groups], optional) the per-station masks.  Returns the
        summed stats (stage_ms summed, too) with 'per_station' = [(n_rays, n_candidate_events)] and 'n_station_calls' /
        'n_groups_offered' (after the station-level selection).  arz_rows = (profile rows, rescale factors) per shower for the
        ARZ models (Station._arz_shower_profiles)."""
        st, ctx, lib = self.station, self.station.ctx, self.station._lib
        n_groups = int(kw.get('n_groups') or n)
        want = kw.pop('want_stats', True)
        dev_keys = ('d_vertex_time', 'n_groups', 'd_group_begin', 'd_max_distance')
        dev_kw = {k: kw.pop(k) for k in dev_keys if k in kw}
        dev_kw.setdefault('n_groups', n_groups)
        d_in = [d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL]
        use_cull = self.cull and dev_kw.get('d_max_distance') is not None
        lane_stations = [st] + list(self.lanes)
        todo = list(range(len(self.centres)) if stations is None else stations)
        if len(todo) < 2:
            lane_stations = lane_stations[:1]
        with_time = dev_kw.get('d_vertex_time') is not None
        scratches = [(_Scratch(q.ctx, n, n_groups, with_time) if use_cull else None) for q in lane_stations]
        results = {}

        def one_station(i, q, scratch):
            """station i on lane station q; the flags are OR-ed into the common mask (or the station's own row)"""
            qlib, qctx = q._lib, q.ctx
            tgt = d_triggered
            if d_station_triggered is not None:
                tgt = d_station_triggered + i * n_groups
                L.check(qlib.nrhip_memset(qctx._h, ctypes.c_void_p(tgt), 0, n_groups))
            if use_cull:
                s_, _, _ = self._station_call(i, n, d_in, tgt, dev_kw, dict(kw, want_stats=want), scratch, arz_rows=arz_rows, st=q)
            else:
                self._move(i, q)
                if arz_rows is not None:
                    L.check(qlib.nrhip_station_set_shower_profiles(q._h, n, L.iptr(arz_rows[0]), L.dptr(arz_rows[1])))
                kw_i = kw
                if kw.get('noise'):   # every station its own noise stream
                    kw_i = dict(kw, noise_seed=(int(kw.get('noise_seed', 0)) + 0x9E3779B97F4A7C15 * (i + 1)) & 0xffffffffffffffff)
                s_ = q.simulate_events_dev(n, *d_in, tgt, accumulate_triggered=True, want_stats=want, **dev_kw, **kw_i)
            if d_station_triggered is not None:
                L.check(qlib.nrhip_mask_or(qctx._h, n_groups, ctypes.c_void_p(d_triggered), ctypes.c_void_p(tgt), 0))
                if len(lane_stations) > 1:
                    qctx.synchronize()
            results[i] = s_

        total, per, offered = None, [], 0
        try:
            L.check(lib.nrhip_memset(ctx._h, ctypes.c_void_p(d_triggered), 0, n_groups))
            if len(lane_stations) == 1:
                for i in todo:
                    one_station(i, st, scratches[0])
            else:
                import threading
                ctx.synchronize()   # the zeroed mask before any lane ORs into it
                it_lock, it = threading.Lock(), iter(todo)
                errors = []

                def worker(q, scratch):
                    try:
                        while True:
                            with it_lock:
                                i = next(it, None)
                            if i is None or errors:
                                return
                            one_station(i, q, scratch)
                    except BaseException as e:   # noqa: B902 -- re-raised on the calling thread
                        errors.append(e)
                threads = [threading.Thread(target=worker, args=(q, sc)) for q, sc in zip(lane_stations, scratches)]
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
                for q in lane_stations:
                    q.ctx.synchronize()
                if errors:
                    raise errors[0]
            for i in todo:
                s_ = results.get(i)
                if s_ is None:
                    continue
                offered += s_['n_events']
                per.append((s_['n_rays'], s_['n_candidate_events']))
                total = s_ if total is None else _add_stats(total, s_)
        finally:
            for sc in scratches:
                if sc is not None:
                    sc.free()
        if want:
            if total is None:
                total = Station.empty_stats()
            trig = np.zeros(n_groups, np.uint8)
            ctx.to_host(trig, d_triggered)
            total.update(n_events=n_groups, n_triggered=int(trig.sum()), per_station=per, n_groups_offered=offered,
                         n_station_calls=len(per))
        return total if want else None

    def simulate_events(self, vertex, zenith, azimuth, energy, shower_type, k_L=None, vertex_time=None, group_id=None,
                        distance_cut_coefficients=None, distance_cut_sum_length=10., arz_iN=None, max_showers_per_call=None,
                        seed=None, rng=None, per_station=True, on_station=None, **kw):
        """Host-array form, the counterpart of Station.simulate_events: returns (triggered [n_groups] bool = any station,
        stats); stats['station_triggered'] [n_st, n_groups] bool when per_station.  on_station(i_station, chunk_slice, Station,
        keep) is called after every (station, chunk) while the station's tables are still fetchable (output writers, tests);
        keep = indices (inside the chunk) of the event groups the station's tables are about (None: all of them)."""
        st, ctx = self.station, self.station.ctx
        vertex = L.f64(vertex).reshape(-1, 3)
        n = len(vertex)
        n_st = len(self.centres)
        per = lambda a: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a), (n,)))
        zenith, azimuth, energy = (per(L.f64(a)) for a in (zenith, azimuth, energy))
        vertex_time = per(None if vertex_time is None else L.f64(vertex_time))
        types = _shower_type_codes(shower_type, n)
        model = kw.get('askaryan_model', 'Alvarez2009')
        is_arz = model in ('ARZ2019', 'ARZ2020')
        kL = np.array(np.broadcast_to(np.nan if k_L is None else L.f64(k_L), (n,)), dtype=np.float64)
        need_kL = model == 'Alvarez2009' and bool(np.any(np.isnan(kL) & (types == SHOWER_TO_INT['EM'])))
        need_iN = is_arz and arz_iN is None
        if rng is None and seed is not None:
            rng = np.random.RandomState(seed)
        if (need_kL or need_iN) and rng is None:
            raise ValueError("missing k_L / arz_iN values need a seed to be drawn in the reference's order")
        if is_arz and getattr(st, '_arz', None) is None:
            raise ValueError("the ARZ models need a shower library: Station.set_arz(nuradiomc_amd.arz.ARZ(library=...))")
        iN = None if arz_iN is None else np.array(np.broadcast_to(arz_iN, (n,)), np.int64)
        if need_iN:
            iN = np.zeros(n, np.int64)
        gid = None if group_id is None else np.asarray(group_id).reshape(-1)
        if gid is not None and len(gid) != n:
            raise ValueError("group_id must have one entry per shower")
        starts = np.arange(n) if gid is None else np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
        if gid is not None and len(np.unique(gid)) != len(starts):
            raise ValueError("showers of one event group must be consecutive")
        n_groups = len(starts)
        if max_showers_per_call is None:
            general = is_arz or getattr(st, '_birefringence_on', False)
            max_showers_per_call = max(1, int((2.5e5 if general else 6e6) / len(st.position)))
        # chunks end at group boundaries
        bounds, a = [0], 0
        gstarts = np.concatenate([starts, [n]])
        while a < n:
            k = np.searchsorted(gstarts, a + max_showers_per_call, side='right') - 1
            b = int(gstarts[k]) if gstarts[k] > a else int(gstarts[np.searchsorted(gstarts, a, side='right')])
            bounds.append(b)
            a = b
        any_trig = np.zeros(n_groups, bool)
        st_trig = np.zeros((n_st, n_groups), bool) if per_station else None
        total, offered, n_calls = None, 0, 0
        for a, b in zip(bounds[:-1], bounds[1:]):
            sl = slice(a, b)
            m = b - a
            g0, g1 = np.searchsorted(starts, a), np.searchsorted(starts, b)
            mg = g1 - g0
            gb = None if gid is None else np.ascontiguousarray(np.concatenate([starts[g0:g1] - a, [m]]), dtype=np.int32)
            md = None
            if distance_cut_coefficients is not None:
                md = distance_cut(vertex[sl], energy[sl], gb, distance_cut_coefficients, distance_cut_sum_length)
            d_in = [ctx.to_device(x) for x in (vertex[sl], zenith[sl], azimuth[sl], energy[sl], types[sl],
                                               np.where(np.isnan(kL[sl]), 1.0, kL[sl]))]
            extra = [ctx.to_device(x) if x is not None else None for x in
                     (None if vertex_time is None else vertex_time[sl], gb, md)]
            d_trig = ctx.malloc(max(mg, 1))
            dev_kw = dict(d_vertex_time=extra[0], n_groups=mg, d_group_begin=extra[1], d_max_distance=extra[2])
            scratch = _Scratch(ctx, m, mg, vertex_time is not None) if (self.cull and md is not None) else None
            try:
                if need_kL or need_iN:
                    first = np.full((n_st, m), -1, np.int32)
                    for i in range(n_st):
                        s_, _, sidx = self._station_call(i, m, d_in, d_trig, dev_kw, dict(kw, select_only=True, want_stats=True),
                                                         scratch, want_index=True)
                        if s_ is None:
                            continue
                        fc = st.fetch('shower_first_channel')
                        if sidx is None:
                            first[i] = fc
                        else:
                            first[i, sidx] = fc
                    order = sequencing.reference_draw_order(first, gb)
                    if need_kL:
                        kL[sl] = sequencing.draw_k_L(kL[sl], energy[sl], types[sl], order, rng)
                        ctx.copy_to_device(d_in[5], np.where(np.isnan(kL[sl]), 1.0, kL[sl]))
                    if need_iN:
                        iN[a + order] = st._arz.draw_profile_numbers(energy[sl][order],
                                                                     ['HAD' if c == 0 else 'EM' for c in types[sl][order]])
                arz_rows = st._arz_shower_profiles(energy[sl], types[sl], iN[sl]) if is_arz else None
                trig = np.zeros(max(mg, 1), np.uint8)
                for i in range(n_st):
                    L.check(st._lib.nrhip_memset(ctx._h, ctypes.c_void_p(d_trig), 0, mg))
                    s_, keep, _ = self._station_call(i, m, d_in, d_trig, dev_kw, kw, scratch,
                                                     want_index=on_station is not None, arz_rows=arz_rows)
                    if s_ is None:
                        continue
                    ctx.to_host(trig, d_trig)
                    t = trig[:mg].astype(bool)
                    any_trig[g0:g1] |= t
                    if per_station:
                        st_trig[i, g0:g1] = t
                    if on_station is not None:
                        on_station(i, sl, st, keep)
                    offered += s_['n_events']
                    n_calls += 1
                    total = s_ if total is None else _add_stats(total, s_)
            finally:
                for p in d_in + [d_trig] + [e for e in extra if e is not None]:
                    ctx.free(p)
                if scratch is not None:
                    scratch.free()
        if total is None:
            total = Station.empty_stats()
        total.update(n_events=n_groups, n_triggered=int(any_trig.sum()), n_groups_offered=offered, n_station_calls=n_calls)
        if per_station:
            total['station_triggered'] = st_trig
        if rng is not None and model == 'Alvarez2009':
            total['k_L'] = kL
        if is_arz:
            total['arz_iN'] = iN
        return any_trig, total


def _add_stats(total, s_):
    for k_, v_ in s_.items():
        if k_ == 'stage_ms':
            total[k_] = {q: total[k_][q] + v_[q] for q in v_}
        elif k_ in ('max_length', 'n_distinct_lengths'):
            total[k_] = max(total[k_], v_)
        elif k_ in ('n_triggered', 'n_events'):
            total[k_] = v_
        elif isinstance(v_, (int, float)):
            total[k_] += v_
    return total
