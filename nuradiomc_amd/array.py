"""StationArray -- the station loop of simulation.run() (NuRadioMC/simulation/simulation.py:1454-1600) for an array of
identical stations (BASELINE configs 3-5: 35 RNO-G-like stations, 200 Gen2-like stations).

"each station is treated independently" (:1500): every event group is offered to every station -- ray tracing for all
its channels with the shower-energy dependent distance cut (:155-163; the station-level quick cut of :1503-1509 is
commented out in the reference: `# continue`), candidate flag, common time grid, channel voltages and trigger per
station -- and the group is kept when any station triggered.  The shower list of a chunk goes to HBM once and serves
every station; ONE Station object (tables, workspace) is moved through the array (Station.move_to).

The stateful random shower parameters (k_L of Alvarez2009 EM showers, ARZ profile numbers) are drawn in the order in which
the reference's loops (group -> station -> channel -> shower) meet the showers: a first pass over all stations only traces
the rays (nrhip_sim_config.select_only), sequencing.reference_draw_order sorts, the host draws, the second pass simulates.
"""
import numpy as np
from . import _lib as L
from . import sequencing
from .station import Station, _shower_type_codes, distance_cut, SHOWER_TO_INT


class StationArray:
    def __init__(self, station, centres, relative_position=None, station_ids=None):
        """station: the Station object that is moved through the array (antennas, orientations, cable delays, filters, tables);
        centres [n_st, 3]: absolute station positions (det.get_absolute_position); relative_position [n_ch, 3]: the channel
        positions inside a station (det.get_relative_position) -- channel c of station i sits at relative_position[c] +
        centres[i], the sum the reference forms (simulation.py:138).  Pass it explicitly: recovering it as station.position -
        centres[0] (the default) is off by an ulp of the station coordinate, and the reference's first ray root is sensitive to
        the last bit of its inputs."""
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

    def __len__(self):
        return len(self.centres)

    def _move(self, i):
        self.station.move_to(self.relative_position + self.centres[i])

    def simulate_events_dev(self, n, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL, d_triggered, stations=None,
                            d_station_triggered=None, **kw):
        """Device-pointer form: the shower list stays in HBM, every station is run on it.  d_triggered (uint8 [n_groups]) receives
        the OR over the stations (accumulated on the device); d_station_triggered (uint8 [n_st][n_groups], optional) the
        per-station masks.  Returns the summed stats (stage_ms summed, too) with 'per_station' = [(n_rays, n_candidate_events,
        n_triggered so far)]."""
        st = self.station
        n_groups = int(kw.get('n_groups') or n)
        total, per = None, []
        want = kw.pop('want_stats', True)
        for k, i in enumerate(range(len(self.centres)) if stations is None else stations):
            self._move(i)
            if d_station_triggered is not None:
                s_ = st.simulate_events_dev(n, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL,
                                            d_station_triggered + i * n_groups, want_stats=want, **kw)
                L.check(st._lib.nrhip_mask_or(st.ctx._h, n_groups, d_triggered, d_station_triggered + i * n_groups, int(k == 0)))
            else:
                s_ = st.simulate_events_dev(n, d_vertex, d_zenith, d_azimuth, d_energy, d_type, d_kL, d_triggered,
                                            accumulate_triggered=(k > 0), want_stats=want, **kw)
            if s_ is None:
                continue
            per.append((s_['n_rays'], s_['n_candidate_events'], s_['n_triggered']))
            total = s_ if total is None else _add_stats(total, s_)
        if total is not None:
            total['per_station'] = per
        return total

    def simulate_events(self, vertex, zenith, azimuth, energy, shower_type, k_L=None, vertex_time=None, group_id=None,
                        distance_cut_coefficients=None, distance_cut_sum_length=10., arz_iN=None, max_showers_per_call=None,
                        seed=None, rng=None, per_station=True, on_station=None, **kw):
        """Host-array form, the counterpart of Station.simulate_events: returns (triggered [n_groups] bool = any station,
        stats); stats['station_triggered'] [n_st, n_groups] bool when per_station.  on_station(i_station, chunk_slice, Station)
        is called after every (station, chunk) while the station's tables are still fetchable (output writers, tests)."""
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
        total = None
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
            try:
                if need_kL or need_iN:
                    first = np.empty((n_st, m), np.int32)
                    for i in range(n_st):
                        self._move(i)
                        st.simulate_events_dev(m, *d_in, d_trig, select_only=True, want_stats=False, **dev_kw, **kw)
                        first[i] = st.fetch('shower_first_channel')
                    order = sequencing.reference_draw_order(first, gb)
                    if need_kL:
                        kL[sl] = sequencing.draw_k_L(kL[sl], energy[sl], types[sl], order, rng)
                        ctx.copy_to_device(d_in[5], np.where(np.isnan(kL[sl]), 1.0, kL[sl]))
                    if need_iN:
                        iN[a + order] = st._arz.draw_profile_numbers(energy[sl][order],
                                                                     ['HAD' if c == 0 else 'EM' for c in types[sl][order]])
                if is_arz:
                    rows, resc = st._arz_shower_profiles(energy[sl], types[sl], iN[sl])
                    L.check(st._lib.nrhip_station_set_shower_profiles(st._h, m, L.iptr(rows), L.dptr(resc)))
                trig = np.zeros(max(mg, 1), np.uint8)
                for i in range(n_st):
                    self._move(i)
                    s_ = st.simulate_events_dev(m, *d_in, d_trig, **dev_kw, **kw)
                    ctx.to_host(trig, d_trig)
                    t = trig[:mg].astype(bool)
                    any_trig[g0:g1] |= t
                    if per_station:
                        st_trig[i, g0:g1] = t
                    if on_station is not None:
                        on_station(i, sl, st)
                    total = s_ if total is None else _add_stats(total, s_)
            finally:
                for p in d_in + [d_trig] + [e for e in extra if e is not None]:
                    ctx.free(p)
        if total is None:
            total = {}
        total['n_events'] = n_groups
        total['n_triggered'] = int(any_trig.sum())
        if per_station:
            total['station_triggered'] = st_trig
        if need_kL:
            total['k_L'] = kL
        if need_iN:
            total['arz_iN'] = iN
        return any_trig, total


def _add_stats(total, s_):
    for k_, v_ in s_.items():
        if k_ == 'stage_ms':
            total[k_] = {q: total[k_][q] + v_[q] for q in v_}
        elif k_ in ('max_length', 'n_distinct_lengths'):
            total[k_] = max(total[k_], v_)
        elif k_ == 'n_triggered':
            total[k_] = v_
        elif isinstance(v_, (int, float)):
            total[k_] += v_
    return total
