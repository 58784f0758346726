"""Golden vectors for split_event_time_diff: event groups whose signals at the station are farther apart in time than the limit
are cut into sub-events by the reference's simulation.group_into_events (simulation.py:906-947); each sub-event then gets the
detector response and the trigger on its own (simulation.run :1566-1600).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_split.py

Groups on station S5 (N = 256): single showers (direct and reflected ray hundreds of ns apart), nu_e CC pairs at one vertex, and two
showers along the track with a vertex-time offset; split_event_time_diff = 300 ns.  Writes tests/golden/chain_split_N256.npz:
per group the number of sub-events and per sub-event L, t_min, trigger, channel maxima, the member signals (channel, shower, ray
solution) and, for the first ones, the full traces."""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def run(n_groups=140, seed=41, N=256, split=300., full=12):
    det = rh.StationS5(n_samples=N, fs=2.0)
    cfg = rh.default_config()
    cfg['split_event_time_diff'] = split
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    rng = np.random.default_rng(seed)
    base = rh.random_events(n_groups, seed, rmax=1800.)
    sh = dict(vertex=[], zenith=[], azimuth=[], energy=[], shower_type=[], vertex_time=[], group=[])
    for g in range(n_groups):
        kind = g % 4
        v, zen, az = base['vertex'][g], base['zenith'][g], base['azimuth'][g]
        e_tot = 10 ** rng.uniform(18.0, 19.3)
        if kind in (0, 1):
            parts = [(v, e_tot, 'HAD', 0.)]
        elif kind == 2:
            y = rng.uniform(0.1, 0.9)
            parts = [(v, y * e_tot, 'HAD', 0.), (v, (1 - y) * e_tot, 'EM', 0.)]
        else:
            d = rng.uniform(100., 900.)
            axis = -np.array([np.sin(zen) * np.cos(az), np.sin(zen) * np.sin(az), np.cos(zen)])
            v2 = v + d * axis
            if v2[2] > -1.:
                v2, d = v.copy(), 0.
            parts = [(v, 0.5 * e_tot, 'HAD', 0.), (v2, 0.5 * e_tot, 'HAD', d / 0.299792458)]
        for (vv, ee, tt, vt) in parts:
            sh['vertex'].append(vv); sh['zenith'].append(zen); sh['azimuth'].append(az); sh['energy'].append(ee)
            sh['shower_type'].append(tt); sh['vertex_time'].append(vt); sh['group'].append(g)
    sh = {k: np.array(v) for k, v in sh.items()}
    n_sh = len(sh['group'])
    k_L = np.full(n_sh, np.nan)
    ev = dict(candidate=np.zeros(n_groups, bool), triggered=np.zeros(n_groups, bool), n_sub=np.zeros(n_groups, np.int32),
              n_rays=np.zeros(n_groups, np.int32))
    sub_rows, member_rows, V_list, V_sub = [], [], [], []
    t0 = time.time()
    for g in range(n_groups):
        idx = np.flatnonzero(sh['group'] == g)
        showers = [rh.make_shower(int(i), sh['vertex'][i], sh['zenith'][i], sh['azimuth'][i], sh['energy'][i],
                                  str(sh['shower_type'][i]), vertex_time=float(sh['vertex_time'][i])) for i in idx]
        o = rh.simulate_event(g, showers, det, prop, ice, cfg, vrms, vrms_e, split=split)
        k_L[idx] = o['k_L_all']
        ev['candidate'][g], ev['triggered'][g], ev['n_rays'][g] = o['candidate'], o['triggered'], len(o['rays'])
        for j, q in enumerate(o.get('sub', [])):
            k = len(sub_rows)
            sub_rows.append((g, j, q['L'], q['t_min'], q['triggered']) + tuple(np.max(np.abs(q['V']), axis=1)))
            for (c, s_, iS) in q['members']:
                member_rows.append((k, c, s_, iS))
            if len(V_sub) < full and len(o['sub']) > 1:
                V_sub.append(k)
                V_list.append(q['V'])
        ev['n_sub'][g] = len(o.get('sub', []))
    sub = np.array(sub_rows)
    print('%d groups, %d showers, %d candidates, %d split into %d sub-events, %d triggered groups, %.1f s' % (
        n_groups, n_sh, ev['candidate'].sum(), (ev['n_sub'] > 1).sum(), ev['n_sub'][ev['n_sub'] > 1].sum(), ev['triggered'].sum(),
        time.time() - t0))
    out = dict(N=N, fs=2.0, vrms=vrms, vrms_efield=vrms_e, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), att_model='SP1', n_freq=25,
               askaryan_model='Alvarez2009', antenna='analytic_VPol', cable_delay=0., split_event_time_diff=split, det_pos=det.pos, det_orientation=np.array(det.orientation),
               vertex=sh['vertex'], zenith=sh['zenith'], azimuth=sh['azimuth'], energy=sh['energy'], shower_type=sh['shower_type'],
               vertex_time=sh['vertex_time'], group=sh['group'], k_L=k_L,
               sub_group=sub[:, 0].astype(np.int64), sub_index=sub[:, 1].astype(np.int64), sub_L=sub[:, 2].astype(np.int64),
               sub_t_min=sub[:, 3], sub_triggered=sub[:, 4].astype(bool), sub_maxV=sub[:, 5:],
               member_sub=np.array([m[0] for m in member_rows]), member_channel=np.array([m[1] for m in member_rows]),
               member_shower=np.array([m[2] for m in member_rows]), member_iS=np.array([m[3] for m in member_rows]),
               V_subs=np.array(V_sub, np.int64), V_offsets=np.cumsum([0] + [v.shape[1] for v in V_list]),
               V_concat=np.concatenate(V_list, axis=1) if V_list else np.zeros((5, 0)))
    out.update({'ev_' + k: v for k, v in ev.items()})
    np.savez_compressed(os.path.join(OUT, 'chain_split_N256.npz'), **out)


if __name__ == '__main__':
    run()
