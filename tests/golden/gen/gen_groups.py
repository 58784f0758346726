"""Golden vectors for MULTI-SHOWER event groups, produced by the reference's own functions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_groups.py

Every event group goes through refharness.simulate_event with the list of its showers, i.e.
simulation.calculate_sim_efield(showers=[...]) per channel (simulation.py:93-292: loop over showers inside), the
per-efield and combined detector response and the simple threshold trigger on station S5.  Groups: single hadronic
showers, HAD + EM at one vertex (nu_e CC), and two showers at different vertices with a vertex-time offset.
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def run(name='groups_N256', n_groups=160, seed=31, N=256, full_events=10, distance_cut=False, loge=(17.3, 18.7)):
    det = rh.StationS5(n_samples=N, fs=2.0)
    cfg = rh.default_config()
    dcut = None
    if distance_cut:
        cfg['speedup']['distance_cut'] = True
        cfg['speedup']['distance_cut_coefficients'] = [-1.56434411e+02, 2.54131322e+01, -1.34932379e+00, 2.39984185e-02]
        cfg['speedup']['distance_cut_sum_length'] = 10.
        dcut = rh.distance_cut_function(cfg)
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    rng = np.random.default_rng(seed)
    base = rh.random_events(n_groups, seed, rmax=2500.)
    sh = dict(vertex=[], zenith=[], azimuth=[], energy=[], shower_type=[], vertex_time=[], group=[])
    for g in range(n_groups):
        kind = g % 5
        v, zen, az = base['vertex'][g], base['zenith'][g], base['azimuth'][g]
        e_tot = 10 ** rng.uniform(*loge)
        if kind in (0, 1):      # one hadronic shower
            parts = [(v, e_tot, 'HAD', 0.)]
        elif kind in (2, 3):    # nu_e CC: hadronic + electromagnetic shower at the same vertex
            y = rng.uniform(0.1, 0.9)
            parts = [(v, y * e_tot, 'HAD', 0.), (v, (1 - y) * e_tot, 'EM', 0.)]
        else:                   # two vertices along the direction of flight, second one later
            d = rng.uniform(20., 300.)
            axis = -np.array([np.sin(zen) * np.cos(az), np.sin(zen) * np.sin(az), np.cos(zen)])
            v2 = v + d * axis
            if v2[2] > -1.:
                v2 = v.copy()
                d = 0.
            parts = [(v, 0.4 * e_tot, 'HAD', 0.), (v2, 0.6 * e_tot, 'HAD', d / 0.299792458)]
        for (vv, ee, tt, vt) in parts:
            sh['vertex'].append(vv); sh['zenith'].append(zen); sh['azimuth'].append(az); sh['energy'].append(ee)
            sh['shower_type'].append(tt); sh['vertex_time'].append(vt); sh['group'].append(g)
    sh = {k: np.array(v) for k, v in sh.items()}
    n_sh = len(sh['group'])
    evo = dict(candidate=np.zeros(n_groups, bool), triggered=np.zeros(n_groups, bool), L=np.zeros(n_groups, np.int64),
               t_min=np.full(n_groups, np.nan), n_rays=np.zeros(n_groups, np.int32), maxV=np.zeros((n_groups, 5)))
    k_L = np.full(n_sh, np.nan)
    V_list, V_ev = [], []
    ray_rows = []
    t0 = time.time()
    for g in range(n_groups):
        idx = np.flatnonzero(sh['group'] == g)
        showers = [rh.make_shower(int(i), sh['vertex'][i], sh['zenith'][i], sh['azimuth'][i], sh['energy'][i],
                                  str(sh['shower_type'][i]), vertex_time=float(sh['vertex_time'][i])) for i in idx]
        o = rh.simulate_event(g, showers, det, prop, ice, cfg, vrms, vrms_e, distance_cut=dcut)
        k_L[idx] = o['k_L_all']
        evo['candidate'][g] = o['candidate']
        evo['triggered'][g] = o['triggered']
        evo['L'][g] = o['L']
        evo['t_min'][g] = o['t_min']
        evo['n_rays'][g] = len(o['rays'])
        for r in o['rays']:
            ray_rows.append((g, r['shower_id'], r['channel'], r['iS'], r['C0'], r['t0'], r['max_efield']))
        if 'V' in o:
            evo['maxV'][g] = np.max(np.abs(o['V']), axis=1)
            if len(V_ev) < full_events and len(idx) > 1:
                V_ev.append(g)
                V_list.append(o['V'])
    print(name, '%d groups, %d showers, %d rays, %d candidates, %d triggered, %.1f s' % (
        n_groups, n_sh, len(ray_rows), evo['candidate'].sum(), evo['triggered'].sum(), time.time() - t0))
    rr = np.array(ray_rows)
    out = dict(N=N, fs=2.0, vrms=vrms, vrms_efield=vrms_e, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]),
               att_model='SP1', n_freq=25, askaryan_model='Alvarez2009', antenna='analytic_VPol', cable_delay=0.,
               distance_cut=distance_cut, distance_cut_coefficients=np.array(cfg['speedup'].get('distance_cut_coefficients', [])),
               det_pos=det.pos, det_orientation=np.array(det.orientation),
               vertex=sh['vertex'], zenith=sh['zenith'], azimuth=sh['azimuth'], energy=sh['energy'],
               shower_type=sh['shower_type'], vertex_time=sh['vertex_time'], group=sh['group'], k_L=k_L,
               ray_group=rr[:, 0].astype(np.int64), ray_shower=rr[:, 1].astype(np.int64), ray_channel=rr[:, 2].astype(np.int64),
               ray_iS=rr[:, 3].astype(np.int64), ray_C0=rr[:, 4], ray_t0=rr[:, 5], ray_max_efield=rr[:, 6],
               V_events=np.array(V_ev, np.int64), V_offsets=np.cumsum([0] + [v.shape[1] for v in V_list]),
               V_concat=np.concatenate(V_list, axis=1) if V_list else np.zeros((5, 0)))
    out.update({'ev_' + k: v for k, v in evo.items()})
    np.savez_compressed(os.path.join(OUT, 'chain_%s.npz' % name), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['groups_N256', 'groups_dcut_N256']
    if 'groups_N256' in which:
        run()
    if 'groups_dcut_N256' in which:  # speedup.distance_cut on, energies where it bites
        run('groups_dcut_N256', n_groups=200, seed=32, distance_cut=True, loge=(16.0, 17.8), full_events=4)
