"""Golden vectors for ARRAYS of stations (BASELINE configs 3, 4, 5), produced by the reference's own functions in the order
of simulation.run() (NuRadioMC/simulation/simulation.py:1454-1600): event group -> station -> channel
(calculate_sim_efield over all showers of the group) -> detector response -> trigger, one RadioShower object per shower
shared by all stations (so the random shower parameters k_L / ARZ profile number are drawn once, at the shower's first
surviving ray, and reused: :221-242).

    cp -r /root/reference /tmp/refcopy
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_array.py [rnog] [rnog_arz_bire] [gen2]

Layouts
* rnog: station positions and the 24-channel layout (relative positions, orientations, cable delays) are read from
  NuRadioReco/detector/RNO_G/RNO_array.json (35 stations; channels defined for the default station 11).  Its antenna
  models are measured patterns the reference downloads; the analytic models stand in: RNOG_vpol -> analytic_VPol,
  RNOG_quadslot -> analytic_HPol, createLPDA -> analytic_LPDA.  The layout is saved to tests/golden/rnog_array_layout.npz
  (data: what bench.py --config 3|4 and the tests build their arrays from).
* gen2: no IceCube-Gen2 detector file exists in the reference; 200 stations on a 1.24 km square grid, each the 5-channel
  dipole string of BASELINE config 2.

Fixtures (tests/golden/array_*.npz): inputs (shower lists with event-group ids), per (group, station): number of rays,
candidate flag, trigger flag, trace length, t_min, per-channel maxima; the k_L / profile numbers the reference drew;
a few complete sets of channel traces.
"""
import json
import os
import pickle
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402
from NuRadioReco.framework.parameters import showerParameters as shp  # noqa: E402
import NuRadioReco.modules.trigger.highLowThreshold  # noqa: E402
import NuRadioReco.detector  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
DCUT = [-1.56434411e+02, 2.54131322e+01, -1.34932379e+00, 2.39984185e-02]   # config_default.yaml speedup.distance_cut_coefficients
ANT = {'RNOG_vpol': 'analytic_VPol', 'RNOG_quad': 'analytic_HPol', 'createLP': 'analytic_LPDA'}


def rnog_layout():
    path = os.path.join(os.path.dirname(NuRadioReco.detector.__file__), 'RNO_G', 'RNO_array.json')
    d = json.load(open(path))
    st = sorted(d['stations'].values(), key=lambda s: s['station_id'])
    centres = np.array([[s['pos_easting'], s['pos_northing'], s['pos_altitude']] for s in st], float)
    ids = np.array([s['station_id'] for s in st])
    ch = sorted(d['channels'].values(), key=lambda c: c['channel_id'])
    rel = np.array([[c['ant_position_x'], c['ant_position_y'], c['ant_position_z']] for c in ch], float)
    ori = np.array([[c['ant_orientation_theta'], c['ant_orientation_phi'], c['ant_rotation_theta'], c['ant_rotation_phi']]
                    for c in ch], float) * units.deg
    ant = np.array([ANT[c['ant_type'][:9] if c['ant_type'].startswith('RNOG') else c['ant_type'][:8]] for c in ch])
    cab = np.array([c.get('cab_time_delay', 0.) for c in ch], float)
    return dict(station_ids=ids, centres=centres, rel_pos=rel, orientation=ori, antenna=ant, cable_delay=cab,
                channel_ids=np.array([c['channel_id'] for c in ch]))


class ArrayDet(rh.StationS5):
    """duck-typed detector of identical stations at `centres`"""

    def __init__(self, lay, n_samples, fs):
        self.lay = lay
        self.n_samples, self.fs = n_samples, fs
        self.ids = [int(i) for i in lay['station_ids']]
        self.centre = {i: np.array(c, float) for i, c in zip(self.ids, lay['centres'])}
        self.pos = np.array(lay['rel_pos'], float)
        self.antenna = [str(a) for a in lay['antenna']]
        self.cable_delay = [float(c) for c in lay['cable_delay']]
        self.orientations = [list(o) for o in lay['orientation']]
        self.station_id = self.ids[0]

    def get_station_ids(self):
        return [self.station_id]     # refharness.simulate_event looks at ONE station: set `station_id` before the call

    def get_absolute_position(self, station_id):
        return self.centre[station_id].copy()


_hl = NuRadioReco.modules.trigger.highLowThreshold.triggerSimulator()


def simulate_group(g, showers, det, prop, ice, cfg, vrms, vrms_e, dcut, trigger):
    """one event group through all stations (simulation.py:1500-1600); returns per-station dicts"""
    res = []
    for sid in det.ids:
        det.station_id = sid
        if trigger is None:
            o = rh.simulate_event(g, showers, det, prop, ice, cfg, vrms, vrms_e, distance_cut=dcut)
        else:
            o = rh.simulate_event(g, showers, det, prop, ice, cfg, vrms, vrms_e, distance_cut=dcut, trigger=trigger)
        res.append(o)
    return res


def pack(name, lay, cfg, N, fs, vrms, vrms_e, ice, sh, results, extra, full_sets=4):
    n_groups, n_st, n_ch = len(results), len(lay['centres']), len(lay['rel_pos'])
    evo = dict(n_rays=np.zeros((n_groups, n_st), np.int32), candidate=np.zeros((n_groups, n_st), bool),
               triggered=np.zeros((n_groups, n_st), bool), L=np.zeros((n_groups, n_st), np.int64),
               t_min=np.full((n_groups, n_st), np.nan), maxV=np.zeros((n_groups, n_st, n_ch)))
    V_list, V_key = [], []
    rows = []
    for g, per in enumerate(results):
        for s, o in enumerate(per):
            evo['n_rays'][g, s] = len(o['rays'])
            evo['candidate'][g, s] = o['candidate']
            evo['triggered'][g, s] = o['triggered']
            evo['L'][g, s] = o['L']
            evo['t_min'][g, s] = o['t_min']
            for r in o['rays']:
                rows.append((g, s, r['shower_id'], r['channel'], r['iS'], r['C0'], r['t0'], r['max_efield']))
            if 'V' in o:
                evo['maxV'][g, s] = np.max(np.abs(o['V']), axis=1)
                if len(V_list) < full_sets and o['triggered']:
                    V_list.append(o['V'])
                    V_key.append((g, s))
    rr = np.array(rows) if rows else np.zeros((0, 8))
    out = dict(N=N, fs=fs, vrms=vrms, vrms_efield=vrms_e, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]),
               att_model=cfg['propagation']['attenuation_model'], n_freq=cfg['propagation']['n_freq'],
               askaryan_model=cfg['signal']['model'], distance_cut_coefficients=np.array(DCUT),
               station_ids=lay['station_ids'], centres=lay['centres'], rel_pos=lay['rel_pos'], orientation=lay['orientation'],
               antenna=lay['antenna'], cable_delay=lay['cable_delay'],
               vertex=sh['vertex'], zenith=sh['zenith'], azimuth=sh['azimuth'], energy=sh['energy'],
               shower_type=sh['shower_type'], vertex_time=sh['vertex_time'], group=sh['group'],
               ray_group=rr[:, 0].astype(np.int64), ray_station=rr[:, 1].astype(np.int64), ray_shower=rr[:, 2].astype(np.int64),
               ray_channel=rr[:, 3].astype(np.int64), ray_iS=rr[:, 4].astype(np.int64), ray_C0=rr[:, 5], ray_t0=rr[:, 6],
               ray_max_efield=rr[:, 7],
               V_keys=np.array(V_key, np.int64).reshape(-1, 2), V_offsets=np.cumsum([0] + [v.shape[1] for v in V_list]),
               V_concat=np.concatenate(V_list, axis=1) if V_list else np.zeros((n_ch, 0)))
    out.update({'ev_' + k: v for k, v in evo.items()})
    out.update(extra)
    np.savez_compressed(os.path.join(OUT, 'array_%s.npz' % name), **out)
    print(name, '%d groups x %d stations: %d rays, %d candidate station-events, %d triggered station-events, %d groups triggered'
          % (n_groups, n_st, len(rows), evo['candidate'].sum(), evo['triggered'].sum(), evo['triggered'].any(axis=1).sum()),
          flush=True)


def draw_showers(n_groups, seed, centres, margin, loge, em_fraction, zmin=-2700.):
    """event groups in the footprint of the array: single hadronic showers and nu_e CC pairs (HAD + EM at one vertex)"""
    rng = np.random.default_rng(seed)
    lo, hi = centres[:, :2].min(axis=0) - margin, centres[:, :2].max(axis=0) + margin
    sh = dict(vertex=[], zenith=[], azimuth=[], energy=[], shower_type=[], vertex_time=[], group=[])
    for g in range(n_groups):
        v = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(lo[1], hi[1]), rng.uniform(zmin, -5.)])
        zen, az = np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi)
        e = 10 ** rng.uniform(*loge)
        if rng.random() < em_fraction:
            y = rng.uniform(0.1, 0.9)
            parts = [(y * e, 'HAD'), ((1 - y) * e, 'EM')]
        else:
            parts = [(e, 'HAD')]
        for ee, tt in parts:
            sh['vertex'].append(v); sh['zenith'].append(zen); sh['azimuth'].append(az); sh['energy'].append(ee)
            sh['shower_type'].append(tt); sh['vertex_time'].append(0.); sh['group'].append(g)
    return {k: np.array(v) for k, v in sh.items()}


def make_showers(sh, g):
    idx = np.flatnonzero(sh['group'] == g)
    return idx, [rh.make_shower(int(i), sh['vertex'][i], sh['zenith'][i], sh['azimuth'][i], sh['energy'][i],
                                str(sh['shower_type'][i]), vertex_time=float(sh['vertex_time'][i])) for i in idx]


def _worker(args):
    (lay, cfg, N, fs, sh, groups) = args
    det = ArrayDet(lay, N, fs)
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    dcut = rh.distance_cut_function(cfg)
    out = []
    for g in groups:
        _, showers = make_showers(sh, g)
        out.append((g, simulate_group(g, showers, det, prop, ice, cfg, vrms, vrms_e, dcut, None)))
    return out


def run_rnog(n_groups=64, seed=41, N=256, fs=2.0, n_proc=8):
    """config 3: hadronic showers only (no random draws), so the event groups are independent and spread over processes"""
    import multiprocessing as mp
    lay = rnog_layout()
    np.savez_compressed(os.path.join(OUT, 'rnog_array_layout.npz'), **lay)
    cfg = rh.default_config(ice_model='greenland_simple', attenuation_model='GL1')
    cfg['speedup'].update(distance_cut=True, distance_cut_coefficients=DCUT, distance_cut_sum_length=10.)
    sh = draw_showers(n_groups, seed, lay["centres"], 800., (17.0, 18.5), 0.)
    det = ArrayDet(lay, N, fs)
    ice, _ = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    t0 = time.time()
    with mp.Pool(n_proc) as pool:
        parts = pool.map(_worker, [(lay, cfg, N, fs, sh, list(range(k, n_groups, n_proc))) for k in range(n_proc)])
    res = dict(sum(parts, []))
    print('rnog: %.0f s' % (time.time() - t0), flush=True)
    pack('rnog', lay, cfg, N, fs, vrms, vrms_e, ice, sh, [res[g] for g in range(n_groups)], {})


def run_rnog_arz_bire(n_groups=12, seed=44, N=4096, fs=2.0, n_stations=35):
    """config 4: ARZ2020 + birefringence (greenland_A) on the same array; sequential (ONE stream of profile numbers)"""
    from NuRadioMC.SignalGen.ARZ import ARZ
    g_arz = np.load(os.path.join(OUT, 'ref_arz.npz'))
    depth = g_arz['lib_depth']
    library = {'EM': {1e18: {'depth': depth, 'charge_excess': list(g_arz['lib_EM_1e18'])},
                      1e16: {'depth': depth, 'charge_excess': list(g_arz['lib_EM_1e16'])}},
               'HAD': {1e18: {'depth': depth, 'charge_excess': list(g_arz['lib_HAD_1e18'])},
                       1e17: {'depth': depth, 'charge_excess': list(g_arz['lib_HAD_1e17'])}}}
    lib_dir = os.path.join(os.path.dirname(ARZ.__file__), 'shower_library')
    default_path = os.path.join(lib_dir, 'library_v1.2.pkl')
    assert default_path.startswith('/tmp/'), default_path   # inside the COPY of the reference tree
    ARZ.ARZ._ARZ__check_and_get_library = lambda self: True   # no download
    with open(default_path, 'wb') as fout:
        pickle.dump(library, fout)
    lay = rnog_layout()
    lay = {k: (v[:n_stations] if k in ('station_ids', 'centres') else v) for k, v in lay.items()}
    cfg = rh.default_config(model='ARZ2020', ice_model='greenland_simple', attenuation_model='GL1')
    cfg['speedup'].update(distance_cut=True, distance_cut_coefficients=DCUT, distance_cut_sum_length=10.)
    cfg['propagation'].update(birefringence=True, birefringence_model='greenland_A', birefringence_propagation='analytical')
    sh = draw_showers(n_groups, seed, lay["centres"], 300., (17.4, 18.4), 0.4, zmin=-1500.)
    det = ArrayDet(lay, N, fs)
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    dcut = rh.distance_cut_function(cfg)
    res, iN = [], np.full(len(sh['group']), -1, np.int64)
    t0 = time.time()
    for g in range(n_groups):
        idx, showers = make_showers(sh, g)
        res.append(simulate_group(g, showers, det, prop, ice, cfg, vrms, vrms_e, dcut, None))
        for i, s_ in zip(idx, showers):
            if s_.has_parameter(shp.charge_excess_profile_id):
                iN[i] = s_[shp.charge_excess_profile_id]
        print('rnog_arz_bire group', g, '%.0f s' % (time.time() - t0), 'rays', sum(len(o['rays']) for o in res[-1]), flush=True)
    pack('rnog_arz_bire', lay, cfg, N, fs, vrms, vrms_e, ice, sh, res,
         dict(arz_iN=iN, seed=cfg['seed'], birefringence_model='greenland_A'), full_sets=2)


def gen2_layout(n_st=200, spacing=1240.):
    side = int(np.ceil(np.sqrt(n_st)))
    centres = np.array([[spacing * (i - (side - 1) / 2), spacing * (j - (side - 1) / 2), 0.]
                        for i in range(side) for j in range(side)])[:n_st]
    d = np.pi / 180
    return dict(station_ids=np.arange(1001, 1001 + n_st), centres=centres,
                rel_pos=np.array([[0., 0., -100. - i] for i in range(5)]), orientation=np.tile([0., 0., 90 * d, 90 * d], (5, 1)),
                antenna=np.array(['analytic_VPol'] * 5), cable_delay=np.zeros(5), channel_ids=np.arange(5))


def run_gen2(n_groups=40, seed=47, N=256, fs=2.0):
    """config 5: 200 stations x 5 channels at the South Pole, showers log-uniform in 1e16 .. 1e20 eV, 45 % nu_e CC pairs whose
    EM showers get their k_L from the reference's generator (seed 1235, fresh stream), 2-of-5 high/low coincidence trigger"""
    lay = gen2_layout()
    cfg = rh.default_config()
    cfg['speedup'].update(distance_cut=True, distance_cut_coefficients=DCUT, distance_cut_sum_length=10.)
    sh = draw_showers(n_groups, seed, lay['centres'], 500., (16., 20.), 0.45)
    det = ArrayDet(lay, N, fs)
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    dcut = rh.distance_cut_function(cfg)
    trig = dict(kind='high_low', threshold_high=3.0 * vrms, threshold_low=-3.0 * vrms, high_low_window=5 * units.ns,
                coinc_window=30 * units.ns, number_concidences=2)
    res, k_L = [], np.full(len(sh['group']), np.nan)
    t0 = time.time()
    for g in range(n_groups):
        idx, showers = make_showers(sh, g)
        res.append(simulate_group(g, showers, det, prop, ice, cfg, vrms, vrms_e, dcut, trig))
        for i, s_ in zip(idx, showers):
            if s_.has_parameter(shp.k_L):
                k_L[i] = s_[shp.k_L]
        print('gen2 group', g, '%.0f s' % (time.time() - t0), 'rays', sum(len(o['rays']) for o in res[-1]), flush=True)
    pack('gen2', lay, cfg, N, fs, vrms, vrms_e, ice, sh, res,
         dict(k_L=k_L, seed=cfg['seed'], trigger_kind='high_low', trigger_n_coincidences=2, trigger_high_low_window=5.,
              trigger_coinc_window=30., trigger_threshold_sigma=3.0))


if __name__ == '__main__':
    which = sys.argv[1:] or ['rnog', 'rnog_arz_bire', 'gen2']
    if 'layout' in which:
        np.savez_compressed(os.path.join(OUT, 'rnog_array_layout.npz'), **rnog_layout())
    if 'gen2' in which:            # first: needs the FRESH Alvarez2009 random stream of this process
        run_gen2()
    if 'rnog' in which:
        run_rnog()
    if 'rnog_arz_bire' in which:
        run_rnog_arz_bire()
