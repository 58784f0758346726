"""The oracle's ARRAY driver (oracle/spectral_oracle.py: simulate_event_group_array = the station loop of simulation.run(),
NuRadioMC/simulation/simulation.py:1454-1600) against fixtures the reference itself produced for whole arrays
(tests/golden/array_*.npz, generator tests/golden/gen/gen_array.py): per (event group, station) ray counts, candidate and trigger
flags, trace lengths, channel maxima.  CPU only."""
import os
import numpy as np
import pytest

from conftest import golden, ROOT
from oracle import spectral_oracle as so

DCUT = [-1.56434411e+02, 2.54131322e+01, -1.34932379e+00, 2.39984185e-02]


def _showers(g, gi, kL):
    idx = np.flatnonzero(g['group'] == gi)
    return [dict(vertex=g['vertex'][i], zenith=float(g['zenith'][i]), azimuth=float(g['azimuth'][i]),
                 energy=float(g['energy'][i]), shower_type=str(g['shower_type'][i]),
                 k_L=None if np.isnan(kL[i]) else float(kL[i]), vertex_time=float(g['vertex_time'][i])) for i in idx]


def _compare(g, groups, stations, kL, trigger=None, amp_tol=5e-3):
    skw = dict(antenna=[str(a) for a in g['antenna']], orientation=g['orientation'], cable_delay=g['cable_delay'],
               n_samples=int(g['N']), fs=float(g['fs']))
    vrms, vrms_e = so.vrms_from_filters(float(g['fs']))
    assert vrms == float(g['vrms']) and vrms_e == float(g['vrms_efield'])
    n_same = n_all = n_cand = n_trig = 0
    for gi in groups:
        res = so.simulate_event_group_array(_showers(g, gi, kL), g['centres'][stations], g['rel_pos'], g['ice'], vrms, vrms_e,
                                            station_kw=skw, att_model=str(g['att_model']), n_freq=int(g['n_freq']),
                                            distance_cut_coefficients=DCUT, trigger=trigger)
        for s, o in zip(stations, res):
            n_all += 1
            if len(o['rays']) != g['ev_n_rays'][gi, s]:
                continue   # the reference's first-root noise changed its solution count
            n_same += 1
            assert o['candidate'] == bool(g['ev_candidate'][gi, s]) and o['triggered'] == bool(g['ev_triggered'][gi, s]), (gi, s)
            if o['candidate']:
                n_cand += 1
                n_trig += o['triggered']
                assert o['L'] == g['ev_L'][gi, s] and abs(o['t_min'] - g['ev_t_min'][gi, s]) < 1e-3
                ref = g['ev_maxV'][gi, s]
                assert np.all(np.abs(np.max(np.abs(o['V']), axis=1) - ref) <= amp_tol * np.max(ref)), (gi, s)
    return n_same / n_all, n_cand, n_trig


def test_rnog_array_vs_reference():
    """35 stations x 24 channels of RNO_array.json, greenland_simple + GL1, distance cut (BASELINE configs[2])"""
    g = golden('array_rnog.npz')
    n_groups, n_st = g['ev_n_rays'].shape
    trig_groups = np.flatnonzero(g['ev_triggered'].any(axis=1))
    groups = sorted(set(trig_groups[:6]) | set(range(0, n_groups, 8)))
    same, n_cand, n_trig = _compare(g, groups, np.arange(n_st), np.ones(len(g['group'])))
    assert same > 0.97 and n_cand >= 10 and n_trig >= 4


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'array_gen2.npz')), reason='fixture not generated')
def test_gen2_array_vs_reference():
    """200 stations x 5 channels, HAD + EM groups with the k_L the reference drew, 2-of-5 high/low coincidence
    (BASELINE configs[4])"""
    g = golden('array_gen2.npz')
    n_groups, n_st = g['ev_n_rays'].shape
    vr = float(g['vrms'])
    trig = dict(trigger='high_low', n_coincidences=int(g['trigger_n_coincidences']),
                threshold_high=float(g['trigger_threshold_sigma']) * vr, threshold_low=-float(g['trigger_threshold_sigma']) * vr,
                high_low_window=float(g['trigger_high_low_window']), coinc_window=float(g['trigger_coinc_window']))
    seen = np.flatnonzero(g['ev_n_rays'].sum(axis=0) > 0)
    gt, st_t = np.nonzero(g['ev_triggered'])     # a subset that holds triggers: every 4th group plus three triggered ones
    groups = sorted(set(range(0, n_groups, 4)) | set(gt[::8][:3].tolist()))
    stations = sorted(set(seen[::3].tolist()) | set(st_t[::8][:3].tolist()))
    same, n_cand, n_trig = _compare(g, groups, stations, g['k_L'], trigger=trig)
    assert same > 0.97 and n_cand >= 5 and n_trig >= 2
