"""Golden vectors for the envelope trigger (NuRadioReco/modules/trigger/envelopeTrigger.py:14-136): per channel the trace is
band-pass filtered (channel.get_filtered_trace(passband, 'butter', order)), |scipy.signal.hilbert| compared with the threshold,
then highLowThreshold.get_majority_logic over the channels.  Events of station S5 (N = 256) through the reference chain with
that trigger; two settings (2-of-5 in 20 ns, 1-of-5).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_envelope.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def run(n=260, seed=53, N=256):
    det = rh.StationS5(n_samples=N, fs=2.0)
    cfg = rh.default_config()
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    ev = rh.random_events(n, seed, rmax=1500., energy=2e18)
    settings = [dict(kind='envelope', passband=[0.13, 0.3], order=4, threshold=2.0 * vrms, coinc_window=20., number_coincidences=2),
                dict(kind='envelope', passband=[0.08, 0.18], order=2, threshold=1.0 * vrms, coinc_window=50., number_coincidences=1)]
    out = dict(N=N, fs=2.0, vrms=vrms, vrms_efield=vrms_e, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), att_model='SP1', n_freq=25,
               askaryan_model='Alvarez2009', antenna='analytic_VPol', cable_delay=0., det_pos=det.pos,
               det_orientation=np.array(det.orientation), vertex=ev['vertex'], zenith=ev['zenith'], azimuth=ev['azimuth'],
               energy=ev['energy'], shower_type=np.array(['HAD'] * n))
    for si, trg in enumerate(settings):
        cand, trig, ttime, nrays, L = (np.zeros(n, bool), np.zeros(n, bool), np.full(n, np.nan), np.zeros(n, np.int32), np.zeros(n, np.int64))
        V_list, V_ev = [], []
        for i in range(n):
            sh = rh.make_shower(i, ev['vertex'][i], ev['zenith'][i], ev['azimuth'][i], ev['energy'][i], 'HAD')
            o = rh.simulate_event(i, sh, det, prop, ice, cfg, vrms, vrms_e, trigger=trg)
            cand[i], trig[i], nrays[i], L[i] = o['candidate'], o['triggered'], len(o['rays']), o['L']
            ttime[i] = o.get('trigger_time', np.nan) - (o['t_min'] if o['candidate'] else 0.)
            if si == 0 and o['candidate'] and len(V_ev) < 40:
                V_ev.append(i)
                V_list.append(o['V'])
        print('setting', si, 'candidates', cand.sum(), 'triggered', trig.sum())
        out.update({'s%d_passband' % si: np.array(trg['passband']), 's%d_order' % si: trg['order'], 's%d_threshold' % si: trg['threshold'],
                    's%d_coinc_window' % si: trg['coinc_window'], 's%d_n_coincidences' % si: trg['number_coincidences'],
                    's%d_triggered' % si: trig, 's%d_trigger_time' % si: ttime})
        if si == 0:
            out.update(ev_candidate=cand, ev_n_rays=nrays, ev_L=L, V_events=np.array(V_ev), V_offsets=np.cumsum([0] + [v.shape[1] for v in V_list]),
                       V_concat=np.concatenate(V_list, axis=1))
    np.savez_compressed(os.path.join(OUT, 'chain_envelope_N256.npz'), **out)


if __name__ == '__main__':
    run()
