"""Golden file of the HDF5 OUTPUT format and of the event-list INPUT format (row f2), produced by
the reference's own code: NuRadioMC/simulation/output_writer_hdf5.outputWriterHDF5 (:95-527) fed as simulation.run() feeds it
(simulation.py:1454-1728) -- event groups built from an input dictionary by simulation.build_NuRadioEvents_from_hdf5 (:659-762),
calculate_sim_efield / detector response / simple threshold trigger (refharness), channelReadoutWindowCutter,
channelSignalReconstructor, _set_event_station_parameters (:1766-1784), add_event_group -> write_output_file.

Needs a REAL h5py (the other generators run with a stub): in the build container

    cp -r /root/reference /tmp/refcopy; mkdir /tmp/shims_noh5; cp -r tests/golden/gen/shims/* /tmp/shims_noh5; rm -r /tmp/shims_noh5/h5py
    mkdir /tmp/shims_noh5/numba; echo 'raise ImportError("off")' > /tmp/shims_noh5/numba/__init__.py   # conda's numba is broken
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/shims_noh5:/tmp/refcopy /opt/conda/bin/python3.9 tests/golden/gen/gen_hdf5.py

Writes tests/golden/ref_hdf5_output.npz: the input event list (the reference's input-file datasets and attributes) and every
dataset / attribute of the file the reference wrote ('out/<path>' and 'attr/<path>@<name>').
"""
import datetime
import os
import sys
import numpy as np
import h5py

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402
import NuRadioReco.framework.event  # noqa: E402
import NuRadioReco.framework.station  # noqa: E402
import NuRadioReco.framework.sim_station  # noqa: E402
import NuRadioReco.modules.channelReadoutWindowCutter  # noqa: E402
import NuRadioReco.modules.channelSignalReconstructor  # noqa: E402
from NuRadioReco.framework.parameters import channelParameters as chp  # noqa: E402
from NuRadioReco.framework.parameters import generatorAttributes as genattrs  # noqa: E402
from NuRadioReco.framework.parameters import showerParameters as shp  # noqa: E402
from NuRadioMC.simulation import simulation  # noqa: E402
from NuRadioMC.simulation.output_writer_hdf5 import outputWriterHDF5  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


class Det(rh.StationS5):
    def get_number_of_channels(self, station_id):
        return len(self.pos)


def event_list(n_groups, seed):
    """an input event list in the reference's format (NuRadioMC/EvtGen/generator.py:1023-1414 writes these datasets): single
    hadronic showers (NC), nu_e CC pairs (hadronic + electromagnetic shower, same vertex) and a later secondary interaction"""
    rng = np.random.default_rng(seed)
    base = rh.random_events(n_groups, seed, rmax=1800.)
    rows = []
    sid = 0
    for g in range(n_groups):
        v, zen, az = base['vertex'][g], base['zenith'][g], base['azimuth'][g]
        e_nu = 10 ** rng.uniform(18.3, 19.5)
        y = rng.uniform(0.1, 0.9)
        kind = g % 4
        if kind in (0, 1):     # NC: one hadronic shower
            parts = [(v, y * e_nu, 'had', 0., 1, 'nc', 14)]
        elif kind == 2:        # nu_e CC
            parts = [(v, y * e_nu, 'had', 0., 1, 'cc', 12), (v, (1 - y) * e_nu, 'em', 0., 1, 'cc', 12)]
        else:                  # nu_tau CC with a second (tau decay) shower further along the track
            d = rng.uniform(30., 250.)
            axis = -np.array([np.sin(zen) * np.cos(az), np.sin(zen) * np.sin(az), np.cos(zen)])
            v2 = v + d * axis
            if v2[2] > -1.:
                v2, d = v.copy(), 0.
            parts = [(v, y * e_nu, 'had', 0., 1, 'cc', 16), (v2, 0.5 * (1 - y) * e_nu, 'had', d / 0.299792458, 2, 'tau_had', 16)]
        for (vv, esh, typ, vt, nint, itype, flav) in parts:
            rows.append(dict(event_group_ids=g + 7, shower_ids=sid, xx=vv[0], yy=vv[1], zz=vv[2], zeniths=zen, azimuths=az,
                             energies=e_nu, shower_energies=esh, shower_type=typ, flavors=flav, n_interaction=nint,
                             interaction_type=itype, inelasticity=y, vertex_times=vt, weights=rng.uniform(0.2, 1.0)))
            sid += 1
    return {k: np.array([r[k] for r in rows]) for k in rows[0]}


def run(n_groups=120, seed=61, N=256, fs=2.0):
    det = Det(n_samples=N, fs=fs, station_id=101)
    cfg = rh.default_config()
    cfg['speedup']['amp_per_ray_solution'] = True
    cfg['weights'] = dict(weight_mode='core_mantle_crust', cross_section_type='ctw')   # config_default.yaml
    ice, prop = rh.make_propagator(cfg, det)
    vrms, vrms_e = rh.vrms_from_filters(cfg)
    fin = event_list(n_groups, seed)
    fin_attrs = dict(n_events=n_groups, fiducial_rmax=1800., fiducial_rmin=0., fiducial_zmax=0., fiducial_zmin=-2700.,
                     rmax=1800., rmin=0., zmax=0., zmin=-2700., Emin=10 ** 18.3, Emax=10 ** 19.5, thetamin=0., thetamax=np.pi,
                     phimin=0., phimax=2 * np.pi, flavors=np.array([12, 14, 16]), deposited=False, volume=np.pi * 1800. ** 2 * 2700.,
                     area=np.pi * 1800. ** 2)
    writer = outputWriterHDF5('/tmp/ref_output.hdf5', cfg, det, [101], number_of_ray_tracing_solutions=2, particle_mode=True)
    cutter = NuRadioReco.modules.channelReadoutWindowCutter.channelReadoutWindowCutter()
    recon = NuRadioReco.modules.channelSignalReconstructor.channelSignalReconstructor()
    ff = np.linspace(0, 0.5 * fs, 10000)
    from NuRadioReco.utilities import signal_processing
    filt = np.ones_like(ff, dtype=complex)
    for kw in rh.FILTERS:
        filt *= signal_processing.get_filter_response(ff, kw['passband'], kw['filter_type'], kw['order'])
    bandwidth = np.trapz(np.abs(filt) ** 2, ff)
    groups = np.unique(fin['event_group_ids'])
    n_trig = 0
    for gid in groups:
        idxs = np.atleast_1d(np.squeeze(np.argwhere(fin['event_group_ids'] == gid)))
        event_group = simulation.build_NuRadioEvents_from_hdf5(fin, fin_attrs, idxs)
        event_group.set_event_time(datetime.datetime(2018, 1, 1))   # simulation.run(): event_group.set_event_time(self._evt_time)
        simulation.calculate_particle_weight(event_group, idxs[0], cfg, fin)   # :880-903, Earth absorption of the primary
        sid = 101
        station = NuRadioReco.framework.station.Station(sid)
        sim_station = NuRadioReco.framework.sim_station.SimStation(sid)
        sim_station.set_is_neutrino()
        station.set_sim_station(sim_station)
        event_group.set_station(station)
        candidate = False
        for ch in det.get_channel_ids(sid):
            ss = simulation.calculate_sim_efield(showers=event_group.get_sim_showers(), station_id=sid, channel_id=ch, det=det,
                                                 propagator=prop, medium=ice, config=cfg,
                                                 min_efield_amplitude=float(cfg['speedup']['min_efield_amplitude']) * vrms_e,
                                                 distance_cut=None)
            candidate = candidate or ss.is_candidate()
            if len(ss.get_electric_fields()) == 0:
                continue
            simulation.apply_det_response_sim(ss, det, cfg, rh.filter_amp)
            station.add_sim_station(ss)
        if len(station.get_sim_station().get_electric_fields()) == 0 or not candidate:
            continue
        events = simulation.group_into_events(station, event_group, True, cfg['split_event_time_diff'])
        buf = {sid: {}}
        for evt in events:
            stn = evt.get_station()
            simulation.apply_det_response(evt, det, cfg, rh.filter_amp, add_noise=False)
            rh._trig.run(evt, stn, det, threshold=3.0 * vrms, triggered_channels=None, number_concidences=1,
                         trigger_name='simple_threshold')
            if not stn.has_triggered():
                continue
            cutter.run(evt, stn, det)
            recon.run(evt, stn, det)
            evt.set_parameter(genattrs.Vrms, vrms)                       # simulation._set_event_station_parameters :1766-1784
            evt.set_parameter(genattrs.dt, 1. / cfg['sampling_rate'])
            evt.set_parameter(genattrs.Tnoise, 300.)
            evt.set_parameter(genattrs.bandwidth, bandwidth)
            for channel in stn.iter_channels():
                channel[chp.Vrms_NuRadioMC_simulation] = vrms
                channel[chp.bandwidth_NuRadioMC_simulation] = bandwidth
            buf[sid][evt.get_id()] = evt
            n_trig += 1
        if buf[sid]:
            writer.add_event_group(buf)
    ok = writer.write_output_file()
    print('groups', len(groups), 'triggered events', n_trig, 'file written', ok)
    out = {'in/' + k: (v.astype('S') if v.dtype.kind == 'U' else v) for k, v in fin.items()}
    out.update({'in_attr/' + k: np.asarray(v) for k, v in fin_attrs.items()})
    out.update(N=N, fs=fs, vrms=vrms, vrms_efield=vrms_e, bandwidth=bandwidth, ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]),
               det_pos=det.pos, station_id=101, seed=cfg['seed'])

    def plain(val):
        a = np.asarray(val.encode() if isinstance(val, str) else val)
        if a.dtype.kind == 'O':
            a = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in a.ravel()]).astype('S').reshape(a.shape)
        return a

    def walk(name, obj):
        if isinstance(obj, h5py.Dataset):
            v = obj[()]
            if v.dtype.kind == 'O':
                v = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v]).astype('S')
            out['out/' + name] = v
        for a, val in obj.attrs.items():
            out['attr/%s@%s' % (name, a)] = plain(val)
    with h5py.File('/tmp/ref_output.hdf5', 'r') as f:
        for a, val in f.attrs.items():
            out['attr/@%s' % a] = plain(val)
        f.visititems(walk)
        print(sorted(k for k in out if k.startswith('out/')))
    np.savez_compressed(os.path.join(OUT, 'ref_hdf5_output.npz'), **out)


if __name__ == '__main__':
    run()
