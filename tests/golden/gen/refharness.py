"""Harness that drives the REAL reference (NuRadioMC pure-Python path) to make golden vectors.

Runs ONLY in the build container:

    cp -r /root/reference /tmp/refcopy          # never import from /root/reference (install.sh side effects)
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/<script>.py

Nothing here ships to the GPU box as product code; the .npz files it writes under
tests/golden/ are data (inputs + expected outputs).

It calls the reference's own functions -- `simulation.calculate_sim_efield`
(NuRadioMC/simulation/simulation.py:93), `apply_det_response_sim` (:465),
`apply_det_response` (:530), `simpleThreshold.triggerSimulator` -- on framework objects,
with a duck-typed detector standing in for the TinyDB detector.
"""
import logging
import numpy as np

logging.disable(logging.WARNING)

from NuRadioReco.utilities import units  # noqa: E402
import datetime  # noqa: E402
import NuRadioReco.framework.particle  # noqa: E402
import NuRadioReco.framework.event  # noqa: E402
import NuRadioReco.framework.station  # noqa: E402
import NuRadioReco.framework.sim_station  # noqa: E402
import NuRadioReco.framework.radio_shower  # noqa: E402
from NuRadioReco.framework.parameters import showerParameters as shp  # noqa: E402
from NuRadioReco.framework.parameters import electricFieldParameters as efp  # noqa: E402
from NuRadioReco.framework.parameters import channelParameters as chp  # noqa: E402
import NuRadioReco.modules.channelBandPassFilter  # noqa: E402
import NuRadioReco.modules.trigger.simpleThreshold  # noqa: E402
import NuRadioReco.modules.trigger.highLowThreshold  # noqa: E402
from NuRadioReco.utilities import signal_processing  # noqa: E402
from NuRadioMC.simulation import simulation  # noqa: E402
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402


class StationS5:
    """Duck-typed detector: one station, `n_ch` analytic dipoles on a vertical string.

    SURVEY.md section 8(d): 5 `analytic_VPol` at (0,0,-100...-104 m), orientation (0,0),
    rotation (90deg,90deg) as in NuRadioMC/test/Veff/dipole_100m.json, cable delay 0.
    """

    def __init__(self, n_samples=4096, fs=2.0, n_ch=5, z_top=-100., dz=-1., station_id=101,
                 antenna='analytic_VPol', cable_delay=0., xy=(0., 0.), orientation=None):
        self.station_id = station_id
        self.n_samples = n_samples
        self.fs = fs
        self.pos = np.array([[xy[0], xy[1], z_top + dz * i] for i in range(n_ch)], dtype=float)
        self.antenna = antenna if isinstance(antenna, (list, tuple)) else [antenna] * n_ch
        self.cable_delay = (list(cable_delay) if hasattr(cable_delay, '__len__') else [cable_delay] * n_ch)
        self.orientation = [0., 0., 90. * units.deg, 90. * units.deg]
        self.orientations = None if orientation is None else [list(o) for o in orientation]  # per channel

    def get_station_ids(self):
        return [self.station_id]

    def get_channel_ids(self, station_id):
        return list(range(len(self.pos)))

    def get_sampling_frequency(self, station_id, channel_id):
        return self.fs

    def get_number_of_samples(self, station_id, channel_id):
        return self.n_samples

    def get_relative_position(self, station_id, channel_id):
        return self.pos[channel_id].copy()

    def get_absolute_position(self, station_id):
        return np.zeros(3)

    def get_cable_delay(self, station_id, channel_id, trigger=False):
        return self.cable_delay[channel_id]

    def get_antenna_model(self, station_id, channel_id, zenith=None):
        return self.antenna[channel_id]

    def get_antenna_orientation(self, station_id, channel_id):
        return list(self.orientation) if self.orientations is None else list(self.orientations[channel_id])

    def get_site(self, station_id):
        return 'southpole'

    def is_channel_noiseless(self, station_id, channel_id):
        return False


def default_config(model='Alvarez2009', ice_model='southpole_2015', fs=2.0, n_freq=25,
                   attenuation_model='SP1', delta_C_cut=0.698):
    return {
        'sampling_rate': fs, 'seed': 1235, 'noise': False, 'split_event_time_diff': 1e6,
        'speedup': {'minimum_weight_cut': 1e-5, 'delta_C_cut': delta_C_cut, 'redo_raytracing': False,
                    'min_efield_amplitude': 2, 'distance_cut': False, 'amp_per_ray_solution': True},
        'propagation': {'module': 'analytic', 'ice_model': ice_model, 'attenuation_model': attenuation_model,
                        'attenuate_ice': True, 'n_freq': n_freq, 'focusing': False, 'focusing_limit': 2,
                        'n_reflections': 0, 'birefringence': False},
        'signal': {'model': model, 'zerosignal': False, 'polarization': 'auto', 'ePhi': 0.,
                   'shower_type': None},
        'trigger': {'noise_temperature': 300, 'Vrms': None},
    }


_bp = NuRadioReco.modules.channelBandPassFilter.channelBandPassFilter()
_trig = NuRadioReco.modules.trigger.simpleThreshold.triggerSimulator()
_hl = NuRadioReco.modules.trigger.highLowThreshold.triggerSimulator()
import NuRadioReco.modules.trigger.envelopeTrigger  # noqa: E402
_env = NuRadioReco.modules.trigger.envelopeTrigger.triggerSimulator()
FILTERS = [dict(passband=[80 * units.MHz, 1000 * units.GHz], filter_type='butter', order=2),
           dict(passband=[0, 500 * units.MHz], filter_type='butter', order=10)]


def filter_amp(evt, station, det):
    """examples/01_Veff_simulation/T02RunSimulation.py:18-22"""
    for kw in FILTERS:
        _bp.run(evt, station, det, **kw)


def vrms_from_filters(config, noise_temperature=300.):
    """simulation.py:1301-1376 for the filter chain above (same 10000-point grid)."""
    ff = np.linspace(0, 0.5 * config['sampling_rate'], 10000)
    filt = np.ones_like(ff, dtype=complex)
    for kw in FILTERS:
        filt *= signal_processing.get_filter_response(ff, kw['passband'], kw['filter_type'], kw['order'])
    bandwidth = np.trapz(np.abs(filt) ** 2, ff)
    max_amp = np.abs(filt).max()
    vrms = signal_processing.calculate_vrms_from_temperature(noise_temperature, bandwidth=bandwidth)
    return vrms, vrms / max_amp / units.m


def distance_cut_function(config):
    """simulation.py:1398-1409"""
    poly = np.polynomial.polynomial.Polynomial(config['speedup']['distance_cut_coefficients'])

    def get_distance_cut(shower_energy):
        if shower_energy <= 0:
            return 100 * units.m
        return max(100 * units.m, 10 ** poly(np.log10(shower_energy)))
    return get_distance_cut


def make_shower(shower_id, vertex, zenith, azimuth, energy, shower_type, vertex_time=0.):
    sh = NuRadioReco.framework.radio_shower.RadioShower(shower_id)
    sh[shp.zenith] = zenith
    sh[shp.azimuth] = azimuth
    sh[shp.energy] = energy
    sh[shp.vertex] = np.array(vertex, dtype=float)
    sh[shp.vertex_time] = vertex_time
    sh[shp.type] = shower_type
    return sh


def random_events(n, seed, rmax=4000., zmin=-2700., energy=3e17, zmax=0.):
    """Synthetic 1 EeV-class event list: uniform in r^2 and z, isotropic."""
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.uniform(0, rmax ** 2, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    z = rng.uniform(zmin, zmax, n)
    zen = np.arccos(rng.uniform(-1, 1, n))
    az = rng.uniform(0, 2 * np.pi, n)
    vertex = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    return dict(vertex=vertex, zenith=zen, azimuth=az, energy=np.full(n, energy))


def make_propagator(config, det):
    ice = medium.get_ice_model(config['propagation']['ice_model'])
    prop = ray.ray_tracing(ice, log_level=logging.ERROR, config=config, detector=det, use_cpp=False,
                           compile_numba=False)
    return ice, prop


def simulate_event(ev_id, shower, det, prop, ice, config, vrms, vrms_efield, trigger_sigma=3.0, distance_cut=None,
                   trigger=None, split=None):
    """One event group through the reference, following simulation.run() (simulation.py:1454-1600).
    `shower` is one RadioShower or the list of showers of the event group.

    Returns a dict of everything the parity tests compare.
    """
    showers = list(shower) if isinstance(shower, (list, tuple)) else [shower]
    sid = det.get_station_ids()[0]
    evt = NuRadioReco.framework.event.Event(ev_id, 0)
    station = NuRadioReco.framework.station.Station(sid)
    sim_station = NuRadioReco.framework.sim_station.SimStation(sid)
    sim_station.set_is_neutrino()
    station.set_sim_station(sim_station)
    evt.set_station(station)
    for sh_ in showers:
        evt.add_sim_shower(sh_)

    out = dict(rays=[], candidate=False, triggered=False, L=0, t_min=np.nan)
    candidate = False
    for ch in det.get_channel_ids(sid):
        ss = simulation.calculate_sim_efield(
            showers=showers, station_id=sid, channel_id=ch, det=det, propagator=prop, medium=ice,
            config=config, min_efield_amplitude=float(config['speedup']['min_efield_amplitude']) * vrms_efield,
            distance_cut=distance_cut)
        if ss.is_candidate():
            candidate = True
        if len(ss.get_electric_fields()) == 0:
            continue
        # capture the efields before the per-efield module (it does not modify them)
        for ef in ss.get_electric_fields():
            rt = ef[efp.raytracing_solution]
            spec = ef.get_frequency_spectrum().copy()
            out['rays'].append(dict(
                channel=ch, iS=ef.get_ray_tracing_solution_id(), shower_id=ef.get_shower_id(),
                C0=rt['ray_tracing_C0'], C1=rt['ray_tracing_C1'], type=rt['ray_tracing_solution_type'],
                reflection=rt.get('ray_tracing_reflection', 0), reflection_case=rt.get('ray_tracing_reflection_case', 1),
                zenith=ef[efp.zenith], azimuth=ef[efp.azimuth], D=ef[efp.nu_vertex_distance],
                T=ef[efp.nu_vertex_propagation_time], view=ef[efp.nu_viewing_angle],
                pol_angle=ef[efp.polarization_angle], launch=np.array(ef[efp.launch_vector]),
                t0=ef.get_trace_start_time(), spec=spec,
                r_theta=ef[efp.reflection_coefficient_theta] if ef.has_parameter(efp.reflection_coefficient_theta) else 1.,
                r_phi=ef[efp.reflection_coefficient_phi] if ef.has_parameter(efp.reflection_coefficient_phi) else 1.,
                max_efield=np.max(np.abs(ef.get_trace()))))
        simulation.apply_det_response_sim(ss, det, config, filter_amp)
        k0 = len(out['rays']) - len(ss.get_electric_fields())
        for k, sc in enumerate(ss.iter_channels()):
            r = out['rays'][k0 + k]
            assert sc.get_id() == r['channel'] and sc.get_ray_tracing_solution_id() == r['iS']
            r['simch_spec'] = sc.get_frequency_spectrum().copy()
            r['simch_t0'] = sc.get_trace_start_time()
            r['max_amp_ray'] = sc[chp.maximum_amplitude_envelope]
            r['signal_time'] = sc[chp.signal_time]
        station.add_sim_station(ss)
    out['candidate'] = candidate
    out['k_L'] = showers[0][shp.k_L] if showers[0].has_parameter(shp.k_L) else np.nan
    out['k_L_all'] = [sh_[shp.k_L] if sh_.has_parameter(shp.k_L) else np.nan for sh_ in showers]
    if len(station.get_sim_station().get_electric_fields()) == 0 or not candidate:
        return out
    if split is not None:
        # simulation.run() :1566-1600: group_into_events, then detector response + trigger per (sub-)event
        evt.set_event_time(datetime.datetime(2018, 1, 1))
        evt.add_particle(NuRadioReco.framework.particle.Particle(0))
        out['sub'] = []
        for sub in simulation.group_into_events(station, evt, True, split):
            stn = sub.get_station()
            simulation.apply_det_response(sub, det, config, filter_amp, add_noise=False)
            _trig.run(sub, stn, det, threshold=trigger_sigma * vrms, triggered_channels=None, number_concidences=1,
                      trigger_name='simple_threshold')
            chans = [stn.get_channel(c) for c in det.get_channel_ids(sid)]
            members = sorted((sc.get_id(), sc.get_shower_id(), sc.get_ray_tracing_solution_id())
                             for sc in stn.get_sim_station().iter_channels())
            out['sub'].append(dict(triggered=bool(stn.has_triggered()), L=chans[0].get_number_of_samples(),
                                   t_min=chans[0].get_trace_start_time(), V=np.array([c.get_trace() for c in chans]),
                                   members=members))
        out['triggered'] = any(q['triggered'] for q in out['sub'])
        return out
    simulation.apply_det_response(evt, det, config, filter_amp, add_noise=False)
    if trigger is not None and trigger.get('kind') == 'envelope':   # envelopeTrigger.triggerSimulator.run (:47-136)
        _env.run(evt, station, det, passband=trigger['passband'], order=trigger['order'], threshold=trigger['threshold'],
                 coinc_window=trigger['coinc_window'], number_coincidences=trigger['number_coincidences'], triggered_channels=None,
                 trigger_name='envelope')
        tr_ = station.get_trigger('envelope')
        out['trigger_time'] = tr_.get_trigger_time() if tr_.has_triggered() else np.nan
    elif trigger is not None and trigger.get('kind') == 'high_low':   # highLowThreshold.triggerSimulator.run (:160-335)
        _hl.run(evt, station, det, threshold_high=trigger['threshold_high'], threshold_low=trigger['threshold_low'],
                high_low_window=trigger['high_low_window'], coinc_window=trigger['coinc_window'],
                number_concidences=trigger['number_concidences'], triggered_channels=None, trigger_name='high_low')
    else:
        _trig.run(evt, station, det, threshold=trigger_sigma * vrms, triggered_channels=None,
                  number_concidences=1, trigger_name='simple_threshold')
    out['triggered'] = bool(station.has_triggered())
    chans = [station.get_channel(c) for c in det.get_channel_ids(sid)]
    out['L'] = chans[0].get_number_of_samples()
    out['t_min'] = chans[0].get_trace_start_time()
    out['V'] = np.array([c.get_trace() for c in chans])
    return out
