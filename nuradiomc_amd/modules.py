"""Drop-in for NuRadioReco.modules.efieldToVoltageConverter.efieldToVoltageConverter
(NuRadioReco/modules/efieldToVoltageConverter.py:23-352): same begin / run / end, same effect on `station`
(one Channel per requested channel id on the event's common time grid, trace_start_time = t_min), the
convolution with the antenna response running on the GPU (Station.efield_to_voltage).

Objects are duck-typed exactly as the reference uses them: `station.get_sim_station().get_electric_fields_for_channels`,
`electric_field.get_trace() / get_trace_start_time() / get_sampling_rate() / get_number_of_samples() / get_position()`,
`electric_field[efp.zenith]`, `det.get_cable_delay / get_relative_position / get_antenna_model /
get_antenna_orientation / get_channel_ids / get_number_of_samples / get_sampling_frequency`.
"""
import time
import logging
import numpy as np
from .station import Station

logger = logging.getLogger('nuradiomc_amd.efieldToVoltageConverter')


def _efield_param(ef, name):
    """electric_field[efp.<name>] with or without NuRadioReco importable"""
    try:
        from NuRadioReco.framework.parameters import electricFieldParameters as efp
        return ef[getattr(efp, name)]
    except ImportError:
        return ef[name]


class efieldToVoltageConverter:
    def __init__(self, log_level=logging.NOTSET, ctx=None, channel_factory=None, antenna_models=None):
        """antenna_models: {name the detector description uses: TabulatedAntenna} for the tabulated patterns (the
        reference's AntennaPatternProvider loads them from its antenna-model directory; analytic names need no entry)"""
        self.__t = 0
        self._antenna_models = dict(antenna_models or {})
        self._ctx = ctx
        self._channel_factory = channel_factory
        self._stations = {}
        logger.setLevel(log_level)
        self.begin()

    def begin(self, debug=False, uncertainty=None, time_resolution=None, pre_pulse_time=200., post_pulse_time=400.,
              caching=True):
        if time_resolution is not None:
            logger.warning("`time_resolution` is deprecated and will be removed in the future. The argument is ignored.")
        self.__debug = debug
        self.__pre_pulse_time = pre_pulse_time
        self.__post_pulse_time = post_pulse_time
        # uncertainties exactly as the reference draws them (efieldToVoltageConverter.py:83-90): the systematic ones once, here, from
        # numpy's global generator in this order -- antenna position offsets 'sys_dx' / 'sys_dy' / 'sys_dz' (drawn and kept; the
        # reference's run() never reads them either), then one gain factor per channel of 'sys_amp'; the statistical gain 'amp' is
        # drawn per (channel, electric field) inside run() (:320-321).  As in the reference the caller's dictionary is modified.
        self.__uncertainty = uncertainty or {}
        for key in ['sys_dx', 'sys_dy', 'sys_dz']:
            if key in self.__uncertainty:
                self.__uncertainty[key] = np.random.normal(0, self.__uncertainty[key])
        if 'sys_amp' in self.__uncertainty:
            for iCh in self.__uncertainty['sys_amp']:
                self.__uncertainty['sys_amp'][iCh] = np.random.normal(1, self.__uncertainty['sys_amp'][iCh])

    def _make_channel(self, channel_id):
        if self._channel_factory is not None:
            return self._channel_factory(channel_id)
        import NuRadioReco.framework.channel
        return NuRadioReco.framework.channel.Channel(channel_id)

    def _station_for(self, det, station_id, channel_ids, n_samples, fs):
        key = (id(det), station_id, tuple(channel_ids), n_samples, fs)
        if key not in self._stations:
            if self._ctx is None:
                from .context import Context
                self._ctx = Context((1.78, 0.423, 77.), 'SP1', device=0)  # the ice model is irrelevant here
            pos = [np.asarray(det.get_relative_position(station_id, c), float) for c in channel_ids]
            self._stations[key] = Station(
                self._ctx, pos, antenna=[self._antenna_models.get(m, m) for m in
                                         (det.get_antenna_model(station_id, c, None) for c in channel_ids)],
                orientation=[det.get_antenna_orientation(station_id, c) for c in channel_ids],
                cable_delay=[det.get_cable_delay(station_id, c) for c in channel_ids], n_samples=n_samples,
                sampling_rate=fs, pre_pulse_time=self.__pre_pulse_time, post_pulse_time=self.__post_pulse_time,
                readout_length=max(det.get_number_of_samples(station_id, c) / det.get_sampling_frequency(station_id, c)
                                   for c in channel_ids), filters=())
        return self._stations[key]

    def run(self, evt, station, det, channel_ids=None):
        t = time.time()
        sim_station = station.get_sim_station()
        sid = sim_station.get_id()
        if len(sim_station.get_electric_fields()) == 0:
            raise LookupError(f"station {station.get_id()} has no efields")
        if channel_ids is None:
            channel_ids = det.get_channel_ids(sid)
        channel_ids = list(channel_ids)
        traces, t0, zen, az, chan = [], [], [], [], []
        n_samples = fs = None
        for i, channel_id in enumerate(channel_ids):
            for ef in sim_station.get_electric_fields_for_channels([channel_id]):
                d = np.linalg.norm(np.asarray(det.get_relative_position(sid, channel_id)) - np.asarray(ef.get_position()))
                if d / 0.001 > 0.01:
                    raise NotImplementedError("efields away from the antenna (air-shower mode) are not provided")
                # gain uncertainties (:320-324): scalar factors on the field's voltage, i.e. on the field itself -- drawn for every
                # electric field the reference's loop meets, in its order (also for those without a start time: zero traces there)
                gain = 1.
                if 'amp' in self.__uncertainty:
                    gain *= np.random.normal(1, self.__uncertainty['amp'][channel_id])
                if 'sys_amp' in self.__uncertainty:
                    gain *= self.__uncertainty['sys_amp'][channel_id]
                if np.isnan(ef.get_trace_start_time()):
                    continue
                tr = np.asarray(ef.get_trace(), float)
                n_samples, fs = tr.shape[-1], ef.get_sampling_rate()
                traces.append(tr[1:3] if gain == 1. else tr[1:3] * gain)
                t0.append(ef.get_trace_start_time())
                zen.append(_efield_param(ef, 'zenith'))
                az.append(_efield_param(ef, 'azimuth'))
                chan.append(i)
        if not traces:
            raise LookupError(f"station {station.get_id()} has no efields")
        st = self._station_for(det, sid, channel_ids, n_samples, fs)
        V, t_min = st.efield_to_voltage(np.array(traces), t0, zen, az, chan)
        for i, channel_id in enumerate(channel_ids):
            channel = self._make_channel(channel_id)
            channel.set_trace(V[i], fs)
            channel.set_trace_start_time(t_min)
            station.add_channel(channel)
        self.__t += time.time() - t

    def end(self):
        from datetime import timedelta
        dt = timedelta(seconds=self.__t)
        logger.info("total time used by this module is {}".format(dt))
        return dt


class efieldToVoltageConverterPerEfield(efieldToVoltageConverter):
    """Drop-in for NuRadioReco.modules.efieldToVoltageConverterPerEfield (efieldToVoltageConverterPerEfield.py:28-101): one
    SimChannel per electric field of the sim station, V(f) = VEL_theta E_theta + VEL_phi E_phi on the efield's own grid,
    nothing below 5 MHz, trace start time = the efield's (no cable delay); the convolution runs on the GPU
    (nrhip_efield_to_voltage on the N-sample grid of the efield).  `sim_channel_factory(channel_id, efield)` builds the
    object to fill (default: NuRadioReco's SimChannel with the efield's shower and ray-tracing ids)."""

    def __init__(self, log_level=logging.NOTSET, ctx=None, sim_channel_factory=None, antenna_models=None):
        super().__init__(log_level=log_level, ctx=ctx, channel_factory=None, antenna_models=antenna_models)
        self._sim_channel_factory = sim_channel_factory

    def _make_sim_channel(self, channel_id, ef):
        if self._sim_channel_factory is not None:
            return self._sim_channel_factory(channel_id, ef)
        import NuRadioReco.framework.sim_channel
        from NuRadioReco.framework.parameters import channelParameters as chp, electricFieldParameters as efp
        sc = NuRadioReco.framework.sim_channel.SimChannel(channel_id, shower_id=ef.get_shower_id(),
                                                          ray_tracing_id=ef.get_ray_tracing_solution_id())
        sc[chp.signal_ray_type] = ef[efp.ray_path_type]
        return sc

    def run(self, evt, station, det):
        sim_station = station.get_sim_station() if hasattr(station, 'get_sim_station') else station
        sid = sim_station.get_id()
        if len(sim_station.get_electric_fields()) == 0:
            raise LookupError(f"station {station.get_id()} has no efields")
        channel_ids = list(det.get_channel_ids(sid))
        for i, channel_id in enumerate(channel_ids):
            for ef in sim_station.get_electric_fields_for_channels([channel_id]):
                d = np.linalg.norm(np.asarray(det.get_relative_position(sid, channel_id)) - np.asarray(ef.get_position()))
                if d / 0.001 > 0.01:
                    raise NotImplementedError("efields away from the antenna (air-shower mode) are not provided")
                tr = np.asarray(ef.get_trace(), float)
                n_samples, fs = tr.shape[-1], ef.get_sampling_rate()
                st = self._station_for(det, sid, channel_ids, n_samples, fs)
                t0 = ef.get_trace_start_time()
                # the efield's own grid: start bin 0, no padding, the cable delay taken out again
                V, _ = st.efield_to_voltage(tr[None, 1:3], [t0], [_efield_param(ef, 'zenith')], [_efield_param(ef, 'azimuth')],
                                            [i], grid=(t0 + st.cable_delay[i], n_samples))
                sc = self._make_sim_channel(channel_id, ef)
                sc.set_trace(V[i], fs)
                sc.set_trace_start_time(t0)
                sim_station.add_channel(sc)
