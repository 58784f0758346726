"""Golden vectors for the digitised phased-array trigger, produced by the reference's own functions: the trigger ADC
(NuRadioReco/modules/analogToDigitalConverter.py: get_digital_trace :254-373 -- resampling to 5 GHz, linear-interpolation
down-sampling to the ADC rate :432-463, perfect_floor_comparator :14-80 with the voltage range from Vrms and trigger_adc_noise_count),
FFT up-sampling (utilities/signal_processing.digital_upsampling :111-190), beam forming with saturation (phasedArrayBase.phase_signals
:183-215) and the power sums with ADC-count rounding (power_sum :217-271), for adc_output 'voltage' and 'counts'.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_pa_adc.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
import NuRadioReco.framework.channel  # noqa: E402
from NuRadioReco.modules.analogToDigitalConverter import analogToDigitalConverter  # noqa: E402
from NuRadioReco.modules.phasedarray.phasedArrayBase import PhasedArrayBase  # noqa: E402
from NuRadioReco.utilities import signal_processing  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


class Det:
    def __init__(self, pos, cable, adc): self.pos, self.cable, self.adc = pos, cable, adc
    def get_relative_position(self, sid, ch): return self.pos[ch]
    def get_cable_delay(self, sid, ch): return self.cable[ch]
    def get_channel(self, sid, ch): return self.adc


class Sta:
    def get_id(self): return 11


rng = np.random.default_rng(29)
adc = analogToDigitalConverter()
out, cases = {}, []
# (channels, samples, sampling rate [GHz], ADC rate [GHz], bits, noise count, up-sampling factor, window, step, beams, output)
for k, (n_ch, n_samples, fs, adc_fs, nbits, ncount, up, window, step, n_beams, output) in enumerate([
        (4, 1456, 2.0, 0.472, 8, 5, 4, 24, 8, 11, 'voltage'), (4, 1456, 2.0, 0.472, 8, 5, 4, 24, 8, 11, 'counts'),
        (4, 2650, 2.0, 0.5, 7, 3, 2, 16, 8, 9, 'counts'), (6, 1000, 2.4, 0.6, 8, 4, 1, 12, 6, 7, 'voltage')]):
    z = -100. + np.sort(rng.uniform(-8., 0., n_ch))[::-1]
    pos = np.stack([np.zeros(n_ch), np.zeros(n_ch), z], axis=1)
    cable = rng.uniform(0., 6., n_ch)
    det = Det(pos, cable, dict(trigger_adc_nbits=nbits, trigger_adc_sampling_frequency=adc_fs, trigger_adc_noise_count=ncount))
    chans = list(range(n_ch))
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), n_beams))
    vrms = 1.3e-5
    pa = PhasedArrayBase()
    pa.begin()
    ev_tr, ev_dig, ev_up, ev_pow = [], [], [], []
    for e in range(5):
        traces = {c: vrms * (rng.normal(0, 1., n_samples) + (9. if e % 2 else 0.) * np.exp(-0.5 * ((np.arange(n_samples) - 500 - 5 * c) / 4.) ** 2)
                             * np.cos(0.9 * np.arange(n_samples))) for c in chans}
        dig, ups = {}, {}
        for c in chans:
            ch = NuRadioReco.framework.channel.Channel(c)
            ch.set_trace(traces[c], fs)
            d, f_adc = adc.get_digital_trace(Sta(), det, ch, Vrms=vrms, trigger_adc=True, adc_type='perfect_floor_comparator',
                                             return_sampling_frequency=True, adc_output=output)
            dig[c] = np.array(d, float)
            u, f_up = signal_processing.digital_upsampling(d, f_adc, upsampling_method='fft', upsampling_factor=up) if up >= 2 else (d, f_adc)
            ups[c] = np.array(u, float)
        rolls = pa.calculate_time_delays(Sta(), det, chans, phasing_angles=angles, ref_index=1.75, sampling_frequency=f_up)
        phased = pa.phase_signals(ups, rolls, adc_output=output, saturation_bits=8)
        powers = [pa.power_sum(coh_sum=p, window=window, step=step, adc_output=output)[0] for p in phased]
        ev_tr.append(np.array([traces[c] for c in chans])); ev_dig.append(np.array([dig[c] for c in chans]))
        ev_up.append(np.array([ups[c] for c in chans])); ev_pow.append(np.array(powers))
    out['traces_%d' % k], out['digital_%d' % k], out['upsampled_%d' % k], out['power_%d' % k] = map(np.array, (ev_tr, ev_dig, ev_up, ev_pow))
    out['rolls_%d' % k] = np.array([[r[c] for c in chans] for r in rolls])
    out['pos_%d' % k], out['cable_%d' % k], out['angles_%d' % k], out['output_%d' % k], out['vrms_%d' % k] = pos, cable, angles, output, vrms
    cases.append((n_ch, n_samples, fs, adc_fs, nbits, ncount, up, window, step, n_beams))
    print(k, output, 'digital', out['digital_%d' % k].shape, 'upsampled', out['upsampled_%d' % k].shape, 'max power', out['power_%d' % k].max())
out['cases'] = np.array(cases, float)
np.savez_compressed(os.path.join(OUT, 'ref_pa_adc.npz'), **out)
