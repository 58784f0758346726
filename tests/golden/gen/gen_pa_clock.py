"""Golden vectors for the clock offset of the trigger ADC (analogToDigitalConverter.get_digital_trace :327-340 with clock_offset, the
argument the phased-array trigger modules hand through: phasedArrayTrigger.py:32,124): the trace is delayed by whole ADC clock cycles
(signal_processing.delay_trace :401-472, cropped) in front of the digitiser.  Produced by the reference's own functions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_pa_clock.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
import NuRadioReco.framework.channel  # noqa: E402
from NuRadioReco.modules.analogToDigitalConverter import analogToDigitalConverter  # noqa: E402
from NuRadioReco.utilities import signal_processing  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


class Det:
    def __init__(self, adc): self.adc = adc
    def get_channel(self, sid, ch): return self.adc


class Sta:
    def get_id(self): return 11


rng = np.random.default_rng(31)
adc = analogToDigitalConverter()
out, cases = {}, []
# (samples, sampling rate [GHz], ADC rate [GHz], bits, noise count, up-sampling factor, clock offset [cycles], output)
for k, (n_samples, fs, adc_fs, nbits, ncount, up, clk, output) in enumerate([
        (1456, 2.0, 0.472, 8, 5, 4, 1, 'counts'), (1456, 2.0, 0.472, 8, 5, 4, 3, 'voltage'), (2650, 2.0, 0.5, 7, 3, 2, 2, 'counts'),
        (1000, 2.4, 0.6, 8, 4, 1, 5, 'counts')]):
    det = Det(dict(trigger_adc_nbits=nbits, trigger_adc_sampling_frequency=adc_fs, trigger_adc_noise_count=ncount))
    vrms = 1.3e-5
    tr, dig, ups, delayed = [], [], [], []
    for e in range(4):
        x = vrms * (rng.normal(0, 1., n_samples) + (9. if e % 2 else 0.) * np.exp(-0.5 * ((np.arange(n_samples) - 500) / 4.) ** 2)
                    * np.cos(0.9 * np.arange(n_samples)))
        ch = NuRadioReco.framework.channel.Channel(0)
        ch.set_trace(x.copy(), fs)
        d, f_adc = adc.get_digital_trace(Sta(), det, ch, Vrms=vrms, trigger_adc=True, adc_type='perfect_floor_comparator',
                                         return_sampling_frequency=True, adc_output=output, clock_offset=clk)
        u = signal_processing.digital_upsampling(d, f_adc, upsampling_method='fft', upsampling_factor=up)[0] if up >= 2 else d
        y, _ = signal_processing.delay_trace(x.copy(), fs, clk / adc_fs)
        tr.append(x); dig.append(np.array(d, float)); ups.append(np.array(u, float)); delayed.append(np.array(y, float))
    out['traces_%d' % k], out['digital_%d' % k], out['upsampled_%d' % k], out['delayed_%d' % k] = map(np.array, (tr, dig, ups, delayed))
    out['output_%d' % k], out['vrms_%d' % k] = output, vrms
    cases.append((n_samples, fs, adc_fs, nbits, ncount, up, clk))
    print(k, output, 'clock offset', clk, 'delayed', out['delayed_%d' % k].shape, 'digital', out['digital_%d' % k].shape, 'upsampled', out['upsampled_%d' % k].shape)
out['cases'] = np.array(cases, float)
np.savez_compressed(os.path.join(OUT, 'ref_pa_clock.npz'), **out)
