"""Golden vectors for the other up-sampling methods and the envelope mode of the phased-array trigger, produced by the reference's own
functions: utilities/signal_processing.digital_upsampling with upsampling_method 'lin' and 'fir' (:111-190, upsampling_fir :192-234:
zero stuffing + scipy.signal.firwin low pass, coefficients rounded to 1 / coeff_gain) on ADC-count and voltage traces, and
PhasedArrayBase.hilbert_envelope (phasedArrayBase.py:337-367: FIR Hilbert transformer with rounded coefficients, max + 3/8 min
magnitude estimate; and the ideal transformer: scipy.signal.hilbert with the exact magnitude) on coherent sums.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_pa_modes.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioReco.modules.phasedarray.phasedArrayBase import PhasedArrayBase  # noqa: E402
from NuRadioReco.utilities import signal_processing  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
rng = np.random.default_rng(41)
out, up_cases, hil_cases = {}, [], []
# (samples, ADC rate [GHz], method, factor, coeff_gain, taps, counts?)
for k, (n, fs, method, up, gain, taps, counts) in enumerate([
        (342, 0.472, 'lin', 4, 1, 45, 1), (342, 0.472, 'lin', 2, 1, 45, 0), (343, 0.5, 'lin', 3, 1, 45, 1),
        (342, 0.472, 'fir', 4, 128, 31, 1), (342, 0.472, 'fir', 2, 1, 45, 0), (660, 0.5, 'fir', 2, 256, 45, 1),
        (343, 0.6, 'fir', 4, 64, 23, 0)]):
    x = 6. * rng.normal(0, 1., n) + 40. * np.exp(-0.5 * ((np.arange(n) - 120) / 3.) ** 2) * np.cos(0.8 * np.arange(n))
    x = np.round(x) if counts else 1.1e-5 * x
    y, new_fs = signal_processing.digital_upsampling(x, fs, upsampling_method=method, upsampling_factor=up, coeff_gain=gain,
                                                     filter_taps=taps)
    up_cases.append([n, fs, {'lin': 1, 'fir': 2}[method], up, gain, taps, counts, new_fs])
    out['up_in_%d' % k] = x
    out['up_out_%d' % k] = np.asarray(y, float)
pa = PhasedArrayBase()
pa.begin()
# (samples, taps, coeff_gain, counts?)
for k, (n, taps, gain, counts) in enumerate([(1368, 31, 128, 1), (1368, 31, 128, 0), (684, 15, 1, 0), (1001, 45, 64, 1)]):
    c = 11. * rng.normal(0, 1., n) + 90. * np.exp(-0.5 * ((np.arange(n) - 400) / 9.) ** 2) * np.cos(0.5 * np.arange(n))
    c = np.round(c) if counts else 1.3e-5 * c
    env = pa.hilbert_envelope(c, adc_output='counts' if counts else 'voltage', ideal_transformer=False, hilbert_n_taps=taps,
                              hilbert_coeff_gain=gain)
    hil_cases.append([n, taps, gain, counts])
    out['hil_in_%d' % k] = c
    out['hil_out_%d' % k] = np.asarray(env, float)
# the ideal transformer (ideal_transformer=True :339-345), appended after the draws above so that their vectors stay what they were
ideal_cases = []
for k, (n, counts) in enumerate([(1368, 1), (1368, 0), (1000, 0), (686, 1)]):
    c = 11. * rng.normal(0, 1., n) + 90. * np.exp(-0.5 * ((np.arange(n) - 400) / 9.) ** 2) * np.cos(0.5 * np.arange(n))
    c = np.round(c) if counts else 1.3e-5 * c
    env = pa.hilbert_envelope(c, adc_output='counts' if counts else 'voltage', ideal_transformer=True)
    ideal_cases.append([n, counts])
    out['ideal_in_%d' % k] = c
    out['ideal_out_%d' % k] = np.asarray(env, float)
out['ideal_cases'] = np.array(ideal_cases, float)
out['up_cases'] = np.array(up_cases, float)
out['hil_cases'] = np.array(hil_cases, float)
np.savez_compressed(os.path.join(OUT, 'ref_pa_modes.npz'), **out)
print('wrote ref_pa_modes.npz', len(up_cases), len(hil_cases), [len(out['up_out_%d' % k]) for k in range(len(up_cases))])
