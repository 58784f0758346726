"""Statistics of the reference's thermal noise (NuRadioReco/modules/channelGenericNoiseAdder.py: bandlimited_noise :66-160, type
'rayleigh', min_freq = 0, max_freq = Nyquist -- the call of simulation.apply_det_response, simulation.py:594-606): 400 traces of
2000 and of 5296 samples at 2 GHz with amplitude 1.  The build's noise uses another random stream (counter based), so what is pinned
is the distribution: per-trace RMS, moments of the samples, moments of the spectral amplitudes, DC / Nyquist conventions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_noise.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
import NuRadioReco.modules.channelGenericNoiseAdder  # noqa: E402
from NuRadioReco.utilities import fft  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
na = NuRadioReco.modules.channelGenericNoiseAdder.channelGenericNoiseAdder()
na.begin(seed=1235)
out = {}
for L in (2000, 5296):
    fs = 2.0
    tr = np.array([na.bandlimited_noise(0., 0.5 * fs, L, fs, 1.0, type='rayleigh') for _ in range(400)])
    spec = np.array([fft.time2freq(t, fs) for t in tr])
    a = np.abs(spec)
    out.update({'L%d_rms' % L: np.sqrt(np.mean(tr ** 2, axis=1)), 'L%d_mean' % L: np.mean(tr, axis=1),
                'L%d_m2' % L: np.mean(tr ** 2), 'L%d_m4' % L: np.mean(tr ** 4),
                'L%d_amp_mean' % L: np.mean(a[:, 1:-1]), 'L%d_amp_m2' % L: np.mean(a[:, 1:-1] ** 2), 'L%d_amp_m4' % L: np.mean(a[:, 1:-1] ** 4),
                'L%d_dc_max' % L: np.max(a[:, 0]), 'L%d_nyq_imag_max' % L: np.max(np.abs(spec[:, -1].imag)),
                'L%d_nyq_m2' % L: np.mean(a[:, -1] ** 2), 'L%d_phase_mean_cos' % L: np.mean(np.cos(np.angle(spec[:, 1:-1]))),
                'L%d_lag1' % L: np.mean(tr[:, 1:] * tr[:, :-1])})
    print(L, 'rms', out['L%d_rms' % L].mean(), 'kurtosis', out['L%d_m4' % L] / out['L%d_m2' % L] ** 2)
np.savez_compressed(os.path.join(OUT, 'ref_noise_stats.npz'), **out)
