"""Golden vectors for per-channel analog chains: a 'gaussian_tapered' band pass on every channel (channelBandPassFilter,
signal_processing.py:310-321: a different response on every trace length) followed by MEASURED amplifier responses that differ
between channels -- channels 0-2 the deep 'iglu' chain, channels 3-4 the 'rno_surface' chain of
NuRadioReco/detector/RNO_G/analog_components.load_amp_response, applied as RNO_G/hardwareResponseIncorporator.run(...,
sim_to_data=True) applies them (:205-215: spectrum *= gain(f, temp) * phase(f)); the module itself insists on a database
detector object, so its three lines are spelled out here around the reference's own response functions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_hw.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402
import gen_chain  # noqa: E402
from NuRadioReco.detector.RNO_G import analog_components  # noqa: E402
from NuRadioReco.utilities import signal_processing, units  # noqa: E402

PASSBAND, ROLL = [0.1, 0.65], 0.02
AMP = ['iglu', 'iglu', 'iglu', 'rno_surface', 'rno_surface']
TEMP = 293.15
_amps = {k: analog_components.load_amp_response(k) for k in set(AMP)}


def filter_amp_hw(evt, station, det):
    rh._bp.run(evt, station, det, passband=PASSBAND, filter_type='gaussian_tapered', roll_width=ROLL)
    for channel in station.iter_channels():
        a = _amps[AMP[channel.get_id()]]
        f = channel.get_frequencies()
        channel.set_frequency_spectrum(channel.get_frequency_spectrum() * (a['gain'](f, TEMP) * a['phase'](f)),
                                       channel.get_sampling_rate())


def vrms_hw(config, channel=0, noise_temperature=300.):
    """simulation.py:1301-1376 with the chain of one channel on the 10000-point grid"""
    ff = np.linspace(0, 0.5 * config['sampling_rate'], 10000)
    a = _amps[AMP[channel]]
    filt = signal_processing.get_filter_response(ff, PASSBAND, 'gaussian_tapered', None, roll_width=ROLL) * a['gain'](ff, TEMP) * a['phase'](ff)
    bandwidth = np.trapz(np.abs(filt) ** 2, ff)
    vrms = signal_processing.calculate_vrms_from_temperature(noise_temperature, bandwidth=bandwidth)
    return vrms, vrms / np.abs(filt).max() / units.m


if __name__ == '__main__':
    rh.filter_amp = filter_amp_hw
    rh.vrms_from_filters = lambda config, noise_temperature=300.: vrms_hw(config)
    gen_chain.run('N256_hw', n_events=220, seed=26, N=256, full_rays=100, full_events=8, rmax=2500.)
    g = dict(np.load(os.path.join(gen_chain.OUT, 'chain_N256_hw.npz')))
    hw_dir = os.path.join(os.path.dirname(analog_components.__file__), 'HardwareResponses')
    for name, fn in (('iglu', 'iglu_drab_placeholder.csv'), ('rno_surface', 'surface_placeholder.csv')):
        t = np.loadtxt(os.path.join(hw_dir, fn), delimiter=',', skiprows=1)
        g['hw_table_' + name] = np.stack([t[:, 0] * units.Hz, t[:, 1], t[:, 2]], axis=1)
    g.update(hw_passband=np.array(PASSBAND), hw_roll_width=ROLL, hw_amp=np.array(AMP), hw_temperature=TEMP)
    np.savez_compressed(os.path.join(gen_chain.OUT, 'chain_N256_hw.npz'), **g)
