"""Golden vectors for the filter responses, produced by the reference's signal_processing.get_filter_response
(NuRadioReco/utilities/signal_processing.py:237-333): butter, butterabs, cheby1, rectangular, gaussian_tapered on L-grids, and
by NuRadioReco/detector/RNO_G/analog_components.load_amp_response (measured amplifier chains 'iglu', 'rno_surface' at two
temperatures: what RNO_G/hardwareResponseIncorporator.get_filter(..., sim_to_data=True) multiplies with).  The
measured tables themselves (data files of the reference, NuRadioReco/detector/RNO_G/HardwareResponses/*.csv, *.s2p) travel
inside the fixture: frequency [GHz], linear gain, phase [rad] as stored (not unwrapped).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_filters.py
"""
import os
import numpy as np
from NuRadioReco.utilities import signal_processing

HERE = os.path.dirname(os.path.abspath(__file__))
ff = np.fft.rfftfreq(5296, 0.5)
specs = [dict(type='butter', passband=(0.08, 1000.), order=2), dict(type='butter', passband=(0., 0.5), order=10),
         dict(type='butterabs', passband=(0.1, 0.7), order=5), dict(type='cheby1', passband=(0.13, 0.3), order=4, rp=0.5),
         dict(type='cheby1', passband=(0., 0.6), order=7, rp=0.1), dict(type='rectangular', passband=(0.096, 0.22), order=0),
         dict(type='rectangular', passband=(0., 0.4), order=0)]
out = dict(ff=ff, specs=np.array([repr(s) for s in specs]))
for i, s in enumerate(specs):
    out['H_%d' % i] = np.asarray(signal_processing.get_filter_response(ff, list(s['passband']), s['type'], s['order'],
                                                                       rp=s.get('rp')), complex)
# gaussian_tapered depends on the grid: three trace lengths, two roll widths
gt = [(5296, (0.08, 0.5), 0.02), (2134, (0.1, 0.6), 0.0025), (256, (0.15, 0.45), 0.03), (8192, (0.0, 0.3), 0.01)]
out['gt_cases'] = np.array([(L, pb[0], pb[1], rw) for L, pb, rw in gt])
for i, (L, pb, rw) in enumerate(gt):
    f_ = np.fft.rfftfreq(L, 0.5)
    out['gt_%d' % i] = np.asarray(signal_processing.get_filter_response(f_, list(pb), 'gaussian_tapered', None, roll_width=rw), complex)
# measured amplifier responses
from NuRadioReco.detector.RNO_G import analog_components
from NuRadioReco.utilities import units
hw_dir = os.path.join(os.path.dirname(analog_components.__file__), 'HardwareResponses')
for name, fn in (('iglu', 'iglu_drab_placeholder.csv'), ('rno_surface', 'surface_placeholder.csv')):
    t = np.loadtxt(os.path.join(hw_dir, fn), delimiter=',', skiprows=1)
    out['hw_table_' + name] = np.stack([t[:, 0] * units.Hz, t[:, 1], t[:, 2]], axis=1)
f_hw = np.fft.rfftfreq(2134, 0.5)
out['hw_ff'] = f_hw
for name in ('iglu', 'rno_surface'):   # 'ULP_216' goes through radiotools.helper.dB_to_linear (un-vendored): left out
    for temp in (293.15, 253.15):
        r = analog_components.load_amp_response(name)
        out['hw_%s_%d' % (name, int(temp))] = r['gain'](f_hw, temp) * r['phase'](f_hw)
np.savez_compressed(os.path.join(HERE, '..', 'ref_filters.npz'), **out)
print('wrote', len(specs), 'filter responses on', len(ff), 'bins')
