"""Golden vectors for the filter responses, produced by the reference's signal_processing.get_filter_response
(NuRadioReco/utilities/signal_processing.py:237-333): butter, butterabs, cheby1, rectangular on an L-grid.

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
np.savez_compressed(os.path.join(HERE, '..', 'ref_filters.npz'), **out)
print('wrote', len(specs), 'filter responses on', len(ff), 'bins')
