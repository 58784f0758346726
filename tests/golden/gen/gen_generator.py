"""Golden event lists from the reference's generator: NuRadioMC/EvtGen/generator.generate_eventlist_cylinder (:1023-1414) with
write_events=False, cross_sections_model='ctw' (the tabulated default is a download), three set-ups.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/shims_noh5:/tmp/refcopy /opt/conda/bin/python3.9 tests/golden/gen/gen_generator.py
(needs a real h5py: the function builds h5py string dtypes for its return value)
"""
import os
import numpy as np
from NuRadioMC.EvtGen import generator

HERE = os.path.dirname(os.path.abspath(__file__))
cases = [dict(n_events=400, Emin=1e17, Emax=1e19, volume=dict(fiducial_rmin=0., fiducial_rmax=3000., fiducial_zmin=-2700., fiducial_zmax=0.),
              seed=11),
         dict(n_events=300, Emin=1e16, Emax=1e20, volume=dict(fiducial_rmin=200., fiducial_rmax=2000., fiducial_zmin=-2000., fiducial_zmax=-100.,
                                                              full_rmax=2500., full_zmin=-2500., x0=100., y0=-50.),
              thetamin=0.3, thetamax=2.5, phimin=0.5, phimax=4., start_event_id=17, flavor=[12, -12, 16], spectrum='E-2.2', seed=12),
         dict(n_events=250, Emin=1e18, Emax=1e18 * 1.0001, volume=dict(fiducial_xmin=-1000., fiducial_xmax=2000., fiducial_ymin=-500., fiducial_ymax=500.,
                                                                         fiducial_zmin=-1500., fiducial_zmax=0.),
              interaction_type='cc', flavor=[12, 14], seed=13)]
out = {}
for i, kw in enumerate(cases):
    data, attrs = generator.generate_eventlist_cylinder('unused', write_events=False, cross_sections_model='ctw', **kw)
    for k, v in data.items():
        v = np.asarray(v)
        out['c%d/%s' % (i, k)] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v]).astype('S') if v.dtype.kind in 'OUS' else v
    for k, v in attrs.items():
        if not isinstance(v, str):
            out['c%d_attr/%s' % (i, k)] = np.asarray(v)
    out['c%d_kwargs' % i] = np.array(repr(kw))
    print(i, len(data['xx']), 'showers', len(np.unique(data['event_group_ids'])), 'groups')
np.savez_compressed(os.path.join(HERE, '..', 'ref_generator.npz'), **out)
