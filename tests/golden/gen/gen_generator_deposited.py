"""Golden event list of the reference's generator with deposited=True (Emin .. Emax are deposited energies; the neutrino energy is
E / y except for nu_e CC, EvtGen/generator.py:199-224, :1247-1252), cross_sections_model='ctw'.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/shims_noh5:/tmp/refcopy /opt/conda/bin/python3.9 tests/golden/gen/gen_generator_deposited.py
(as gen_generator.py)  -> tests/golden/ref_generator_deposited.npz
"""
import os
import numpy as np
from NuRadioMC.EvtGen import generator

HERE = os.path.dirname(os.path.abspath(__file__))
kw = dict(n_events=400, Emin=1e17, Emax=1e19, volume=dict(fiducial_rmin=0., fiducial_rmax=3000., fiducial_zmin=-2700., fiducial_zmax=0.),
          seed=31, deposited=True)
data, attrs = generator.generate_eventlist_cylinder('unused', write_events=False, cross_sections_model='ctw', **kw)
out = {}
for k, v in data.items():
    v = np.asarray(v)
    out['c0/%s' % k] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v]).astype('S') if v.dtype.kind in 'OUS' else v
out['c0_kwargs'] = np.array(repr(kw))
out['c0_attr_deposited'] = np.asarray(attrs['deposited'])
print(len(data['xx']), 'showers', np.asarray(data['energies']).min(), np.asarray(data['energies']).max())
np.savez_compressed(os.path.join(HERE, '..', 'ref_generator_deposited.npz'), **out)
