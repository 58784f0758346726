"""Golden event list of the reference's generator with cross_sections_model='hedis_bgr18' (charged / neutral current from the
integrated table, inelasticity from its cumulative distribution; NuRadioMC/utilities/inelasticities.py:54-157) on the SYNTHETIC
table tests/golden/bgr18_synthetic.npz (gen_hedis.py; the real file is a download): the reference's one np.load / os.path.exists
of the data file's name is redirected to it.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/shims_noh5:/tmp/refcopy /opt/conda/bin/python3.9 tests/golden/gen/gen_generator_hedis.py
(/tmp/refcopy: an unmodified copy of /root/reference; a real h5py is needed, as for gen_generator.py)  -> tests/golden/ref_generator_hedis.npz
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
table = os.path.join(HERE, '..', 'bgr18_synthetic.npz')
real_load, real_exists = np.load, os.path.exists
is_table = lambda p: str(p).endswith('BGR18_dsigma_dy_H2O.npz')
np.load = lambda p, *a, **k: real_load(table if is_table(p) else p, *a, **k)
os.path.exists = lambda p: True if is_table(p) else real_exists(p)
from NuRadioMC.EvtGen import generator

cases = [dict(n_events=500, Emin=1e15, Emax=1e20, volume=dict(fiducial_rmin=0., fiducial_rmax=3000., fiducial_zmin=-2700., fiducial_zmax=0.),
              seed=21),
         dict(n_events=300, Emin=1e17, Emax=5e21, volume=dict(fiducial_rmin=0., fiducial_rmax=1000., fiducial_zmin=-1000., fiducial_zmax=0.),
              flavor=[12, -16], interaction_type='cc', seed=22)]      # energies above the table's last node: its last row
out = {}
for i, kw in enumerate(cases):
    data, attrs = generator.generate_eventlist_cylinder('unused', write_events=False, cross_sections_model='hedis_bgr18', **kw)
    for k, v in data.items():
        v = np.asarray(v)
        out['c%d/%s' % (i, k)] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v]).astype('S') if v.dtype.kind in 'OUS' else v
    out['c%d_kwargs' % i] = np.array(repr(kw))
    print(i, len(data['xx']), 'showers', np.mean(np.asarray(data['inelasticity'])), np.unique(np.asarray(data['interaction_type']), return_counts=True))
np.savez_compressed(os.path.join(HERE, '..', 'ref_generator_hedis.npz'), **out)
