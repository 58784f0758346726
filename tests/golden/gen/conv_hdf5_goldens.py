"""Convert the reference's own SingleEvents golden HDF5 into a small .npz fixture.

Run with the conda interpreter (the only one with h5py in the build container):

    /opt/conda/bin/python3.9 tests/golden/gen/conv_hdf5_goldens.py

Source: NuRadioMC/test/SingleEvents/1e18_output_reference.hdf5 (compared by the reference in
NuRadioMC/test/SingleEvents/T04validate_allmost_equal.py:143-207).  Pure data conversion.
"""
import os
import h5py
import numpy as np

REF = os.environ.get('NRMC_REFCOPY', '/tmp/refcopy')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')

f = h5py.File(os.path.join(REF, 'NuRadioMC/test/SingleEvents/1e18_output_reference.hdf5'), 'r')
g = f['station_101']
out = {}
for k in ['xx', 'yy', 'zz', 'zeniths', 'azimuths', 'shower_energies', 'shower_ids', 'vertex_times']:
    out[k] = f[k][:]
out['shower_type'] = np.array([s.decode() if isinstance(s, bytes) else str(s) for s in f['shower_type'][:]])
for k in ['ray_tracing_C0', 'ray_tracing_C1', 'ray_tracing_solution_type', 'ray_tracing_reflection',
          'ray_tracing_reflection_case', 'launch_vectors', 'receive_vectors', 'travel_times', 'travel_distances',
          'polarization', 'focusing_factor', 'shower_id']:
    out['st_' + k] = g[k][:]
out['antenna_positions'] = g.attrs['antenna_positions']
out['ice_model'] = 'ARAsim_southpole'
np.savez_compressed(os.path.join(OUT, 'ref_single_events_1e18.npz'), **out)
print('wrote ref_single_events_1e18.npz', {k: np.shape(v) for k, v in out.items()})
