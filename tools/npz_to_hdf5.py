#!/usr/bin/env python
"""Turn the .npz twin of an output file (nuradiomc_amd.output.OutputFile.save_npz: datasets 'name' / 'station_<id>/name',
attributes 'attr/<group>@<name>') into the HDF5 file itself -- for machines whose simulation interpreter has no h5py.  Needs only
numpy + h5py and no part of this package:

    python tools/npz_to_hdf5.py output.npz output.hdf5 [--prefix out/]
"""
import sys
import numpy as np
import h5py


def convert(src, dst, prefix=''):
    g = np.load(src)
    with h5py.File(dst, 'w') as f:
        for k in g.files:
            v = g[k]
            if k.startswith('attr/'):
                grp, name = k[5:].split('@', 1)
                obj = f if grp == '' else f.require_group(grp)
                if v.dtype.kind == 'S':
                    v = v.astype(str)
                obj.attrs[name] = (v.tolist() if v.ndim else str(v[()])) if v.dtype.kind == 'U' else v
            elif k.startswith(prefix) and not k.startswith(('in/', 'in_attr/')):
                name = k[len(prefix):]
                if v.dtype.kind in 'SU':   # variable-length UTF-8 strings like the reference's writer
                    f[name] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v], dtype=h5py.string_dtype(encoding='utf-8'))
                elif v.ndim or v.dtype.kind != 'O':
                    f[name] = v


if __name__ == '__main__':
    prefix = sys.argv[sys.argv.index('--prefix') + 1] if '--prefix' in sys.argv else ''
    convert(sys.argv[1], sys.argv[2], prefix)
    print('wrote', sys.argv[2])
