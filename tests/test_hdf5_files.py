"""The HDF5 files themselves: tools/npz_to_hdf5.py writes the reference's output layout,
nuradiomc_amd.output.EventList.from_hdf5 reads the reference's input layout.  Needs an interpreter with h5py -- this one or the
build container's /opt/conda/bin/python3.9 (run as a subprocess); skipped where neither has it."""
import os
import subprocess
import sys
import numpy as np
import pytest

from conftest import golden, ROOT


def _h5py_python():
    for exe in (sys.executable, '/opt/conda/bin/python3.9'):
        if os.path.exists(exe) and subprocess.run([exe, '-c', 'import h5py, numpy'], capture_output=True).returncode == 0:
            return exe
    return None


PY = _h5py_python()
pytestmark = pytest.mark.skipif(PY is None, reason='no interpreter with h5py')

_DUMP = r'''
import sys, numpy as np, h5py
sys.path.insert(0, sys.argv[3])
out = {}
with h5py.File(sys.argv[1], 'r') as f:
    def walk(name, obj):
        if isinstance(obj, h5py.Dataset):
            v = obj[()]
            if v.dtype.kind == 'O':
                v = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in v]).astype('S')
            out['out/' + name] = v
        for a, val in obj.attrs.items():
            out['attr/%s@%s' % (name, a)] = np.asarray(val if not isinstance(val, str) else val.encode())
    for a, val in f.attrs.items():
        out['attr/@' + a] = np.asarray(val if not isinstance(val, str) else val.encode())
    f.visititems(walk)
out = {k: (np.array([str(x) for x in v.ravel()]).astype('S').reshape(v.shape) if v.dtype.kind == 'O' else v) for k, v in out.items()}
np.savez(sys.argv[2], **out)
'''

_INPUT = r'''
import sys, numpy as np, h5py
sys.path.insert(0, sys.argv[3])
g = np.load(sys.argv[1])
with h5py.File(sys.argv[2], 'w') as f:          # an input file as NuRadioMC/EvtGen/generator.py writes it
    for k in g.files:
        if k.startswith('in/'):
            v = g[k]
            f[k[3:]] = np.array([x.decode() for x in v], dtype=h5py.string_dtype(encoding='utf-8')) if v.dtype.kind == 'S' else v
        elif k.startswith('in_attr/'):
            f.attrs[k[8:]] = g[k][()]
from nuradiomc_amd.output import EventList
ev = EventList.from_hdf5(sys.argv[2])
np.savez(sys.argv[2] + '.npz', n=len(ev), vertex=ev.vertex, types=ev.shower_type_codes(), gid=ev.data['event_group_ids'],
         energies=ev.data['shower_energies'], n_events=ev.attrs['n_events'], vt=ev.data['vertex_times'])
'''


def test_output_file_roundtrip(tmp_path):
    g = golden('ref_hdf5_output.npz')
    src, dst, back = str(tmp_path / 'o.npz'), str(tmp_path / 'o.hdf5'), str(tmp_path / 'b.npz')
    np.savez(src, **{k: g[k] for k in g.files if k.startswith(('out/', 'attr/'))})
    r = subprocess.run([PY, os.path.join(ROOT, 'tools', 'npz_to_hdf5.py'), src, dst, '--prefix', 'out/'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([PY, '-c', _DUMP, dst, back, ROOT], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    b = np.load(back)
    want = [k for k in g.files if k.startswith(('out/', 'attr/'))]
    assert sorted(b.files) == sorted(want)
    for k in want:
        assert b[k].dtype.kind == g[k].dtype.kind and b[k].shape == g[k].shape, k
        assert np.array_equal(b[k], g[k], equal_nan=b[k].dtype.kind == 'f'), k


def test_event_list_from_hdf5(tmp_path):
    g = golden('ref_hdf5_output.npz')
    path = str(tmp_path / 'in.hdf5')
    r = subprocess.run([PY, '-c', _INPUT, os.path.join(ROOT, 'tests', 'golden', 'ref_hdf5_output.npz'), path, ROOT],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    b = np.load(path + '.npz')
    assert b['n'] == len(g['in/xx']) and b['n_events'] == g['in_attr/n_events']
    assert np.array_equal(b['vertex'], np.stack([g['in/xx'], g['in/yy'], g['in/zz']], axis=1))
    assert np.array_equal(b['types'], np.array([0 if t == b'had' else 1 for t in g['in/shower_type']]))
    assert np.array_equal(b['gid'], g['in/event_group_ids']) and np.array_equal(b['energies'], g['in/shower_energies'])
    assert np.array_equal(b['vt'], g['in/vertex_times'])


def test_output_window_helpers():
    """Host-side pieces of simulate_to_output: the read-out window (channelReadoutWindowCutter: np.roll, then the first samples) without the
    rolled copy, and the maxima / Hilbert envelopes of all windows on a pool of threads (real transforms) against scipy.signal.hilbert."""
    from scipy.signal import hilbert
    from nuradiomc_amd import output as o
    rng = np.random.default_rng(5)
    V = rng.normal(size=(5, 100))
    for tb, nw, pre in ((3, 40, 10), (90, 64, 5), (50, 120, 7), (0, 100, 0), (99, 2, 110)):
        assert np.array_equal(o._readout_window(V, tb, nw, pre), np.roll(V, -(tb - pre), axis=-1)[..., :nw]), (tb, nw, pre)
    for n in (4096, 1000, 777, 6):
        x = rng.normal(size=(3, 5, n))
        assert np.max(np.abs(o._hilbert_envelope(x) - np.abs(hilbert(x, axis=-1)))) < 1e-13
    W = [rng.normal(size=(5, 512)) for _ in range(150)]
    amp, env = o._window_maxima(W, step=16)
    assert np.array_equal(amp, np.abs(np.array(W)).max(axis=-1))
    assert np.max(np.abs(env - np.abs(hilbert(np.array(W), axis=-1)).max(axis=-1))) < 1e-13
