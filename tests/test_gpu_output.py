"""Rows f2 / f3 of SURVEY.md section 8: the event-list input format and the HDF5 output layout.  The fixture
tests/golden/ref_hdf5_output.npz holds an input event list in the reference's format and every dataset / attribute of the file
the reference's own outputWriterHDF5 wrote for it (generator: tests/golden/gen/gen_hdf5.py, run with a real h5py);
nuradiomc_amd.output.simulate_to_output has to produce the same tables from the same list."""
import numpy as np
import pytest

import nuradiomc_amd
from nuradiomc_amd import output
from conftest import golden

pytestmark = pytest.mark.gpu


def test_output_tables_like_the_reference_writer(gpu_ctx_factory):
    g = golden('ref_hdf5_output.npz')
    ev = output.EventList({k[3:]: g[k] for k in g.files if k.startswith('in/')},
                          {k[8:]: g[k][()] for k in g.files if k.startswith('in_attr/')})
    ctx = gpu_ctx_factory(g['ice'], 'SP1')
    st = nuradiomc_amd.Station(ctx, g['det_pos'], n_samples=int(g['N']), sampling_rate=float(g['fs']))
    out = output.simulate_to_output(st, ev, station_ids=[int(g['station_id'])], seed=int(g['seed']))
    ref = {k[4:]: g[k] for k in g.files if k.startswith('out/')}
    assert set(out.datasets) == set(ref), (sorted(set(ref) - set(out.datasets)), sorted(set(out.datasets) - set(ref)))
    # the same showers and events, in the same order
    for k in ('shower_ids', 'event_group_ids', 'station_101/shower_id', 'station_101/event_group_ids', 'station_101/event_ids',
              'station_101/event_group_id_per_shower', 'station_101/event_id_per_shower', 'flavors', 'n_interaction'):
        assert np.array_equal(out.datasets[k], ref[k]), k
        assert out.datasets[k].dtype.kind == ref[k].dtype.kind, k
    for k in ('shower_type', 'interaction_type'):
        assert [str(x) for x in np.asarray(out.datasets[k]).astype(str)] == [x.decode() for x in ref[k]], k
    for k in ('triggered', 'multiple_triggers', 'station_101/triggered', 'station_101/multiple_triggers',
              'station_101/triggered_per_event', 'station_101/multiple_triggers_per_event'):
        assert out.datasets[k].dtype == bool and np.array_equal(out.datasets[k], ref[k]), k
    tol = {'station_101/maximum_amplitudes': 5e-3, 'station_101/maximum_amplitudes_envelope': 5e-3,
           'station_101/max_amp_shower_and_ray': 5e-3, 'weights': 1e-6}
    for k, r in ref.items():
        o = np.asarray(out.datasets[k])
        if r.dtype.kind != 'f':
            continue
        assert o.shape == r.shape, (k, o.shape, r.shape)
        assert np.array_equal(np.isnan(o), np.isnan(r)), k      # the same NaN padding
        m = ~np.isnan(r)
        if 'time' in k:       # ~1e4 ns: the reference's first-root noise moves arrival times by ~1e-3 ns
            assert np.max(np.abs(o[m] - r[m])) < 5e-3 if 'trigger' not in k and 'shower_and_ray' not in k else \
                np.max(np.abs(o[m] - r[m])) <= 0.5 + 1e-9, k     # trigger / envelope-maximum times: at most one sample
        elif k == 'weights':   # exp(-column density / interaction length): compare the exponents (weights down to 1e-300)
            big = r[m] > 1e-290
            lo, lr = np.log(np.maximum(o[m][big], 1e-320)), np.log(r[m][big])
            # 1e-6 for the weights the reference simulates at all (>= speedup.minimum_weight_cut = 1e-5); below, the chord's last
            # 500 m sample sits on the surface and its density class hangs on the last bit of np.dot -- a BLAS property (this
            # fixture was written under another numpy build than the golden weights of test_gpu_earth.py)
            bad = np.abs(lo - lr) > np.where(r[m][big] >= 1e-5, 1e-6, 2e-5 * np.abs(lr))
            assert not bad.any(), (k, o[m][big][bad], r[m][big][bad])
            assert np.all(o[m][~big] <= 1e-280)
        else:
            rt = tol.get(k, 1e-6)
            assert np.all(np.abs(o[m] - r[m]) <= rt * np.maximum(np.abs(r[m]), 1e-300) + (1e-7 if 'vector' in k or 'polar' in k else 0)), k   # unit vectors: absolute
    # the random shower parameters the reference drew from its seed
    # (1e-13: the fixture was written under another numpy build, whose pow may differ in the last bit)
    assert np.allclose(out.datasets['shower_realization_Alvarez2009'], ref['shower_realization_Alvarez2009'], rtol=1e-13, atol=0,
                       equal_nan=True)
    # attributes
    assert [x.decode() for x in g['attr/@trigger_names']] == list(out.attrs[('', 'trigger_names')])
    for name in ('Vrms', 'dt', 'Tnoise', 'bandwidth'):
        assert abs(float(out.attrs[('', name)]) - float(g['attr/@' + name])) <= 1e-12 * abs(float(g['attr/@' + name])), name
    for name in ('Vrms', 'bandwidth', 'antenna_positions'):
        assert np.allclose(out.attrs[('station_101', name)], g['attr/station_101@' + name], rtol=1e-12), name
    for name in ('n_events', 'fiducial_rmax', 'Emin', 'volume'):
        assert out.attrs[('', name)] == g['attr/@' + name][()]
    assert out.datasets['triggered'].sum() >= 15


def test_device_readout_windows_equal_the_host_path(gpu_ctx_factory, monkeypatch):
    """nrhip_readout_windows (trigger bin, read-out window, maximum and Hilbert-envelope maximum on the device) against the host path
    the writer used until round 4 (the traces fetched, numpy / scipy.fft) -- still taken for windows that are no power of two:
    every dataset of the output equal, the maxima to 1e-12."""
    g = golden('ref_hdf5_output.npz')
    ev = output.EventList({k[3:]: g[k] for k in g.files if k.startswith('in/')},
                          {k[8:]: g[k][()] for k in g.files if k.startswith('in_attr/')})
    ctx = gpu_ctx_factory(g['ice'], 'SP1')
    st = nuradiomc_amd.Station(ctx, g['det_pos'], n_samples=int(g['N']), sampling_rate=float(g['fs']))
    a = output.simulate_to_output(st, ev, station_ids=[int(g['station_id'])], seed=int(g['seed']))
    monkeypatch.setenv('NRHIP_OUTPUT_HOST_WINDOWS', '1')
    b = output.simulate_to_output(st, ev, station_ids=[int(g['station_id'])], seed=int(g['seed']))
    assert set(a.datasets) == set(b.datasets)
    for k in a.datasets:
        x, y = np.asarray(a.datasets[k]), np.asarray(b.datasets[k])
        if x.dtype.kind == 'f':
            assert np.array_equal(np.isnan(x), np.isnan(y)), k
            m = ~np.isnan(x)
            assert np.all(np.abs(x[m] - y[m]) <= 1e-12 * np.maximum(np.abs(y[m]), 1e-300)), k
        else:
            assert np.array_equal(x, y), k
    assert a.datasets['station_101/maximum_amplitudes_envelope'].shape[0] >= 15
    # a read-out window that is no power of two takes the host path by itself
    c = output.simulate_to_output(st, ev, station_ids=[int(g['station_id'])], seed=int(g['seed']), detector_n_samples=int(g['N']) - 2)
    assert c.datasets['station_101/maximum_amplitudes'].shape == a.datasets['station_101/maximum_amplitudes'].shape


def test_output_tables_of_an_array(gpu_ctx_factory):
    """simulate_to_output on a two-station array (the second station 1.5 km away): the tables of station 101 are those of the
    single-station run, station 102 has its own, a shower is stored once at the top level with its earliest trigger time."""
    g = golden('ref_hdf5_output.npz')
    ev = output.EventList({k[3:]: g[k] for k in g.files if k.startswith('in/')},
                          {k[8:]: g[k][()] for k in g.files if k.startswith('in_attr/')})
    ctx = gpu_ctx_factory(g['ice'], 'SP1')
    st = nuradiomc_amd.Station(ctx, g['det_pos'], n_samples=int(g['N']), sampling_rate=float(g['fs']))
    kw = dict(seed=int(g['seed']))
    single = output.simulate_to_output(st, ev, station_ids=[101], **kw)
    # (the random k_L of the electromagnetic showers are drawn in the reference's order, which walks the stations: a second station
    # changes the draws -- so the array run is given the k_L the single-station run drew, and draws only what that run never needed)
    data = dict(ev.data)
    data['shower_realization_Alvarez2009'] = single.stats['k_L']
    ev2 = output.EventList(data, ev.attrs)
    centres = np.array([[0., 0., 0.], [1500., 300., 0.]])
    arr = nuradiomc_amd.StationArray(st, centres, relative_position=g['det_pos'], station_ids=[101, 102], cull=False)
    both = output.simulate_to_output(arr, ev2, **kw)
    for k, v in single.datasets.items():
        if k.startswith('station_101/'):
            a, b = np.asarray(v), np.asarray(both.datasets[k])
            assert a.shape == b.shape, k
            if a.dtype.kind == 'f':
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-12, atol=0), k
            else:
                assert np.array_equal(a, b), k
    assert 'station_102/maximum_amplitudes' in both.datasets and len(both.datasets['station_102/event_group_ids']) >= 3
    # top level: every shower of the single-station file is there, once; triggered showers of station 102 join
    ids1, ids2 = single.datasets['shower_ids'], both.datasets['shower_ids']
    assert len(np.unique(ids2)) == len(ids2) and np.all(np.isin(ids1, ids2)) and len(ids2) > len(ids1)
    t1 = dict(zip(ids1.tolist(), single.datasets['trigger_times'][:, 0].tolist()))
    t2 = dict(zip(ids2.tolist(), both.datasets['trigger_times'][:, 0].tolist()))
    for k_, v_ in t1.items():
        if not np.isnan(v_):
            assert t2[k_] <= v_ + 1e-9
