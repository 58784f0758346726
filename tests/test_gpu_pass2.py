"""Pass 2 on the device (Station.triggered_pass_dev): the triggered groups of a resident list are gathered in HBM and run again
with every channel trace kept -- same traces, trigger bins and selection as the host-side pass 2 of nuradiomc_amd.output
(re-upload of the triggered showers, dump_traces)."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('flavour', ['had', 'mixed'])
def test_device_pass2_equals_host_pass2(gpu_ctx_factory, flavour):
    n = 40000
    wl = bench.make_workload(2, n, 10, flavour)
    ctx = gpu_ctx_factory(wl['ice'], wl['att_model'])
    st = bench.build_array(ctx, wl)
    d = bench.upload_events(ctx, wl)
    try:
        s1 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], n_groups=d['n_groups'], d_group_begin=d['gb'])
        mask = np.zeros(d['n_groups'], np.uint8)
        ctx.to_host(mask, d['trig'])
        assert mask.sum() == s1['n_triggered'] > 100
        s2, d_keep, nk = st.triggered_pass_dev(d['n'], *d['in'], d['trig'], n_groups=d['n_groups'], d_group_begin=d['gb'])
        assert nk == mask.sum() and s2['n_events'] == nk and s2['n_triggered'] == nk   # every selected group triggers again
        keep = np.zeros(nk, np.int32)
        ctx.to_host(keep, d_keep)
        assert np.array_equal(keep, np.flatnonzero(mask))
        A = {k: st.fetch(k).copy() for k in ('item_event', 'trace', 'trace_offset', 'ev_trigger_bin', 'ev_L', 'ev_t_min')}
    finally:
        bench.free_events(ctx, d)
    # host-side pass 2: the triggered showers uploaded again
    ev = wl['events']
    rows = np.flatnonzero(np.isin(ev['group'], keep))
    t2, _ = st.simulate_events(ev['vertex'][rows], ev['zenith'][rows], ev['azimuth'][rows], ev['energy'][rows], ev['shower_type'][rows],
                               ev['k_L'][rows], group_id=ev['group'][rows], dump_traces=True)
    assert t2.all()
    for k, v in A.items():
        assert np.array_equal(v, st.fetch(k)), k
    assert len(A['trace']) > 5 * 4096 * nk * 0.5


@pytest.mark.parametrize('flavour', ['had', 'mixed'])
def test_traces_emitted_when_an_event_triggers(gpu_ctx_factory, flavour):
    """emit_traces: the convolution kernel writes all channel traces of an event the moment it triggers.  Same mask as without;
    the blocks equal the traces of the dump_traces pass over the same events bit for bit; a buffer that is too small is reported
    (n_emit_overflow) and the events that did fit are still right."""
    n = 40000
    wl = bench.make_workload(2, n, 10, flavour)
    ctx = gpu_ctx_factory(wl['ice'], wl['att_model'])
    st = bench.build_array(ctx, wl)
    d = bench.upload_events(ctx, wl)
    try:
        kw = dict(n_groups=d['n_groups'], d_group_begin=d['gb'])
        s0 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], **kw)
        m0 = np.zeros(d['n_groups'], np.uint8)
        ctx.to_host(m0, d['trig'])
        s1 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], emit_traces=True, **kw)
        m1 = np.zeros(d['n_groups'], np.uint8)
        ctx.to_host(m1, d['trig'])
        assert np.array_equal(m0, m1) and s1['n_triggered'] == s0['n_triggered'] > 100
        assert s1['n_emitted_events'] == s1['n_triggered'] and s1['n_emit_overflow'] == 0
        got = st.triggered_traces()
        assert sorted(got) == list(np.flatnonzero(m1))
        L1 = st.fetch('ev_L').copy()
        # the reference traces: dump_traces pass over the triggered groups
        s2, d_keep, nk = st.triggered_pass_dev(d['n'], *d['in'], d['trig'], **kw)
        keep = np.zeros(nk, np.int32)
        ctx.to_host(keep, d_keep)
        item_event, tr, off = st.fetch('item_event'), st.fetch('trace'), st.fetch('trace_offset')
        n_ch = len(wl['rel_pos'])
        assert np.array_equal(item_event, np.arange(nk))
        for i, g in enumerate(keep):
            ref = np.array([tr[off[i * n_ch + c]:off[i * n_ch + c + 1]] for c in range(n_ch)])
            assert ref.shape == got[int(g)].shape == (n_ch, L1[g]) and np.array_equal(ref, got[int(g)]), g
        # a buffer for about a third of the triggered events
        cap = int(s1['n_emitted_samples'] // 3)
        s3 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], emit_traces=True, emit_capacity_samples=cap, **kw)
        assert s3['n_triggered'] == s0['n_triggered'] and s3['n_emit_overflow'] > 0
        assert s3['n_emitted_events'] + s3['n_emit_overflow'] == s3['n_triggered']
        part = st.triggered_traces()
        assert len(part) == s3['n_emitted_events'] and all(np.array_equal(v, got[k]) for k, v in part.items())
    finally:
        bench.free_events(ctx, d)


def test_device_pass2_carries_the_noise_of_pass1(gpu_ctx_factory):
    """With thermal noise the traces pass 2 stores must be the ones pass 1 decided on: the noise of an event group is keyed by its
    id in the ORIGINAL list, not by its place in the compact list of the triggered groups (nrhip_index_to_i64 /
    nrhip_gather_i64).  Pass-2 traces == the traces of a dump_traces run over the whole list, with running ids and with ids given
    by the caller."""
    n = 12000
    wl = bench.make_workload(2, n, 10, 'had')
    ctx = gpu_ctx_factory(wl['ice'], wl['att_model'])
    st = bench.build_array(ctx, wl)
    st.set_noise(300.)
    d = bench.upload_events(ctx, wl)
    n_ch = len(wl['rel_pos'])
    ids = (np.arange(n, dtype=np.int64) * 7 + 1000003)
    d_ids = ctx.to_device(ids)
    try:
        for idkw in (dict(noise_group_offset=5000), dict(d_noise_group_id=d_ids)):
            kw = dict(n_groups=d['n_groups'], d_group_begin=d['gb'], noise=True, noise_seed=77, **idkw)
            st.simulate_events_dev(d['n'], *d['in'], d['trig'], dump_traces=True, **kw)
            m_full = np.zeros(n, np.uint8)
            ctx.to_host(m_full, d['trig'])
            ie, tr, off, tb = (st.fetch(k).copy() for k in ('item_event', 'trace', 'trace_offset', 'ev_trigger_bin'))
            s1 = st.simulate_events_dev(d['n'], *d['in'], d['trig'], **kw)
            mask = np.zeros(n, np.uint8)
            ctx.to_host(mask, d['trig'])
            assert np.array_equal(mask, m_full) and s1['n_triggered'] > 20
            s2, d_keep, nk = st.triggered_pass_dev(d['n'], *d['in'], d['trig'], **kw)
            assert nk == mask.sum() and s2['n_triggered'] == nk
            keep = np.zeros(nk, np.int32)
            ctx.to_host(keep, d_keep)
            ie2, tr2, off2, tb2 = (st.fetch(k) for k in ('item_event', 'trace', 'trace_offset', 'ev_trigger_bin'))
            assert np.array_equal(ie2, np.arange(nk))
            where = {int(g): i for i, g in enumerate(ie)}
            for i, g in enumerate(keep):
                j = where[int(g)]
                a = tr[off[j * n_ch]:off[(j + 1) * n_ch]]
                b = tr2[off2[i * n_ch]:off2[(i + 1) * n_ch]]
                assert len(a) == len(b) and np.array_equal(a, b), (g, idkw)
            assert np.array_equal(tb2[:nk], tb[keep])
    finally:
        ctx.free(d_ids)
        bench.free_events(ctx, d)
