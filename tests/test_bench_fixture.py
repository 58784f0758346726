"""The headline workload (bench.py config 2, 1e6-event list, seed 10) pinned on the REFERENCE: its first 24 000 event groups went
through the reference's own simulation functions (tests/golden/gen/gen_bench.py -> chain_bench_N4096.npz: 81 433 rays, 2014
candidate events, 222 triggers; 17.4 minutes on 8 cores = 23 events/s).  CPU: the oracle against the fixture; GPU (-m gpu): the batched HIP path, in the production
mode bench.py times, against the fixture."""
import numpy as np
import pytest
from conftest import golden

import bench


def _fixture():
    g = golden('chain_bench_N4096.npz')
    K = len(g['zenith'])
    v, z, a = bench.make_events(int(g['n_list']), int(g['seed']))
    # the fixture IS the head of the list bench.py times
    assert np.array_equal(v[:K], g['vertex']) and np.array_equal(z[:K], g['zenith']) and np.array_equal(a[:K], g['azimuth'])
    assert float(g['energy']) == bench.ENERGY and int(g['N']) == bench.N_SAMPLES and float(g['fs']) == bench.FS
    return g, K


def _compare(g, K, n_rays, cand, trig, L, maxV_of, what, two_sided=False):
    """decisions exact wherever the ray counts agree; returns the observed maxima"""
    same = n_rays == g['ev_n_rays'][:K]
    frac_diff = 1. - same.mean()
    # observed: 13 of 24 000 = 5.4e-4, on every one of them the reference is short of a TRUE root (tests/test_true_roots.py): since
    # round 5 oracle and kernels hold the true solution set, so a count can differ in one direction only.
    # two_sided (the reference-procedure finder): the acceptance test of the first root falls either way with the last bits of
    # exp / log, so a count may be one more or one fewer
    assert frac_diff <= 1.2e-3, frac_diff
    if not two_sided:
        assert np.all(n_rays >= g['ev_n_rays'][:K])
    assert np.array_equal(cand[same], g['ev_candidate'][:K][same])
    assert np.array_equal(trig[same], g['ev_triggered'][:K][same])
    both = same & cand
    assert np.array_equal(L[both], g['ev_L'][:K][both])
    worst = 0.
    for ev in np.flatnonzero(both):
        got = maxV_of(ev)
        if got is None:
            continue
        ref = g['ev_maxV'][ev]
        m = np.isfinite(got) & (got >= 0)
        if m.any():
            worst = max(worst, float(np.max(np.abs(got[m] - ref[m])) / np.max(ref)))
    assert worst <= 1.9e-3, worst                  # observed 9.4e-4 GPU on 24 000 events, 3.5e-4 oracle on 1500 (a 1e-7 shift of T is ~3e-3 rad at 500 MHz)
    print('%s: %d events, ray counts differ on %d, %d candidates, %d triggers (reference %d), max |dV| / max|V| = %.2e'
          % (what, K, int((~same).sum()), int(cand.sum()), int(trig.sum()), int(g['ev_triggered'].sum()), worst))
    return frac_diff, worst


def test_oracle_vs_reference_on_the_bench_list():
    from oracle import spectral_oracle as so
    g, K = _fixture()
    K = 1500
    st = so.Station(bench.CHANNELS, n_samples=bench.N_SAMPLES, fs=bench.FS)
    vrms, vrms_e = so.vrms_from_filters(bench.FS)
    assert abs(vrms - float(g['vrms'])) <= 1e-12 * vrms and abs(vrms_e - float(g['vrms_efield'])) <= 1e-12 * vrms_e
    n_rays, cand, trig, L = np.zeros(K, int), np.zeros(K, bool), np.zeros(K, bool), np.zeros(K, int)
    V = {}
    for i in range(K):
        o = so.simulate_event(g['vertex'][i], g['zenith'][i], g['azimuth'][i], bench.ENERGY, 'HAD', None, st, bench.ICE, vrms, vrms_e)
        n_rays[i], cand[i], trig[i], L[i] = len(o['rays']), o['candidate'], o['triggered'], o.get('L', 0)
        if 'V' in o:
            V[i] = np.max(np.abs(o['V']), axis=1)
    _compare(g, K, n_rays, cand, trig, L, lambda ev: V.get(ev), 'oracle vs reference')
    assert trig.sum() >= 10


@pytest.mark.gpu
@pytest.mark.parametrize('production', [True, False])
def test_gpu_vs_reference_on_the_bench_list(gpu_ctx_factory, production):
    g, K = _fixture()
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = bench.build_array(ctx, bench.make_workload(2, 1000, 10))   # the station bench.py times
    assert abs(st.vrms - float(g['vrms'])) <= 1e-12 * st.vrms
    n = K
    trig, stats = st.simulate_events(g['vertex'], g['zenith'], g['azimuth'], np.full(n, bench.ENERGY), np.zeros(n, np.int32),
                                     np.ones(n), dump_traces=not production, no_pruning=not production)
    n_rays = st.fetch('ev_n_rays')[:n]
    cand = st.fetch('ev_candidate')[:n].astype(bool)
    L = st.fetch('ev_L')[:n]
    item_event = st.fetch('item_event')
    maxV = st.fetch('item_maxV').reshape(len(item_event), -1)
    row = {int(ev): i for i, ev in enumerate(item_event)}
    _compare(g, n, n_rays, cand, trig.astype(bool), L, lambda ev: maxV[row[ev]] if ev in row else None,
             'GPU (%s) vs reference' % ('production' if production else 'exhaustive'))
    # ray tables of the events with equal ray counts: types exact, C0 / D / T at 1e-6 (north_star)
    same = n_rays == g['ev_n_rays'][:n]
    ray_ev = st.fetch('ray_event')[:stats['n_rays']]
    keep = same[ray_ev]
    ref_keep = same[g['ray_event']]
    for k in ('C0', 'D'):
        got, ref = st.fetch('ray_' + k)[:stats['n_rays']][keep], g['ray_' + k][ref_keep]
        rel = np.max(np.abs(got - ref) / np.abs(ref))
        print('ray_%s max rel %.2e' % (k, rel))
        assert rel <= 4.4e-7, (k, rel)   # observed on 81 433 rays: 1.5e-7 (C0), 2.2e-7 (D); north_star: 1e-6
    assert np.array_equal(st.fetch('ray_channel')[:stats['n_rays']][keep], g['ray_channel'][ref_keep])
    assert np.array_equal(st.fetch('ray_solution')[:stats['n_rays']][keep], g['ray_iS'][ref_keep])


@pytest.mark.gpu
def test_gpu_reference_finder_on_the_bench_list(gpu_ctx_factory):
    """The 24 000-event fixture with the reference's own finder (NRHIP_FINDER_REFERENCE): ray counts within the two-sided noise of
    the acceptance test, decisions exact wherever they agree, and the number of events on which the two finders differ printed."""
    g, K = _fixture()
    ctx = gpu_ctx_factory(bench.ICE, 'SP1', ray_finder='reference')
    st = bench.build_array(ctx, bench.make_workload(2, 1000, 10))
    args = (g['vertex'], g['zenith'], g['azimuth'], np.full(K, bench.ENERGY), np.zeros(K, np.int32), np.ones(K))
    trig, stats = st.simulate_events(*args)
    n_rays = st.fetch('ev_n_rays')[:K].copy()
    cand = st.fetch('ev_candidate')[:K].astype(bool)
    L = st.fetch('ev_L')[:K]
    item_event = st.fetch('item_event')
    maxV = st.fetch('item_maxV').reshape(len(item_event), -1)
    row = {int(ev): i for i, ev in enumerate(item_event)}
    _compare(g, K, n_rays, cand, trig.astype(bool), L, lambda ev: maxV[row[ev]] if ev in row else None,
             'GPU (reference finder) vs reference', two_sided=True)
    ref_counts = g['ev_n_rays'][:K]
    ctx.set_ray_finder('true_roots')
    trig_t, _ = st.simulate_events(*args)
    n_rays_t = st.fetch('ev_n_rays')[:K]
    print('ray counts differ from the reference on %d events (reference finder: %d more, %d fewer) and on %d (true-root finder, never fewer); '
          'the two finders differ on %d events; triggers %d / %d (reference 222)'
          % ((n_rays != ref_counts).sum(), (n_rays > ref_counts).sum(), (n_rays < ref_counts).sum(), (n_rays_t != ref_counts).sum(),
             (n_rays != n_rays_t).sum(), trig.sum(), trig_t.sum()))
    assert np.all(n_rays_t >= n_rays)
    ctx.set_ray_finder('reference')


@pytest.mark.gpu
@pytest.mark.parametrize('given', ['C0', 'C0+D+T'])
def test_gpu_on_the_reference_rays_of_the_bench_list(gpu_ctx_factory, given):
    """The headline workload on the reference's OWN rays: the 81 433 rays the reference found for the first
    24 000 events of the bench list go into the batched path (nrhip_sim_config.given_C0 [+ given_D / given_T]); decisions exact on
    all 24 000 events; with the launch parameters alone path lengths / travel times to 2.5e-7 and channel maxima to the reference's
    own rounding noise in T (tests/test_gpu_chain.py::test_reference_rays_through_the_batched_path), with its D and T as well the
    channel maxima of all 2014 candidate events at 1e-6 against the reference's numbers."""
    from test_gpu_chain import reference_rays_table, check_against_reference_on_its_rays
    g, K = _fixture()
    ctx = gpu_ctx_factory(bench.ICE, 'SP1')
    st = bench.build_array(ctx, bench.make_workload(2, 1000, 10))
    tab, ref_rays = reference_rays_table(g, K, len(bench.CHANNELS))
    kw = dict(given_C0=tab[0]) if given == 'C0' else dict(given_C0=tab[0], given_D=tab[1], given_T=tab[2])
    trig, stats = st.simulate_events(g['vertex'], g['zenith'], g['azimuth'], np.full(K, bench.ENERGY), np.zeros(K, np.int32),
                                     np.ones(K), dump_traces=True, no_pruning=True, **kw)
    tol, tol_path = (5e-4, 2.5e-7) if given == 'C0' else (1e-6, 1e-9)
    worst, _ = check_against_reference_on_its_rays(g, st, trig, stats, K, ref_rays, 'bench list given ' + given, tol, tol_path)
    assert int(trig.sum()) == 222 and stats['n_candidate_events'] == 2014
