"""HIP ray tracer + attenuation through the C ABI vs the oracle and the committed golden vectors."""
import numpy as np
import pytest
from conftest import golden, max_rel
from oracle import raytrace_oracle as orc
from test_oracle_golden import _subset_ok, compare_with_reference_two_sided

pytestmark = pytest.mark.gpu


def _compare_ray_tables(o, g, count_tol=0.008):
    """against the reference's outputs.  Since round 5 the kernels hold the TRUE solution set (tests/test_true_roots.py): a count
    differs only where the reference lost a root to its `fun < 1e-7` test (observed 0.2 % on fixture A, 0 on B, 0.4 % on C), and
    every solution of the reference is one of the GPU's"""
    bad = o['n_sol'] != g['n_sol']
    assert bad.mean() <= count_tol, "solution-count mismatches: %d of %d" % (bad.sum(), len(bad))
    for i in np.where(bad)[0]:
        assert o['n_sol'][i] > g['n_sol'][i]
        assert _subset_ok(g['C0'][i], o['C0'][i])
    ok = ~bad
    assert np.array_equal(o['type'][ok], g['type'][ok])
    assert max_rel(o['C0'][ok], g['C0'][ok]) < 2e-7   # observed 3.3e-8: the distance of the reference's hybr iterate from the root
    for k in ('D', 'T'):
        rel = np.abs(o[k][ok] - g[k][ok]) / np.abs(g[k][ok])
        rel = rel[np.isfinite(rel)]
        assert rel.max() < 8e-6 and (rel > 1e-6).mean() <= 0.0026, k   # observed 4.1e-6 on ONE ray of fixture C (it turns 1e-7 m from the receiver's depth: against 60-digit arithmetic the reference is 2.4e-6 off, we 1.7e-6 the other way -- tools/true_roots.py), 1.2e-7 elsewhere
    for k in ('launch', 'receive'):
        assert np.array_equal(np.isnan(o[k][ok]), np.isnan(g[k][ok]))
        assert np.nanmax(np.abs(o[k][ok] - g[k][ok])) < 4e-7   # observed 1.8e-7
    assert np.nanmax(np.abs(o['C1'][ok] - g['C1'][ok])) < 7.5e-4   # observed 3.5e-4 m
    assert max_rel(o['refl_angle'][ok], g['refl_angle'][ok]) < 1e-6
    return bad.sum()


def _assert_identical_to_oracle(o, ref):
    """The kernels and the oracle use the same bit-reproducible exp / log and the same operation order, so the
    whole ray table is EQUAL, not close: counts, types and every float64 bit (NaN padding included).  Only the
    reflection angle goes through libm's atan2 on both sides (output only)."""
    assert np.array_equal(o['n_sol'], ref['n_sol'])
    assert np.array_equal(o['type'], ref['type'])
    for k in ('C0', 'C1', 'D', 'T', 'launch', 'receive'):
        assert np.array_equal(o[k], ref[k], equal_nan=True), k
    assert np.array_equal(np.isnan(o['refl_angle']), np.isnan(ref['refl_angle']))
    assert np.nanmax(np.abs(o['refl_angle'] - ref['refl_angle']), initial=0.) < 1e-14


@pytest.mark.parametrize('name', ['A', 'B', 'C'])
def test_find_solutions_vs_reference_fixture(gpu_ctx_factory, name):
    g = golden('raytrace_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    o = ctx.find_solutions_batch(g['x1'], g['x2'])
    _compare_ray_tables(o, g)
    _assert_identical_to_oracle(o, orc.raytrace_batch(g['x1'], g['x2'], g['ice']))


@pytest.mark.parametrize('name', ['A', 'B', 'C'])
def test_reference_finder_vs_reference_fixture(gpu_ctx_factory, name):
    """NRHIP_FINDER_REFERENCE (Context(..., ray_finder='reference'), nrhip_ctx_set_ray_finder): the reference's procedure and
    acceptance test to the letter for every pair (analyticraytracing.py:1476-1547).  Against the reference's own tables at the
    tolerances of rounds 1-4 (counts may differ EITHER way on <= 0.3 % of the pairs: the acceptance test is a coin flip on the last
    bits of exp / log), and equal to the checker in the same mode bit for bit.  Prints the count mismatches of both finders."""
    g = golden('raytrace_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']), ray_finder='reference')
    assert ctx.ray_finder == 'reference'
    o = ctx.find_solutions_batch(g['x1'], g['x2'])
    n_more, n_fewer = compare_with_reference_two_sided(o, g)
    with orc.reference_procedure():
        _assert_identical_to_oracle(o, orc.raytrace_batch(g['x1'], g['x2'], g['ice']))
    # the same context switched to the default finder and back
    ctx.set_ray_finder('true_roots')
    t = ctx.find_solutions_batch(g['x1'], g['x2'])
    _assert_identical_to_oracle(t, orc.raytrace_batch(g['x1'], g['x2'], g['ice']))
    ctx.set_ray_finder('reference')
    assert np.array_equal(ctx.find_solutions_batch(g['x1'], g['x2'])['C0'], o['C0'], equal_nan=True)
    d = t['n_sol'] - g['n_sol']
    print('fixture %s, %d pairs: reference finder %d longer / %d shorter than the reference; true-root finder %d longer / %d shorter'
          % (name, len(d), n_more, n_fewer, (d > 0).sum(), (d < 0).sum()))
    assert np.all(o['n_sol'] <= t['n_sol'])
    with pytest.raises(ValueError):
        ctx.set_ray_finder('newton')


def test_reference_finder_random_geometries_and_reflections(gpu_ctx_factory):
    """The reference finder on 1e5 random pairs (shallow and deep receivers, degenerate pairs) and on the calls with a reflective
    layer: equal to the checker in the same mode bit for bit; its list is a subset of the true-root finder's."""
    ice = (1.78, 0.423, 77.)
    rng = np.random.default_rng(2026)
    n = 100000
    r, ph = np.sqrt(rng.uniform(0, 5000. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    x1 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, -0.01, n)], axis=1)
    x2 = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.01, 14 * ice[2], n)], axis=1)
    x2[:300, :2] = x1[:300, :2]
    x2[300:600, 2] = x1[300:600, 2]
    ctx = gpu_ctx_factory(ice, 'SP1', ray_finder='reference')
    o = ctx.find_solutions_batch(x1, x2)
    with orc.reference_procedure():
        ref = orc.raytrace_batch(x1, x2, ice)
    _assert_identical_to_oracle(o, ref)
    t = orc.raytrace_batch(x1, x2, ice)
    short = o['n_sol'] < t['n_sol']
    print('reference finder short of the true set on %d of %d pairs (%.2f %%)' % (short.sum(), n, 100 * short.mean()))
    assert np.all(o['n_sol'] <= t['n_sol']) and 0 < short.mean() < 0.02
    for i in np.flatnonzero(short)[:200]:
        assert _subset_ok(o['C0'][i], t['C0'][i])
    # with a reflective layer (Moore's Bay fixture): the plain call of the set loses the sign-change rescue as well
    g = golden('ref_mooresbay.npz')
    ctxm = gpu_ctx_factory(g['ice'], 'MB1', ray_finder='reference')
    m = len(g['points'])
    zr = float(g['z_reflection'])
    xr = np.tile(g['x_receiver'], (m, 1))
    om = ctxm.find_solutions_reflections_batch(g['points'], xr, 2, zr)
    with orc.reference_procedure():
        refm = orc.raytrace_batch_refl(g['points'], xr, g['ice'], 2, zr)
    for k in ('n_sol', 'type', 'reflection', 'reflection_case'):
        assert np.array_equal(om[k], refm[k]), k
    assert np.array_equal(om['C0'], refm['C0'], equal_nan=True)
    got = np.where(np.isnan(om['C0']), 0., om['C0'])
    np.testing.assert_allclose(got, g['ref_C0'], rtol=1e-6, atol=0)      # T06unit_test_C0_mooresbay.py:47
    assert np.array_equal(om['n_sol'], g['n_sol'])


def test_find_solutions_vs_oracle_survey_geometry(gpu_ctx_factory):
    """2e4 events x 5 channels of the BASELINE config-2 geometry, outer-product addressing."""
    rng = np.random.default_rng(123)
    n = 20000
    r = np.sqrt(rng.uniform(0, 4000. ** 2, n))
    ph = rng.uniform(0, 2 * np.pi, n)
    vert = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, 0, n)], axis=1)
    chan = np.array([[0., 0., -100. - i] for i in range(5)])
    ice = (1.78, 0.423, 77.)
    ctx = gpu_ctx_factory(ice)
    o = ctx.find_solutions_batch(vert, chan, outer=True)
    ref = orc.raytrace_batch(np.repeat(vert, 5, axis=0), np.tile(chan, (n, 1)), ice)
    _assert_identical_to_oracle(o, ref)
    assert (o['n_sol'] == 2).mean() > 0.3


def test_find_solutions_edge_cases(gpu_ctx_factory):
    ice = (1.78, 0.423, 77.)
    ctx = gpu_ctx_factory(ice)
    # empty batch
    o = ctx.find_solutions_batch(np.zeros((0, 3)), np.zeros((0, 3)))
    assert o['n_sol'].shape == (0,)
    # swapped end points (emitter above receiver), vertical pair, shadow-zone pair (no solution)
    x1 = np.array([[100., 50., -10.], [0., 0., -500.], [3900., 0., -5.], [-300., 200., -1200.]])
    x2 = np.array([[0., 0., -300.], [0.5, 0., -100.], [0., 0., -100.], [0., 0., -100.]])
    o = ctx.find_solutions_batch(x1, x2)
    ref = orc.raytrace_batch(x1, x2, ice)
    _assert_identical_to_oracle(o, ref)
    assert o['n_sol'][2] == 0 and np.all(np.isnan(o['C0'][2])) and np.all(o['type'][2] == 0)


@pytest.mark.parametrize('name', ['A', 'B', 'C'])
def test_attenuation_vs_reference_fixture(gpu_ctx_factory, name):
    g = golden('raytrace_%s.npz' % name)
    ctx = gpu_ctx_factory(g['ice'], str(g['att_model']))
    att = g['att']
    na, _, nf = att.shape
    x1 = np.repeat(g['x1'][:na], 2, axis=0)
    x2 = np.repeat(g['x2'][:na], 2, axis=0)
    C0 = g['C0'][:na].reshape(-1)
    out, nev = ctx.attenuation_batch(x1, x2, C0, g['fcoarse'], return_neval=True)
    ref = att.reshape(na * 2, nf)
    assert max_rel(out, ref) < 1e-6
    # identical adaptive decisions and identical bits as the oracle's QUADPACK restatement
    out_o, nev_o = orc.attenuation_batch(x1, x2, C0, g['ice'], str(g['att_model']), g['fcoarse'], return_neval=True)
    m = np.isfinite(ref)
    assert np.array_equal(nev[m], nev_o[m])
    assert np.array_equal(out, out_o, equal_nan=True)


def test_attenuation_length_models(gpu_ctx_factory):
    z = np.linspace(-2800., 5., 57)
    f = np.array([0.05, 0.3, 0.999, 1.0, 1.7])[:, None] * np.ones_like(z)[None, :]
    for model in ('SP1', 'GL1', 'MB1', 'GL2'):
        ctx = gpu_ctx_factory((1.78, 0.423, 77.), model)
        got = ctx.attenuation_length(z[None, :] * np.ones_like(f), f)
        ref = orc.attenuation_length(z[None, :] * np.ones_like(f), f, model)
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        m = np.isfinite(ref)
        assert np.max(np.abs(got[m] - ref[m]) / np.abs(ref[m])) < 1e-12, model


def test_gl3_attenuation(gpu_ctx_factory):
    """GL3: attenuation length from the depth table and the segment-sum path integral (incl. the QUADPACK run on ds around
    the turning depth, with and without the break point inside the segment) -- bit-equal to the oracle, which is pinned
    against the reference (tests/test_oracle_golden.py::test_gl3_attenuation_vs_reference); and directly vs the fixture."""
    g = golden('ref_gl3.npz')
    orc.set_gl3_table(g['gl3_table'])
    ctx = gpu_ctx_factory(g['ice'], 'GL3', gl3_table=g['gl3_table'])
    zz = g['z_probe']
    for j, f in enumerate(g['f_probe']):
        L = ctx.attenuation_length(zz, np.full(len(zz), f))
        assert np.array_equal(L, orc.attenuation_length(zz, np.full(len(zz), f), 'GL3'))
    rng = np.random.default_rng(8)
    n = 4000
    r, ph = np.sqrt(rng.uniform(0, 3000. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    x1 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2900., -5., n)], axis=1)
    x2 = np.stack([np.zeros(n), np.zeros(n), rng.choice([-3., -60., -100., -400.], n)], axis=1)
    t = ctx.find_solutions_batch(x1, x2)
    n_checked = 0
    for s in range(2):
        sel = np.flatnonzero(t['n_sol'] > s)
        a_gpu, ne_gpu = ctx.attenuation_batch(x1[sel], x2[sel], t['C0'][sel, s], g['fcoarse'], return_neval=True)
        a_ref, ne_ref = orc.attenuation_batch(x1[sel], x2[sel], t['C0'][sel, s], g['ice'], 'GL3', g['fcoarse'], return_neval=True)
        assert np.array_equal(ne_gpu, ne_ref)
        assert np.array_equal(a_gpu, a_ref), np.max(np.abs(a_gpu - a_ref) / a_ref)
        n_checked += len(sel)
    assert n_checked > 3000
    for s in range(2):  # the reference's own numbers
        sel = np.flatnonzero(~np.isnan(g['C0'][:, s]))
        a = ctx.attenuation_batch(g['x1'][sel], g['x2'][sel], g['C0'][sel, s], g['fcoarse'])
        assert np.max(np.abs(a - g['att'][sel, s]) / g['att'][sel, s]) < 1e-10
    with pytest.raises(ValueError):
        import nuradiomc_amd
        nuradiomc_amd.Context(g['ice'], 'GL3')


@pytest.mark.gpu
def test_bottom_reflections_vs_reference_table_and_oracle(gpu_ctx_factory):
    """Reflections off the bottom of the ice shelf: the reference's own golden table reference_C0_MooresBay.pkl (1000
    vertices, n_reflections = 2, up to 10 solutions) at the reference test's tolerance, and every record bit-for-bit /
    to 1e-9 against the oracle; then the records from given solutions (set_solution) and the attenuation along the path
    segments (MB1) against the reference's values."""
    g = golden('ref_mooresbay.npz')
    ctx = gpu_ctx_factory(g['ice'], 'MB1')
    n = len(g['points'])
    zr = float(g['z_reflection'])
    x2 = np.tile(g['x_receiver'], (n, 1))
    o = ctx.find_solutions_reflections_batch(g['points'], x2, 2, zr)
    got = np.where(np.isnan(o['C0']), 0., o['C0'])
    np.testing.assert_allclose(got, g['ref_C0'], rtol=1e-6, atol=0)      # T06unit_test_C0_mooresbay.py:47
    assert np.array_equal(o['n_sol'], g['n_sol'])
    ref = orc.raytrace_batch_refl(g['points'], x2, g['ice'], 2, zr)
    for k in ('n_sol', 'type', 'reflection', 'reflection_case', 'n_surface', 'n_segments', 'surface_mask'):
        assert np.array_equal(o[k], ref[k]), k
    assert np.array_equal(o['C0'], ref['C0'], equal_nan=True)              # same arithmetic: same bits
    m = ~np.isnan(ref['C0'])
    for k in ('C1', 'D', 'T'):
        assert np.array_equal(np.isnan(o[k]), ~m) and max_rel(o[k][m], ref[k][m]) < 1e-9, k
    for k in ('launch', 'receive'):
        assert np.max(np.abs(o[k][m] - ref[k][m])) < 1e-12, k
    has = m & (ref['n_surface'] > 0)
    assert np.array_equal(np.isnan(o['refl_angle']), ~has) and np.max(np.abs(o['refl_angle'][has] - ref['refl_angle'][has])) < 1e-12
    # the outer-product form (every vertex with the one receiver) gives the same tables
    o2 = ctx.find_solutions_reflections_batch(g['points'][:200], g['x_receiver'][None], 2, zr, outer=True)
    for k in o2:
        assert np.array_equal(o2[k], o[k][:200], equal_nan=True), k
    # records of given solutions vs the reference's own values
    nf = int(g['n_full'])
    sol = {k: g[k][:nf] for k in ('n_sol', 'C0', 'reflection', 'reflection_case')}
    r = ctx.find_solutions_reflections_batch(g['points'][:nf], x2[:nf], 2, zr, solutions=sol)
    mf = ~np.isnan(g['C0'][:nf])
    assert np.array_equal(r['type'][mf], g['type'][:nf][mf])
    assert max_rel(r['D'][mf], g['D'][mf]) < 1e-7 and max_rel(r['T'][mf], g['T'][mf]) < 1e-7
    assert np.max(np.abs(r['launch'][mf] - g['launch'][mf])) < 1e-12 and np.max(np.abs(r['receive'][mf] - g['receive'][mf])) < 1e-12
    assert np.array_equal(r['n_surface'][mf], np.sum(~np.isnan(g['refl_angle']), axis=2)[mf])
    idx = np.argwhere(mf)
    att = ctx.attenuation_reflections_batch(g['points'][idx[:, 0]], x2[idx[:, 0]], g['C0'][:nf][mf], g['reflection'][:nf][mf],
                                            g['reflection_case'][:nf][mf], zr, g['fcoarse'])
    assert np.max(np.abs(att - g['att'][mf]) / g['att'][mf]) < 1e-6
    # n_reflections = 0 reproduces the plain solution finder
    p = ctx.find_solutions_batch(g['points'][:300], x2[:300])
    q = ctx.find_solutions_reflections_batch(g['points'][:300], x2[:300], 0, zr)
    # (to the 1e-7 of the reference's first root: with a reflective layer every call is the reference's procedure -- hybr + two Brent
    # searches --, the plain finder takes every root from a bracket since round 5; counts agree unless the procedure loses a root)
    same = p['n_sol'] == q['n_sol']
    assert same.mean() >= 0.99 and np.all(p['n_sol'] >= q['n_sol'])
    assert np.array_equal(p['type'][same], q['type'][same])
    for k in ('C0', 'D', 'T'):
        assert max_rel(p[k][same], q[k][same]) < (2e-7 if k == 'C0' else 1e-6), k
    for k in ('launch', 'receive'):
        assert np.nanmax(np.abs(p[k][same] - q[k][same])) < 4e-7, k
    with pytest.raises(Exception, match='reflective layer'):
        ctx.find_solutions_reflections_batch(g['points'][:2], x2[:2], 1, 0.)


def test_deep_receivers_keep_the_reference_procedure(gpu_ctx_factory):
    """Pairs whose upper end point lies deeper than 10 z_0 are flagged by the finder without the hybr stage and served by the
    reference's procedure in a second launch (raytrace_roots_kernel, only_flagged): a list that mixes shallow and deep receivers
    equals the oracle's tables bit for bit, which takes the same decision per pair."""
    ice = (1.78, 0.51, 37.25)   # greenland_simple: 10 z_0 = 372.5 m
    rng = np.random.default_rng(77)
    n = 6000
    r, ph = np.sqrt(rng.uniform(0, 3000. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    x1 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, -1., n)], axis=1)
    x2 = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-900., -1., n)], axis=1)
    deep = np.maximum(x1[:, 2], x2[:, 2]) < -372.5
    assert 0.2 < deep.mean() < 0.8
    ctx = gpu_ctx_factory(ice, 'GL1')
    o = ctx.find_solutions_batch(x1, x2)
    ref = orc.raytrace_batch(x1, x2, ice)
    assert np.all(o['n_sol'] >= 0)
    _assert_identical_to_oracle(o, ref)
    assert o['n_sol'][deep].sum() > 100 and o['n_sol'][~deep].sum() > 100


@pytest.mark.parametrize('ice,model', [((1.78, 0.423, 77.), 'SP1'), ((1.78, 0.51, 37.25), 'GL1'), ((1.78, 0.46, 34.5), 'MB1')])
def test_find_solutions_random_geometries(gpu_ctx_factory, ice, model):
    """1e5 random pairs per ice model -- end points anywhere between the surface and 2.7 km, receivers on both sides of the 10 z_0
    line that separates the bracketed finder from the reference's procedure, pairs above each other, at equal depth, metres apart --
    equal to the oracle bit for bit (both finders, both launches, the same decision per pair)."""
    rng = np.random.default_rng(int(ice[2] * 100))
    n = 100000
    z0 = ice[2]
    r, ph = np.sqrt(rng.uniform(0, 5000. ** 2, n)), rng.uniform(0, 2 * np.pi, n)
    x1 = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700, -0.01, n)], axis=1)
    x2 = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.01, 14 * z0, n)], axis=1)
    x2[:500, :2] = x1[:500, :2]                                  # vertically above / below each other
    x2[500:1000, 2] = x1[500:1000, 2]                            # equal depth
    x1[1000:1500] = x2[1000:1500] + rng.uniform(-3, 3, (500, 3)) * [1, 1, 0.2]   # metres apart
    x1[:, 2] = np.minimum(x1[:, 2], -0.01)
    ctx = gpu_ctx_factory(ice, model)
    o = ctx.find_solutions_batch(x1, x2)
    ref = orc.raytrace_batch(x1, x2, ice)
    _assert_identical_to_oracle(o, ref)
    assert np.bincount(o['n_sol'], minlength=3)[2] > 0.2 * n
