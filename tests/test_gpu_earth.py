"""Earth-absorption weights on the GPU (nrhip_earth_weights_batch) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
from conftest import golden, max_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import nuradiomc_amd as nr
    c = nr.Context((1.78, 0.423, 77.), 'SP1')
    yield c
    c.close() if hasattr(c, 'close') else None


def test_weights_vs_reference_and_oracle(ctx):
    """All four modes of earth_attenuation.get_weight (tests/golden/ref_earth_weights.npz: the reference run event by
    event) through the drop-in get_weight with arrays, and event by event with scalars as simulation.py:897-903 calls it.
    Tolerance 1e-6 relative (north_star); slant depths 1e-12: the samples and their layers are the same doubles, only the
    order of the trapezoid sum differs."""
    from nuradiomc_amd import earth_attenuation as ea
    from oracle import earth_oracle as eo
    g = golden('ref_earth_weights.npz')
    n = len(g['zenith'])
    for mode in ('simple', 'core_mantle_crust_simple', 'core_mantle_crust', 'PREM'):
        w = ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode=mode, cross_section_type='ctw',
                          vertex_position=g['vertex'], phi_nu=g['azimuth'], ctx=ctx)
        ref = g['weight_' + mode]
        big = ref > 1e-100
        assert max_rel(w[big], ref[big]) < 1e-6, mode
        assert np.max(np.abs(w - ref)) < 1e-9, mode
        wo = eo.get_weight(g['zenith'], g['azimuth'], g['energy'], g['flavor'], g['vertex'], mode)
        assert max_rel(w[big], wo[big]) < 1e-6, mode
        wg = ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode=mode, cross_section_type='ghandi',
                           vertex_position=g['vertex'], phi_nu=g['azimuth'], ctx=ctx)
        ref = g['weight_ghandi_' + mode]
        assert max_rel(wg[ref > 1e-100], ref[ref > 1e-100]) < 1e-6 and np.max(np.abs(wg - ref)) < 1e-9, mode
    for i in (0, 1, 2, 7, 42, 103):
        w1 = ea.get_weight(float(g['zenith'][i]), float(g['energy'][i]), int(g['flavor'][i]), mode='core_mantle_crust',
                           cross_section_type='ctw', vertex_position=g['vertex'][i], phi_nu=float(g['azimuth'][i]), ctx=ctx)
        assert isinstance(w1, float) and abs(w1 - g['weight_core_mantle_crust'][i]) < 1e-9
    d = np.stack([np.sin(g['zenith']) * np.cos(g['azimuth']), np.sin(g['zenith']) * np.sin(g['azimuth']), np.cos(g['zenith'])], axis=1)
    for cls, name in ((ea.CoreMantleCrustModel, 'core_mantle_crust'), (ea.PREM, 'PREM')):
        sd = cls(ctx).slant_depth(g['vertex'], d)
        ref = g['slant_depth_' + name]
        assert np.array_equal(sd == 0, ref == 0)
        assert max_rel(sd[ref > 0], ref[ref > 0]) < 1e-12, name   # incl. the chords of 2 .. 5 samples: same surface sample
        assert abs(cls(ctx).slant_depth(g['vertex'][7], d[7]) - ref[7]) <= 1e-12 * ref[7]


def test_given_cross_sections(ctx):
    """Tabulated cross sections evaluated by the caller (NRHIP_XS_GIVEN): the 'csms' table (cc + nc of each event, from
    tests/golden/ref_csms.npz: the reference's values) through every mode, against the oracle with the same values."""
    from nuradiomc_amd import earth_attenuation as ea, cross_sections as xs
    from oracle import earth_oracle as eo
    g = golden('ref_earth_weights.npz')
    n = len(g['zenith'])
    sigma = xs.csms(g['energy'], np.full(n, 'cc'), g['flavor']) + xs.csms(g['energy'], np.full(n, 'nc'), g['flavor'])
    assert np.all(sigma > 0)
    for mode in ('simple', 'core_mantle_crust_simple', 'core_mantle_crust', 'PREM'):
        w = ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode=mode, vertex_position=g['vertex'], phi_nu=g['azimuth'], ctx=ctx,
                          cross_section=sigma)
        wo = eo.get_weight(g['zenith'], g['azimuth'], sigma, g['flavor'], g['vertex'], mode, cross_section_type='given')
        big = wo > 1e-100
        assert big.sum() > n // 4 and max_rel(w[big], wo[big]) < 1e-6 and np.max(np.abs(w - wo)) < 1e-9, mode
        wc = ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode=mode, vertex_position=g['vertex'], phi_nu=g['azimuth'], ctx=ctx)
        assert 0.2 < np.median(np.log(w[big & (wc > 1e-100) & (w < 0.99)]) / np.log(wc[big & (wc > 1e-100) & (w < 0.99)])) < 5   # same physics, another table


def test_hedis_cross_sections_from_a_data_file(ctx):
    """cross_section_type='hedis_bgr18' end to end: the table file (here the synthetic one of tests/golden, whose reference values
    test_oracle_golden checks) -> host integration / interpolation -> NRHIP_XS_GIVEN -> weights, against the oracle fed with the
    reference's own cross sections of the same events; mode 'simple' asks for flavor 0, which the file does not hold (the reference
    fails there too)."""
    import os
    from nuradiomc_amd import earth_attenuation as ea, cross_sections as xs
    from oracle import earth_oracle as eo
    g = golden('ref_earth_weights.npz')
    xs.set_bgr18_file(os.path.join(os.path.dirname(__file__), 'golden', 'bgr18_synthetic.npz'))
    try:
        sigma = xs.get_nu_cross_section(g['energy'], g['flavor'], 'total', 'hedis_bgr18')
        for mode in ('core_mantle_crust_simple', 'core_mantle_crust', 'PREM'):
            w = ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode=mode, vertex_position=g['vertex'], phi_nu=g['azimuth'],
                              ctx=ctx, cross_section_type='hedis_bgr18')
            wo = eo.get_weight(g['zenith'], g['azimuth'], sigma, g['flavor'], g['vertex'], mode, cross_section_type='given')
            big = wo > 1e-100
            assert big.sum() > len(w) // 4 and max_rel(w[big], wo[big]) < 1e-6 and np.max(np.abs(w - wo)) < 1e-9, mode
        with pytest.raises(ValueError):
            ea.get_weight(g['zenith'], g['energy'], g['flavor'], mode='simple', ctx=ctx, cross_section_type='hedis_bgr18')
    finally:
        xs.set_bgr18_file(None)


def test_errors_and_edges(ctx):
    from nuradiomc_amd import earth_attenuation as ea
    import nuradiomc_amd as nr
    assert ea.get_weight(2., 1e18, 12, mode='None') == 1.
    with pytest.raises(NotImplementedError):
        ea.get_weight(2., 1e18, 12, mode='two_layers', ctx=ctx)
    with pytest.raises(FileNotFoundError):
        ea.get_weight(2., 1e18, 12, mode='PREM', cross_section_type='hedis_bgr18', vertex_position=np.zeros(3), phi_nu=0.,
                      ctx=ctx)                                                              # a download of the reference: no file here
    # 'csms' has no rows for inttype='total' (what get_interaction_length asks for): cross section 0, weight 1 -- as the reference
    assert ea.get_weight(np.full(3, 2.), np.full(3, 1e18), np.full(3, 12), mode='simple', cross_section_type='csms', ctx=ctx).tolist() == [1., 1., 1.]
    assert len(ea.get_weight(np.zeros(0), np.zeros(0), np.zeros(0, int), mode='simple', ctx=ctx)) == 0
    # below 1e4 GeV the parametrisation is not valid: NaN (cross_sections.py:69-76)
    assert np.isnan(ea.get_weight(2., 1e12, 12, mode='simple', ctx=ctx))
    # a chord that leaves the Earth at once (vertex above the surface looking up): column density 0, weight 1
    assert ea.PREM(ctx).slant_depth(np.array([0., 0., 10.]), np.array([0., 0., 1.])) == 0.
    with pytest.raises(nr.NrhipError):
        ctx.earth_weights_batch(np.ones(2), np.full(2, 1e18), np.full(2, 12), 5)
    with pytest.raises(NotImplementedError):
        ctx.earth_weights_batch(np.ones(2), np.full(2, 1e18), np.full(2, 12), 0, cross_section_type='hedis_bgr18')


def test_full_size_properties(ctx):
    """1e6 events (BASELINE config 2's event count): the weights are 1-periodic quantities of the chord only -- (i) in a
    constant-density sphere the column density is rho * chord length up to half a trapezoid cell, (ii) the weight is
    monotone in the cross section (energy), (iii) rotating vertex and direction about the vertical leaves it unchanged
    up to the rounding of the surface sample (half a cell of crust)."""
    from nuradiomc_amd import earth_attenuation as ea
    rng = np.random.default_rng(5)
    n = 1_000_000
    zen = np.arccos(rng.uniform(-1., 1., n))
    az = rng.uniform(0., 2 * np.pi, n)
    vertex = np.stack([rng.uniform(-3e3, 3e3, n), rng.uniform(-3e3, 3e3, n), -rng.uniform(1., 2700., n)], axis=1)
    d = np.stack([np.sin(zen) * np.cos(az), np.sin(zen) * np.sin(az), np.cos(zen)], axis=1)
    R = 6.3710e6
    rho = 3.0 * 6.241509744511525e+33 / 0.01 ** 3
    _, sd = ctx.earth_weights_batch(zen, np.full(n, 1e18), np.full(n, 12), 2, endpoint=vertex, direction=d,
                                    model=(R, [R], [[rho, 0., 0., 0.]]), return_slant_depth=True)
    e = vertex + np.array([0., 0., R])
    dot = np.sum(e * d, axis=1)
    chord = np.maximum(-dot + np.sqrt(dot ** 2 - np.sum(e ** 2, axis=1) + R ** 2), 0.)   # a vertex 1 m deep and 3.7 km off the origin is outside the sphere
    # only the surface half-cell may be missing (a cell is < 1000 m: n_steps - 1 = floor(chord / 500 m)); chords < 500 m count 0
    assert np.all(np.abs(sd - rho * chord) <= rho * 500. * 1.001)
    assert np.all(sd <= rho * chord * (1. + 1e-5))   # the chord length itself carries the cancellation of dot^2 - |e|^2 + R^2
    assert np.all(sd >= 0)
    w1 = ea.get_weight(zen, np.full(n, 1e17), np.full(n, 12), mode='PREM', vertex_position=vertex, phi_nu=az, ctx=ctx)
    w2 = ea.get_weight(zen, np.full(n, 1e19), np.full(n, 12), mode='PREM', vertex_position=vertex, phi_nu=az, ctx=ctx)
    assert np.all(w2 <= w1) and np.all(w1 <= 1.) and np.all(w2 >= 0.)
    assert w2.mean() < w1.mean() < 1.
