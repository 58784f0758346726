"""nuradiomc_amd.generator.generate_eventlist_cylinder against event lists the reference's generator produced from the same
seeds (tests/golden/ref_generator.npz, generator tests/golden/gen/gen_generator.py): the same random stream, hence the same
list -- positions, directions, flavours, energies, interaction types, inelasticities, shower rows, ids, attributes."""
import numpy as np
from conftest import golden
from nuradiomc_amd import generator


def test_event_lists_like_the_reference_generator():
    g = golden('ref_generator.npz')
    for i in range(3):
        kw = eval(str(g['c%d_kwargs' % i]))
        ev = generator.generate_eventlist_cylinder(**kw)
        keys = [k[len('c%d/' % i):] for k in g.files if k.startswith('c%d/' % i)]
        assert sorted(keys) == sorted(ev.data), (sorted(keys), sorted(ev.data))
        for k in keys:
            ref, got = g['c%d/%s' % (i, k)], ev.data[k]
            if ref.dtype.kind == 'S':
                assert [x.decode() for x in ref] == [str(x) for x in got], k
            elif ref.dtype.kind == 'f':
                assert np.allclose(got, ref, rtol=1e-13, atol=0), k       # pow / log of another numpy build: last bit
            else:
                assert np.array_equal(got, ref), k
        for k in [k for k in g.files if k.startswith('c%d_attr/' % i)]:
            name = k.split('/', 1)[1]
            if name.startswith('NuRadioMC'):
                continue
            assert np.allclose(np.asarray(ev.attrs[name], float), np.asarray(g[k], float), rtol=1e-13), name
        em = np.array([str(t) for t in ev.data['shower_type']]) == 'em'
        assert em.sum() > 10 and len(np.unique(ev.data['event_group_ids'])) == ev.attrs['n_events']


def test_event_lists_with_the_tabulated_hedis_model():
    """cross_sections_model='hedis_bgr18' (cc / nc from the integrated table, inelasticity through its cumulative distribution at
    the energy node above the event) against the reference's generator on the synthetic table of tests/golden
    (ref_generator_hedis.npz, generator tests/golden/gen/gen_generator_hedis.py): same draws in the same order, hence the same
    interaction types; inelasticities to 1e-10 (the cumulative sums are formed in another order)."""
    import os
    from nuradiomc_amd import cross_sections as xs
    g = golden('ref_generator_hedis.npz')
    xs.set_bgr18_file(os.path.join(os.path.dirname(__file__), 'golden', 'bgr18_synthetic.npz'))
    try:
        for i in range(2):
            kw = eval(str(g['c%d_kwargs' % i]))
            ev = generator.generate_eventlist_cylinder(cross_sections_model='hedis_bgr18', **kw)
            keys = [k[len('c%d/' % i):] for k in g.files if k.startswith('c%d/' % i)]
            assert sorted(keys) == sorted(ev.data)
            for k in keys:
                ref, got = g['c%d/%s' % (i, k)], ev.data[k]
                if ref.dtype.kind == 'S':
                    assert [x.decode() for x in ref] == [str(x) for x in got], k
                elif ref.dtype.kind == 'f':
                    assert np.allclose(got, ref, rtol=1e-10 if k in ('inelasticity', 'shower_energies') else 1e-13, atol=0), k
                else:
                    assert np.array_equal(got, ref), k
            y = ev.data['inelasticity']
            assert np.all((y > 0) & (y < 1)) and 0.1 < y.mean() < 0.4
    finally:
        xs.set_bgr18_file(None)
    import pytest
    with pytest.raises(FileNotFoundError):
        generator.generate_eventlist_cylinder(10, 1e17, 1e18, dict(fiducial_rmin=0., fiducial_rmax=1e3, fiducial_zmin=-1e3, fiducial_zmax=0.),
                                              seed=1, cross_sections_model='hedis_bgr18')
    with pytest.raises(NotImplementedError):
        generator.generate_eventlist_cylinder(10, 1e17, 1e18, dict(fiducial_rmin=0., fiducial_rmax=1e3, fiducial_zmin=-1e3, fiducial_zmax=0.),
                                              seed=1, cross_sections_model='csms')


def test_event_list_with_deposited_energies():
    """deposited=True (Emin .. Emax are deposited energies; the neutrino energy is E / y except for nu_e CC; generator.py:199-224,
    :1247-1252) against the reference's generator (ref_generator_deposited.npz, tests/golden/gen/gen_generator_deposited.py)"""
    g = golden('ref_generator_deposited.npz')
    kw = eval(str(g['c0_kwargs']))
    ev = generator.generate_eventlist_cylinder(**kw)
    keys = [k[len('c0/'):] for k in g.files if k.startswith('c0/')]
    assert sorted(keys) == sorted(ev.data) and bool(ev.attrs['deposited']) and bool(g['c0_attr_deposited'])
    for k in keys:
        ref, got = g['c0/' + k], ev.data[k]
        if ref.dtype.kind == 'S':
            assert [x.decode() for x in ref] == [str(x) for x in got], k
        elif ref.dtype.kind == 'f':
            assert np.allclose(got, ref, rtol=1e-13, atol=0), k
        else:
            assert np.array_equal(got, ref), k
    had = np.array([str(t) for t in ev.data['shower_type']]) == 'had'
    nue_cc = (np.abs(ev.data['flavors']) == 12) & (np.array([str(t) for t in ev.data['interaction_type']]) == 'cc')
    assert np.all(ev.data['shower_energies'][had & ~nue_cc] <= 1e19 * (1 + 1e-12)) and ev.data['energies'].max() > 1e20
