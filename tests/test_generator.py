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
