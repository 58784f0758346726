"""N > 1 path on CPU: world size 2, gloo.  Shard ranges tile the event list; the all-gather of triggered masks
reassembles the global mask in event order on every rank."""
import os
import sys
import pytest
import numpy as np
import torch.multiprocessing as mp
from conftest import ROOT
from nuradiomc_amd.comm import shard_range


def test_shard_ranges_tile_the_event_list():
    for n in (0, 1, 7, 1000, 1000003):
        for W in (1, 2, 3, 8):
            r = [shard_range(n, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(W - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, n, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from nuradiomc_amd import comm
    from torch_gather import gather_triggered
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    full = (np.arange(n) * 7919 % 13 == 0).astype(np.uint8)   # what a single process would have produced
    a, b = comm.shard_range(n, rank, world)
    # the PRODUCT's communicator (nuradiomc_amd.comm.Comm; no device here, so its collectives go over its TCP star): shard ->
    # barrier -> gather of the masks -> counters, the sequence bench.py runs on every rank
    c = comm.Comm(None, rank, world, addr='127.0.0.1', port=port + 1000, backend='tcp')
    c.barrier()
    got = c.allgather_masks(full[a:b].copy(), b - a, n)
    tot = c.allreduce_sum([int(full[a:b].sum()), b - a])
    c.barrier()
    ok = bool(np.array_equal(got, full)) and [int(v) for v in tot] == [int(full.sum()), n] and c.mode == 'tcp'
    c.close()
    # cross-check: the same shards through torch.distributed on gloo (tests/torch_gather.py, the twin of allgather_masks)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    twin = gather_triggered(full[a:b], n, dist=dist)
    q.put((rank, ok and bool(np.array_equal(twin, got))))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_triggered_world2_gloo():
    """World size 2 on CPU: the product's Comm gathers the sharded trigger masks of one list into the one-process mask on every
    rank and sums the counters; torch.distributed (gloo) gathers the same shards to the same mask."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, 1001, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}


def test_distance_cut_host_logic():
    """nuradiomc_amd.station.distance_cut (host side of speedup.distance_cut) against the reference-generated fixture's
    inputs: per shower max(100 m, 10 ** poly(log10(E_sum))) with the energy sum over showers of the same group whose
    distance to the group's first vertex differs by < 10 m (simulation.py:125-131, :155-163, :1398-1409)."""
    import numpy as np
    from conftest import golden
    from nuradiomc_amd.station import distance_cut
    g = golden('chain_groups_dcut_N256.npz')
    gid = g['group']
    first = np.flatnonzero(np.concatenate([[True], gid[1:] != gid[:-1]]))
    gb = np.concatenate([first, [len(gid)]]).astype(np.int32)
    coef = g['distance_cut_coefficients']
    got = distance_cut(g['vertex'], g['energy'], gb, coef, 10.)
    poly = np.polynomial.polynomial.Polynomial(coef)
    for a, b in zip(gb[:-1], gb[1:]):
        d = np.linalg.norm(g['vertex'][a:b] - g['vertex'][a], axis=1)
        for i in range(a, b):
            e_sum = g['energy'][a:b][np.abs(d - d[i - a]) < 10.].sum()
            assert abs(got[i] - max(100., 10 ** poly(np.log10(e_sum)))) <= 4e-16 * got[i]  # vector vs scalar pow: 1 ulp
    # no groups: every shower on its own
    single = distance_cut(g['vertex'], g['energy'], None, coef)
    assert np.allclose(single, np.maximum(100., 10 ** poly(np.log10(g['energy']))), rtol=4e-16, atol=0)
    assert distance_cut(np.zeros((1, 3)), [0.], None, coef)[0] == 100.


def _id_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from nuradiomc_amd import comm
    payload = bytes(range(128)) if rank == 0 else b''
    got = comm.exchange_bytes(payload, rank, world, addr='127.0.0.1', port=port, timeout=60.)
    q.put((rank, got == bytes(range(128))))


def _tcp_worker(rank, world, port, backend, q, allow_tcp=True):
    sys.path.insert(0, ROOT)
    from nuradiomc_amd import comm
    n_total = 1003
    a, b = comm.shard_range(n_total, rank, world)
    full = (np.arange(n_total) % 7 == 0).astype(np.uint8)
    try:
        c = comm.Comm(None, rank, world, addr='127.0.0.1', port=port, backend=backend, allow_tcp=allow_tcp)
    except comm.CommError as e:
        q.put((rank, False, [], 0., 'error: ' + str(e)[:40]))
        return
    try:
        # ONE rank holds something that is no shard_range shard: refused on EVERY rank (the sizes are agreed on before the
        # collective, so nobody is left waiting in it), not silently permuted
        c.allgather_masks(full[a:b].copy(), b - a + (1 if rank == 1 else 0), n_total)
        q.put((rank, False, [], 0., 'no ValueError'))
        return
    except ValueError:
        pass
    c.barrier()
    mask = c.allgather_masks(full[a:b].copy(), b - a, n_total)
    tot = c.allreduce_sum([int(full[a:b].sum()), rank])
    mx = c.allreduce_max([float(rank) + 0.5])
    c.barrier()
    mode = c.mode
    c.close()
    q.put((rank, bool(np.array_equal(mask, full)), [int(v) for v in tot], float(mx[0]), mode))


@pytest.mark.parametrize('backend', ['tcp', 'rccl'])
def test_comm_tcp_star_world3(backend):
    """The host-side stand-in of the communicator (nuradiomc_amd.comm.Comm, mode 'tcp'): the sharded masks of ONE list are
    gathered in rank order, sums and maxima agree on every rank -- world size 3 on the loopback, no GPU.  With backend 'rccl'
    requested and no device context the vote fails on every rank; with allow_tcp the same path is taken (what a node without a
    working RCCL may be told to fall back to instead of hanging), without it every rank raises CommError."""
    import multiprocessing
    ctx = multiprocessing.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + (0 if backend == 'tcp' else 1)
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, backend, q)) for r in range(3)]
    for p in procs[1:] + procs[:1]:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    n_set = int((np.arange(1003) % 7 == 0).sum())
    for rank, ok, tot, mx, mode in res:
        assert ok and tot == [n_set, 3] and mx == 2.5 and mode == 'tcp', (rank, ok, tot, mx, mode)
    if backend == 'rccl':   # the default: a failed RCCL vote is an error on every rank, not a silent TCP run
        procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port + 2, backend, q, False)) for r in range(3)]
        for p in procs[1:] + procs[:1]:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(60)
        assert all(mode.startswith('error: nuradiomc_amd.comm: RCCL did not come up') for _, _, _, _, mode in res), res


def test_comm_id_exchange_and_shard_helpers():
    """nuradiomc_amd.comm (the RCCL path of bench.py, no torch): the 128-byte communicator id reaches every rank over TCP (world
    size 3 on the loopback), shard_chunks tiles a sorted list round-robin, the sequencing helpers order and split as the
    reference does."""
    import multiprocessing
    from nuradiomc_amd import comm, sequencing
    ctx = multiprocessing.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_id_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs[1:] + procs[:1]:      # the clients may come up before the server
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True, 2: True}
    # chunked round-robin sharding of a sorted list: a partition, chunk c -> rank c % W
    n, W, chunk = 100003, 4, 1000
    parts = [comm.shard_chunks(n, r, W, chunk) for r in range(W)]
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(n))
    assert all(np.all((p // chunk) % W == r) for r, p in enumerate(parts))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= chunk
    # order in which the reference meets the showers: group -> station -> channel -> shower
    first = np.array([[-1, 3, 0, 2, -1, 1],     # station 0: first channel with a kept ray per shower (-1: none)
                      [0, 0, 1, -1, -1, 0]])    # station 1
    order = sequencing.reference_draw_order(first, np.array([0, 3, 6]))   # two groups of three showers
    assert list(order) == [2, 1, 0, 5, 3]       # group 0: station 0 ch 0 (sh 2), ch 3 (sh 1), then station 1 ch 0 (sh 0); ...
    # split_event_time_diff: gaps larger than the limit start a new sub-event (simulation.group_into_events :906-947)
    t = np.array([100., 5000., 130., 5100., 9999.])
    assert list(sequencing.split_event_times(t, 1000.)) == [0, 1, 0, 1, 2]
    assert list(sequencing.split_event_times(t, 1e6)) == [0, 0, 0, 0, 0]


def _run_bench(argv, env=None, timeout=900):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_launches_ranks():
    """`bench.py --gpus N` without a launcher starts N ranks itself (the reference's runner.py:53-87 starts N processes over a
    split list); under a launcher whose WORLD_SIZE contradicts --gpus it exits non-zero instead of printing a line for a run that
    never had N ranks.  --dry-run: no GPU, the ranks meet on the product's TCP star."""
    import json
    r = _run_bench(['--gpus', '3', '--dry-run', '--events', '1001'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [q for q in r.stdout.splitlines() if q.startswith('{')]
    assert len(lines) == 1      # ONE line: rank 0's
    j = json.loads(lines[0])
    assert j['n_gpus'] == 3 and j['ranks'] == 3 and j['rank_sum'] == 3 and j['local_rank_sum'] == 3 and j['mask_ok']
    r = _run_bench(['--gpus', '4', '--dry-run'], env=dict(WORLD_SIZE='2', RANK='0'))
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr and not r.stdout.strip()
    # a failing rank fails the launch (here: no such option)
    r = _run_bench(['--gpus', '2', '--dry-run', '--events', 'x'])
    assert r.returncode != 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_the_launcher():
    """shard -> barrier -> gather -> counters on hardware: `bench.py --gpus 2 --allow-tcp --scaling strong` on a one-GPU box (RCCL
    refuses two ranks on one device, the vote fails, the collectives go over the star) gathers the same mask as the one-rank run
    of the same list."""
    import json
    common = ['--events', '200000', '--scaling', 'strong', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
    import tempfile
    sha_file = os.path.join(tempfile.mkdtemp(), 'expected_mask_sha16.json')
    os.environ['NRHIP_EXPECTED_SHA_JSON'] = sha_file     # (the committed file holds the hash of the full 1e6-event list)
    try:
        one = _run_bench(['--gpus', '1', '--write-expected-sha'] + common)
        assert one.returncode == 0, one.stderr[-2000:]
        two = _run_bench(['--gpus', '2', '--allow-tcp'] + common)
        assert two.returncode == 0, two.stderr[-2000:]
        # the N-rank line checks itself against the one-rank hash: a wrong hash on file makes the run fail
        known = json.load(open(sha_file))
        json.dump({k: 'deadbeefdeadbeef' for k in known}, open(sha_file, 'w'))
        wrong = _run_bench(['--gpus', '2', '--allow-tcp'] + common)
        assert wrong.returncode != 0 and 'DIFFERS from the one-rank mask' in wrong.stderr
    finally:
        del os.environ['NRHIP_EXPECTED_SHA_JSON']
    j1 = json.loads([q for q in one.stdout.splitlines() if q.startswith('{')][-1])
    j2 = json.loads([q for q in two.stdout.splitlines() if q.startswith('{')][-1])
    assert j1['n_gpus'] == 1 and j2['n_gpus'] == 2
    assert j2['config']['collectives'] in ('tcp', 'rccl')   # 'rccl' on a box with two GPUs
    assert j1['config']['n_triggered_all'] == j2['config']['n_triggered_all'] > 0
    assert j1['config']['gathered_mask_sha16'] == j2['config']['gathered_mask_sha16']
    assert j2['config']['gathered_mask_check'].startswith('equal to the one-rank mask')
    assert j2['config']['all_ranks']['n_pairs'] == j1['config']['all_ranks']['n_pairs']
    # without --allow-tcp a run that cannot bring RCCL up on every rank exits non-zero (one GPU: two ranks on one device)
    from nuradiomc_amd import _lib
    if _lib.load().nrhip_device_count() == 1:
        bad = _run_bench(['--gpus', '2'] + common)
        assert bad.returncode != 0 and 'RCCL did not come up' in bad.stderr
