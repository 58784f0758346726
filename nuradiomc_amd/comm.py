"""Multi-GPU plumbing without PyTorch: one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the
launcher's environment), RCCL bound through the C ABI (nrhip_comm_*, include/nrhip.h).

The reference scales out as independent processes over a split event list (NuRadioMC/utilities/runner.py:9-15); here the
split is `shard_range` and the merge of the outputs is ONE all-gather of the triggered masks over xGMI.

The only thing that has to travel between the processes on the host is RCCL's 128-byte communicator id: rank 0 serves it on
a TCP socket (MASTER_ADDR, NRHIP_COMM_PORT or MASTER_PORT + 1000), the other ranks fetch it.  Standard library only.
"""
import ctypes
import os
import socket
import time
import numpy as np
from . import _lib as L

ID_BYTES = 128

L._OPTIONAL.update({
    'nrhip_comm_get_unique_id': (ctypes.c_int, [ctypes.c_void_p]),
    'nrhip_comm_create': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, L.c_void_pp]),
    'nrhip_comm_destroy': (None, [ctypes.c_void_p]),
    'nrhip_comm_barrier': (ctypes.c_int, [ctypes.c_void_p]),
    'nrhip_comm_allgather_u8': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'nrhip_comm_allreduce_i64_sum': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    'nrhip_comm_allreduce_f64_max': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    'nrhip_mask_or': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    'nrhip_cull_groups': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, L.c_double_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'nrhip_select_groups': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'nrhip_gather_groups': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 20),
    'nrhip_index_to_i64': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'nrhip_gather_i64': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'nrhip_memset': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64]),
    'nrhip_mask_scatter_or': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
})


def shard_range(n_events, rank, world_size):
    """contiguous, balanced: the first n % W ranks get one extra event"""
    base, extra = divmod(int(n_events), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_chunks(n_events, rank, world_size, chunk):
    """Chunked round-robin assignment for SORTED event lists (energy- or position-ordered inputs would load contiguous shards
    unequally): chunk c of `chunk` events goes to rank c % W.  Returns the index array of this rank's events (ascending)."""
    n_events, chunk = int(n_events), max(1, int(chunk))
    idx = np.arange(n_events)
    return idx[(idx // chunk) % world_size == rank]


def env_rank():
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run, mpirun-style wrappers)"""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def exchange_bytes(payload, rank, world_size, addr=None, port=None, timeout=300.):
    """rank 0 hands `payload` (bytes) to every other rank over TCP; returns it on every rank."""
    if world_size == 1:
        return payload
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(port or os.environ.get('NRHIP_COMM_PORT', 0) or int(os.environ.get('MASTER_PORT', 29500)) + 1000)
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world_size)
        srv.settimeout(timeout)
        try:
            for _ in range(world_size - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(len(payload).to_bytes(4, 'little') + payload)
        finally:
            srv.close()
        return payload
    t_end = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.) as c:
                c.settimeout(timeout)
                buf = b''
                while len(buf) < 4:
                    d = c.recv(4 - len(buf))
                    if not d:
                        raise ConnectionError("peer closed")
                    buf += d
                n = int.from_bytes(buf, 'little')
                out = b''
                while len(out) < n:
                    d = c.recv(n - len(out))
                    if not d:
                        raise ConnectionError("peer closed")
                    out += d
                return out
        except (ConnectionRefusedError, ConnectionError, socket.timeout, OSError):
            if time.time() > t_end:
                raise
            time.sleep(0.05)


class _Star:
    """Persistent TCP star through rank 0 (a few bytes per call): the rendezvous of the communicator id, the vote on whether
    RCCL came up on every rank, and the host-side stand-in for the scalar collectives when it did not."""

    def __init__(self, rank, world_size, addr=None, port=None, timeout=300.):
        self.rank, self.world_size = rank, world_size
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port or os.environ.get('NRHIP_COMM_PORT', 0) or int(os.environ.get('MASTER_PORT', 29500)) + 1000)
        self.peers = {}
        self.sock = None
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world_size)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < world_size - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    r = int.from_bytes(self._recv(conn, 4), 'little')
                    self.peers[r] = conn
            finally:
                srv.close()
        else:
            t_end = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.)
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.sendall(int(rank).to_bytes(4, 'little'))
                    self.sock = c
                    break
                except (ConnectionRefusedError, ConnectionError, socket.timeout, OSError):
                    if time.time() > t_end:
                        raise
                    time.sleep(0.05)

    @staticmethod
    def _recv(c, n):
        out = b''
        while len(out) < n:
            d = c.recv(n - len(out))
            if not d:
                raise ConnectionError("peer closed")
            out += d
        return out

    def _send_msg(self, c, payload):
        c.sendall(len(payload).to_bytes(8, 'little') + payload)

    def _recv_msg(self, c):
        return self._recv(c, int.from_bytes(self._recv(c, 8), 'little'))

    def gather(self, payload):
        """rank 0: list of every rank's bytes (by rank); other ranks: None"""
        if self.rank == 0:
            out = [payload] + [None] * (self.world_size - 1)
            for r, c in self.peers.items():
                out[r] = self._recv_msg(c)
            return out
        self._send_msg(self.sock, payload)
        return None

    def broadcast(self, payload):
        if self.rank == 0:
            for c in self.peers.values():
                self._send_msg(c, payload)
            return payload
        return self._recv_msg(self.sock)

    def allgather(self, payload):
        parts = self.gather(payload)
        if self.rank == 0:
            blob = b''.join(len(q).to_bytes(8, 'little') + q for q in parts)
        else:
            blob = b''
        blob = self.broadcast(blob)
        out, o = [], 0
        while o < len(blob):
            n = int.from_bytes(blob[o:o + 8], 'little')
            out.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return out

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


class CommError(RuntimeError):
    pass


class Comm:
    """RCCL communicator of one process (one GPU) over nrhip_comm_*; world_size 1 needs no RCCL and every call is local.

    With more than one rank the processes first meet on a TCP star through rank 0 (communicator id, then a vote): only if RCCL
    loads on EVERY rank is the communicator created, and only if it came up on every rank is it used (`mode == 'rccl'`).
    Otherwise every rank raises CommError (a run that was asked for RCCL must not print a line that never touched xGMI) -- unless
    `allow_tcp=True` / NRHIP_ALLOW_TCP=1: then (`mode == 'tcp'`, said loudly on stderr and in bench.py's JSON) the few scalars
    and the masks travel over the star, so that a sharded run on a node without a working RCCL completes instead of hanging in
    a half-built communicator.  `backend='tcp'` asks for the star directly (hosts without RCCL, CPU tests of the sharding
    logic)."""

    def __init__(self, ctx, rank=None, world_size=None, addr=None, port=None, force_rccl=False, backend=None, allow_tcp=None):
        r, _, w = env_rank()
        self.ctx = ctx
        self.rank = r if rank is None else int(rank)
        self.world_size = w if world_size is None else int(world_size)
        self._lib = L.load() if ctx is not None else None
        self._h = None
        self._star = None
        self.mode = 'local'
        backend = backend or os.environ.get('NRHIP_COMM_BACKEND', 'rccl')
        if allow_tcp is None:
            allow_tcp = os.environ.get('NRHIP_ALLOW_TCP', '0') not in ('', '0')
        if self.world_size == 1 and force_rccl:   # a one-rank communicator (tests of the binding on a single GPU)
            uid = (ctypes.c_uint8 * ID_BYTES)()
            L.check(self._lib.nrhip_comm_get_unique_id(uid))
            h = ctypes.c_void_p()
            L.check(self._lib.nrhip_comm_create(ctx._h, uid, self.rank, self.world_size, ctypes.byref(h)))
            self._h = h
            self.mode = 'rccl'
        elif self.world_size > 1:
            self._star = _Star(self.rank, self.world_size, addr, port)
            self.mode = 'tcp'
            uid = (ctypes.c_uint8 * ID_BYTES)()
            ok = 0
            if backend == 'rccl' and self._lib is not None:
                ok = int(self._lib.nrhip_comm_get_unique_id(uid) == 0)   # loads RCCL: a local call on every rank
            votes = self._star.allgather(bytes([ok]) + (bytes(uid) if self.rank == 0 else b''))
            if all(v[0] == 1 for v in votes):
                uid = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(votes[0][1:1 + ID_BYTES])
                h = ctypes.c_void_p()
                up = int(self._lib.nrhip_comm_create(ctx._h, uid, self.rank, self.world_size, ctypes.byref(h)) == 0)
                if all(v[0] == 1 for v in self._star.allgather(bytes([up]))):
                    self._h = h
                    self.mode = 'rccl'
                elif up:
                    self._lib.nrhip_comm_destroy(h)
            if self.mode == 'tcp' and backend == 'rccl':
                why = self._lib.nrhip_last_error().decode() if self._lib is not None else 'no device context'
                if not allow_tcp:   # every rank takes this branch (the votes are the same everywhere): all exit non-zero
                    self.close()
                    raise CommError("nuradiomc_amd.comm: RCCL did not come up on every rank (%s); pass allow_tcp=True / "
                                    "NRHIP_ALLOW_TCP=1 (bench.py --allow-tcp) to run the collectives over the TCP star instead" % why)
                if self.rank == 0:
                    import sys
                    print("nuradiomc_amd.comm: RCCL did not come up on every rank -- collectives go over the TCP star (%s)" % why,
                          file=sys.stderr)

    def barrier(self):
        if self.ctx is not None and self._h is None:
            self.ctx.synchronize()
        if self._h is not None:
            L.check(self._lib.nrhip_comm_barrier(self._h))
        elif self._star is not None:
            self._star.allgather(b'')

    def allgather_masks(self, d_local, n_local, n_total):
        """All-gather the per-rank DEVICE uint8 masks of a list of n_total events sharded with shard_range: returns the full host
        mask [n_total] (same on every rank).  The one collective of the path (padded to the largest shard).  Lists dealt with
        shard_chunks (round-robin chunks) are NOT contiguous: gather those per chunk or scatter by index on the host."""
        W = self.world_size
        sizes = [shard_range(n_total, k, W)[1] - shard_range(n_total, k, W)[0] for k in range(W)]
        # the ranks agree on the shard sizes BEFORE the collective: a mismatch on one rank alone would leave the others waiting in
        # the all-gather until its timeout, so every rank learns every n_local (over the star, 8 bytes) and all raise together
        held = [int(n_local)]
        if self._star is not None:
            held = [int.from_bytes(q, 'little') for q in self._star.allgather(int(n_local).to_bytes(8, 'little'))]
        bad = [k for k in range(W) if held[k] != sizes[k]]   # (the star exists whenever W > 1: len(held) == W)
        if bad:
            raise ValueError("allgather_masks: rank %d holds %d events, shard_range(%d, %d, %d) has %d -- the gather is defined for "
                             "contiguous shard_range shards only" % (bad[0], held[bad[0]], n_total, bad[0], W, sizes[bad[0]]))
        if self._h is None:
            if isinstance(d_local, np.ndarray):   # a host mask (tcp backend without a device context)
                out = np.ascontiguousarray(d_local[:n_local], np.uint8)
            else:
                out = np.zeros(n_local, np.uint8)
                self.ctx.to_host(out, d_local)
            if self._star is None:
                return out
            return np.concatenate([np.frombuffer(q, np.uint8) for q in self._star.allgather(out.tobytes())])
        pad = max(sizes)
        d_send = self.ctx.malloc(max(pad, 1))
        d_recv = self.ctx.malloc(max(W * pad, 1))
        try:
            L.check(self._lib.nrhip_mask_or(self.ctx._h, n_local, ctypes.c_void_p(d_send), ctypes.c_void_p(d_local), 1))
            L.check(self._lib.nrhip_comm_allgather_u8(self._h, ctypes.c_void_p(d_send), ctypes.c_void_p(d_recv), pad))
            host = np.zeros(W * pad, np.uint8)
            self.ctx.to_host(host, d_recv)
        finally:
            self.ctx.free(d_send)
            self.ctx.free(d_recv)
        host = host.reshape(W, pad)
        return np.concatenate([host[k, :sizes[k]] for k in range(W)])

    def _host_reduce(self, v, op):
        parts = [np.frombuffer(q, v.dtype) for q in self._star.allgather(v.tobytes())]
        return op(np.stack(parts), axis=0)

    def allreduce_sum(self, values):
        """element-wise sum of a small int64 vector over the ranks"""
        v = np.ascontiguousarray(values, np.int64)
        if self._h is None:
            return v.copy() if self._star is None else self._host_reduce(v, np.sum)
        d = self.ctx.to_device(v)
        try:
            L.check(self._lib.nrhip_comm_allreduce_i64_sum(self._h, ctypes.c_void_p(d), len(v)))
            out = np.zeros_like(v)
            self.ctx.to_host(out, d)
        finally:
            self.ctx.free(d)
        return out

    def allreduce_max(self, values):
        v = np.ascontiguousarray(values, np.float64)
        if self._h is None:
            return v.copy() if self._star is None else self._host_reduce(v, np.max)
        d = self.ctx.to_device(v)
        try:
            L.check(self._lib.nrhip_comm_allreduce_f64_max(self._h, ctypes.c_void_p(d), len(v)))
            out = np.zeros_like(v)
            self.ctx.to_host(out, d)
        finally:
            self.ctx.free(d)
        return out

    def close(self):
        if self._h is not None:
            self._lib.nrhip_comm_destroy(self._h)
            self._h = None
        if self._star is not None:
            self._star.close()
            self._star = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
