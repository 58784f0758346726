"""Event sharding across ranks (one process per GPU) and the gather of triggered masks -- the torch.distributed twin of
nuradiomc_amd.comm (which binds RCCL directly and is what bench.py uses on GPUs): kept for hosts that already run a
torch.distributed process group and for the CPU tests of the sharding logic (gloo).

Events are independent, so rank r of W owns the contiguous index range shard_range(n, r, W); nothing is exchanged
during the compute.  `gather_triggered` is the only collective: an all-gather of the per-rank uint8 masks
(RCCL over xGMI with backend 'nccl' on GPUs, gloo in the CPU tests)."""
import numpy as np


from .comm import shard_range, shard_chunks  # noqa: F401  (the product path: nuradiomc_amd.comm.Comm over RCCL)


def gather_triggered(local_mask, n_events, dist=None, device=None):
    """all-gather variable-length uint8 masks into the full [n_events] mask (same on every rank)"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local_mask, np.uint8).copy()
    W, r = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_events, k, W)[1] - shard_range(n_events, k, W)[0] for k in range(W)]
    pad = max(sizes)
    t = torch.zeros(pad, dtype=torch.uint8, device=device)
    if isinstance(local_mask, torch.Tensor):
        t[:sizes[r]] = local_mask[:sizes[r]]
    else:
        t[:sizes[r]] = torch.from_numpy(np.ascontiguousarray(local_mask, dtype=np.uint8)).to(t.device)[:sizes[r]]
    out = torch.empty(W * pad, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(W, pad)
    return np.concatenate([out[k, :sizes[k]] for k in range(W)])
