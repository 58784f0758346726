"""torch.distributed twin of nuradiomc_amd.comm.Comm.allgather_masks -- TEST infrastructure (the product binds RCCL directly and
never imports torch): lets the CPU suite run the N > 1 gather of the sharded trigger masks on gloo, world size 2."""
import numpy as np
from nuradiomc_amd.comm import shard_range


def gather_triggered(local_mask, n_events, dist=None, device=None):
    """all-gather variable-length uint8 masks into the full [n_events] mask (same on every rank)"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local_mask, np.uint8).copy()
    W, r = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_events, k, W)[1] - shard_range(n_events, k, W)[0] for k in range(W)]
    pad = max(sizes)
    t = torch.zeros(pad, dtype=torch.uint8, device=device)
    t[:sizes[r]] = torch.from_numpy(np.ascontiguousarray(local_mask, dtype=np.uint8)).to(t.device)[:sizes[r]]
    out = torch.empty(W * pad, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(W, pad)
    return np.concatenate([out[k, :sizes[k]] for k in range(W)])
