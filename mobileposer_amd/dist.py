"""Multi-GPU: one process per GPU, independent sequences sharded across ranks.

The reference has no distributed code at all; the path shards over independent units
(sequences / streams never interact), so the only collective is ONE broadcast of the flat weight blob
(26.7 MB fp32) from rank 0 over RCCL/xGMI at start-up.  Nothing is exchanged per step.
"""
import numpy as np
import torch

from .manifest import n_params
from .model_utils import state_dict_to_blob


def shard_range(n_items, rank, world):
    """Contiguous split of ``n_items`` sequences: rank r gets [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(state_dict, device, src=0):
    """Rank ``src`` passes its state dict (others pass None); every rank gets the blob as a device tensor
    (cuda -> backend nccl = RCCL; cpu -> gloo in the CPU tests)."""
    import torch.distributed as dist
    device = torch.device(device)
    n = n_params()
    if dist.get_rank() == src:
        blob = torch.from_numpy(state_dict_to_blob(state_dict)).to(device)
    else:
        blob = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(blob, src=src)
    return blob


def broadcast_model(state_dict, smpl, device, src=0):
    """The whole start-up exchange of SURVEY.md 8(e) as ONE broadcast: the flat weight blob followed by the SMPL constants
    the path uses (24 parent indices + 24 x 3 joint positions, as fp32).  Rank ``src`` passes its state dict and SMPL dict
    (keys 'J', 'kintree_table'), the others pass None.  Returns (weight blob on ``device``, smpl dict for
    ParametricModel(data=...))."""
    import torch.distributed as dist
    device = torch.device(device)
    n = n_params()
    if dist.get_rank() == src:
        parent = np.asarray(smpl["kintree_table"])[0].astype(np.int64)
        parent[0] = -1
        tail = np.concatenate((parent.astype(np.float32), np.asarray(smpl["J"], dtype=np.float32).reshape(-1)))
        buf = torch.from_numpy(np.concatenate((state_dict_to_blob(state_dict), tail))).to(device)
    else:
        buf = torch.empty(n + 24 + 72, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    tail = buf[n:].cpu().numpy()
    parent = np.rint(tail[:24]).astype(np.int64)
    out = {"J": tail[24:].reshape(24, 3).copy(), "kintree_table": np.stack((parent, np.arange(24)))}
    if dist.get_rank() == src:                       # the source keeps its mesh data (evaluator), the others have none
        out = dict(smpl, **out)
    return buf[:n], out


def gather_counts(local_frames, local_seconds, device):
    """All-gather per-rank (frames, seconds) for the scaling report; returns a [world, 2] float64 array."""
    import torch.distributed as dist
    t = torch.tensor([float(local_frames), float(local_seconds)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])
