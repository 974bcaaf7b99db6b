"""Multi-GPU partitioning of a ray fan: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

Rays are independent (the reference loops ``for angle in angles``, simulation.py:89-94), so a fan shards
with no data-path exchange until the heat map is assembled.  Exactly two collectives exist on this path:

  C1  all-reduce(min/max) of the 4 plot-range doubles  (RayPack2D.add range fold, ray.py:466-472) -- fixes
      the global heat-map shape before anything is binned;
  C2  all-reduce(sum) of the nx*nz float64 heat map      (``heatmap_full +=`` across rays, simulation.py:131).

Both are latency-bound over xGMI (4 doubles; ~0.6 MB at 50 m x 25 m cells), one RCCL call each.
The functions work on any backend (the CPU tests run them under gloo with world_size 2).
"""

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_reduce(t, op):
    """dist.all_reduce that also works for device tensors under a CPU-only backend (gloo: staged through the host)."""
    dist = _dist()
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def active():
    """True inside an initialised process group of more than one rank (RT2D_FORCE_COLLECTIVES=1: of any size, so that
    the collectives can be rehearsed over RCCL on a single GPU)."""
    import os
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('RT2D_FORCE_COLLECTIVES') == '1'


def shard_angles(angles, rank=None, world_size=None, mode='interleave'):
    """This rank's share of a fan.

    'interleave' (default): ray j goes to rank j % world -- every rank sees the whole angular range, so
    steep (short) and shallow (long) rays are spread evenly.  'block': contiguous blocks, which keeps
    neighbouring lanes of a wavefront closest in length.  Returns (angles, global indices)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    angles = np.asarray(angles, dtype=np.float64)
    idx = np.arange(angles.shape[0])
    if mode == 'interleave':
        sel = idx[rank::world_size]
    elif mode == 'block':
        per = (angles.shape[0] + world_size - 1) // world_size
        sel = idx[rank * per:(rank + 1) * per]
    else:
        raise ValueError(mode)
    return angles[sel], sel


def allreduce_plot_range(range_min_plot, range_max_plot, device=None):
    """C1: global plot range.  Inputs are the local RayPack2D.range_min_plot / range_max_plot (2 doubles each)."""
    import torch
    dist = _dist()
    lo = np.asarray(range_min_plot, dtype=np.float64)
    hi = np.asarray(range_max_plot, dtype=np.float64)
    if not active():
        return lo.copy(), hi.copy()
    # one collective: max over (-min, max)
    t = torch.tensor(np.concatenate((-lo, hi)), dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    t = t.cpu().numpy()
    return -t[:2], t[2:]


def heatmap_shape(range_min_plot, range_max_plot, res):
    """ceil(size_plot / res) + 1 (simulation.py:117)."""
    size_plot = np.asarray(range_max_plot) - np.asarray(range_min_plot)
    return (np.ceil(size_plot / np.asarray(res)).astype(int) + 1).astype(np.int64)


def allreduce_heatmap(H):
    """C2: sum the per-rank heat maps in place (a torch tensor on this rank's device)."""
    dist = _dist()
    if active():
        _all_reduce(H, dist.ReduceOp.SUM)
    return H


def allgather_topk(dist_local, angles_local, k, index_local=None):
    """C3 (sweep + refine): the rays whose distance is among the k smallest DISTINCT distances of the whole fan -- the
    reference's selection rule (simulation.py:241-249: it keys its rays by distance and keeps the k smallest keys, so rays
    at equal distance stay together) -- sorted by distance, equal distances in fan order.  Each rank contributes the rays
    within ITS k smallest distinct distances (a superset of its share of the answer; NaN = never reaches the target
    abscissa), the candidates are all-gathered (padded to the largest contribution) and the rule is applied again on every
    rank.  index_local: the rays' positions in the whole fan (shard_angles' second result), used to order ties; None = the
    local positions.  Returns (distances, angles)."""
    import torch
    dist = _dist()
    d = torch.as_tensor(dist_local, dtype=torch.float64)
    a = torch.as_tensor(angles_local, dtype=torch.float64, device=d.device)
    idx = torch.arange(d.numel(), device=d.device) if index_local is None else torch.as_tensor(index_local, device=d.device)
    idx = idx.to(torch.float64)                                   # exact below 2^53; travels in the same tensor

    def select(dd, aa, ii):
        valid = ~torch.isnan(dd)
        if not bool(valid.any()):
            return dd[:0], aa[:0], ii[:0]
        uniq = torch.unique(dd[valid])                            # ascending
        kth = uniq[min(k, uniq.numel()) - 1]
        sel = torch.nonzero(valid & (dd <= kth)).flatten()
        order = torch.argsort(ii[sel], stable=True)               # fan order first ...
        sel = sel[order]
        sel = sel[torch.argsort(dd[sel], stable=True)]            # ... then by distance, stable: ties stay in fan order
        return dd[sel], aa[sel], ii[sel]

    d, a, idx = select(d, a, idx)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = d.device
        count = torch.tensor([d.numel()], dtype=torch.int64, device=dev)
        _all_reduce(count, dist.ReduceOp.MAX)
        m = max(1, int(count.item()))          # (no rank has a valid distance: still a non-empty gather; the NaN padding is filtered by select)
        cand = torch.full((3, m), float('nan'), dtype=torch.float64, device=dev)
        cand[0, :d.numel()], cand[1, :d.numel()], cand[2, :d.numel()] = d, a, idx
        if cand.is_cuda and dist.get_backend() == 'gloo':
            cand = cand.cpu()
        parts = [torch.empty_like(cand) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, cand)
        cand = torch.cat(parts, dim=1).to(dev)
        d, a, idx = select(cand[0], cand[1], cand[2])
    return d, a
