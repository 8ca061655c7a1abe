"""Multi-GPU composition of the hot path: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU unit tests).

* Batch scalar multiplication / batch verification are embarrassingly parallel: shard by index
  (`shard_range`), no data-path collective.
* MSM is a sum of independent terms: every rank runs the full single-GPU pipeline
  on its shard of (points, scalars), then ONE all-gather of the per-rank partial sums
  (one affine point + infinity flag: 97-193 bytes per rank) and a combine step - the
  partials are summed by the same MSM path with unit scalars, so the combine also runs in the
  HIP kernels.  RCCL cannot reduce with a group law, hence all-gather + local add rather than
  all-reduce.  The exchange is latency-bound (KBs over xGMI); no bucket-sized traffic moves.
"""
import numpy as np

from ._native import POINT_BYTES


def shard_range(n, rank, world):
    """Contiguous [lo, hi) shard of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def all_gather_partials(partial_wire, is_inf, device=None):
    """All-gather one (affine point bytes, infinity flag) per rank -> uint8 array [world, PB+1]."""
    import torch
    dist = _dist()
    pb = partial_wire.shape[0]
    mine = np.zeros((pb + 1,), dtype=np.uint8)
    mine[:pb] = partial_wire
    mine[pb] = 1 if is_inf else 0
    if dist is None or dist.get_world_size() == 1:
        return mine.reshape(1, -1)
    t = torch.from_numpy(mine)
    if device is not None and dist.get_backend() == "nccl":
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return torch.stack(parts).cpu().numpy()


def combine_partials(engine, curve, gathered):
    """Sum of the per-rank partial points, through the engine's MSM with unit scalars."""
    pb = POINT_BYTES[curve]
    pts = np.ascontiguousarray(gathered[:, :pb])
    ones = np.zeros((pts.shape[0], 32), dtype=np.uint8)
    ones[:, 0] = 1
    return engine.msm(curve, pts, ones)


def msm_sharded(engine, curve, n_local, d_points, d_scalars, stream=None, device=None):
    """MSM over the union of all ranks' shards.  `d_points` / `d_scalars` are this rank's
    device-resident shard (raw pointers).  Every rank returns the same (affine bytes, is_inf)."""
    part, part_inf = engine.msm_dev(curve, n_local, d_points, d_scalars, stream)
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return part, part_inf
    gathered = all_gather_partials(part, part_inf, device)
    return combine_partials(engine, curve, gathered)


def msm_sharded_host(engine, curve, points_wire, scalars_wire, device=None):
    """Host-buffer variant: every rank holds the full arrays and takes its index shard."""
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    lo, hi = shard_range(points_wire.shape[0], rank, world)
    part, part_inf = engine.msm(curve, points_wire[lo:hi], scalars_wire[lo:hi])
    if world == 1:
        return part, part_inf
    gathered = all_gather_partials(part, part_inf, device)
    return combine_partials(engine, curve, gathered)
