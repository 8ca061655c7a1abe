"""Multi-GPU composition of the hot path: one process per GPU, launched by torch.distributed.

* Batch scalar multiplication / batch verification are embarrassingly parallel: shard by index
  (`shard_range`), no data-path collective.
* MSM is a sum over points (src/abstract/curve.ts:863-905): every rank runs the single-GPU
  pipeline on its shard up to the grouped window sums; the exchange and the combine happen INSIDE the
  C ABI (`ncg_msm_sharded_dev`, csrc/comm.hip): one ncclAllGather of ~18 KB per rank on device buffers
  over xGMI, a one-wave kernel adding the per-rank arrays, one finish.  RCCL cannot reduce with a group
  law, hence all-gather + local add.  This module only boot-straps the communicator: rank 0 makes the
  RCCL unique id and torch.distributed broadcasts its 128 bytes.
* Without RCCL (backend "gloo": several ranks sharing one GPU, hosts without xGMI): the SAME native pipeline
  with a host-staged exchange - `ncg_msm_shard_local_dev` returns this rank's slot (window-plan header +
  grouped window sums, a fixed size per curve), torch.distributed all-gathers the slots, and
  `ncg_msm_shard_combine` runs the header check, the adding kernel and the finish that the RCCL path runs
  after its own all-gather.
* Strong scaling of ONE MSM: `msm_sharded_windows` - every rank holds all points (replicated array or a resident set
  per GPU) and runs a range of the WINDOWS (curve.ts:886-902: windows are independent until the final chain); the
  slots are concatenated instead of added.  Same two transports.
* `n_max` (the largest shard, which fixes the window plan on every rank) is found with one MAX all-reduce
  when the caller does not pass it, so ragged shards always agree on the plan.
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


_native_comm_failed = False


def init_comm(engine, device=None, single_ok=False):
    """Collective: give `engine` an RCCL communicator spanning the torch.distributed world (backend nccl).
    Returns True if the native multi-GPU path is active, False for single-rank / gloo dry runs (single_ok=True builds
    the communicator for a world of one rank too: the same id broadcast, ncclCommInitRank and all-gather calls, which is
    how the 1-GPU box exercises this path beside torch's own RCCL communicator - tests/test_gpu_multi.py).  If the
    communicator cannot be made on ANY rank (RCCL library not loadable, ncclCommInitRank error) every rank
    learns it through one all-reduce, the reason goes to stderr once, and the callers use the
    torch.distributed exchange instead - still device buffers over RCCL, only outside the C ABI."""
    global _native_comm_failed
    import torch
    dist = _dist()
    if dist is None or (dist.get_world_size() == 1 and not single_ok) or dist.get_backend() != "nccl" or _native_comm_failed:
        return False
    if engine.has_comm() and engine.comm_size() == dist.get_world_size():
        return True
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = torch.zeros(128, dtype=torch.uint8, device=device)
    err = None
    if rank == 0:
        try:
            uid = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).to(device)
        except Exception as e:  # noqa: BLE001 - reported below, on every rank
            err = e
    bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
    dist.broadcast(bad, 0)
    if int(bad.item()) == 0:
        dist.broadcast(uid, 0)
        try:
            engine.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
        except Exception as e:  # noqa: BLE001
            err = e
        bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if int(bad.item()) != 0:
        import sys
        _native_comm_failed = True
        engine.comm_destroy()   # (no-op without one) this rank's communicator may exist while another rank's does not
        print("[noble-gpu] rank %d: native RCCL communicator unavailable (%s); exchanging through torch.distributed"
              % (rank, err if err else "another rank failed"), file=sys.stderr, flush=True)
        return False
    return True


def all_gather_partials(partial_wire, is_inf, device=None):
    """(dry-run path) all-gather one (affine point bytes, infinity flag) per rank -> uint8 [world, PB+1]."""
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
    """(dry-run path) sum of the per-rank partial points by pairwise additions on the engine."""
    pb = POINT_BYTES[curve]
    pts = np.ascontiguousarray(gathered[:, :pb])
    while pts.shape[0] > 1:
        half = pts.shape[0] // 2
        summed, _ = engine.add_pairs_batch(curve, pts[:half], pts[half:2 * half])
        pts = np.concatenate([summed, pts[2 * half:]], axis=0)
    out = np.ascontiguousarray(pts[0])
    ident = np.zeros((pb,), np.uint8)
    if pb == 64 and curve == 1:          # Edwards identity is (0, 1)
        ident[32] = 1
    return out, bool((out == ident).all())


def _agree_n_max(n_local, device=None):
    """largest shard over all ranks (one small all-reduce)."""
    import torch
    dist = _dist()
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device if (device is not None and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def all_gather_slots(slot, device=None):
    """all-gather one fixed-size slot (uint8 array) per rank -> uint8 [world, slot_bytes] in rank order."""
    import torch
    dist = _dist()
    t = torch.from_numpy(slot)
    if device is not None and dist.get_backend() == "nccl":
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return torch.stack(parts).cpu().numpy()


def msm_sharded(engine, curve, n_local, d_points, d_scalars, stream=None, device=None, n_max=0):
    """MSM over the union of all ranks' shards.  `d_points` / `d_scalars` are this rank's device-resident
    shard (raw pointers).  Every rank returns the same (affine bytes, is_inf).  n_max = 0: the largest
    shard is agreed with one all-reduce (ranks may hold different counts)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return engine.msm_dev(curve, n_local, d_points, d_scalars, stream)
    if n_max == 0:
        n_max = _agree_n_max(n_local, device)
    if init_comm(engine, device):
        return engine.msm_sharded_dev(curve, n_local, d_points, d_scalars, stream, n_max)
    return _staged_exchange(engine, curve, n_max, device, stream,
                            lambda: engine.msm_shard_local_dev(curve, n_local, d_points, d_scalars, stream, n_max))


def _staged_exchange(engine, curve, n_plan, device, stream, local_phase):
    """Host-staged exchange: local phase -> all-gather of the fixed-size slots -> combine on every rank.  A rank whose
    local phase fails must still enter the all-gather (the others are already waiting in it): it posts a poisoned slot,
    which fails the header check of every rank's combine, and re-raises its own error afterwards.  A scalar outside the
    group order is not such a failure: the native local phase records it in the slot header and EVERY rank's combine
    raises 'invalid scalar' (the reference's validateMSMScalars fails the whole call, curve.ts:398-404)."""
    err = None
    try:
        slot = local_phase()
    except Exception as e:  # noqa: BLE001 - re-raised below, after the collective
        err = e
        slot = np.full((engine.msm_shard_slot_bytes(curve),), 0xFF, dtype=np.uint8)
    slots = all_gather_slots(slot, device)
    if err is not None:
        raise err
    return engine.msm_shard_combine(curve, n_plan, slots, stream)


def msm_sharded_windows(engine, curve, n, d_points, d_scalars, stream=None, device=None, resident=None):
    """ONE n-point MSM cut by WINDOWS over the ranks (strong scaling): every rank holds all n points - `d_points`, or
    `resident` (a ResidentPoints of this rank's engine holding the same set) - and all n scalars; rank r runs its range
    of the windows, the slots are concatenated (precomputed sets: added), every rank returns the same (bytes, is_inf).
    RCCL backend: inside the C ABI (ncg_msm_sharded_windows_dev); otherwise the host-staged exchange."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        if resident is not None:
            return resident.msm_dev(d_scalars, stream)
        return engine.msm_dev(curve, n, d_points, d_scalars, stream)
    if init_comm(engine, device):
        return engine.msm_sharded_windows_dev(curve, n, d_points, d_scalars, stream, resident)
    rank, world = dist.get_rank(), dist.get_world_size()
    c = resident.curve if resident is not None else curve
    n_all = len(resident) if resident is not None else n
    return _staged_exchange(engine, c, n_all, device, stream,
                            lambda: engine.msm_shard_windows_local_dev(curve, n, rank, world, d_points, d_scalars, stream, resident))


def msm_sharded_host(engine, curve, points_wire, scalars_wire, device=None):
    """Host-buffer variant: every rank holds the full arrays and takes its index shard (dry-run path:
    partial points exchanged by torch.distributed)."""
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    lo, hi = shard_range(points_wire.shape[0], rank, world)
    part, part_inf = engine.msm(curve, points_wire[lo:hi], scalars_wire[lo:hi])
    if world == 1:
        return part, part_inf
    return combine_partials(engine, curve, all_gather_partials(part, part_inf, device))
