"""world_size-2 gloo tests of the multi-rank MSM that run without a GPU.

The per-rank engine is the HOST TWIN of the native one (tests/hosttest.py -> csrc/hosttest.hip): the per-shard
window sums are computed naively with the kernels' group-law templates on the CPU, everything around them is the
code the GPU path runs - the window plan from n_max (csrc/msm_plan.hpp), the fixed-size slot with its plan header
(csrc/msm_shard.hpp), the header check, the order in which the shards' sums are added and the host finish
(csrc/msm_finish.hpp, bls_host64.hpp).  `noble_curves_amd.distributed.msm_sharded` drives it exactly as it drives
the native engine on a gloo backend: agree n_max, local phase, all-gather of the slots, combine on every rank.
Ragged shards (sizes straddling a power of two - the case ADVICE r02 flagged), an empty shard and a rank that
passes a wrong n_max are covered.  The NATIVE engine runs the same flow on the GPU box (tests/test_gpu_multi.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostTwinEngine:
    """Same method names / argument meaning as NativeEngine's sharded-MSM entry points; `d_points` / `d_scalars`
    are host addresses here (test infrastructure only)."""

    def __init__(self, shared=False):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hosttest
        self.ht = hosttest
        self.shared = shared

    def comm_size(self):
        return 1

    def msm_shard_local_dev(self, curve, n_local, d_points, d_scalars, stream=None, n_max=0):
        return self.ht.msm_shard_local(curve, n_local, n_max or n_local, d_points, d_scalars)

    def msm_shard_slot_bytes(self, curve):
        return self.ht.msm_shard_slot_bytes(curve)

    def msm_shard_windows_local_dev(self, curve, n, part, nparts, d_points, d_scalars, stream=None, resident=None):
        # `shared`: the twin of a rank whose resident set is PRECOMPUTED (its slots are added, not concatenated)
        return self.ht.msm_shard_windows_local(curve, n, part, nparts, d_points, d_scalars, shared=self.shared)

    def msm_shard_combine(self, curve, n_max, slots, stream=None):
        from noble_curves_amd._native import POINT_BYTES
        return self.ht.msm_shard_combine(curve, n_max, slots, POINT_BYTES[curve])


def _worker_windows(rank, world, port, curve_name, n, q, bad_at, c_forced=0, shared=False):
    """window-sharded mode: every rank holds ALL n points and scalars, rank r computes its range of the windows."""
    if c_forced:
        os.environ["NCG_MSM_C"] = str(c_forced)     # the library's public knob (csrc/knobs.hpp): the plan of a 2^20 / 2^18-point MSM
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
    from noble_curves_amd import _native
    from noble_curves_amd.distributed import msm_sharded_windows
    curve = getattr(_native, curve_name)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    from oracle.curves import makeRng
    rng = makeRng(0x57A6)
    ks = [rng.rndBelow(order - 1) + 1 for _ in range(n)]
    sc = [0 if i % 5 == 0 else (order - 1 if i % 7 == 0 else rng.rndBelow(order)) for i in range(n)]
    if bad_at is not None:
        sc[bad_at] = order          # outside the contract: EVERY rank must fail (validateMSMScalars, curve.ts:398-404)
    pts_w = points_to_wire(curve, [Pt.BASE.multiplyUnsafe(k) for k in ks])
    sc_w = scalars_to_wire(sc)
    eng = HostTwinEngine(shared=shared)
    exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order).toAffine()
    try:
        out, inf = msm_sharded_windows(eng, curve, n, pts_w.ctypes.data, sc_w.ctypes.data)
        q.put((rank, wire_to_affine(curve, out) == exp, inf, None))
    except ValueError as e:
        q.put((rank, False, False, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, curve_name, sizes, q, wrong_n_max, bad_at=None, c_forced=0):
    if c_forced:
        os.environ["NCG_MSM_C"] = str(c_forced)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
    from noble_curves_amd import _native
    from noble_curves_amd.distributed import msm_sharded
    curve = getattr(_native, curve_name)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    from oracle.curves import makeRng
    rng = makeRng(0xD157)
    n = sum(sizes)
    ks = [rng.rndBelow(order - 1) + 1 for _ in range(n)]
    sc = [0 if i % 5 == 0 else (order - 1 if i % 7 == 0 else rng.rndBelow(order)) for i in range(n)]
    if bad_at is not None:
        sc[bad_at] = order          # only the rank that owns this index sees it; every rank must raise
    lo = sum(sizes[:rank])
    hi = lo + sizes[rank]
    pts_w = points_to_wire(curve, [Pt.BASE.multiplyUnsafe(k) for k in ks[lo:hi]]) if hi > lo else np.zeros((0, 1), np.uint8)
    sc_w = scalars_to_wire(sc[lo:hi]) if hi > lo else np.zeros((0, 32), np.uint8)
    eng = HostTwinEngine()
    exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order).toAffine()
    try:
        n_max = 0
        if wrong_n_max and rank == 1:
            n_max = 1 << 12           # this rank plans other windows than rank 0
        elif wrong_n_max:
            n_max = max(sizes)
        out, inf = msm_sharded(eng, curve, hi - lo, pts_w.ctypes.data, sc_w.ctypes.data, n_max=n_max)
        q.put((rank, wire_to_affine(curve, out) == exp, inf, None))
    except ValueError as e:
        q.put((rank, False, False, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def _run_windows(curve_name, n, world, bad_at=None, c_forced=0, shared=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hosttest
    hosttest.lib()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_windows, args=(r, world, port, curve_name, n, q, bad_at, c_forced, shared)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    return res


def _run(curve_name, sizes, wrong_n_max=False, bad_at=None, c_forced=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hosttest
    hosttest.lib()           # (re)build the twin once, here, not inside the workers' time limit
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = len(sizes)
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve_name, sizes, q, wrong_n_max, bad_at, c_forced)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    return res


def test_shard_range_covers_everything():
    from noble_curves_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 - a0 - (b1 - b0) in (0, 1)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("curve_name,sizes", [
    ("BLS12_381_G1", (33, 31)),       # ragged, straddling 2^5: the plans would differ without the agreed n_max
    ("BLS12_381_G2", (9, 4)),
    ("SECP256K1", (16, 0)),           # an empty shard contributes identities
    ("ED25519", (5, 12)),
])
def test_msm_sharded_two_ranks_gloo_native_combine_twin(curve_name, sizes):
    res = _run(curve_name, sizes)
    assert all(r[1] for r in res), res
    assert not any(r[2] for r in res)


@pytest.mark.timeout(300)
def test_msm_sharded_rejects_disagreeing_plans():
    """rank 1 passes another n_max: every rank must get the header-check error, not a wrong sum (and, on the RCCL
    path, not a collective with mismatched counts: the slot size does not depend on the plan)."""
    res = _run("BLS12_381_G1", (20, 20), wrong_n_max=True)
    assert all((not r[1]) and r[3] and "all ranks must pass the same curve and n_max" in r[3] for r in res), res


@pytest.mark.timeout(300)
@pytest.mark.parametrize("curve_name,n,world", [
    ("BLS12_381_G1", 40, 2),          # c = 2: 128 windows, 64 per rank
    ("SECP256K1", 24, 3),             # uneven ranges
    ("ED25519", 11, 2),
])
def test_msm_window_sharded_ranks_gloo_native_assembly_twin(curve_name, n, world):
    """Strong-scaling mode: rank r computes windows [w0, w0 + cnt) of ALL points, the slots are concatenated by the
    shipped assembly (csrc/msm_shard.hpp) and every rank runs the Horner finish (curve.ts:886-902)."""
    res = _run_windows(curve_name, n, world)
    assert all(r[1] for r in res), res
    assert not any(r[2] for r in res)


@pytest.mark.timeout(300)
def test_out_of_range_scalar_fails_every_rank_not_only_the_owner():
    """the verdict of validateMSMScalars travels in the slot header, so no rank leaves the exchange early
    and all of them raise - window mode (all ranks see the scalar) and point mode (only the owner of the slice does)."""
    res = _run_windows("BLS12_381_G1", 20, 2, bad_at=13)
    assert all((not r[1]) and r[3] and "invalid scalar at index 13" in r[3] for r in res), res


# ---- N = 8: the configuration `bench.py --gpus 8` runs -------------------------------------------------------
# The plans are the ones the 2^20-point G1 MSM (c = 16: 16 windows) and the 2^18-point G2 MSM (c = 13: 20 windows) use on the
# device, forced through the library's public NCG_MSM_C knob so that a few dozen points exercise the window ranges 8 ranks get.

def test_window_ranges_of_the_bench_plans_over_eight_ranks():
    """csrc/msm_shard.hpp msm_shard_window_range through the twin's slot headers: G1 16 windows -> 2 per rank, G2 20 windows ->
    3,3,3,3,2,2,2,2; 24 parts of a 16-window plan: the last 8 parts get none."""
    import struct
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hosttest
    from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
    from noble_curves_amd import _native
    from noble_curves_amd._native import POINT_BYTES
    old = os.environ.get("NCG_MSM_C")
    try:
        for curve, c, nparts, want in ((_native.BLS12_381_G1, 16, 8, [2] * 8), (_native.BLS12_381_G2, 13, 8, [3, 3, 3, 3, 2, 2, 2, 2]),
                                       (_native.BLS12_381_G1, 16, 24, [1] * 16 + [0] * 8), (_native.SECP256K1, 16, 8, None),
                                       (_native.ED25519, 14, 8, None)):
            os.environ["NCG_MSM_C"] = str(c)
            Pt = ORACLE_CURVE[curve]
            order = Pt.Fn.ORDER
            n = 6
            ks = [3 + 7 * i for i in range(n)]
            sc = [order - 1, 0, 1, (1 << 200) + 12345, order >> 1, 0xFFFF_FFFF_FFFF]
            pts_w = points_to_wire(curve, [Pt.BASE.multiplyUnsafe(k) for k in ks])
            sc_w = scalars_to_wire(sc)
            plan = hosttest.msm_plan(curve, n)
            assert plan["c"] == c
            slots = [hosttest.msm_shard_windows_local(curve, n, r, nparts, pts_w.ctypes.data, sc_w.ctypes.data) for r in range(nparts)]
            hdr = [struct.unpack("<8I", bytes(s[:32])) for s in slots]
            got = [h[5] for h in hdr]                       # FinHeader.wcnt
            if want is not None:
                assert plan["nwin"] == sum(want) and got == want, (curve, got)
            assert sum(got) == plan["nwin"] and max(got) - min(got) <= 1
            w = 0
            for h in hdr:                                   # the ranges tile [0, nwin) in rank order
                assert h[4] == w and h[0] == c and h[1] == plan["nwin"]
                w += h[5]
            out, inf = hosttest.msm_shard_combine(curve, n, __import__("numpy").stack(slots), POINT_BYTES[curve])
            exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order).toAffine()
            assert not inf and wire_to_affine(curve, out) == exp, curve
            # a precomputed set (shared buckets): one grouped-sum array per part, added
            slots = [hosttest.msm_shard_windows_local(curve, n, r, nparts, pts_w.ctypes.data, sc_w.ctypes.data, shared=True) for r in range(nparts)]
            assert all(struct.unpack("<8I", bytes(s[:32]))[6] == 2 for s in slots)      # SHARD_WINDOWS_SHARED
            out, inf = hosttest.msm_shard_combine(curve, n, __import__("numpy").stack(slots), POINT_BYTES[curve])
            assert not inf and wire_to_affine(curve, out) == exp, curve
            # a missing part is an error, not a wrong sum
            with pytest.raises(ValueError, match="all ranks must pass the same curve and n_max"):
                hosttest.msm_shard_combine(curve, n, __import__("numpy").stack(slots[:-1] if got[-1] else slots[:got.index(0)][:-1]), POINT_BYTES[curve])
    finally:
        if old is None:
            os.environ.pop("NCG_MSM_C", None)
        else:
            os.environ["NCG_MSM_C"] = old


@pytest.mark.timeout(600)
@pytest.mark.parametrize("curve_name,n,c_forced,shared", [
    ("BLS12_381_G1", 24, 16, False),      # the 2^20 plan: 16 windows, 2 per rank, concatenated
    ("BLS12_381_G2", 10, 13, False),      # the 2^18 plan: 20 windows, 3,3,3,3,2,2,2,2
    ("BLS12_381_G1", 12, 16, True),       # precomputed set: every rank's slot is ONE grouped-sum array, the 8 slots are added
])
def test_msm_window_sharded_eight_ranks_gloo(curve_name, n, c_forced, shared):
    res = _run_windows(curve_name, n, 8, c_forced=c_forced, shared=shared)
    assert all(r[1] for r in res), res
    assert not any(r[2] for r in res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("curve_name,sizes,c_forced", [
    ("BLS12_381_G1", (5, 0, 9, 1, 0, 8, 3, 7), 16),     # ragged shards incl. two empty ones, the 2^20 plan
    ("BLS12_381_G2", (2, 3, 0, 1, 4, 1, 0, 2), 13),
    ("SECP256K1", (3, 3, 3, 3, 3, 3, 3, 2), 0),
    ("ED25519", (0, 0, 0, 9, 0, 0, 0, 0), 0),            # seven ranks contribute identities only
])
def test_msm_point_sharded_eight_ranks_gloo(curve_name, sizes, c_forced):
    res = _run(curve_name, sizes, c_forced=c_forced)
    assert all(r[1] for r in res), res
    assert not any(r[2] for r in res)


@pytest.mark.timeout(600)
def test_out_of_range_scalar_fails_all_eight_ranks_in_both_modes():
    res = _run_windows("BLS12_381_G1", 20, 8, bad_at=13, c_forced=16)
    assert all((not r[1]) and r[3] and "invalid scalar at index 13" in r[3] for r in res), res
    # point mode: index 17 of the whole array is index 2 of rank 5's slice (sizes 3 each); only that rank sees it
    res = _run("BLS12_381_G1", (3,) * 8, bad_at=17, c_forced=16)
    assert all((not r[1]) and r[3] and "invalid scalar at index 2 of shard 5" in r[3] for r in res), res
