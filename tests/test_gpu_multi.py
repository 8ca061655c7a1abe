"""Multi-GPU MSM behind the C ABI (include/ncg.h "multi-GPU MSM", csrc/comm.hip) on the 1-GPU box:

* the per-shard phase + multi-GPU combine kernel + finish with 2..8 shards on one GPU (ncg_msm_split_dev:
  everything of ncg_msm_sharded_dev except the all-gather), all four curves, ragged and empty shards;
* the RCCL communicator path with one rank (ncg_comm_unique_id / ncg_comm_init / ncg_msm_sharded_dev);
* the single-process device-set context with one device (ncg_multi_init / ncg_msm_multi);
* the host-staged exchange (ncg_msm_shard_local_dev / ncg_msm_shard_combine) in one process: G = 2, 3, 8 slots of
  ragged and empty shards through the NATIVE header check, adding kernel and finish; disagreeing plans rejected;
* two ranks sharing the GPU over gloo with the REAL engine driving that same exchange (distributed.msm_sharded),
  ragged shards straddling a power of two, n_max agreed by all-reduce.
Results are compared with the oracle (pippenger, src/abstract/curve.ts:863-905) and the single-GPU MSM."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, POINT_BYTES, SECP256K1, Engine, MultiEngine
from oracle import curve as OC
from oracle.curves import makeRng

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(curve, n, seed):
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    rng = makeRng(seed)
    pts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(order - 1) + 1) for _ in range(n)]
    sc = [0 if i % 7 == 3 else rng.rndBelow(order) for i in range(n)]
    if n > 4:
        pts[2] = Pt.ZERO
        pts[4] = pts[1]
        sc[4] = (order - sc[1]) % order          # cancels entry 1
    exp = OC.pippenger(Pt, pts, sc)
    return points_to_wire(curve, pts), scalars_to_wire(sc), exp


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


@pytest.mark.parametrize("curve", [SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2])
def test_split_pipeline_matches_oracle(curve):
    eng = get_engine()
    n = 150 if curve != BLS12_381_G2 else 70
    pw, sw, exp = _case(curve, n, 0x5A4D + curve)
    dp, ds = _dev(pw), _dev(sw)
    single, _ = eng.msm_dev(curve, n, dp.data_ptr(), ds.data_ptr())
    assert wire_to_affine(curve, single) == exp.toAffine()
    for parts in (1, 2, 3, 8):
        got, inf = eng.msm_split_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
        assert wire_to_affine(curve, got) == exp.toAffine(), parts
        assert inf == exp.is0()
    # more shards than points: the trailing shards are empty
    got, _ = eng.msm_split_dev(curve, 5, 8, dp.data_ptr(), ds.data_ptr())
    Pt = ORACLE_CURVE[curve]
    pts = [Pt.fromAffine(wire_to_affine(curve, pw[i])) for i in range(5)]
    sc = [int.from_bytes(sw[i].tobytes(), "little") for i in range(5)]
    assert wire_to_affine(curve, got) == OC.pippenger(Pt, pts, sc).toAffine()


def test_split_pipeline_full_size_identity():
    """2^18 G1 points in 8 shards (the per-GPU share of configs[3]): the progression identity of
    test/slow-curves.test.ts:185-252, and equality with the unsharded MSM."""
    import torch
    import bench
    from oracle.curves import BLS_R, BlsG1
    eng = get_engine()
    dev = torch.device("cuda", 0)
    n = 1 << 18
    rng = makeRng(0x6D736D0000000003)
    a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
    pts, pks = bench.gen_points(eng, BLS12_381_G1, BlsG1, n, a, b, dev, None)
    sc = bench.gen_scalars(n, 254, 777, dev)
    sc[::17] = 0
    ks = bench.scalars_to_ints(sc)
    exp = BlsG1.BASE.multiplyUnsafe(sum(k * p for k, p in zip(ks, pks)) % BLS_R).toAffine()
    for parts in (2, 8):
        got, _ = eng.msm_split_dev(BLS12_381_G1, n, parts, pts.data_ptr(), sc.data_ptr())
        assert wire_to_affine(BLS12_381_G1, got) == exp, parts


def test_rccl_communicator_single_rank():
    """ncg_comm_unique_id -> ncg_comm_init(1 rank) -> ncg_msm_sharded_dev: RCCL is loaded, the
    communicator is created, and the collective entry point returns the MSM."""
    eng = Engine(0)
    try:
        uid = Engine.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(1, 0, uid)
        assert eng.comm_size() == 1
        pw, sw, exp = _case(BLS12_381_G1, 90, 0xC0FFEE)
        dp, ds = _dev(pw), _dev(sw)
        got, inf = eng.msm_sharded_dev(BLS12_381_G1, 90, dp.data_ptr(), ds.data_ptr())
        assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine() and inf == exp.is0()
        # a rank holding fewer points than the largest shard plans with n_max
        got, _ = eng.msm_sharded_dev(BLS12_381_G1, 90, dp.data_ptr(), ds.data_ptr(), n_max=4096)
        assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine()
        with pytest.raises(Exception, match="n_local > n_max"):
            eng.msm_sharded_dev(BLS12_381_G1, 90, dp.data_ptr(), ds.data_ptr(), n_max=10)
    finally:
        eng.close()


@pytest.mark.parametrize("curve", [SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2])
def test_host_staged_exchange_native_combine(curve):
    """ncg_msm_shard_local_dev per shard -> slots concatenated on the host -> ncg_msm_shard_combine: the code path of G
    ranks with a non-RCCL transport, here with the shards of one process."""
    eng = get_engine()
    n = 130 if curve != BLS12_381_G2 else 66
    pw, sw, exp = _case(curve, n, 0x51075 + curve)
    dp, ds = _dev(pw), _dev(sw)
    pb = POINT_BYTES[curve]
    assert eng.msm_shard_slot_bytes(curve) % 256 == 0
    for cuts in ((65, n), (64, n), (33, 64, n), (n, n), tuple(range(17, n, 17)) + (n,)):   # ragged, straddling 2^6, an empty shard
        sizes = [b - a for a, b in zip((0,) + cuts[:-1], cuts)]
        n_max = max(sizes)
        slots, lo = [], 0
        for m in sizes:
            slots.append(eng.msm_shard_local_dev(curve, m, dp.data_ptr() + lo * pb, ds.data_ptr() + lo * 32, None, n_max))
            lo += m
        got, inf = eng.msm_shard_combine(curve, n_max, np.stack(slots))
        assert wire_to_affine(curve, got) == exp.toAffine(), sizes
        assert inf == exp.is0()
    # a shard planned with another n_max is refused by the header check
    a = eng.msm_shard_local_dev(curve, 40, dp.data_ptr(), ds.data_ptr(), None, 40)
    b = eng.msm_shard_local_dev(curve, 40, dp.data_ptr() + 40 * pb, ds.data_ptr() + 40 * 32, None, 1 << 14)
    with pytest.raises(Exception, match="all ranks must pass the same curve and n_max"):
        eng.msm_shard_combine(curve, 40, np.stack([a, b]))
    # every shard empty: the identity
    got, inf = eng.msm_shard_combine(curve, 0, np.stack([eng.msm_shard_local_dev(curve, 0, 0, 0)] * 2))
    assert inf


def test_multi_engine_one_device():
    m = MultiEngine([0])
    try:
        assert m.devices() == 1
        for curve in (SECP256K1, BLS12_381_G1, BLS12_381_G2, ED25519):
            pw, sw, exp = _case(curve, 60, 0xBEEF + curve)
            got, inf = m.msm(curve, pw, sw)
            assert wire_to_affine(curve, got) == exp.toAffine() and inf == exp.is0()
        got, inf = m.msm(BLS12_381_G1, np.zeros((0, 96), np.uint8), np.zeros((0, 32), np.uint8))
        assert inf and not got.any()
    finally:
        m.close()


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noble_curves_amd import get_engine as ge
    from noble_curves_amd.distributed import msm_sharded, shard_range
    eng = ge(0)                                   # both ranks on the one GPU of the box
    pw, sw, exp = _case(BLS12_381_G1, n, 0xD157)
    lo, hi = (0, 129) if rank == 0 else (129, n)    # ragged: 129 and 71 points plan different windows on their own
    dp, ds = torch.from_numpy(pw[lo:hi].copy()).cuda(), torch.from_numpy(sw[lo:hi].copy()).cuda()
    out, inf = msm_sharded(eng, BLS12_381_G1, hi - lo, dp.data_ptr(), ds.data_ptr(), None, torch.device("cuda", 0))
    q.put((rank, wire_to_affine(BLS12_381_G1, out) == exp.toAffine(), bool(inf)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_the_gpu_over_gloo_with_the_native_engine():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 200, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res) and not any(r[2] for r in res)
