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
        assert eng.comm_count() == (1, 0)          # ncclCommCount / ncclCommUserRank of the communicator itself
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


@pytest.mark.parametrize("curve", [SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2])
def test_window_sharded_pipeline_matches_oracle(curve):
    """Strong-scaling mode (include/ncg.h "WINDOW-sharded mode"): part r runs a range of the windows of ALL points
    (curve.ts:886-902), the slots are concatenated, every part count incl. more parts than windows; the host-staged
    twin in one process; resident sets in their generic, endomorphism and precomputed plans."""
    eng = get_engine()
    n = 150 if curve != BLS12_381_G2 else 70
    pw, sw, exp = _case(curve, n, 0x77AD + curve)
    dp, ds = _dev(pw), _dev(sw)
    for parts in (1, 2, 3, 8, 64):
        got, inf = eng.msm_split_windows_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
        assert wire_to_affine(curve, got) == exp.toAffine(), parts
        assert inf == exp.is0()
        assert eng.msm_last_plan()["nwin_total"] >= eng.msm_last_plan()["nwin"]
    for nparts in (2, 5):
        slots = [eng.msm_shard_windows_local_dev(curve, n, r, nparts, dp.data_ptr(), ds.data_ptr()) for r in range(nparts)]
        got, inf = eng.msm_shard_combine(curve, n, np.stack(slots))
        assert wire_to_affine(curve, got) == exp.toAffine() and inf == exp.is0(), nparts
        with pytest.raises(Exception, match="all ranks must pass the same curve and n_max"):
            eng.msm_shard_combine(curve, n, np.stack(slots[:-1]))                 # a window range is missing
        with pytest.raises(Exception, match="all ranks must pass the same curve and n_max"):
            eng.msm_shard_combine(curve, n, np.stack(slots[::-1]))                # ranges out of rank order
    rs = eng.upload_points(curve, pw)
    got, _ = eng.msm_split_windows_dev(curve, n, 3, 0, ds.data_ptr(), resident=rs)
    assert wire_to_affine(curve, got) == exp.toAffine()
    rs.free()
    got, inf = eng.msm_split_windows_dev(curve, 0, 4, 0, 0)
    assert inf


def test_out_of_range_scalar_travels_in_the_slot_header():
    """the local phase of the rank that owns a scalar >= the group order returns its slot (so the
    rank still takes part in the exchange), and the combine fails on EVERY rank - here: with every slot order."""
    eng = get_engine()
    Pt = ORACLE_CURVE[BLS12_381_G1]
    pw, sw, exp = _case(BLS12_381_G1, 64, 0xBAD5)
    sw = sw.copy()
    sw[40] = np.frombuffer(int(Pt.Fn.ORDER).to_bytes(32, "little"), dtype=np.uint8)
    dp, ds = _dev(pw), _dev(sw)
    a = eng.msm_shard_local_dev(BLS12_381_G1, 32, dp.data_ptr(), ds.data_ptr(), None, 32)
    b = eng.msm_shard_local_dev(BLS12_381_G1, 32, dp.data_ptr() + 32 * 96, ds.data_ptr() + 32 * 32, None, 32)   # holds index 40
    with pytest.raises(Exception, match="invalid scalar at index 8 of shard 1"):
        eng.msm_shard_combine(BLS12_381_G1, 32, np.stack([a, b]))
    with pytest.raises(Exception, match="invalid scalar at index 8 of shard 0"):
        eng.msm_shard_combine(BLS12_381_G1, 32, np.stack([b, a]))
    with pytest.raises(Exception, match="invalid scalar at index 40"):
        eng.msm_split_windows_dev(BLS12_381_G1, 64, 4, dp.data_ptr(), ds.data_ptr())
    with pytest.raises(Exception, match=r"invalid scalar at index 40 \("):         # one caller: the index in ITS arrays (curve.ts:402)
        eng.msm_split_dev(BLS12_381_G1, 64, 2, dp.data_ptr(), ds.data_ptr())
    with pytest.raises(Exception, match=r"invalid scalar at index 40 \("):
        eng.msm_split_dev(BLS12_381_G1, 64, 5, dp.data_ptr(), ds.data_ptr())           # slices of 13: index 1 of shard 3
    with pytest.raises(Exception, match="invalid scalar at index 40"):
        eng.msm_dev(BLS12_381_G1, 64, dp.data_ptr(), ds.data_ptr())
    eng.msm_async_submit(0, BLS12_381_G1, 64, dp.data_ptr(), ds.data_ptr())
    with pytest.raises(Exception, match="invalid scalar at index 40"):
        eng.msm_async_collect(0, BLS12_381_G1)


def test_several_msms_in_flight():
    """ncg_msm_async_submit / _collect: four MSMs of different curves and sizes enqueued on the four lanes before any is
    collected; collected in another order; lanes reused; a busy lane and an idle lane are refused."""
    eng = get_engine()
    assert eng.msm_async_lanes() >= 2
    jobs = []
    for lane, (curve, n) in enumerate([(BLS12_381_G1, 300), (SECP256K1, 111), (BLS12_381_G2, 60), (ED25519, 77)]):
        pw, sw, exp = _case(curve, n, 0xA5 + lane)
        dp, ds = _dev(pw), _dev(sw)
        jobs.append((lane, curve, n, dp, ds, exp))
    for rnd in range(3):
        for lane, curve, n, dp, ds, exp in jobs:
            eng.msm_async_submit(lane, curve, n, dp.data_ptr(), ds.data_ptr())
        with pytest.raises(Exception, match="has an MSM in flight"):
            eng.msm_async_submit(2, BLS12_381_G1, 10, jobs[0][3].data_ptr(), jobs[0][4].data_ptr())
        for lane, curve, n, dp, ds, exp in (jobs[::-1] if rnd % 2 else jobs):
            got, inf = eng.msm_async_collect(lane, curve)
            assert wire_to_affine(curve, got) == exp.toAffine() and inf == exp.is0(), (rnd, lane)
    with pytest.raises(Exception, match="nothing was submitted"):
        eng.msm_async_collect(1, SECP256K1)
    # resident sets (incl. a precomputed one) and the empty MSM on a lane
    pw, sw, exp = _case(BLS12_381_G1, 5000, 0x5E7)
    rs = eng.upload_points(BLS12_381_G1, pw)
    ds = _dev(sw)
    for stage in range(3):
        if stage == 1:
            assert rs.verify_subgroup() == -1
        if stage == 2:
            assert rs.precompute()
        eng.msm_async_submit(0, BLS12_381_G1, 0, 0, ds.data_ptr(), resident=rs)
        eng.msm_async_submit(1, BLS12_381_G1, 0, 0, ds.data_ptr(), resident=rs)
        for lane in (0, 1):
            got, inf = eng.msm_async_collect(lane, BLS12_381_G1)
            assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine(), stage
        got, _ = eng.msm_split_windows_dev(BLS12_381_G1, 0, 8, 0, ds.data_ptr(), resident=rs)
        assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine(), stage
        # the parts of ONE window-sharded MSM in flight on the lanes, their slots combined (NCG_MSM_ASYNC_PART)
        for r in range(4):
            eng.msm_async_submit(r, BLS12_381_G1, 0, 0, ds.data_ptr(), resident=rs, flags=eng.async_part(r, 4))
        with pytest.raises(Exception, match="use ncg_msm_async_collect_slot"):
            eng.msm_async_collect(0, BLS12_381_G1)
        slots = [eng.msm_async_collect_slot(r, BLS12_381_G1) for r in range(4)]
        got, _ = eng.msm_shard_combine(BLS12_381_G1, 5000, np.stack(slots))
        assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine(), stage
    rs.free()
    eng.msm_async_submit(3, BLS12_381_G1, 0, 0, 0)
    got, inf = eng.msm_async_collect(3, BLS12_381_G1)
    assert inf and not got.any()


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


def _nccl_single_worker(port, q):
    """One-rank torch.distributed group on the nccl (= RCCL) backend with the engine's own communicator beside it."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from noble_curves_amd import get_engine as ge
        from noble_curves_amd._native import Engine as _E
        ASYNC_WINDOWS = _E.ASYNC_WINDOWS
        from noble_curves_amd.distributed import init_comm
        eng = ge(0)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                           # torch's own communicator is up first
        assert init_comm(eng, dev, single_ok=True)   # id broadcast over torch's group, ncclCommInitRank inside libncg
        assert eng.has_comm() and eng.comm_size() == 1 and eng.comm_count() == (1, 0)
        assert init_comm(eng, dev, single_ok=True)   # second call: nothing to do
        n = 300
        pw, sw, exp = _case(BLS12_381_G1, n, 0xACC1)
        dp, ds = torch.from_numpy(pw).to(dev), torch.from_numpy(sw).to(dev)
        want = exp.toAffine()
        out, inf = eng.msm_sharded_dev(BLS12_381_G1, n, dp.data_ptr(), ds.data_ptr())          # all-gather of 1 slot
        ok = wire_to_affine(BLS12_381_G1, out) == want and not inf
        out, inf = eng.msm_sharded_windows_dev(BLS12_381_G1, n, dp.data_ptr(), ds.data_ptr())  # window mode
        ok = ok and wire_to_affine(BLS12_381_G1, out) == want
        for lane in (0, 1, 2):                       # collectives of the asynchronous lanes funnel through one stream
            eng.msm_async_submit(lane, BLS12_381_G1, n, dp.data_ptr(), ds.data_ptr(), flags=ASYNC_WINDOWS)
        for lane in (0, 1, 2):
            out, inf = eng.msm_async_collect(lane, BLS12_381_G1)
            ok = ok and wire_to_affine(BLS12_381_G1, out) == want
        dist.all_reduce(t)                           # and torch's communicator still works afterwards
        torch.cuda.synchronize()
        ok = ok and float(t[0]) == 1.0
        eng.comm_destroy()
        assert not eng.has_comm()
        dist.destroy_process_group()
        q.put(("ok" if ok else "wrong result", ""))
    except Exception as e:  # noqa: BLE001 - reported to the parent
        import traceback
        q.put(("error", "%s\n%s" % (e, traceback.format_exc())))


@pytest.mark.timeout(300)
def test_native_communicator_beside_torch_nccl_group():
    """The path `bench.py --gpus N` takes on a multi-GPU node, as far as one GPU can run it: torch.distributed on the
    nccl backend, distributed.init_comm (unique id from rank 0 broadcast over torch's group, ncclCommInitRank inside
    libncg.so, the RCCL library shared with torch), then the point-sharded, window-sharded and in-flight collectives."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_worker, args=(port, q))
    p.start()
    status, detail = q.get(timeout=240)
    p.join(60)
    assert status == "ok", detail
    assert p.exitcode == 0
