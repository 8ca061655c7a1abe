"""world_size-2 gloo test of the multi-rank MSM glue (shard -> local MSM -> exchange -> combine) that runs
without a GPU.  The native engine needs a GPU, so here the per-rank engine is an oracle-backed stand-in
with the same `msm` / `add_pairs_batch` signatures: this checks shard_range, the collective and the
combine order.  The NATIVE multi-rank paths are covered on the GPU box by tests/test_gpu_multi.py
(2 ranks sharing the GPU over gloo with the real engine; the RCCL communicator path with one rank; the
per-shard phase + combine kernel with 2-8 shards on one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Stand-in engine: MSM through the CPU oracle (test infrastructure only)."""

    def msm(self, curve, points, scalars):
        from helpers import ORACLE_CURVE, affine_to_wire, wire_to_affine
        from oracle import curve as C
        Pt = ORACLE_CURVE[curve]
        pts = [Pt.fromAffine(wire_to_affine(curve, row)) for row in points]
        sc = [int.from_bytes(bytes(row), "little") for row in scalars]
        r = C.pippenger(Pt, pts, sc)
        aff = r.toAffine()
        return np.frombuffer(affine_to_wire(curve, aff), dtype=np.uint8).copy(), r.is0()

    def add_pairs_batch(self, curve, a, b, subtract=False):
        from helpers import ORACLE_CURVE, affine_to_wire, wire_to_affine
        Pt = ORACLE_CURVE[curve]
        out = np.zeros_like(a)
        inf = np.zeros((a.shape[0],), np.uint8)
        for i in range(a.shape[0]):
            p, q = Pt.fromAffine(wire_to_affine(curve, a[i])), Pt.fromAffine(wire_to_affine(curve, b[i]))
            r = p.subtract(q) if subtract else p.add(q)
            out[i] = np.frombuffer(affine_to_wire(curve, r.toAffine()), dtype=np.uint8)
            inf[i] = r.is0()
        return out, inf


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import points_to_wire, scalars_to_wire, wire_to_affine
    from noble_curves_amd._native import BLS12_381_G1
    from noble_curves_amd.distributed import msm_sharded_host, shard_range
    from oracle.curves import BLS_R, BlsG1, makeRng
    rng = makeRng(0xD157)
    ks = [rng.rndBelow(BLS_R - 1) + 1 for _ in range(n)]
    pts = [BlsG1.BASE.multiplyUnsafe(k) for k in ks]
    sc = [0 if i % 5 == 0 else rng.rndBelow(BLS_R) for i in range(n)]
    out, inf = msm_sharded_host(OracleEngine(), BLS12_381_G1, points_to_wire(BLS12_381_G1, pts), scalars_to_wire(sc))
    exp = BlsG1.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % BLS_R).toAffine()
    lo, hi = shard_range(n, rank, world)
    q.put((rank, wire_to_affine(BLS12_381_G1, out) == exp, inf, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from noble_curves_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 - a0 - (b1 - b0) in (0, 1)


@pytest.mark.timeout(180)
def test_msm_sharded_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 13, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and not any(r[2] for r in res)
    assert {r[3] for r in res} == {(0, 7), (7, 13)}
