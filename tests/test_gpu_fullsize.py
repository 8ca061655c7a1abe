"""Parity at BASELINE.json's full sizes through size-independent identities (the oracle cannot
finish these sizes in seconds):
  * secp256k1 batch multiply, 2^20 pairs: sum_i k_i*P_i with P_i = (a+i*b)G must equal
    (sum k_i (a+i*b) mod n) G - the sum is taken through the MSM path, and a sample of outputs is
    compared with the oracle's C restatement;
  * bls12-381 G1 MSM 2^20 / G2 MSM 2^18: the arithmetic-progression construction of the
    reference's test/slow-curves.test.ts:185-252 (every 17th scalar zero);
  * ed25519 batch verify 2^18: verdicts known by construction (valid unless corrupted).
Inputs are generated on the device with the batch-multiply kernel itself (bench.py helpers)."""
import numpy as np
import pytest
import torch

import bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle import cport
from oracle.curves import BLS_R, BlsG1, BlsG2, SECP256K1_N, Secp256k1, makeRng

pytestmark = pytest.mark.gpu


def _setup():
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    return get_engine(0), dev, st.cuda_stream


def test_secp256k1_2p20_checksum_and_sample():
    eng, dev, stream = _setup()
    n = 1 << 20
    rng = makeRng(0x5EC9)
    a, b = rng.rndBelow(SECP256K1_N - 1) + 1, rng.rndBelow(SECP256K1_N - 1) + 1
    pts, pks = bench.gen_points(eng, SECP256K1, Secp256k1, n, a, b, dev, stream)
    sc = bench.gen_scalars(n, 255, 42, dev, edge_order=SECP256K1_N)
    out = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    inf = torch.empty((n,), dtype=torch.uint8, device=dev)
    eng.mul_var_batch_dev(SECP256K1, n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), inf.data_ptr(), stream)
    torch.cuda.synchronize()
    ks = bench.scalars_to_ints(sc)
    expect = sum(k * p for k, p in zip(ks, pks)) % SECP256K1_N
    ones = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    ones[:, 0] = 1
    tot, _ = eng.msm_dev(SECP256K1, n, out.data_ptr(), ones.data_ptr(), stream)
    assert wire_to_affine(SECP256K1, tot) == Secp256k1.BASE.multiplyUnsafe(expect).toAffine()
    assert int(inf.sum().item()) == 1 and int(inf[0].item()) == 1
    # every item pinned: a RANDOM linear combination sum_i r_i * out_i == (sum_i r_i k_i p_i) G with 64-bit r_i - a plain sum could
    # hide two compensating wrong items, this cannot (one wrong item changes it unless r_i = 0 mod its order)
    rr = np.random.RandomState(0xC0FFEE).randint(1, 1 << 62, size=n, dtype=np.int64)
    rw = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    rw[:, :8] = torch.from_numpy(rr.view(np.uint8).reshape(n, 8)).to(dev)
    tot_r, _ = eng.msm_dev(SECP256K1, n, out.data_ptr(), rw.data_ptr(), stream)
    expect_r = sum(int(r) * k * p for r, k, p in zip(rr, ks, pks)) % SECP256K1_N
    assert wire_to_affine(SECP256K1, tot_r) == Secp256k1.BASE.multiplyUnsafe(expect_r).toAffine()
    idx = np.unique(np.concatenate([np.arange(64), np.arange(n - 64, n), np.random.RandomState(1).randint(0, n, 4096)]))
    o_c, i_c = cport.multiply_unsafe("secp256k1", pts[idx].cpu().numpy(), sc[idx].cpu().numpy())
    assert np.array_equal(o_c, out[idx].cpu().numpy()) and np.array_equal(i_c, inf[idx].cpu().numpy())
    # fixed-base path at full size: sum_i k_i*G == (sum k_i) G
    eng.mul_base_batch_dev(SECP256K1, n, sc.data_ptr(), out.data_ptr(), inf.data_ptr(), stream)
    torch.cuda.synchronize()
    tot, _ = eng.msm_dev(SECP256K1, n, out.data_ptr(), ones.data_ptr(), stream)
    assert wire_to_affine(SECP256K1, tot) == Secp256k1.BASE.multiplyUnsafe(sum(ks) % SECP256K1_N).toAffine()


@pytest.mark.parametrize("curve,Pt,log2n", [(BLS12_381_G1, BlsG1, 20), (BLS12_381_G2, BlsG2, 18)])
def test_bls_msm_fullsize_progression(curve, Pt, log2n):
    eng, dev, stream = _setup()
    n = 1 << log2n
    rng = makeRng(0x6D5 + curve)
    a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
    pts, pks = bench.gen_points(eng, curve, Pt, n, a, b, dev, stream)
    sc = bench.gen_scalars(n, 254, 7 + curve, dev)
    sc[::17] = 0
    ks = bench.scalars_to_ints(sc)
    expect = Pt.BASE.multiplyUnsafe(sum(k * p for k, p in zip(ks, pks)) % BLS_R).toAffine()
    got, ginf = eng.msm_dev(curve, n, pts.data_ptr(), sc.data_ptr(), stream)
    assert wire_to_affine(curve, got) == expect and not ginf
    # the same MSM with every scalar equal (one bucket per window): sum P_i * k
    same = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    same[:, 0] = 0x1D
    same[:, 5] = 0x03
    kk = 0x1D + (0x03 << 40)
    got, _ = eng.msm_dev(curve, n, pts.data_ptr(), same.data_ptr(), stream)
    assert wire_to_affine(curve, got) == Pt.BASE.multiplyUnsafe(sum(pks) * kk % BLS_R).toAffine()


def test_ed25519_verify_full_size_both_modes():
    """configs[2] as SURVEY 8d specifies it: 2^18 (signature, message, public key) triples with distinct keys,
    1/64 corrupted (R, s or message), the reference's 196 zip215.json cases appended; verified from the
    MESSAGES (SHA-512 on the device) in ZIP-215 and strict mode.  Verdicts: by construction for the synthetic
    part in both modes, zip215.json's `valid_zip215` for the appended cases, the oracle for their strict-mode
    verdicts and for a sample of the synthetic ones."""
    from oracle.curves import Ed25519
    from oracle.edwards import eddsa_verify
    eng, dev, st = _setup()
    nv = 1 << 18
    eb = bench.make_ed25519_batch(eng, nv, 0, dev, st)
    d_ok = torch.empty((nv,), dtype=torch.uint8, device=dev)
    P = lambda t: t.data_ptr()  # noqa: E731
    eng.ed25519_verify_batch_msgs_dev(nv, P(eb["d_sig"]), P(eb["d_pk"]), P(eb["d_blob"]), P(eb["d_off"]), True, P(d_ok), st)
    torch.cuda.synchronize()
    got = d_ok.cpu().numpy().astype(bool)
    assert np.array_equal(got, eb["expect"])
    assert int((~got[:eb["tail"]]).sum()) == eb["tail"] // 64        # exactly the corrupted ones fail
    eng.ed25519_verify_batch_msgs_dev(nv, P(eb["d_sig"]), P(eb["d_pk"]), P(eb["d_blob"]), P(eb["d_off"]), False, P(d_ok), st)
    torch.cuda.synchronize()
    strict = d_ok.cpu().numpy().astype(bool)
    tail = eb["tail"]
    assert np.array_equal(strict[:tail], eb["expect"][:tail])
    for j in range(eb["nz"]):
        i = tail + j
        assert strict[i] == eddsa_verify(Ed25519, eb["sig"][i].tobytes(), b"Zcash", eb["pk"][i].tobytes(), zip215=False), j
    for i in list(range(0, 12)) + [63, 127, 191, tail - 1]:
        assert got[i] == eddsa_verify(Ed25519, eb["sig"][i].tobytes(), eb["msgs"][i], eb["pk"][i].tobytes(), zip215=True)
    # the pre-hashed entry point agrees (challenges from the device hash)
    d_k = torch.empty((nv, 32), dtype=torch.uint8, device=dev)
    eng.ed25519_challenge_batch_dev(nv, P(eb["d_sig"]), P(eb["d_pk"]), P(eb["d_blob"]), P(eb["d_off"]), P(d_k), st)
    eng.ed25519_verify_batch_dev(nv, P(eb["d_sig"]), P(eb["d_pk"]), P(d_k), True, P(d_ok), st)
    torch.cuda.synchronize()
    assert np.array_equal(d_ok.cpu().numpy().astype(bool), eb["expect"])
