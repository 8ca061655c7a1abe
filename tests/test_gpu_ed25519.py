"""GPU parity: ed25519 batch verification against the oracle (edwards.ts:942-989 restated) and the
reference's vectors: cr.yp.to sign.input, the 196 ZIP-215 cases in both modes, eprint 2020/1244
edge cases, scalar-boundary signatures."""
import pytest

from noble_curves_amd import ed25519 as ed
from oracle.curves import Ed25519
from oracle.edwards import eddsa_verify

from helpers import load_golden
from test_host_logic import _ed_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("zip215", [True, False])
def test_verify_batch_matches_oracle(zip215):
    cases = _ed_cases()
    got = ed.verify_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], zip215=zip215)
    for (sig, msg, pk), g in zip(cases, got):
        assert g == eddsa_verify(Ed25519, sig, msg, pk, zip215=zip215), (sig.hex(), pk.hex())


def test_zip215_json_verdicts():
    """test/ed25519.test.ts:393-405."""
    vs = load_golden("ed25519_zip215.json")
    got = ed.verify_batch([bytes.fromhex(v["sig_bytes"]) for v in vs], [b"Zcash"] * len(vs),
                          [bytes.fromhex(v["vk_bytes"]) for v in vs])
    assert got == [v["valid_zip215"] for v in vs]


def test_sign_input_vectors_and_single_verify():
    rows = load_golden("ed25519_vectors.json")
    sigs = [bytes.fromhex(r["sig"]) for r in rows]
    msgs = [bytes.fromhex(r["msg"]) for r in rows]
    pks = [bytes.fromhex(r["pk"]) for r in rows]
    assert all(ed.verify_batch(sigs, msgs, pks))
    assert all(ed.verify_batch(sigs, msgs, pks, zip215=False))
    assert ed.verify(sigs[3], msgs[3], pks[3]) and not ed.verify(sigs[3], msgs[4], pks[3])
    assert ed.verify_batch([], [], []) == []
    with pytest.raises(ValueError):
        ed.verify(sigs[0][:63], msgs[0], pks[0])


@pytest.mark.gpu
def test_wycheproof_old_vectors_gpu():
    """test/ed25519.test.ts:420-444 through the batch verifier (signatures of the wrong length never reach the
    device: the shim rejects them like the reference's abytes check)."""
    from noble_curves_amd import ed25519 as ed
    rows = load_golden("ed25519_wycheproof_old.json")
    good = [r for r in rows if len(r["sig"]) == 128]
    got = ed.verify_batch([bytes.fromhex(r["sig"]) for r in good], [bytes.fromhex(r["msg"]) for r in good],
                          [bytes.fromhex(r["pk"]) for r in good])
    for r, v in zip(good, got):
        assert v == (r["result"] in ("valid", "acceptable")), r["comment"]
    assert all(r["result"] == "invalid" for r in rows if len(r["sig"]) != 128)


def test_chosen_challenges_exercise_the_scalar_halving():
    """ncg_ed25519_verify_batch takes the challenge k as data: signatures built for chosen k (edges of the
    truncated Euclidean algorithm of ed_halve.hpp: tiny, around 2^127, L - 1, values >= L, the golden-ratio worst
    case, k with one long quotient) must verify, corrupted ones must not, and a torsion component added to R or A
    keeps the cofactored verdict - all against the equation [8](sB - kA - R) == O evaluated by the oracle."""
    from decimal import Decimal, getcontext
    import numpy as np
    from noble_curves_amd import get_engine
    from oracle.curves import ED25519_L as L, makeRng
    getcontext().prec = 120
    gold = int(Decimal(L) * (Decimal(5).sqrt() - 1) / 2)
    rng = makeRng(0x1A1F)
    B = Ed25519.BASE
    ks = [0, 1, 2, 3, 2 ** 127 - 1, 2 ** 127, 2 ** 127 + 1, 2 ** 126, 2 ** 128 + 1, 2 ** 200 + 1, L - 1, L - 2, L // 2, L // 3, gold,
          L - gold, L, L + 5, 2 ** 255 + 19, 2 ** 256 - 1, 2 ** 252]
    ks += [rng.rndBelow(L) for _ in range(107)]
    tors = []                                       # 8-torsion points: [L]P for curve points P outside the subgroup
    for y in range(2, 60):
        try:
            P = Ed25519.fromBytes(y.to_bytes(32, "little"))
        except Exception:
            continue
        T = P.multiplyUnsafe(L - 1).add(P)
        if not T.is0():
            tors.append(T)
    assert len(tors) >= 4
    sigs, pks, kw, exp = [], [], [], []
    for i, k in enumerate(ks):
        a, r = rng.rndBelow(L - 1) + 1, rng.rndBelow(L - 1) + 1
        A, R = B.multiplyUnsafe(a), B.multiplyUnsafe(r)
        s = (r + k * a) % L
        mode = i % 4 if i >= 21 else 0
        if mode == 1:
            s = (s + 1) % L                         # wrong s
        elif mode == 2 and tors:
            R = R.add(tors[i % len(tors)])          # torsion on R: still valid under the cofactored equation
        elif mode == 3 and tors:
            A = A.add(tors[i % len(tors)])          # torsion on A: [8][k]A unchanged
        lhs = B.multiplyUnsafe(s).subtract(A.multiplyUnsafe(k % L)).subtract(R)
        exp.append(lhs.clearCofactor().is0())
        sigs.append(R.toBytes() + s.to_bytes(32, "little"))
        pks.append(A.toBytes())
        kw.append(k.to_bytes(32, "little"))
    got = get_engine().ed25519_verify_batch(np.frombuffer(b"".join(sigs), np.uint8), np.frombuffer(b"".join(pks), np.uint8),
                                            np.frombuffer(b"".join(kw), np.uint8), zip215=True)
    assert list(got) == exp
    assert sum(exp) > 90 and not all(exp)


def test_fresh_context_sizes_its_own_scratch():
    """A context whose FIRST call is a large ed25519 verification must size the per-item table scratch for it (two
    8-entry tables per item) - not inherit a buffer that an earlier, larger call happened to leave behind."""
    import numpy as np
    from noble_curves_amd._native import Engine
    rows = load_golden("ed25519_vectors.json")[:64]
    reps = 128
    sigs = np.frombuffer(b"".join(bytes.fromhex(r["sig"]) for r in rows), np.uint8).reshape(-1, 64)
    pks = np.frombuffer(b"".join(bytes.fromhex(r["pk"]) for r in rows), np.uint8).reshape(-1, 32)
    msgs = [bytes.fromhex(r["msg"]) for r in rows]
    blob = np.frombuffer(b"".join(msgs) * reps, np.uint8)
    off = np.cumsum([0] + [len(m) for m in msgs] * reps).astype(np.uint64)
    eng = Engine(0)
    ok = eng.ed25519_verify_batch_msgs(np.tile(sigs, (reps, 1)), np.tile(pks, (reps, 1)), blob, off, zip215=True)
    assert ok.all() and ok.shape[0] == 64 * reps
