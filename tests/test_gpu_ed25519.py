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
