"""The N-API addon + JS shim (addon/): reference call shapes from Node.  Without a GPU the smoke
script checks loading and pre-crossing validation; on the GPU box it also checks k*G, batch
multiply and pippenger against the reference's vectors.

`ref_dropin_test.mjs` registers the REFERENCE's own Point classes (from oracle/_ref/refjs.bundle, the type-stripped
copy of the reference's sources - test infrastructure) with the shim and compares every redirected call with the
reference's own pippenger / multiplyUnsafe / multiply / ed25519.verify on the same objects."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "addon")


def _run():
    if not shutil.which("node") or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("node / N-API headers not available")
    if not os.path.exists(os.path.join(ADDON, "noble_gpu.node")):
        subprocess.check_call(["make", "-C", ADDON], stdout=subprocess.DEVNULL)
    r = subprocess.run(["node", "smoke_test.js"], cwd=ADDON, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def _run_dropin():
    import sys
    sys.path.insert(0, ROOT)
    from oracle import refjs
    if not shutil.which("node") or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("node / N-API headers not available")
    if not refjs.available():
        pytest.skip("oracle/_ref/refjs.bundle not built (needs /root/reference at build time)")
    if not os.path.exists(os.path.join(ADDON, "noble_gpu.node")):
        subprocess.check_call(["make", "-C", ADDON], stdout=subprocess.DEVNULL)
    r = subprocess.run(["node", os.path.join(ADDON, "ref_dropin_test.mjs"), refjs.ref_dir(), os.path.join(ROOT, "tests", "golden")],
                       cwd=ADDON, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_shim_rejects_arguments_exactly_like_the_reference_classes():
    """argument errors of curve.ts:393,399-402,875 / weierstrass.ts:904,920 / edwards.ts:561,573: message AND error class equal
    to what the reference itself throws on the same call, for all four registered reference classes (no GPU needed)."""
    out = _run_dropin()
    for name in ("secp256k1", "bls12_381.G1", "bls12_381.G2", "ed25519"):
        assert name + ": " in out


@pytest.mark.gpu
def test_reference_point_classes_through_the_addon():
    out = _run_dropin()
    assert "reference drop-in OK" in out, out
    for name in ("secp256k1: OK", "bls12_381.G1: OK", "bls12_381.G2: OK", "ed25519: OK", "ed25519.verify: OK"):
        assert name in out, out


def test_addon_loads_and_validates():
    assert "validation OK" in _run()


@pytest.mark.gpu
def test_addon_gpu_smoke():
    assert "GPU smoke OK" in _run()
