"""The reference's own `pippenger` export REDIRECTED to the GPU ("switches to GPU above a size threshold").

oracle/_ref/refjs.bundle holds a second copy of the reference (`js_hooked/`) with the 12-line MSM-backend patch of INTEGRATION.md
applied to src/abstract/curve.ts (oracle/ref_js/downlevel.py --gpu-hook) plus the reference's own test / benchmark files for the
path.  addon/ref_redirect_test.mjs installs the shim as the backend of the reference's four Point classes and runs, unmodified,
benchmark/msm_timings.ts (its own `check`), benchmark/bls12-381.ts:64-79, the selected tests of test/point.test.ts and ed25519.verify
over test/vectors/ed25519/vectors.txt - counting the calls that took the GPU path."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "addon")


def _hooked():
    sys.path.insert(0, ROOT)
    from oracle import refjs
    if not shutil.which("node") or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("node / N-API headers not available")
    if not refjs.available() or refjs.hooked_dir() is None:
        pytest.skip("oracle/_ref/refjs.bundle (with js_hooked/) not built - needs /root/reference at build time")
    if not os.path.exists(os.path.join(ADDON, "noble_gpu.node")):
        subprocess.check_call(["make", "-C", ADDON], stdout=subprocess.DEVNULL)
    return refjs.hooked_dir()


def _run(extra=(), timeout=1500):
    d = _hooked()
    r = subprocess.run(["node", os.path.join(ADDON, "ref_redirect_test.mjs"), d] + list(extra), cwd=ADDON, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_hook_is_inert_without_a_backend_and_keeps_the_reference_checks_first():
    """no GPU needed: below minPoints the reference's loop runs, its argument errors and its empty-input return come before the
    backend, uninstall restores the loop, and there is no CPU fallback behind an installed backend"""
    out = _run()
    assert "hook checks OK" in out or "reference redirect OK" in out, out


def test_the_patch_is_the_one_integration_md_shows():
    """INTEGRATION.md quotes the diff `downlevel.py --print-hook-diff` produces against the reference's file"""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference is absent here")
    diff = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_js", "downlevel.py"), "--print-hook-diff"],
                          capture_output=True, text=True, check=True).stdout
    added = [ln[1:] for ln in diff.splitlines() if ln.startswith("+") and not ln.startswith("+++")]
    assert 10 <= len(added) <= 14 and not any(ln.startswith("-") and not ln.startswith("---") for ln in diff.splitlines())
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for ln in added:
        assert ln.strip() in doc, ln


@pytest.mark.gpu
def test_reference_tests_and_benchmarks_run_on_the_gpu_through_the_reference_export(tmp_path):
    table = tmp_path / "threshold.json"
    out = _run(["--threshold-table", str(table)])
    for mark in ("A msm_timings.ts: its own check passed", "B bls12-381.ts MSM pippenger x32768", "C test/point.test.ts: 36 tests passed",
                 "D ed25519 vectors.txt: 1280 verifications", "reference redirect OK"):
        assert mark in out, out
    rows = json.loads(table.read_text())["rows"]
    assert len(rows) == 32
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        shutil.copy(str(table), os.path.join(keep, "js_threshold.json"))
        with open(os.path.join(keep, "js_redirect.log"), "w") as f:
            f.write(out)
