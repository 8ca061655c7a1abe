"""The N-API addon + JS shim (addon/): reference call shapes from Node.  Without a GPU the smoke
script checks loading and pre-crossing validation; on the GPU box it also checks k*G, batch
multiply and pippenger against the reference's vectors."""
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


def test_addon_loads_and_validates():
    assert "validation OK" in _run()


@pytest.mark.gpu
def test_addon_gpu_smoke():
    assert "GPU smoke OK" in _run()
