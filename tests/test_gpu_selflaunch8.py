"""`bench.py --gpus 8` as a tested configuration BEFORE an 8-GPU node runs it (VERDICT r05 #1b).

The driver launches `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...`; without a launcher bench.py
starts the same command itself (`self_launch`).  On a 1-GPU box the eight ranks share the GPU and exchange the MSM slots over
gloo (RCCL refuses two ranks on one device) - every rank-dependent piece of the run is the one an 8-GPU node executes: eight
shards of the batch workloads, 2 windows per rank of the 16-window G1 plan and 3,3,3,3,2,2,2,2 of the 20-window G2 plan in the
strong blocks, the point-sharded MSM over eight slices, the barrier / max-over-ranks timing, rank 0's single JSON line.  Only the
transport differs (host-staged slots instead of the ncclAllGather inside the C ABI - that call runs in
tests/test_gpu_multi.py::test_native_communicator_beside_torch_nccl_group with the one rank a 1-GPU box offers).
Why eight contexts cannot use the native communicator on one device: ncclCommInitRank rejects duplicate devices ("Duplicate GPU
detected"), so `ncg_multi_init` with 8 x the same device id creates the contexts but no communicator can span them; the one-process
form is covered by tests/test_gpu_multi.py::test_multi_engine_one_device."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_bench_gpus8_self_launched_as_eight_gloo_ranks_prints_one_line(tmp_path):
    out = tmp_path / "gpus8.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--quick-verify",
                        "--no-cpu-baseline", "--no-live-pmc", "--out", str(out)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-3000:]                       # rank 0's line and nothing else on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    for key, nwin, counts in (("msm_g1_strong", 16, [2] * 8), ("msm_g2_strong", 20, [3, 3, 3, 3, 2, 2, 2, 2])):
        b = line[key]
        assert b["scaling"] == "strong" and b["mode"] == "windows"
        assert b["ms_per_msm"] > 0 and b["ms_per_msm_n1"] > 0 and b["world"] == 8
        assert abs(b["speedup_vs_n1"] - b["ms_per_msm_n1"] / b["ms_per_msm"]) < 1e-3
        assert b["in_flight"]["ms_per_msm"] > 0
    full = json.loads(out.read_text().strip().splitlines()[-1])
    ex = full["extra"]
    for key, nwin, counts in (("msm_g1_strong", 16, [2] * 8), ("msm_g2_strong", 20, [3, 3, 3, 3, 2, 2, 2, 2])):
        b = ex[key]
        assert b["window_plan"]["nwin"] == nwin
        assert b["world"] == 8 and b["windows_per_rank"] == counts, b.get("windows_per_rank")
        assert "gloo" in b["transport"]
        assert b["by_points"]["mode"] == "points" and b["by_points"]["ms_per_msm"] > 0
        assert b["pipelined"]["ms_per_msm"] > 0 and b["pipelined"]["depth"] == 3 and len(b["pipelined"]["completion_intervals_ms"]) == 2
    for key in ("msm_g1", "msm_g2"):
        assert ex[key]["scaling"] == "weak" and ex[key]["total_points"] == 8 * ex[key]["points_per_gpu"]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        shutil.copy(str(out), os.path.join(keep, "gpus8_selflaunch_gloo.json"))
