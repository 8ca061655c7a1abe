"""The line bench.py prints must be something the driver can keep: one compact JSON object, well under the 8 KB
stdout tail it parses (BENCH_r04 came back `parsed: null` from a 36.7 KB line).  Built here from committed full
result objects of real runs (single GPU; the multi-rank dry run with the strong-scaling blocks)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_compact  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _full(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r06_bench_full.json", "r05_bench_full.json", "r04_bench_line_last_tree.json", "r04_bench_line.json", "r03_bench_line.json"])
def test_compact_line_fits_and_keeps_the_contract(name):
    full = _full(name)
    line = bench_compact.dumps(bench_compact.compact_line(full, "gpurun_out/bench_full.json"))
    assert "\n" not in line
    assert len(line) < bench_compact.LIMIT < 8192
    c = json.loads(line)
    for k in CONTRACT:
        assert k in c, k
    assert c["metric"] == full["metric"] and c["n_gpus"] == full["n_gpus"] and c["steps"] == full["steps"]
    assert abs(c["value"] - full["value"]) <= 1e-6 * full["value"]
    assert abs(c["ms_per_step"] - full["ms_per_step"]) <= 1e-6 * full["ms_per_step"]
    assert c["config"]["workload"]
    rf = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert "mad_frac" in rf["valu"]
    cb = c["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    # both halves of BASELINE's metric, flat
    for key in ("msm_g1", "msm_g2"):
        b = c[key]
        for k in ("value", "unit", "ms_per_msm", "roofline", "cpu_baseline"):
            assert k in b, (key, k)
        src = (full.get("extra") or {}).get(key) or full[key]
        assert abs(b["ms_per_msm"] - src["ms_per_msm"]) <= 1e-5 * src["ms_per_msm"]
    assert "ms_per_batch" in c["ed25519"] and "ms_per_transform" in c["ntt"]


@pytest.mark.parametrize("name", ["r05_bench_dist_dry_run.json", "r05_gpus2_selflaunch_gloo.json", "r04_bench_dist_dry_run.json"])
def test_compact_line_of_a_multi_rank_run_keeps_the_strong_blocks(name):
    full = _full(name)
    c = bench_compact.compact_line(full)
    assert len(bench_compact.dumps(c)) < bench_compact.LIMIT
    for key in ("msm_g1_strong", "msm_g2_strong"):
        assert c[key]["scaling"] == "strong" and c[key]["mode"] and c[key]["ms_per_msm"] > 0
        if name.startswith("r05"):      # round 5: the in-run one-GPU time of the same MSM and the speed-up against it
            assert c[key]["ms_per_msm_n1"] > 0 and abs(c[key]["speedup_vs_n1"] - c[key]["ms_per_msm_n1"] / c[key]["ms_per_msm"]) < 1e-3
    if name == "r05_bench_dist_dry_run.json":
        assert c["msm_g1_strong"]["rccl_ranks"] == 1     # ncclCommCount of the one-rank communicator the dry run creates


def test_committed_driver_line_is_what_the_compactor_makes_of_the_committed_full_object():
    """profiles/r06_bench_line.json is the stdout of the driver's command, profiles/r06_bench_full.json its --out file"""
    with open(os.path.join(ROOT, "profiles", "r06_bench_line.json")) as f:
        text = f.read()
    assert text.count("\n") == 1 and len(text) < bench_compact.LIMIT
    line = json.loads(text)
    again = bench_compact.compact_line(_full("r06_bench_full.json"), line.get("full"))
    assert again == line


def _fractions(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _fractions(v, path + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _fractions(v, path + "/%d" % i)
    elif isinstance(o, (int, float)) and not isinstance(o, bool) and path.rsplit("/", 1)[-1].endswith("frac"):
        yield path, o


@pytest.mark.parametrize("name", ["r06_bench_full.json", "r06_bench_line.json"])
def test_nothing_called_a_fraction_exceeds_one(name):
    """issue_frac read 1.005 / 1.046.  From round 6 every key ending in `frac` is a rate over the ceiling
    measured for that instruction class (mad_frac from the SQ_INSTS_VALU_INT64 counter, plain_frac from SQ_INSTS_VALU minus it),
    and the sum of the two is reported as pipe_demand_sum - explicitly not a fraction."""
    full = _full(name)
    fr = list(_fractions(full))
    assert len(fr) >= 5
    for path, v in fr:
        assert 0.0 <= v <= 1.0, (path, v)
    assert "issue_frac" not in json.dumps(full)
    valu = full["roofline"]["valu"]
    assert "mad_frac" in valu
    if name == "r06_bench_full.json":
        assert valu["mad_source"].startswith("SQ_INSTS_VALU_INT64") and 0.8 < valu["counter_over_static"] < 1.1
        assert abs(valu["pipe_demand_sum"] - valu["mad_frac"] - valu["plain_frac"]) < 1e-9
        for key in ("msm_g1", "msm_g2", "ed25519_verify", "ntt_fr"):
            v = full["extra"][key]["roofline"]["valu"]
            assert v["mad_source"].startswith("SQ_INSTS_VALU_INT64"), key


def test_compact_line_never_exceeds_the_limit_even_with_bloated_input():
    full = _full("r04_bench_line_last_tree.json")
    full["config"]["workload"] = "x" * 5000
    full["data"] = "y" * 5000
    for e in full["extra"].values():
        if isinstance(e, dict):
            e["note"] = "z" * 20000
    assert len(bench_compact.dumps(bench_compact.compact_line(full))) < bench_compact.LIMIT


def test_single_workload_lines_keep_what_the_ab_scripts_read():
    full = _full("r04_bench_line_last_tree.json")
    e = dict(full["extra"]["msm_g1"])
    e.update({"n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": None, "config": {"workload": "msm_g1"}})
    c = bench_compact.compact_line(e)
    assert c["ms_per_msm"] > 0 and c["resident_subgroup_set"]["ms_per_msm"] > 0


def test_power_state_of_the_headline_travels_in_the_line():
    """bench.py's untimed power_state leg (shader clock / package power the box holds under the headline workload): the compact line keeps the
    medians and the clock-normalised time, and still fits."""
    full = _full("r06_bench_full.json")
    full["power_state"] = {"sclk_mhz": [2264.0, 2272.0, 2278.0], "power_w": [1272.0, 1285.0, 1297.0], "power_cap_w": 1400.0, "samples": 28,
                           "seconds": 1.2, "source": "hwmon", "ms_per_step_x_sclk": full["ms_per_step"] * 2272.0}
    line = bench_compact.dumps(bench_compact.compact_line(full, "gpurun_out/bench_full.json"))
    assert len(line) < bench_compact.LIMIT
    ps = json.loads(line)["power_state"]
    assert ps["sclk_mhz_med"] == 2272.0 and ps["power_w_med"] == 1285.0
    assert abs(ps["ms_per_step_x_sclk"] - full["ms_per_step"] * 2272.0) < 1.0
    # a box without readable hwmon files: the block is simply absent
    full.pop("power_state")
    assert "power_state" not in json.loads(bench_compact.dumps(bench_compact.compact_line(full, None)))
