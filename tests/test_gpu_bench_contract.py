"""bench.py prints ONE JSON line with the contract's keys (metric, value, roofline, cpu_baseline, ...)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-spp", "1", "--repeats", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["metric"] == "Mpaths/sec (paths = pixels x bounces x spp) at 1280x720, 8 bounces"
    assert d["unit"] == "Mpaths/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # one step = one wavefront batch of 32 iterations of the whole frame
    assert d["config"]["iterations_per_step"] == 32 and d["config"]["paths_per_step_nominal"] == 1280 * 720 * 8 * 32
    assert abs(d["value"] - 1280 * 720 * 8 * 32 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    assert d["value"] > 1000.0                                   # north star: >= 1.0 Gpaths/s
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["frac"] < 1
    # every timed launch carries a full batch, so the committed PMC figures describe the launches that were timed:
    # measured HBM traffic within 25 % of the algorithmic bytes, and the kernel cannot beat its own VALU issue bound
    assert rf["iterations_per_launch"] == 32 and rf["launches"] == 3 * 8
    if rf["traffic"] is not None:
        assert 0.9 < rf["traffic_over_algorithmic"] < 1.25
        assert 0 < rf["valu"]["frac_of_issue_bound"]["v_fma_f32"] < 1
    # the timed block of exactly `steps` steps is repeated inside the run: the median block's wall is the line's, the spread beside it
    assert d["repeats"] == 5 and len(d["ms_per_step_blocks"]) == 5
    assert d["ms_per_step"] == sorted(d["ms_per_step_blocks"])[2]
    assert d["ms_per_step_min"] == min(d["ms_per_step_blocks"]) and d["ms_per_step_max"] == max(d["ms_per_step_blocks"])
    assert d["value_min"] <= d["value"] <= d["value_max"]
    # the line says what the box it ran on does on a fixed job (boxes of the pool differ by up to 1.8x)
    assert d["box_calibration"]["result_checked"] is True and d["box_calibration"]["algorithmic_GBps"] > 500
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert cb["cpu_model"] and cb["host_cores"] >= 1 and cb["pinned_to_core"] is not None
    # BASELINE config C1 (sphere.txt 400x400, 1 spp, depth 4) on the CPU, in full
    assert cb["c1"]["value"] > 0 and cb["c1"]["runs"] == 15 and "400x400" in cb["c1"]["config"]
