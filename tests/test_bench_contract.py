"""bench.py keeps the driver's contract: one JSON line with the agreed keys."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "roofline_valu"}


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    """Default run = BASELINE configs[3] at full size, every step's audio delivered to
    the host, the steps the oracle golden covers compared hash by hash."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3",
                        "--no-cpu-baseline", "--no-engine"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    # (round 6) the driver reads the last 8 KB of stdout: the line is short, everything else is in the side file
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert json.loads(json.dumps(d)) == d
    assert REQUIRED <= set(d), REQUIRED - set(d)
    full = json.load(open(os.path.join(ROOT, d["details"])))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "timing" in full["roofline"]
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "voice-samples/s" and d["value"] > 0 and d["parity_vs_golden"] is True
    assert "configs[3]" in d["config"]["workload"] and d["config"]["voices_per_gpu"] == 65536
    # the golden covers 64 steps (round 4: every step of the default run): here step 0, three warm-up steps and all
    # six timed steps are compared hash by hash
    assert d["parity"]["golden_steps_compared"] == 10 and d["parity"]["timed_steps_covered_by_golden"] == 6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    # (the line carries 6 significant digits - bench._r - so frac and achieved / peak agree to that, not to 1e-9)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 * rf["frac"] and rf["kernel"] == "k_leaf_osc2pan"
    assert d["roofline_valu"]["bound"] == "valu-issue"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["realtime"]["fragment_ms_p99"] > 0
    for k in ("configs[1]", "configs[2]"):
        assert d["other_configs"][k]["value"] > 0 and d["other_configs"][k]["parity_vs_golden"] is True


@pytest.mark.gpu
def test_bench_line_is_alone_on_stdout_under_torchrun():
    """The driver launches N>1 through torch.distributed.run; RCCL writes a version
    banner to fd 1 through C stdio.  The contract line must be the ONLY line on
    stdout (bench.py points fd 1 at stderr until it prints it)."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--config", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, A2AMD_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines[:-1]
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["value"] > 0


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.parametrize("name", ["r06_bench_default_details.json", "r06_bench_default_final_library_details.json", "r05_bench_default.json", "r05_bench_n_gt_1_path_one_rank.json",
                                  "r04_bench_cfg5_private_waves.json", "r04_bench_cfg4_all_on_one_gpu.json"])
def test_contract_line_of_a_full_run_stays_short(name):
    """Round 5's driver record had parsed: null - the line had grown to 24 KB and the driver keeps 8 KB of stdout.
    bench.contract_line() of everything a full default run measures (the round-5 runs kept under profiles/ are such
    dictionaries: all engine cells, both other configs, the sweeps) is under 4 096 characters, is JSON, and carries
    every key the contract and the measurement row name."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", name)) as f:
        full = json.load(f)
    line = bench.contract_line(full)
    s = json.dumps(line)
    assert len(s) <= bench.LINE_LIMIT < 4096
    d = json.loads(s)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
            "algorithmic_bytes_per_launch"} <= set(d["roofline"])
    assert d["value"] == pytest.approx(full["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert "workload" in d["config"] and "model" not in d["config"]
    if "cpu_baseline" in full:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    if "engine_in_loop" in full:
        assert d["engine_in_loop"]["hash_equal"] is True and len(d["engine_in_loop"]["cells"]) == len(full["engine_in_loop"]["cases"])
        if "value_a2_run" in full:      # (the N = 1 default run)
            assert d["value_a2_run"]["hash_equal"] is True
