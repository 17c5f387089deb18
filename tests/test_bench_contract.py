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
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "voice-samples/s" and d["value"] > 0 and d["parity_vs_golden"] is True
    assert "configs[3]" in d["config"]["workload"] and d["config"]["voices_per_gpu"] == 65536
    # the golden covers 64 steps (round 4: every step of the default run): here step 0, three warm-up steps and all
    # six timed steps are compared hash by hash
    assert d["parity"]["golden_steps_compared"] == 10 and d["parity"]["timed_steps_covered_by_golden"] == 6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["kernel"] == "k_leaf_osc2pan"
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
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["value"] > 0


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
