"""bench.py's one JSON line — the contract the driver reads — at toy sizes, for every configuration it offers: the N = 1 form
(`python bench.py`), the shipped launch (`--config launch`: micro-batches, frozen prefix, fused Adam), mDT-large shapes, fp8 operands.
The numbers mean nothing at these sizes; the fields, their types and the legs that must have run (roofline from live HIP events,
cpu_baseline from the oracle, the numerics self-check) do."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--trees", "2", "--nodes", "8", *extra],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,dtype", [((), "bf16"), (("--config", "launch"), "bf16"), (("--config", "large"), "bf16"), (("--dtype", "fp8", "--no-selfcheck"), "fp8")])      # the self-check's fp8 gates are calibrated at the full size
def test_bench_line_contract(extra, dtype):
    d = run_bench(*extra, "--no-cpu-baseline")
    assert d["metric"] == "discussion-tree comments/sec fwd+bwd" and d["unit"] == "comments/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["dtype"] == dtype
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["config"]["comments_per_step_per_gpu"] / d["ms_per_step"] * 1e3) <= 0.02 * d["value"]
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] in (2500.0, 5000.0)
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches"] > 0
    if "--no-selfcheck" not in extra:
        assert d["selfcheck"]["kernels"] == "passed"


def test_bench_cpu_baseline_leg_runs_the_oracle():
    d = run_bench("--trees", "1", "--nodes", "4")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "comments/s" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle" in cb["sample"]
