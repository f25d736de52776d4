"""Data-parallel gradient exchange on the GPU with the REAL model and the two-stream tape: two ranks on one card (gloo
between them, both on cuda:0), launched as fresh child processes (nothing is exec'ed from a process that holds the GPU).
The children run under a hard timeout and with HIP's default of 4 hardware queues per process (two processes share the
card, tests/ddp_gpu_worker.py).  MDT_SKIP_MULTIPROC=1 skips it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("MDT_SKIP_MULTIPROC") == "1", reason="MDT_SKIP_MULTIPROC=1")
def test_two_rank_gradient_exchange_matches_single_process():
    import signal
    env = dict(os.environ, MDT_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "ddp_gpu_worker.py")]
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group started above
        out, err = proc.communicate()
        pytest.fail("two-rank worker did not finish in 300 s:\n" + out[-2000:] + err[-2000:])
    assert proc.returncode == 0, out[-3000:] + err[-3000:]
    assert "DDP_GPU_OK" in out, out[-3000:]


@pytest.mark.skipif(os.environ.get("MDT_SKIP_MULTIPROC") == "1", reason="MDT_SKIP_MULTIPROC=1")
@pytest.mark.parametrize("extra", [[], ["--dtype", "fp8", "--with-optimizer"]], ids=["bf16", "fp8+optimizer"])
def test_bench_gpus_2_as_typed_on_one_card(extra):
    """`python3 bench.py --gpus 2` exactly as the driver types it (no torch.distributed.run in front): bench.py starts its
    two ranks itself (both on cuda:0 for the rehearsal, so gloo between them: RCCL refuses two ranks on one device) and prints ONE line with n_gpus = 2, the
    exchange self-check green — also with fp8 operands and the optimizer inside the step (the check's
    repeated step must not see the previous repetition's scales / weight update)."""
    import json
    import signal
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MDT_SINGLE_DEVICE="1", MDT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--trees", "2", "--nodes", "16",
           "--no-gemm-timer", "--no-selfcheck"] + extra
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail("bench.py --gpus 2 did not finish in 420 s:\n" + out[-2000:] + err[-2000:])
    assert proc.returncode == 0, out[-3000:] + err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    o = json.loads(lines[0])
    d = o["distributed"]
    assert o["n_gpus"] == 2 and d["world_seen_by_backend"] == 2 and d["exchange_check"]["ok"], d
    assert len(d["per_rank_host"]["packer_host_ms_max"]) == 2 and o["value"] > 0
