"""Data-parallel gradient exchange on the GPU with the two-stream tape: two ranks on one card (gloo), launched as
separate processes.  Opt-in (MDT_RUN_MULTIPROC=1): it starts child processes, which a test process that has already
initialised the GPU should not do on a shared box; run it on its own:
  MDT_RUN_MULTIPROC=1 python -m pytest tests/test_ddp_gpu.py -m gpu -q"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("MDT_RUN_MULTIPROC") != "1", reason="opt-in: spawns two GPU processes (MDT_RUN_MULTIPROC=1)")
def test_two_rank_gradient_exchange_matches_single_process():
    env = dict(os.environ, MDT_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "ddp_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DDP_GPU_OK" in r.stdout, r.stdout[-3000:]
