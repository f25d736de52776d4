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
