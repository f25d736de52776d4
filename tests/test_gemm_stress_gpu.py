"""The 8-wave persistent GEMM under the jittered stress of tools/gemm_stress.py: 264 tiles on 256 workgroups at
K = 256 — the launch of tests/test_dropout_gpu.py that was seen to fail intermittently on one box — and at K = 768, five epilogue
forms, production and jitter builds, every output bit-compared with the 128 x 128 kernel.  A hard gate: no retries."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_persistent_gemm_is_bit_equal_to_the_tile_kernel_under_jitter():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_stress
    argv, sys.argv = sys.argv, ["gemm_stress.py", "120", "--quick"]
    try:
        assert gemm_stress.main() == 0
    finally:
        sys.argv = argv
