"""A handful of GEMM launches for PMC collection (rocprofv3 --pmc ...). MDT_GEMM_DIAG variants via --diag."""
import os
import sys

if "--diag" in sys.argv:
    os.environ["MDT_GEMM_DIAG"] = sys.argv[sys.argv.index("--diag") + 1]
import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402

M = 212992
bf = torch.bfloat16
cases = [(M, 3072, 768, 0, 0), (M, 768, 3072, 0, 0), (M, 768, 3072, 0, 1), (M, 3072, 768, 0, 1)]
for (m, n, k, ta, tb) in cases:
    a = torch.randn(m, k, device="cuda", dtype=bf)
    b = torch.randn(k, n, device="cuda", dtype=bf) if tb else torch.randn(n, k, device="cuda", dtype=bf)
    out = torch.empty(m, n, device="cuda", dtype=bf)
    for _ in range(3):
        ops.gemm(a, b, trans_b=bool(tb), out=out)
    torch.cuda.synchronize()
    del a, b, out
dy = torch.randn(M, 3072, device="cuda", dtype=bf)
x = torch.randn(M, 768, device="cuda", dtype=bf)
c = torch.zeros(3072, 768, device="cuda", dtype=torch.float32)
for _ in range(3):
    ops.gemm(dy, x, trans_a=True, trans_b=True, out=c, epilogue=ops.EPI_ATOMIC, split_k=7)
torch.cuda.synchronize()
