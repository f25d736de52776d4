"""One GEMM shape, a few launches, for SQ counter passes (rocprofv3 --pmc ...)."""
import sys
import torch
sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops
M, N, K = 212992, 768, 3072
bf = torch.bfloat16
a = torch.randn(M, K, device="cuda", dtype=bf); b = torch.randn(N, K, device="cuda", dtype=bf)
out = torch.empty(M, N, device="cuda", dtype=bf)
for _ in range(4):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
