"""Which kernels the vendor library picks for the bench's forward GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
bf = torch.bfloat16
M = 106496
for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(M, k, device="cuda", dtype=bf)
    w = torch.randn(n, k, device="cuda", dtype=bf)
    out = torch.empty(M, n, device="cuda", dtype=bf)
    for _ in range(20):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
