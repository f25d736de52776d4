"""Does the leading dimension of the streamed operand (row stride of A, of C) matter?  Same GEMM with rows 0, 64 and 128 elements apart
beyond the contraction length.  GPU box only."""
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

bf = torch.bfloat16
M = 106496
for (N, K) in [(2304, 768), (768, 768), (768, 3072), (3072, 768)]:
    w = torch.randn(N, K, device="cuda", dtype=bf)
    bias = torch.randn(N, device="cuda", dtype=bf)
    for pad_a in (0, 32, 64, 128):
        for pad_c in (0, 64):
            a = torch.randn(M, K + pad_a, device="cuda", dtype=bf)[:, :K]
            c = torch.empty(M, N + pad_c, device="cuda", dtype=bf)[:, :N]
            t = timeit(lambda: ops.gemm(a, w, bias=bias, out=c), iters=20, warm=5)
            print(f"N={N:5d} K={K:5d} lda=K+{pad_a:3d} ldc=N+{pad_c:3d}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:6.0f} TF/s", flush=True)
