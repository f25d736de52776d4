"""Streaming kernels at mDT-base shapes: LayerNorm forward / backward (plain, fused dropped copy + column sums, with
residual add), column sums; algorithmic TB/s.  GPU box only."""
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

bf = torch.bfloat16
M, D = 106496, 768
x = torch.randn(M, D, device="cuda", dtype=bf)
g = torch.randn(D, device="cuda", dtype=bf)
b = torch.randn(D, device="cuda", dtype=bf)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12)
t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-12, out=y), iters=30)
print(f"ln fwd                     {t*1e6:7.1f} us  {2*x.numel()*2/t/1e12:5.2f} TB/s")
dy = torch.randn(M, D, device="cuda", dtype=bf)
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dgamma=dg, dbeta=db), iters=30)
print(f"ln bwd plain               {t*1e6:7.1f} us  {3*x.numel()*2/t/1e12:5.2f} TB/s")
add = torch.randn(M, D, device="cuda", dtype=bf)
t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, add=add, dgamma=dg, dbeta=db), iters=30)
print(f"ln bwd + add               {t*1e6:7.1f} us  {4*x.numel()*2/t/1e12:5.2f} TB/s")
cs = torch.zeros(D, device="cuda")
t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dgamma=dg, dbeta=db, drop_p=0.4, drop_seed=3, colsum=cs, want_dropped=True), iters=30)
print(f"ln bwd + dropped + colsum  {t*1e6:7.1f} us  {4*x.numel()*2/t/1e12:5.2f} TB/s")
t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, add=add, dgamma=dg, dbeta=db, drop_p=0.4, drop_seed=3, colsum=cs, want_dropped=True), iters=30)
print(f"ln bwd + add + dropped     {t*1e6:7.1f} us  {5*x.numel()*2/t/1e12:5.2f} TB/s")
for N in (768, 2304):
    z = torch.randn(M, N, device="cuda", dtype=bf)
    o = torch.zeros(N, device="cuda")
    t = timeit(lambda: ops.colsum(z, out=o), iters=30)
    print(f"colsum N={N:5d}             {t*1e6:7.1f} us  {z.numel()*2/t/1e12:5.2f} TB/s")
z = torch.randn(M, D, device="cuda", dtype=bf)
t = timeit(lambda: ops.dropout(z, 0.4, 5), iters=30)
print(f"dropout                    {t*1e6:7.1f} us  {2*z.numel()*2/t/1e12:5.2f} TB/s")
t = timeit(lambda: z.clone(), iters=30)
print(f"torch clone (copy)         {t*1e6:7.1f} us  {2*z.numel()*2/t/1e12:5.2f} TB/s")
