"""Context only: torch.matmul (hipBLASLt / rocBLAS) on the bench's GEMM shapes, same box, same random data."""
import sys
import torch
sys.path.insert(0, ".")
from tools.kbench import timeit
from multimodaldiscussiontransformer_amd import ops

bf = torch.bfloat16
for M in (212992, 106496):
    for (n, k, name) in [(2304, 768, "qkv fwd"), (768, 768, "out fwd"), (3072, 768, "ffn1 fwd"), (768, 3072, "ffn2 fwd")]:
        a = torch.randn(M, k, device="cuda", dtype=bf)
        w = torch.randn(n, k, device="cuda", dtype=bf)
        out = torch.empty(M, n, device="cuda", dtype=bf)
        t_v = timeit(lambda: torch.matmul(a, w.t(), out=out))
        t_o = timeit(lambda: ops.gemm(a, w, out=out))
        fl = 2 * M * n * k
        print(f"M={M} {name:9s}: vendor {fl/t_v/1e12:7.1f} TF/s   ours {fl/t_o/1e12:7.1f} TF/s", flush=True)
    # wgrad shape
    dy = torch.randn(M, 3072, device="cuda", dtype=bf); x = torch.randn(M, 768, device="cuda", dtype=bf)
    c = torch.zeros(3072, 768, device="cuda", dtype=torch.float32)
    t_v = timeit(lambda: torch.matmul(dy.t(), x))
    t_o = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, out=c, epilogue=ops.EPI_ATOMIC, split_k=7))
    fl = 2 * M * 3072 * 768
    print(f"M={M} ffn1 wgrad: vendor {fl/t_v/1e12:7.1f} TF/s (bf16 out)   ours {fl/t_o/1e12:7.1f} TF/s (fp32 accumulate into the arena)", flush=True)

# the same launches WITH the work their epilogues carry in a training step: the vendor path needs separate elementwise passes
import torch.nn.functional as F
M = 106496
a = torch.randn(M, 768, device="cuda", dtype=bf)
w1, b1 = torch.randn(3072, 768, device="cuda", dtype=bf), torch.randn(3072, device="cuda", dtype=bf)
aux = torch.empty(M, 3072, device="cuda", dtype=bf); out = torch.empty(M, 3072, device="cuda", dtype=bf)
t_v = timeit(lambda: F.gelu(torch.addmm(b1, a, w1.t())))
t_o = timeit(lambda: ops.gemm(a, w1, bias=b1, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD))
fl = 2 * M * 3072 * 768
print(f"M={M} ffn1 fwd + bias + GELU: vendor addmm + F.gelu {fl/t_v/1e12:7.1f} TF/s (no saved derivative)   ours, one launch, + saved derivative {fl/t_o/1e12:7.1f} TF/s")
wo, bo = torch.randn(768, 768, device="cuda", dtype=bf), torch.randn(768, device="cuda", dtype=bf)
res = torch.randn(M, 768, device="cuda", dtype=bf); o2 = torch.empty(M, 768, device="cuda", dtype=bf)
t_v = timeit(lambda: res + F.dropout(torch.addmm(bo, a, wo.t()), 0.4, True))
t_o = timeit(lambda: ops.gemm(a, wo, bias=bo, residual=res, out=o2, drop_p=0.4, drop_seed=3))
fl = 2 * M * 768 * 768
print(f"M={M} out fwd + bias + dropout + residual: vendor addmm + dropout + add {fl/t_v/1e12:7.1f} TF/s   ours, one launch {fl/t_o/1e12:7.1f} TF/s")
