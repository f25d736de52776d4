"""8-bit operands against bf16 on the encoder's big forward GEMM shapes, same box, same call (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from tools.kbench import timeit
from multimodaldiscussiontransformer_amd import ops

bf = torch.bfloat16
M = 110592
for (n, k, name) in [(2304, 768, "qkv fwd"), (768, 768, "out fwd"), (3072, 768, "ffn1 fwd"), (768, 3072, "ffn2 fwd")]:
    a = torch.randn(M, k, device="cuda", dtype=bf)
    w = torch.randn(n, k, device="cuda", dtype=bf) * 0.05
    bias = torch.randn(n, device="cuda", dtype=bf)
    out = torch.empty(M, n, device="cuda", dtype=bf)
    sa, sw = (448.0 / a.abs().max().float()).reshape(1), (448.0 / w.abs().max().float()).reshape(1)
    ia, iw = 1 / sa, 1 / sw
    a8, w8 = ops.fp8_quantize(a, 0, scale=sa), ops.fp8_quantize(w, 0, scale=sw)
    amax = torch.zeros(1, device="cuda")
    t_b = timeit(lambda: ops.gemm(a, w, bias=bias, out=out))
    t_8 = timeit(lambda: ops.gemm_fp8(a8, w8, ia, iw, bias=bias, out=out))
    t_q = timeit(lambda: ops.fp8_quantize(a, 0, scale=sa, amax=amax, out=a8))
    fl = 2 * M * n * k
    print(f"M={M} {name:9s}: bf16 {fl/t_b/1e12:7.1f} TF/s ({t_b*1e6:7.1f} us)   fp8 {fl/t_8/1e12:7.1f} TF/s ({t_8*1e6:7.1f} us)   "
          f"quantise A {t_q*1e6:6.1f} us ({M*k*3/t_q/1e12:.2f} TB/s)   fp8 + quantise vs bf16: {t_b/(t_8+t_q):.2f}x", flush=True)
