"""The three 8-bit GEMM sites of a training step (QKV forward, GELU forward with its saved derivative, fc2's input gradient with
saved-derivative multiply + column sums) on the 8-wave 16x16x32 kernel (MDT_GEMM_F8W=0), the 4-wave 16x16x128 kernel (=1) and
the bf16 kernels, back to back in one process.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, ".")
from tools.kbench import timeit  # noqa: E402
from multimodaldiscussiontransformer_amd import _lib as L, ops  # noqa: E402

bf = torch.bfloat16
M = int(os.environ.get("M", 106496))
g = torch.Generator(device="cuda").manual_seed(3)
r = lambda *s: torch.randn(*s, device="cuda", dtype=bf, generator=g)
sc = lambda x, fmax: (fmax / x.abs().max().float()).reshape(1)


def arm(f8w):
    os.environ["MDT_GEMM_F8W"] = str(f8w)
    L.reload_env()


x, dy = r(M, 768), r(M, 768) * 0.1
wq, w1, w2t = r(2304, 768) * 0.05, r(3072, 768) * 0.05, r(3072, 768) * 0.05
bq, b1 = r(2304), r(3072)
saved = r(M, 3072)
aux = torch.empty(M, 3072, device="cuda", dtype=bf)
cs = torch.zeros(3072, device="cuda")
sx, sd = sc(x, 448.0), sc(dy, 57344.0)
x8, d8 = ops.fp8_quantize(x, 0, scale=sx), ops.fp8_quantize(dy, 1, scale=sd)
q = {n: (ops.fp8_quantize(w, 0, scale=sc(w, 448.0)), 1 / sc(w, 448.0)) for n, w in (("q", wq), ("1", w1), ("2", w2t))}
sites = [
    ("qkv fwd (bias)", 2304, lambda: ops.gemm(x, wq, bias=bq), lambda: ops.gemm_fp8(x8, q["q"][0], 1 / sx, q["q"][1], bias=bq)),
    ("fc1 fwd (bias + gelu + saved)", 3072, lambda: ops.gemm(x, w1, bias=b1, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD),
     lambda: ops.gemm_fp8(x8, q["1"][0], 1 / sx, q["1"][1], bias=b1, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)),
    ("fc2 dgrad (x saved + colsum)", 3072, lambda: ops.gemm(dy, w2t.t().contiguous(), trans_b=True, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs),
     lambda: ops.gemm_fp8(d8, q["2"][0], 1 / sd, q["2"][1], a_format=1, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs)),
]
# round 3: the K = 4 D sites fed by their producers, and what writing the fp8 copy costs the producers
h = r(M, 3072)
w2, b2, resid = r(768, 3072) * 0.03, r(768), r(M, 768)
w1t = r(768, 3072) * 0.05                      # fc1's weight transposed: the k-contiguous B operand of dX = dU W1
du = r(M, 3072) * 0.1
sh, sdu = sc(h, 448.0), sc(du, 57344.0)
h8, du8 = ops.fp8_quantize(h, 0, scale=sh), ops.fp8_quantize(du, 1, scale=sdu)
q["2f"] = (ops.fp8_quantize(w2, 0, scale=sc(w2, 448.0)), 1 / sc(w2, 448.0))
q["1t"] = (ops.fp8_quantize(w1t, 0, scale=sc(w1t, 448.0)), 1 / sc(w1t, 448.0))
out8 = torch.empty(M, 3072, device="cuda", dtype=torch.uint8)
amax = torch.zeros(1, device="cuda")
sites += [
    ("fc2 fwd K3072 (bias+drop+res)", 768 * 4, lambda: ops.gemm(h, w2, bias=b2, residual=resid, drop_p=0.1, drop_seed=5),
     lambda: ops.gemm_fp8(h8, q["2f"][0], 1 / sh, q["2f"][1], bias=b2, residual=resid, drop_p=0.1, drop_seed=5)),
    ("fc1 dgrad K3072 (res)", 768 * 4, lambda: ops.gemm(du, w1t.t().contiguous(), trans_b=True, residual=resid),
     lambda: ops.gemm_fp8(du8, q["1t"][0], 1 / sdu, q["1t"][1], a_format=1, residual=resid)),
    ("fc1 fwd + e4m3 copy of h", 3072, lambda: ops.gemm(x, w1, bias=b1, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD),
     lambda: (ops.gemm_fp8(x8, q["1"][0], 1 / sx, q["1"][1], bias=b1, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, q8_out=out8, q8_format=0, q8_scale=sh, q8_amax=amax)
              if os.environ["MDT_GEMM_F8W"] == "1" else ops.gemm_fp8(x8, q["1"][0], 1 / sx, q["1"][1], bias=b1, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD))),
    ("fc2 dgrad + e5m2 copy of du", 3072, lambda: ops.gemm(dy, w2t.t().contiguous(), trans_b=True, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs),
     lambda: (ops.gemm_fp8(d8, q["2"][0], 1 / sd, q["2"][1], a_format=1, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs, q8_out=out8, q8_format=1, q8_scale=sdu, q8_amax=amax)
              if os.environ["MDT_GEMM_F8W"] == "1" else ops.gemm_fp8(d8, q["2"][0], 1 / sd, q["2"][1], a_format=1, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs))),
]
for name, n, f_bf, f_8 in sites:
    fl = 2.0 * M * n * 768
    t_b = timeit(f_bf)
    res = []
    for w in (0, 1, 0, 1):
        arm(w)
        res.append(timeit(f_8))
    print(f"{name:32s} bf16 {t_b*1e6:7.1f} us {fl/t_b/1e12:6.0f} TF/s | 8-wave fp8 {res[0]*1e6:7.1f} {res[2]*1e6:7.1f} us | 4-wave block-MFMA fp8 "
          f"{res[1]*1e6:7.1f} {res[3]*1e6:7.1f} us {fl/res[3]/1e12:6.0f} TF/s  x{res[2]/res[3]:.2f} over 8-wave, x{t_b/res[3]:.2f} over bf16", flush=True)
