"""Numerics of the big-tile GEMM paths against torch (GPU box)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402

torch.manual_seed(0)
bf = torch.bfloat16
for (M, N, K) in [(16640, 1024, 256), (16677, 1024, 256), (16677, 1024, 768), (70000, 768, 128), (66000, 3072, 768)]:
    a = torch.randn(M, K, device="cuda", dtype=bf)
    b = (torch.randn(N, K, device="cuda") * 0.2).to(bf)
    bias = torch.randn(N, device="cuda", dtype=bf)
    ref = a.float() @ b.float().t()
    out = ops.gemm(a, b)
    e0 = (out.float() - ref).abs().max().item()
    out = ops.gemm(a, b, bias=bias)
    e1 = (out.float() - ref - bias.float()).abs().max().item()
    m = ops.dropout_mask(M * N, 0.3, 99).view(M, N).float() / 0.7
    out = ops.gemm(a, b, bias=bias, drop_p=0.3, drop_seed=99)
    e2 = (out.float() - (ref + bias.float()) * m).abs().max().item()
    aux = torch.empty(M, N, device="cuda", dtype=bf)
    out = ops.gemm(a, b, bias=bias, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=0.3, drop_seed=99)
    e3 = (out.float() - F.gelu(ref + bias.float()) * m).abs().max().item()
    bt = b.t().contiguous()
    g = torch.randn(M, N, device="cuda", dtype=bf)
    out = ops.gemm(g, b.t().contiguous().t().contiguous(), trans_b=True) if False else ops.gemm(a, bt, trans_b=True)
    e4 = (out.float() - ref).abs().max().item()
    print(f"M={M} N={N} K={K} persist={os.environ.get('MDT_GEMM_PERSIST', '1')}: plain {e0:.3f} bias {e1:.3f} dropout {e2:.3f} gelu+auxgrad+dropout {e3:.3f} NT {e4:.3f}  (|ref| max {ref.abs().max().item():.1f})", flush=True)
