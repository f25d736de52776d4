"""Micro-benchmarks of the individual kernels at mDT-base (C2) shapes. GPU box only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = "cuda"
    bf = torch.bfloat16
    M = 212992
    import os
    print("== GEMM bf16 (default dispatch) ==")
    gemm_section(M, dev, bf)
    if "--gemm-only" not in sys.argv:
        rest(M, dev, bf)


def gemm_section(M, dev, bf):
    for (m, n, k, ta, tb, name) in [
        (M, 2304, 768, 0, 0, "qkv fwd"), (M, 768, 768, 0, 0, "out fwd"), (M, 3072, 768, 0, 0, "ffn1 fwd"),
        (M, 768, 3072, 0, 0, "ffn2 fwd"), (M, 768, 2304, 0, 1, "qkv dgrad"), (M, 768, 3072, 0, 1, "ffn1 dgrad"),
        (M, 3072, 768, 0, 1, "ffn2 dgrad"),
    ]:
        a = torch.randn(m, k, device=dev, dtype=bf)
        b = torch.randn(k, n, device=dev, dtype=bf) if tb else torch.randn(n, k, device=dev, dtype=bf)
        out = torch.empty(m, n, device=dev, dtype=bf)
        t = timeit(lambda: ops.gemm(a, b, trans_b=bool(tb), out=out))
        print(f"{name:12s} M={m} N={n} K={k}: {t*1e3:8.3f} ms  {2*m*n*k/t/1e12:7.1f} TF/s")
        del a, b, out
    from multimodaldiscussiontransformer_amd.engine import _split_k
    for (n, k, name) in [(2304, 768, "qkv wgrad"), (768, 768, "out wgrad"), (3072, 768, "ffn1 wgrad"),
                         (768, 3072, "ffn2 wgrad")]:
        sk = _split_k(n, k, M)
        dy = torch.randn(M, n, device=dev, dtype=bf)
        x = torch.randn(M, k, device=dev, dtype=bf)
        c = torch.zeros(n, k, device=dev, dtype=torch.float32)
        t = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, out=c, epilogue=ops.EPI_ATOMIC, split_k=sk))
        print(f"{name:12s} N={n} K={k} red={M} split{sk}: {t*1e3:8.3f} ms  {2*M*n*k/t/1e12:7.1f} TF/s")
        del dy, x, c
    print("-- epilogues (ffn1 fwd bias+gelu+aux) --")
    a = torch.randn(M, 768, device=dev, dtype=bf); b = torch.randn(3072, 768, device=dev, dtype=bf)
    bias = torch.randn(3072, device=dev, dtype=bf); aux = torch.empty(M, 3072, device=dev, dtype=bf)
    out = torch.empty(M, 3072, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm(a, b, bias=bias, aux=aux, out=out, epilogue=ops.EPI_GELU))
    print(f"ffn1 gelu+aux: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.gemm(a, b, bias=bias, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_DROPOUT, drop_p=0.3, drop_seed=5))
    print(f"ffn1 gelu+aux+dropout: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    dy = torch.randn(M, 768, device=dev, dtype=bf); w2 = torch.randn(768, 3072, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm(dy, w2, trans_b=True, aux=aux, out=out, epilogue=ops.EPI_DGELU | ops.EPI_DROPOUT, drop_p=0.3, drop_seed=5))
    print(f"ffn2 dgrad dgelu+dropout: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.gemm(a, b, bias=bias, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=0.3, drop_seed=5))
    print(f"ffn1 gelu+AUX_GRAD+dropout: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.gemm(dy, w2, trans_b=True, aux=aux, out=out, epilogue=ops.EPI_MULAUX))
    print(f"ffn2 dgrad MULAUX: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    res = torch.randn(M, 768, device=dev, dtype=bf); w3 = torch.randn(768, 3072, device=dev, dtype=bf); b3 = torch.randn(768, device=dev, dtype=bf)
    o3 = torch.empty(M, 768, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm(out, w3, bias=b3, residual=res, out=o3, epilogue=ops.EPI_DROPOUT, drop_p=0.4, drop_seed=5))
    print(f"ffn2 fwd bias+dropout+residual: {t*1e3:8.3f} ms {2*M*3072*768/t/1e12:7.1f} TF/s")
    del a, b, bias, aux, out


def rest(M, dev, bf):
    print("== attention bf16 ==")
    for (nseq, S, name) in [(2048, 104, "bert"), (512, 201, "vit"), (32, 65, "graph")]:
        H, hd = 12, 64
        qkv = torch.randn(nseq * S, 3 * H * hd, device=dev, dtype=bf)
        kw = {}
        if name == "graph":
            kw = dict(attn_bias=torch.zeros(nseq, S, S, device=dev), spatial_pos=torch.randint(1, 22, (nseq, S - 1, S - 1), device=dev, dtype=torch.int32),
                      sp_table=torch.randn(512, H, device=dev, dtype=bf), virt=torch.randn(H, device=dev, dtype=bf),
                      key_pad=torch.zeros(nseq, S, device=dev, dtype=torch.uint8))
        out, lse = ops.attention_fwd(qkv, nseq, S, H, **kw)
        t = timeit(lambda: ops.attention_fwd(qkv, nseq, S, H, **kw))
        fl = 4 * nseq * H * S * S * hd
        print(f"{name:6s} fwd nseq={nseq} S={S}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s (algorithmic)")
        dout = torch.randn(nseq * S, H * hd, device=dev, dtype=bf)
        ex = {}
        if name == "graph":
            ex = dict(d_sp_table=torch.zeros(512, H, device=dev), d_virt=torch.zeros(H, device=dev))
        import os
        os.environ["MDT_ATTN_BWD"] = "v1"
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, **kw, **ex))
        os.environ.pop("MDT_ATTN_BWD")
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        print(f"{name:6s} bwd (v1): {t*1e3:8.3f} ms  {2.5*fl/t/1e12:7.1f} TF/s (algorithmic 10 S^2 d)")
        os.environ["MDT_ATTN_BWD"] = "v2"
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, **kw, **ex))
        os.environ.pop("MDT_ATTN_BWD")
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        print(f"{name:6s} bwd (v2): {t*1e3:8.3f} ms  {2.5*fl/t/1e12:7.1f} TF/s")
        os.environ["MDT_ATTN_BWD"] = "v3"
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, **kw, **ex))
        os.environ.pop("MDT_ATTN_BWD")
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        print(f"{name:6s} bwd (v3): {t*1e3:8.3f} ms  {2.5*fl/t/1e12:7.1f} TF/s")
    print("== layernorm bf16 ==")
    x = torch.randn(M, 768, device=dev, dtype=bf); g = torch.ones(768, device=dev, dtype=bf); bb = torch.zeros(768, device=dev, dtype=bf)
    y, mean, rstd = ops.layernorm_fwd(x, g, bb, 1e-12)
    t = timeit(lambda: ops.layernorm_fwd(x, g, bb, 1e-12, out=y))
    print(f"ln fwd: {t*1e3:8.3f} ms  {2*x.numel()*2/t/1e12:6.2f} TB/s")
    dg = torch.zeros(768, device=dev); db = torch.zeros(768, device=dev); dx = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm_bwd(y, x, g, mean, rstd, dgamma=dg, dbeta=db, dx=dx))
    print(f"ln bwd: {t*1e3:8.3f} ms  {3*x.numel()*2/t/1e12:6.2f} TB/s")
    t = timeit(lambda: ops.colsum(x, out=dg))
    print(f"colsum: {t*1e3:8.3f} ms  {x.numel()*2/t/1e12:6.2f} TB/s")


if __name__ == "__main__":
    main()
