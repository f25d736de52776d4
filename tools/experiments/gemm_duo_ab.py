"""A/B of the persistent GEMM's two forms on the (shape, epilogue) combinations of a training step, in one process:
MDT_GEMM_DUO=0 (one 8-wave workgroup per CU, 256 x 256 tiles) against MDT_GEMM_DUO=1 (two independent 4-wave
workgroups per CU, 128 x 256 tiles).  Outputs must be bit-identical (same k order per element).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import _lib as L, ops  # noqa: E402

bf = torch.bfloat16
dev = "cuda"
VARIANTS = [(0, None), (1, 0), (1, None), (1, 40)]


def timeit(fn, iters=20, warm=5):
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


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    L.reload_env()


def combos(M):
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, dtype=bf, generator=g)
    x768, x2304, x3072 = r(M, 768), r(M, 2304), r(M, 3072)
    out = []
    # name, a, b, kwargs
    out.append(("qkv fwd      bias", x768, r(2304, 768), dict(bias=r(2304))))
    out.append(("o fwd        bias+res+drop", x768, r(768, 768), dict(bias=r(768), residual=r(M, 768), drop_p=0.4, drop_seed=3)))
    out.append(("fc1 fwd      bias+gelu+auxgrad", x768, r(3072, 768), dict(bias=r(3072), aux=torch.empty(M, 3072, device=dev, dtype=bf),
                                                                         epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)))
    out.append(("fc2 fwd      bias+res+drop", x3072, r(768, 3072), dict(bias=r(768), residual=r(M, 768), drop_p=0.4, drop_seed=4)))
    out.append(("dqkv NT      plain", x2304, r(2304, 768), dict(trans_b=True)))
    out.append(("do NT        plain", x768, r(768, 768), dict(trans_b=True)))
    out.append(("dfc1 NT      res", x3072, r(3072, 768), dict(trans_b=True, residual=r(M, 768))))
    out.append(("dfc2 NT      mulaux+colsum", x768, r(768, 3072), dict(trans_b=True, aux=r(M, 3072), epilogue=ops.EPI_MULAUX,
                                                                     colsum=torch.zeros(3072, device=dev))))
    return out


def main():
    Ms = [106496, 102912] if "--both" in sys.argv else [106496]
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = a.split("=", 1)[1]
    for M in Ms:
        print(f"== M = {M}")
        tot = [0.0] * len(VARIANTS)
        for name, a, b, kw in combos(M):
            if only and only not in name:
                continue
            N = b.shape[1] if kw.get("trans_b") else b.shape[0]
            K = a.shape[1]
            res = []
            for duo, skew in VARIANTS:
                setenv(MDT_GEMM_DUO=duo, MDT_GEMM_DUO_SKEW=skew)
                kw2 = dict(kw)
                if "aux" in kw and not (kw.get("epilogue", 0) & ops.EPI_MULAUX):
                    kw2["aux"] = torch.empty_like(kw["aux"])
                if "colsum" in kw:
                    kw2["colsum"] = torch.zeros_like(kw["colsum"])
                o = ops.gemm(a, b, **kw2)
                torch.cuda.synchronize()
                keep = [o.clone()]
                if "aux" in kw2 and not (kw.get("epilogue", 0) & ops.EPI_MULAUX):
                    keep.append(kw2["aux"].clone())
                if "colsum" in kw2:
                    keep.append(kw2["colsum"].clone())
                out = torch.empty_like(o)
                t = timeit(lambda: ops.gemm(a, b, out=out, **kw2))
                res.append((t, keep))
                del o, out
            same = all(all(torch.equal(x, y) if x.dtype != torch.float32 else torch.allclose(x, y, rtol=1e-3, atol=1e-2)
                           for x, y in zip(res[0][1], r[1])) for r in res[1:])
            fl = 2.0 * M * N * K
            for i, r in enumerate(res):
                tot[i] += r[0]
            cells = " | ".join(f"{r[0]*1e6:7.1f} us {fl/r[0]/1e12:5.0f} TF/s x{res[0][0]/r[0]:.3f}" for r in res)
            print(f"{name:34s} N={N:5d} K={K:5d}  {cells}  {'same' if same else 'DIFFERENT'}", flush=True)
            del res
        print("variants (MDT_GEMM_DUO, MDT_GEMM_DUO_SKEW):", VARIANTS, " sums (ms):", [round(t * 1e3, 3) for t in tot])
    setenv(MDT_GEMM_DUO=None, MDT_GEMM_DUO_SKEW=None)


if __name__ == "__main__":
    main()
