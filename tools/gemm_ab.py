"""A/B of library switches on the (shape, epilogue) combinations of a training step's GEMMs, in one process:
    python tools/gemm_ab.py MDT_GEMM_NT=0 MDT_GEMM_NT=1 MDT_GEMM_NT=3 [--both] [--only=fc1]
Every argument NAME=VALUE[,NAME=VALUE...] is one variant (the first is the baseline); outputs of all variants must be
bit-identical.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import _lib as L, ops  # noqa: E402

bf = torch.bfloat16
dev = "cuda"
VARIANTS = [dict(kv.split("=", 1) for kv in a.split(",")) for a in sys.argv[1:] if "=" in a and not a.startswith("--")] or [{}]
NAMES = sorted({k for v in VARIANTS for k in v})


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
    if "--epi" in sys.argv:      # what the epilogue variants cost on one short-K shape (N = K = 768)
        out = []
        w, wt = r(768, 768), r(768, 768)
        out.append(("NN plain(generic)", x768, w, dict()))
        out.append(("NN bias", x768, w, dict(bias=r(768))))
        out.append(("NN bias+res+drop", x768, w, dict(bias=r(768), residual=r(M, 768), drop_p=0.4, drop_seed=3)))
        out.append(("NT plain", x768, wt, dict(trans_b=True)))
        out.append(("NT res", x768, wt, dict(trans_b=True, residual=r(M, 768))))
    if "--n1024" in sys.argv:    # four column tiles: the 32 workgroups of an XCD are 8 row-panel groups of 4 (MDT_GEMM_DIAG=256)
        out = [("NN bias N1024", x768, r(1024, 768), dict(bias=r(1024))),
               ("NN bias+res+drop N1024", x768, r(1024, 768), dict(bias=r(1024), residual=r(M, 1024), drop_p=0.4, drop_seed=3)),
               ("NN plain N1024 K3072", x3072, r(1024, 3072), dict())]
    if "--wgrad" in sys.argv:    # the split-K weight gradients of a block (dW = dY^T X, fp32 atomics, bias gradient riding): MDT_GEMM_W4=0 | 2
        out = []
        for (nm, dy_, x_) in [("wgrad qkv", x2304, x768), ("wgrad o", x768, x768), ("wgrad fc1", x3072, x768), ("wgrad fc2", x768, x3072)]:
            n_out, k_in = dy_.shape[1], x_.shape[1]
            tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
            split = max(1, min(1024 // tiles, M // 1024))
            out.append((f"{nm} split{split}", dy_, x_, dict(trans_a=True, trans_b=True, epilogue=ops.EPI_ATOMIC, split_k=split,
                                                           asum=torch.zeros(n_out, device=dev), out=torch.zeros(n_out, k_in, device=dev))))
    if "--nn-dgrad" in sys.argv:   # the input-gradient launches against a pre-transposed weight copy (k-contiguous B) beside the k-major form
        out = []
        for (nm, xin, n_out, k_in, extra) in [("dqkv plain", x2304, 768, 2304, {}), ("do plain", x768, 768, 768, {}),
                                              ("dfc1 res", x3072, 768, 3072, dict(residual=r(M, 768))),
                                              ("dfc2 mulaux+colsum", x768, 3072, 768, dict(aux=r(M, 3072), epilogue=ops.EPI_MULAUX, colsum=torch.zeros(3072, device=dev)))]:
            w = r(k_in, n_out)                       # W stored [K][N]: the k-major operand of dX = dY W
            out.append((nm + " NT", xin, w, dict(trans_b=True, **extra)))
            out.append((nm + " NN", xin, w.t().contiguous(), dict(**extra)))
    return out


def main():
    Ms = [106496, 102912] if "--both" in sys.argv else [106496]
    for a in sys.argv[1:]:
        if a.startswith("--M="):
            Ms = [int(x) for x in a.split("=", 1)[1].split(",")]
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
            if kw.get("trans_a"):                 # weight gradient: [rows, N_out]^T [rows, K_in] — flops 2 rows N_out K_in
                wg_rows, wg_n = a.shape
                res = []
                setenv(**{k: VARIANTS[0].get(k) for k in NAMES})
                timeit(lambda: ops.gemm(a, b, **dict(kw)), iters=10)
                for var in VARIANTS:
                    setenv(**{k: var.get(k) for k in NAMES})
                    kw["out"].zero_(); kw["asum"].zero_()
                    ops.gemm(a, b, **dict(kw))
                    torch.cuda.synchronize()
                    keep = [kw["out"].clone(), kw["asum"].clone()]
                    t = timeit(lambda: ops.gemm(a, b, **dict(kw)))
                    res.append((t, keep))
                same = all(all(torch.allclose(x, y, rtol=1e-3, atol=2e-2) for x, y in zip(res[0][1], r[1])) for r in res[1:])
                fl = 2.0 * wg_rows * wg_n * N
                for i, r in enumerate(res):
                    tot[i] += r[0]
                cells = " | ".join(f"{r[0]*1e6:7.1f} us {fl/r[0]/1e12:5.0f} TF/s x{res[0][0]/r[0]:.3f}" for r in res)
                print(f"{name:34s} dW {wg_n:5d}x{N:5d} rows={wg_rows}  {cells}  {'same' if same else 'DIFFERENT'}", flush=True)
                continue
            res = []
            setenv(**{k: VARIANTS[0].get(k) for k in NAMES})      # throw-away pass: the first timed variant of a row otherwise
            timeit(lambda: ops.gemm(a, b, **dict(kw)), iters=10)  # pays the clock ramp / cold caches (3-10 % on these launches)
            for var in VARIANTS:
                setenv(**{k: var.get(k) for k in NAMES})
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
        print("variants:", VARIANTS, " sums (ms):", [round(t * 1e3, 3) for t in tot])
    setenv(**{k: None for k in NAMES})


if __name__ == "__main__":
    main()
