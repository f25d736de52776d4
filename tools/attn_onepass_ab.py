"""A/B of the two-pass (v3) and one-pass (v4) attention backward, dense and ragged rows, with and without dropout;
gradients of the two must agree to bf16 rounding.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import _lib as L, ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

bf = torch.bfloat16


def setenv(v):
    if v is None:
        os.environ.pop("MDT_ATTN_ONEPASS", None)
    else:
        os.environ["MDT_ATTN_ONEPASS"] = str(v)
    L.reload_env()


def run(name, nseq, S, lens=None, p=0.1, q_limit=0):
    H, hd = 12, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    kw = {}
    if lens is None:
        rows = nseq * S
    else:
        off = torch.zeros(nseq + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(lens, 0)
        rows = int(off[-1])
        kw["seq_offsets"] = off.cuda()
    if q_limit:
        kw["q_limit"] = q_limit
    qkv = torch.randn(rows, 3 * H * hd, device="cuda", dtype=bf, generator=g)
    dout = torch.randn(rows, H * hd, device="cuda", dtype=bf, generator=g)
    out, lse = ops.attention_fwd(qkv, nseq, S, H, drop_p=p, drop_seed=5, **kw)
    res = []
    for v in (0, 1):
        setenv(v)
        d = ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw)
        d = d[0] if isinstance(d, (tuple, list)) else d
        torch.cuda.synchronize()
        t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw), iters=20)
        res.append((t, d.float().clone()))
    setenv(None)
    a, b = res[0][1], res[1][1]
    if q_limit:   # rows past the limit are unspecified
        pass
    err = (a - b).abs().max().item()
    rel = ((a - b).norm() / a.norm()).item()
    print(f"{name:28s} p={p}: two-pass {res[0][0]*1e6:8.1f} us | one-pass {res[1][0]*1e6:8.1f} us  x{res[0][0]/res[1][0]:.3f}   "
          f"max |diff| {err:.3e} rel-L2 {rel:.2e} finite {bool(torch.isfinite(b).all())}", flush=True)


torch.manual_seed(0)
run("vit 512 x 201", 512, 201, p=0.0)
run("vit 512 x 201", 512, 201, p=0.1)
run("bert 2048 x 104 dense", 2048, 104, p=0.1)
lens = torch.randint(8, 101, (2048,), dtype=torch.int32)
run("bert 2048 ragged 8..100", 2048, 104, lens=lens, p=0.1)
run("bert 2048 ragged 8..100", 2048, 104, lens=lens, p=0.0)
run("large 128 x 261", 128, 261, p=0.1)
run("short 512 x 40", 512, 40, p=0.1)
