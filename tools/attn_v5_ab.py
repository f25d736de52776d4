"""One-pass attention backward on long rows: v4 (one workgroup per item, MDT_ATTN_ONEPASS=4) against the persistent v5
(default); gradients must be bit-identical.  GPU box only."""
import hashlib
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


def run(name, nseq, S, lens=None, p=0.3, q_limit=0):
    H, hd = 12, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    kw = {}
    rows = nseq * S
    if lens is not None:
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
    for v in (4, None):
        setenv(v)
        d = ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw)[0]
        torch.cuda.synchronize()
        t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw), iters=30)
        res.append((t, d.clone()))
    setenv(None)
    same = all(torch.equal(res[0][1], r[1]) for r in res[1:])
    print(f"{name:30s} p={p}: v4 {res[0][0]*1e6:7.1f} us | v5 {res[1][0]*1e6:7.1f} us  x{res[0][0]/res[1][0]:.3f}  identical {same}", flush=True)


torch.manual_seed(0)
run("bert 2048 x 104 padded", 2048, 104)
tl = torch.randint(10, 103, (2048,), generator=torch.Generator().manual_seed(1)).to(torch.int32)
run("bert 2048 ragged 10-102", 2048, 104, lens=tl)
tl2 = torch.randint(65, 103, (1000,), generator=torch.Generator().manual_seed(2)).to(torch.int32)
run("bert 1000 ragged 65-102", 1000, 104, lens=tl2)
run("vit 512 x 197", 512, 197)
run("vit 512 x 197", 512, 197, p=0.0)
run("vit 37 x 197 (tail)", 37, 197)
run("rows 300 x 208", 300, 208)
run("rows 300 x 130 (9 tiles)", 300, 130)
lens = torch.randint(120, 209, (400,), generator=torch.Generator().manual_seed(1)).to(torch.int32)
run("ragged 400 x 120-208", 400, 208, lens=lens)
run("vit 512 x 197 q_limit 16", 512, 197, q_limit=16)
