"""Timing + checksums of the hot attention launches (ViT rows, padded and ragged BERT rows; forward and the one-pass
backward, dropout on) for A/B runs of two library builds (tools/ab_libs.sh): the checksums must not move.  GPU box only."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

bf = torch.bfloat16


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def run(name, nseq, S, lens=None, p=0.3):
    H, hd = 12, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    kw = {}
    rows = nseq * S
    if lens is not None:
        off = torch.zeros(nseq + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(lens, 0)
        rows = int(off[-1])
        kw["seq_offsets"] = off.cuda()
    qkv = torch.randn(rows, 3 * H * hd, device="cuda", dtype=bf, generator=g)
    dout = torch.randn(rows, H * hd, device="cuda", dtype=bf, generator=g)
    out, lse = ops.attention_fwd(qkv, nseq, S, H, drop_p=p, drop_seed=5, **kw)
    d = ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw)
    d = d[0] if isinstance(d, (tuple, list)) else d
    torch.cuda.synchronize()
    tf = timeit(lambda: ops.attention_fwd(qkv, nseq, S, H, drop_p=p, drop_seed=5, **kw), iters=30)
    tb = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5, **kw), iters=30)
    gb = rows * H * hd * 2 / 1e9
    print(f"{name:26s} rows {rows:7d}  fwd {tf*1e6:7.1f} us ({4*gb/tf/1e3:4.2f} TB/s)  bwd {tb*1e6:7.1f} us ({8*gb/tb/1e3:4.2f} TB/s)  "
          f"sha fwd {digest(out, lse)} bwd {digest(d)}", flush=True)


torch.manual_seed(0)
run("vit 512 x 197", 512, 197)
run("bert 2048 x 104 padded", 2048, 104)
lens = torch.randint(10, 103, (2048,), generator=torch.Generator().manual_seed(1)).to(torch.int32)
run("bert 2048 ragged 10-102", 2048, 104, lens=lens)
short = torch.randint(10, 65, (2048,), generator=torch.Generator().manual_seed(2)).to(torch.int32)
run("bert 2048 ragged 10-64", 2048, 64, lens=short)
