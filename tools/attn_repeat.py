"""Bitwise repeatability of the attention backward on a ragged, length-binned launch set (the production text path): any
difference between repetitions, or a non-finite value, is a race or an uninitialised read.  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import ops  # noqa: E402
from multimodaldiscussiontransformer_amd.data.packer import RaggedText  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    nseq, H, hd = 256, 12, 64
    D = H * hd
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(12, 105, (nseq,), generator=g, dtype=torch.int64)
    off = torch.zeros(nseq + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(lens, 0).int()
    rows = int(off[-1])
    sl, order = torch.sort(lens, stable=True)
    qkv = (torch.randn(rows, 3 * D, generator=g) * 0.7).to(torch.bfloat16).cuda()
    dout = torch.randn(rows, D, generator=g).to(torch.bfloat16).cuda()
    S = int(lens.max())
    bins = []
    lo = 0
    for cap in (64, S):
        hi = int(np.searchsorted(sl.numpy(), cap, side="right")) if cap != S else nseq
        if hi > lo:
            bins.append((order[lo:hi].int().cuda(), cap))
        lo = hi
    qlim = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    kw = dict(drop_p=p, drop_seed=11, seq_offsets=off.cuda(), bins=bins)
    if qlim:
        kw["q_limit"] = qlim
        keep = torch.zeros(rows, dtype=torch.bool)
        for s_ in range(nseq):
            keep[int(off[s_]):int(off[s_]) + min(qlim, int(lens[s_]))] = True
        dout = dout * keep.cuda()[:, None]            # rows the forward skipped carry no gradient
    out, lse = ops.attention_fwd(qkv, nseq, S, H, **kw)
    # the two forms of delta against each other (same launch set): close everywhere, finite
    from multimodaldiscussiontransformer_amd import _lib
    res = {}
    for e in ("1", "0"):
        os.environ["MDT_ATTN_EXACT_DELTA"] = e
        _lib.reload_env()
        res[e] = ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, **kw)[0].float()
    os.environ.pop("MDT_ATTN_EXACT_DELTA")
    _lib.reload_env()
    dd = (res["1"] - res["0"]).abs()
    print(f"q_limit {qlim}: delta summed in the kernel vs from the bf16 output: max |diff| {float(dd.max()):.3e} at |g| max {float(res['0'].abs().max()):.3e}, "
          f"rel-L2 {float(dd.norm() / res['0'].norm()):.3e}, finite {bool(torch.isfinite(res['1']).all())} / {bool(torch.isfinite(res['0']).all())}", flush=True)
    if float(dd.max()) > 0.05 * float(res["0"].abs().max()):
        idx = (dd > 0.05 * float(res["0"].abs().max())).nonzero()
        seqs = sorted({int(np.searchsorted(off.numpy(), int(r_), side="right") - 1) for r_ in idx[:2000, 0].tolist()})
        print(f"  LARGE differences in {idx.shape[0]} elements: sequences {seqs[:12]} lengths {[int(lens[s]) for s in seqs[:12]]} rows-in-seq "
              f"{sorted({int(r_) - int(off[np.searchsorted(off.numpy(), int(r_), side='right') - 1]) for r_ in idx[:200, 0].tolist()})[:20]} columns {int(idx[:, 1].min())}-{int(idx[:, 1].max())}", flush=True)
    ref = None
    bad = nonfinite = 0
    for r in range(reps):
        d, _ = ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, **kw)
        if not bool(torch.isfinite(d.float()).all()):
            nonfinite += 1
        if ref is None:
            ref = d.clone()
        elif not torch.equal(d.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            if bad <= 3:
                diff = (d.float() - ref.float()).abs()
                idx = diff.nonzero()
                seqs = sorted({int(np.searchsorted(off.numpy(), int(r_), side="right") - 1) for r_ in idx[:2000, 0].tolist()})
                print(f"  rep {r}: {int((diff > 0).sum())} elements differ (max {float(diff.max()):.3e}), sequences {seqs[:10]} lengths {[int(lens[s]) for s in seqs[:10]]}, "
                      f"columns {int(idx[:, 1].min())}-{int(idx[:, 1].max())}", flush=True)
    print(f"attention backward, {nseq} ragged sequences in {len(bins)} length bins, dropout {p}, MDT_ATTN_EXACT_DELTA={os.environ.get('MDT_ATTN_EXACT_DELTA', '1')}: "
          f"{bad} of {reps - 1} repetitions differ, {nonfinite} with non-finite values", flush=True)


if __name__ == "__main__":
    main()
