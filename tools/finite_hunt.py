"""Find the first kernel launch of a training step whose output holds a non-finite (or absurdly large) value.  Every function of
multimodaldiscussiontransformer_amd.ops is wrapped: after the launch, every tensor it returned or was handed as an output is
checked (device sync per launch: slow, timing changes — run a second copy beside it to keep the contention that made round 3's
intermittent garbage tiles show: DESIGN.md §4).  GPU box only:  python tools/finite_hunt.py [--reps 30] [--no-sync-check]"""
import argparse
import functools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = [sys.argv[0]], sys.argv[1:]
import bench  # noqa: E402
from multimodaldiscussiontransformer_amd import ops  # noqa: E402

LIMIT = 1e6
log = []


def bad(t):
    if not torch.is_tensor(t) or not t.is_cuda or not t.is_floating_point() or t.numel() == 0:
        return False
    f = t.float() if t.dtype != torch.float32 else t
    return (not bool(torch.isfinite(f).all())) or float(f.abs().max()) > LIMIT


def wrap(name, fn):
    @functools.wraps(fn)
    def w(*a, **kw):
        r = fn(*a, **kw)
        outs = list(r) if isinstance(r, (tuple, list)) else [r]
        outs += [kw.get(k) for k in ("out", "aux", "dx", "dgamma", "dbeta", "colsum", "asum", "d_sp_table", "d_virt")]
        if name == "row_axpby":
            outs.append(a[0])
        if name == "attention_fwd" and kw.get("q_limit", 0):
            outs = []         # rows beyond the query limit are unspecified by contract (nobody reads them): not a finding
        for i, t in enumerate(outs):
            if bad(t):
                ins = [(tuple(x.shape), str(x.dtype).replace("torch.", ""), bad(x)) for x in list(a) + list(kw.values()) if torch.is_tensor(x)]
                where = ""
                if name == "gemm" and t.dim() == 2:
                    first = t.clone()
                    cs = kw.get("colsum")
                    if cs is not None:
                        kw = dict(kw, colsum=torch.zeros_like(cs))
                    again = fn(*a, **dict(kw, out=torch.empty_like(t)))          # the same launch once more
                    torch.cuda.synchronize()
                    d = (first.float() - again.float())
                    nz = (d != 0) | ~torch.isfinite(d)
                    idx = nz.nonzero()
                    tiles = sorted({(int(r) // 256, int(c) // 256) for r, c in idx[:: max(1, idx.shape[0] // 5000)].tolist()})
                    where = (f"; against the same launch repeated: {idx.shape[0]} elements differ, rows {int(idx[:, 0].min())}-{int(idx[:, 0].max())}, cols "
                             f"{int(idx[:, 1].min())}-{int(idx[:, 1].max())}, 256x256 tiles (row, col) {tiles[:16]}{' ...' if len(tiles) > 16 else ''}; repeat bad? {bad(again)}; "
                             f"sample first/again {first[idx[0, 0], idx[0, 1]].item()} / {again[idx[0, 0], idx[0, 1]].item()}")
                raise RuntimeError(f"FIRST BAD OUTPUT: ops.{name} output #{i} shape {tuple(t.shape)} {t.dtype}; launch #{len(log)}{where}; "
                                   f"inputs (shape, dtype, bad?) {ins}; scalar kwargs { {k: v for k, v in kw.items() if not torch.is_tensor(v) and not isinstance(v, (list, tuple))} }")
        log.append(name)
        return r
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--tag", default="")
    a = ap.parse_args(argv)
    from multimodaldiscussiontransformer_amd import engine, synthetic
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    for n in dir(ops):
        f = getattr(ops, n)
        if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__ and isinstance(f, type(main)):
            setattr(ops, n, wrap(n, f))
    args = argparse.Namespace(config="base", num_fusion_layers=5, freeze_initial_encoders=False, dropout=0.4, attention_dropout=0.3, act_dropout=0.3)
    torch.manual_seed(1234)
    model = GraphormerModel.build_model(bench.base_args(args), task=None).cuda().bfloat16()
    model.train()
    model.prepare_main_grads()
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
    trees = synthetic.make_trees(4, 32, seed=1234, seq_len=100, image_frac=0.25, image_size=224, shape="bushy")
    pb = pack_batch(trees, spatial_pos_max=5)
    for rep in range(a.reps):
        torch.manual_seed(4242)
        model.zero_main_grads()
        log.clear()
        try:
            loss, n, lg = crit(model, {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
        except RuntimeError as e:
            print(f"[{a.tag}] rep {rep}: {e}", flush=True)
            print(f"[{a.tag}] launches before it: ... {log[-12:]}", flush=True)
            return
        torch.cuda.synchronize()
        g = model.main_grad_flat
        if bad(g):
            print(f"[{a.tag}] rep {rep}: arena bad at the END of the step although no checked launch output was ({len(log)} launches)", flush=True)
            return
    print(f"[{a.tag}] {a.reps} steps clean ({len(log)} checked launches per step)", flush=True)


if __name__ == "__main__":
    main()
