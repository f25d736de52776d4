"""The fp8 model run several times per route (operands quantised by their producers / by stand-alone passes) from identical weights and
seeds: logits and gradients of every run against the first of its route and against the other route.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import fp8  # noqa: E402
from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy  # noqa: E402
from multimodaldiscussiontransformer_amd.data.packer import pack_batch  # noqa: E402
from multimodaldiscussiontransformer_amd.models import GraphormerModel  # noqa: E402
from tests.test_oracle_golden import full_case  # noqa: E402
from tests.util_model import fill_hash_weights, model_args  # noqa: E402


def run(fused, steps=3):
    fname, hp, trees, over = full_case("C2")
    fp8.FUSED_Q = fused
    torch.manual_seed(11)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model, overrides=over)
    model = model.cuda().bfloat16().train()
    model.prepare_main_grads()
    st = model.enable_fp8()
    names = {id(p): n for n, p in model.named_parameters()}
    try:
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
        per_step = []
        for step in range(steps):
            torch.manual_seed(100 + step)
            model.zero_main_grads()
            loss, n, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
            torch.cuda.synchronize()
            per_step.append((float(loss), model.main_grad_flat.clone(), st.scale[:len(st.sites)].clone(), st.amax[:len(st.sites)].clone()))
        with torch.no_grad():
            logits, _ = model(pb.batched_data)
        torch.cuda.synchronize()
        slots = {}
        for p in model.parameters():
            mg = getattr(p, "main_grad", None)
            if mg is not None:
                slots[names[id(p)]] = ((mg.data_ptr() - model.main_grad_flat.data_ptr()) // 4, mg.numel())
        return logits.float().cpu(), per_step, slots, list(st.sites)
    finally:
        fp8.ACTIVE = None


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ref = {}
    for rep in range(reps):
        for fused in (True, False):
            lg, per_step, slots, sites = run(fused)
            tag = "fused" if fused else "alone"
            if fused not in ref:
                ref[fused] = (lg, per_step)
                print(f"[{tag} {rep}] reference: logits sum {float(lg.double().sum()):.10f}, losses {[round(s[0], 6) for s in per_step]}", flush=True)
                continue
            rl, rs = ref[fused]
            msg = [f"logits max|d| {float((lg - rl).abs().max()):.3e}"]
            for k, (a, b) in enumerate(zip(per_step, rs)):
                gd = float((a[1] - b[1]).norm() / b[1].norm())
                sd = int((a[2] != b[2]).sum())
                msg.append(f"step {k}: loss d {a[0] - b[0]:+.2e} grad rel {gd:.2e} scales differing {sd}")
                if gd > 1e-5 and k == 0:
                    rows = sorted(((float((a[1][o:o + n] - b[1][o:o + n]).norm() / (b[1][o:o + n].norm() + 1e-30)), nm) for nm, (o, n) in slots.items()), reverse=True)
                    msg.append("worst " + "; ".join(f"{nm.replace('encoder.graph_encoder.', '')} {d:.2e}" for d, nm in rows[:4]))
                if sd and k < 2:
                    bad = (a[2] != b[2]).nonzero().flatten().tolist()
                    msg.append("sites " + str([str(sites[i])[:60] for i in bad[:4]]))
            print(f"[{tag} {rep}] " + " | ".join(msg), flush=True)
    lf, ls = ref[True][0], ref[False][0]
    print(f"fused vs alone (first runs): logits max|d| {float((lf - ls).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
