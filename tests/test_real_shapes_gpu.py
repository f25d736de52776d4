"""Model-level parity at the REAL geometries of BASELINE.json (not the toy fixtures of test_model_gpu.py):

  "C2"  configs[1]: BERT-base + ViT-B/16 split 6 + 6, FFN 3072, 6 executed graph layers, L = 100 (S = 104 / ragged),
        224-px images (P = 197, S = 201), one bushy 64-comment tree (T = 65), 25 % image comments — the shapes at
        which the 256x256 persistent GEMM, the split-K weight gradient, attn_bwd_v3's 64-key chunking and the ragged
        offsets engage.  Checked against the REAL reference's outputs (tests/golden/full_c2_real.npz, made by
        oracle/gen_golden.py) and, tensor by tensor, against the oracle.
  "C4"  configs[3] mDT-large shapes: D 1024, 16 heads, FFN 4096, ViT-L/14 (P = 257, S = 261), one 128-comment
        deep-thread tree (T = 129, banded -inf mask).  Layers cut to 2 + 2 so the CPU oracle takes seconds; the
        reference cannot run D != 768, so the oracle (pinned at D = 768 by the goldens) is the checker.
  "C1"  configs[0] EXACTLY, as a full model on the HIP path: Tiny mDT — D 128, BERT-mini split 2 + 2 (2 heads of 64, FFN
        512), 2 executed graph layers (8 heads of 16), text only, 8 bushy 16-comment trees, L = 100.  The reference
        hard-codes 768, so the D-parametric oracle — pinned at D = 768 by the goldens and at
        D = 128 per module by graph_modules_d128.npz — is the checker.
  "C4F" configs[3] at its FULL 12 + 12 depth (24 BERT-large / ViT-L/14 blocks each way, 12 executed graph layers), bf16,
        one 128-comment deep thread: what depth compounding at D = 1024 does to the bf16 path, measured once against the
        fp32 oracle on the same bf16-rounded weights (a ~2-minute CPU pass).
  "M"   tiny shapes, mixed predictions (TP / FP / FN / TN all non-zero).
  "LAUNCH" the configuration the reference SHIPS (sample_run.sh:3 = `run_train.sh 8 4 5 2 2 0`, run_train.sh:41-65): BERT-base +
        ViT-B/16 split 3 + 9, fusion stacks of 2 (uneven last stack), graph stacks of 2 (10 executed graph layers), graph FFN
        768, --freeze_initial_encoders (the frozen prefix gets no gradient and its adjoint never runs), two trees of the
        launch's 12-tree batch at L = 100 / 224 px.  Checked against the REAL reference (tests/golden/full_launch.npz).

fp32: north_star's 1e-3 gate on logits and on EVERY parameter gradient, both text layouts; counters / F1 exact.
bf16: logits within 0.05, every parameter gradient within a relative L2 of BF16_GRAD_REL_L2 of the fp32 oracle run on
the same bf16-rounded weights (magnitude AND direction), small tensors included.
"""
import os

import numpy as np
import pytest
import torch

from oracle import mdt_ref_cpu as R
from oracle import structure as S
from tests.test_oracle_golden import full_case
from tests.util_model import fill_hash_weights, model_args, named_canonical_params, split_qkv_grad

pytestmark = pytest.mark.gpu

BF16_GRAD_REL_L2 = 5e-2      # per parameter tensor, vs the fp32 oracle on bf16-rounded weights (observed: <= 3.5e-2, the
                             # query / key projections of the deepest pre-fusion layers; most tensors 1-2e-2)
BF16_LOGIT_ABS = 5e-2

_ORACLE = {}


def oracle_run(kind, rounded: bool):
    """fp32 oracle forward + backward of a case (cached: the C2 / C4 passes take ~10-20 s of host CPU each)."""
    key = (kind, rounded)
    if key not in _ORACLE:
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        fname, hp, trees, over = full_case(kind)
        W = R.make_weights(hp, overrides=over)
        if rounded:
            W = {n: w.detach().bfloat16().float().requires_grad_(not R.is_frozen(hp, n)) for n, w in W.items()}
        ref_b = S.collate(trees, 5)
        batch = R.to_torch_batch(ref_b)
        logits, glob = R.model_forward(W, hp, batch)
        loss, counters = R.node_cross_entropy(logits, batch["y"], batch["y_mask"], hp)
        loss.backward()
        grads = {n: (None if w.grad is None else w.grad.detach()) for n, w in W.items()}
        _ORACLE[key] = dict(hp=hp, trees=trees, over=over, fname=fname, ref_b=ref_b, logits=logits.detach(),
                            glob=glob.detach(), loss=float(loss), counters=counters, grads=grads)
    return _ORACLE[key]


def product_run(kind, dtype, ragged, main_grad=False):
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    fname, hp, trees, over = full_case(kind)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model, overrides=over)
    model = model.cuda().to(dtype)
    model.train()                                            # dropout p = 0 everywhere; train mode like the launch
    model.encoder.graph_encoder.ragged_tokens = ragged
    if main_grad:
        model.prepare_main_grads()
    pb = pack_batch(trees, 5)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
    loss, sample_size, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
    loss.backward()
    with torch.no_grad():
        logits, glob = model(pb.batched_data)
    torch.cuda.synchronize()
    return model, pb, float(loss.detach()), sample_size, log, logits, glob


@pytest.mark.parametrize("ragged", [False, True], ids=["padded", "ragged"])
@pytest.mark.parametrize("kind", ["M", "C1", "C2", "C4", "LAUNCH"])
def test_fp32_real_shapes_vs_reference_golden_and_oracle(golden_dir, kind, ragged):
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    o = oracle_run(kind, rounded=False)
    model, pb, loss, sample_size, log, logits, glob = product_run(kind, torch.float32, ragged)
    for k in ("attn_bias", "spatial_pos", "in_degree", "x_token_mask", "x", "x_attention_mask", "x_image_indexes", "y_mask"):
        assert np.array_equal(pb.batched_data[k].cpu().numpy(), o["ref_b"][k]), k           # integer side: bit-exact
    grads = {n: p.grad for n, p in named_canonical_params(model).items()}
    lg = logits.cpu()
    if o["fname"] is not None:                               # the real reference's outputs
        g = np.load(os.path.join(golden_dir, o["fname"]))
        np.testing.assert_allclose(lg.numpy(), g["logits"], atol=1e-3)
        np.testing.assert_allclose(glob.cpu().numpy(), g["enc/global"], atol=1e-3)
        assert abs(loss - float(g["loss"])) <= 4e-2          # fp16 loss value: 1 ulp at ~27 is 1.6e-2
        assert sample_size == int(g["sample_size"])
        for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
            assert int(log[k]) == int(g["log/" + k]), k
        m = GraphPredictionNodeCrossEntropy.compute_metrics([log])
        for k in ("accuracy", "recall", "precision", "f1"):
            assert abs(m[k] - float(g["metric/" + k])) < 1e-6, k
        assert 0.0 < m["f1"] < 1.0 and 0.0 < m["precision"] < 1.0 and 0.0 < m["recall"] < 1.0   # non-degenerate fixture
        n_checked = 0
        for key in [k for k in g.files if k.startswith("gnorm/")]:
            name = key[len("gnorm/"):]
            gn = float(g[key])
            gr = split_qkv_grad(name, grads)
            if gn < 0:
                assert gr is None or float(gr.abs().max()) == 0.0, f"{name}: the reference gives no gradient"
                continue
            assert gr is not None, name
            n_checked += 1
            assert abs(float(gr.double().norm()) - gn) <= 1e-3 * max(1.0, gn), (name, float(gr.norm()), gn)
            d = float(np.abs(gr.flatten()[:64].float().cpu().numpy() - g["gslice/" + name]).max())
            assert d <= 1e-3 * max(1.0, gn), (name, d)
        assert n_checked == int(g["n_trainable_with_grad"])
    # the oracle, full tensors
    assert float((lg - o["logits"]).abs().max()) < 1e-3
    assert float((glob.cpu() - o["glob"]).abs().max()) < 1e-3
    for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
        assert int(log[k]) == o["counters"][k], k
    c = o["counters"]
    assert c["num_positive_correct"] > 0 and c["num_pred_positive"] > c["num_positive_correct"] \
        and c["total_positive"] > c["num_positive_correct"], "fixture predictions must be mixed (TP, FP, FN > 0)"
    worst = ("", 0.0)
    n = 0
    for name, ref in o["grads"].items():
        gr = split_qkv_grad(name, grads)
        if ref is None:
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        assert gr is not None, name
        n += 1
        scale = max(1.0, float(ref.abs().max()))
        err = float((gr.float().cpu() - ref).abs().max()) / scale
        if err > worst[1]:
            worst = (name, err)
        assert err <= 1e-3, (name, err)
    print(f"[{kind} {'ragged' if ragged else 'padded'}] logits |err| {float((lg - o['logits']).abs().max()):.2e}; "
          f"{n} parameter gradients, worst {worst[1]:.2e} ({worst[0]})")


@pytest.mark.parametrize("kind", ["M", "C2", "C4", "LAUNCH"])
def test_bf16_real_shapes_vs_fp32_oracle(kind):
    """The production dtype at the real shapes: bf16 MFMA GEMMs / attention, fp32 softmax, LayerNorm statistics and
    gradient arena, ragged text, against the fp32 oracle on the same bf16-rounded weights.  Every parameter gradient is
    held to a relative L2 error (magnitude and direction), biases / LayerNorm / tables included."""
    o = oracle_run(kind, rounded=True)
    model, pb, loss, sample_size, log, logits, glob = product_run(kind, torch.bfloat16, True, main_grad=True)
    lg = logits.float().cpu()
    d_logit = float((lg - o["logits"]).abs().max())
    assert d_logit < BF16_LOGIT_ABS, d_logit
    assert abs(loss - o["loss"]) < 0.15 + 0.01 * abs(o["loss"])
    # predictions may differ only where the fp32 margin is inside the bf16 logit tolerance
    margin = (o["logits"][:, 1] - o["logits"][:, 0])
    pred_ref, pred = margin > 0, (lg[:, 1] - lg[:, 0]) > 0
    clear = margin.abs() > 2 * BF16_LOGIT_ABS
    assert bool((pred_ref[clear] == pred[clear]).all())
    assert int(log["total_positive"]) == o["counters"]["total_positive"]
    grads = {n: getattr(p, "main_grad", None) for n, p in named_canonical_params(model).items()}
    rows = []
    for name, ref in o["grads"].items():
        if ref is None:
            continue
        gr = split_qkv_grad(name, grads)
        assert gr is not None, name
        rn = float(ref.double().norm())
        if rn < 1e-6:
            # mathematically zero (a key bias shifts every score of a row alike): what is left is rounding noise
            assert float(gr.float().norm()) < 5e-3, name
            continue
        # the classifier bias gradient is a sum of ~n_labels terms of magnitude ~0.5 that cancel to ~0.1: its error is
        # measured against the scale of what is summed, not against the cancelled result
        floor = 0.5 if name == "node_classifier.bias" else 0.0
        rel = float((gr.float().cpu().double() - ref.double()).norm()) / max(rn, floor)
        rows.append((rel, name, rn, ref.numel()))
    rows.sort(reverse=True)
    print(f"[{kind} bf16] logits |err| {d_logit:.3e}; {len(rows)} gradients; worst rel-L2: "
          + "; ".join(f"{n} {r:.3e} (|g| {rn:.2e}, {ne} el)" for r, n, rn, ne in rows[:6]))
    bad = [(n, r) for r, n, _, _ in rows if r > BF16_GRAD_REL_L2]
    assert not bad, bad[:10]


def test_bf16_large_config_at_full_depth_vs_fp32_oracle():
    """configs[3]'s 12 + 12 blocks and 12 graph layers in bf16 against the fp32 oracle (same
    bf16-rounded weights).  Gates are the C2 / C4 ones for the logits; for gradients the per-tensor relative L2 may grow
    with depth — the bound below is what a first-layer tensor is allowed after 24 blocks, the printed table is the
    measurement."""
    o = oracle_run("C4F", rounded=True)
    model, pb, loss, sample_size, log, logits, glob = product_run("C4F", torch.bfloat16, True, main_grad=True)
    lg = logits.float().cpu()
    d_logit = float((lg - o["logits"]).abs().max())
    span = float(o["logits"].max() - o["logits"].min())
    grads = {n: getattr(p, "main_grad", None) for n, p in named_canonical_params(model).items()}
    rows = []
    for name, ref in o["grads"].items():
        if ref is None:
            continue
        gr = split_qkv_grad(name, grads)
        assert gr is not None, name
        rn = float(ref.double().norm())
        if rn < 1e-6:
            continue
        floor = 0.5 if name == "node_classifier.bias" else 0.0
        # query / key projections: with hash weights the softmax of the deep pre-fusion blocks is nearly uniform and the signal
        # through it almost vanishes — |g_query| is 1e-4 ... 1e-5 of |g_value| of the same block (measured: 8.4e-3 against 229
        # in text layer 11) — so bf16's rowsum(dO o O) rounding (the flash-attention form of the softmax gradient), harmless
        # at the block's gradient scale, is a large FRACTION of that remnant.  Their error is therefore measured against
        # max(|g|, 1e-3 |g_value|): a thousandth of the block's attention-gradient scale.
        for a, b in ((".query.", ".value."), (".key.", ".value."), (".q_proj.", ".v_proj."), (".k_proj.", ".v_proj.")):
            if a in name and o["grads"].get(name.replace(a, b)) is not None:
                floor = max(floor, 1e-3 * float(o["grads"][name.replace(a, b)].double().norm()))
        rows.append((float((gr.float().cpu().double() - ref.double()).norm()) / max(rn, floor), name, rn, ref.numel()))
    rows.sort(reverse=True)
    big = [r for r in rows if r[3] >= 1 << 16]
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    def vnorm(n):        # |gradient| of the value projection of the same block: the scale of that block's attention gradients
        for a, b in ((".query.", ".value."), (".key.", ".value."), (".q_proj.", ".v_proj."), (".k_proj.", ".v_proj.")):
            if a in n:
                r = o["grads"].get(n.replace(a, b))
                return None if r is None else float(r.double().norm())
        return None
    print(f"[C4F bf16, 12 + 12 blocks, 12 graph layers] logits |err| {d_logit:.3e} (logit span {span:.3f}); {len(rows)} gradients, "
          f"median rel-L2 {med:.3e}; worst: " + "; ".join(f"{n} {r:.3e} (|g| {rn:.2e}, |g_value| {vnorm(n)}, {ne} el)" for r, n, rn, ne in rows[:8]))
    assert d_logit < BF16_LOGIT_ABS, d_logit
    assert abs(loss - o["loss"]) < 0.15 + 0.01 * abs(o["loss"])
    margin = (o["logits"][:, 1] - o["logits"][:, 0])
    clear = margin.abs() > 2 * BF16_LOGIT_ABS
    assert bool(((margin > 0)[clear] == ((lg[:, 1] - lg[:, 0]) > 0)[clear]).all())
    assert med < 3e-2, med
    # measured on MI355X: median 2.6e-2; every tensor but the deep pre-fusion query / key projections <= 7e-2; those (their |g| sits
    # AT the 1e-3 |g_value| floor: text layers 6-8) 7e-2 ... 1.07e-1
    is_qk = lambda n: any(t in n for t in (".query.", ".key.", ".q_proj.", ".k_proj."))
    print("[C4F] worst query / key tensors: " + "; ".join(f"{n} {r:.3e}" for r, n, _, _ in [x for x in big if is_qk(x[1])][:4]))
    # Round 4 measured what that q / k remnant's error is NOT: these text rows (48 tokens, one unbinned launch) ran on the fp32-scratch
    # kernel before and run on attn_bwd_v4x now — both sum delta = sum P o dP in fp32 — and the table is the same to three digits
    # (layer 7 query: 1.07e-1 in round 3, 1.067e-1 now).  So delta's rounding was never what these tensors see.  What is left is the
    # bf16 STORAGE of q / k / v themselves: the value rows of a collapsed block differ by ~1e-3 of their norm, so dP_ij - delta_i =
    # dO_i . (V_j - O_i) is a difference of bf16-rounded rows (2^-9 each) and a quarter of it is rounding — an fp32 oracle on the same
    # weights keeps those activations in fp32.  Only storing qkv wider would remove it; the allowance for exactly these tensors stays.
    bad = [(n, r) for r, n, _, _ in big if r > (3 if is_qk(n) else 2) * BF16_GRAD_REL_L2]
    assert not bad, bad[:10]


def test_tree_with_more_than_271_comments_fp32_vs_oracle():
    """The reference takes any tree size (dense attention); here a 300-comment thread rides the key-chunked graph-attention
    kernels (csrc/attention_long.hip) inside the full model: logits and every parameter gradient against the oracle."""
    from multimodaldiscussiontransformer_amd import synthetic
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from oracle import cases
    hp = cases.tiny_hparams("A")
    rng = np.random.Generator(np.random.PCG64(300))
    trees = [synthetic.make_tree(300, rng, seq_len=8, vocab_size=hp.vocab_size, image_frac=0.01, image_size=hp.image_size, shape="deep", min_len=2),
             synthetic.make_tree(5, rng, seq_len=8, vocab_size=hp.vocab_size, image_frac=0.0, image_size=hp.image_size, min_len=2)]
    for i, t in enumerate(trees):
        n = len(t["parent"])
        t["y_mask"][:] = False
        lab = list(range(0, n, 7))
        t["y_mask"][lab] = True
        t["y"] = np.asarray([(k + i) % 2 for k in range(len(lab))], dtype=np.float32)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda().train()
    pb = pack_batch(trees, 5)
    assert pb.T == 301
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
    loss, n_lab, log = crit(model, {"nsamples": 2, "net_input": {"batched_data": pb.batched_data}})
    loss.backward()
    with torch.no_grad():
        logits, glob = model(pb.batched_data)
    W = R.make_weights(hp)
    batch = R.to_torch_batch(S.collate(trees, 5))
    lo, go = R.model_forward(W, hp, batch)
    ol, counters = R.node_cross_entropy(lo, batch["y"], batch["y_mask"], hp)
    ol.backward()
    assert float((logits.cpu() - lo.detach()).abs().max()) < 1e-3
    assert float((glob.cpu() - go.detach()).abs().max()) < 1e-3
    for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
        assert int(log[k]) == counters[k], k
    grads = {k: p.grad for k, p in named_canonical_params(model).items()}
    for name, w in W.items():
        if w.grad is None:
            continue
        gr = split_qkv_grad(name, grads)
        assert gr is not None, name
        assert float((gr.float().cpu() - w.grad).abs().max()) <= 1e-3 * max(1.0, float(w.grad.abs().max())), name


def test_sequences_beyond_272_tokens_fp32_vs_oracle():
    """VERDICT r2 missing #6: the reference's BERT takes up to max_position_embeddings = 512 tokens per comment
    (multigraphormer_graph_encoder.py:236-245) and a ViT as many patches as its image has; the single-pass attention kernels
    hold 272.  A batch beyond that runs in the padded layout on the key-chunked kernels (needs_long_attention): comments of up
    to 300 tokens (S = 304) and 272-px images (P = 290, S = 294) through the full model, logits and every parameter gradient
    against the oracle."""
    from multimodaldiscussiontransformer_amd import synthetic
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = R.hparams(dim=128, enc_heads=2, graph_heads=2, enc_ffn=256, graph_ffn=128, text_layers=4, vit_layers=4, num_fusion_layers=1,
                   num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4, vocab_size=600, max_pos=512, image_size=272, patch=16,
                   pos_weight=1.5, neg_weight=1.0)
    rng = np.random.Generator(np.random.PCG64(512))
    trees = [synthetic.make_tree(5, rng, seq_len=300, vocab_size=hp.vocab_size, image_frac=0.4, image_size=hp.image_size, min_len=150),
             synthetic.make_tree(3, rng, seq_len=300, vocab_size=hp.vocab_size, image_frac=0.0, image_size=hp.image_size, min_len=280)]
    for i, t in enumerate(trees):
        n = len(t["parent"])
        t["y_mask"][:] = True
        t["y"] = np.asarray([(k + i) % 2 for k in range(n)], dtype=np.float32)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda().train()
    ge = model.encoder.graph_encoder
    pb = pack_batch(trees, 5)
    assert ge.needs_long_attention(pb) and ge.ragged_tokens and ge.prune_last_layer      # the defaults stay on; the batch overrides them
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
    loss, n_lab, log = crit(model, {"nsamples": 2, "net_input": {"batched_data": pb.batched_data}})
    loss.backward()
    with torch.no_grad():
        logits, glob = model(pb.batched_data)
    W = R.make_weights(hp)
    batch = R.to_torch_batch(S.collate(trees, 5))
    lo, go = R.model_forward(W, hp, batch)
    ol, counters = R.node_cross_entropy(lo, batch["y"], batch["y_mask"], hp)
    ol.backward()
    assert float((logits.cpu() - lo.detach()).abs().max()) < 1e-3
    assert float((glob.cpu() - go.detach()).abs().max()) < 1e-3
    for k in ("ncorrect", "total_positive"):
        assert int(log[k]) == counters[k], k
    grads = {k: p.grad for k, p in named_canonical_params(model).items()}
    n = 0
    for name, w in W.items():
        if w.grad is None:
            continue
        gr = split_qkv_grad(name, grads)
        assert gr is not None, name
        n += 1
        assert float((gr.float().cpu() - w.grad).abs().max()) <= 1e-3 * max(1.0, float(w.grad.abs().max())), name
    assert n > 60
    # a comment longer than BERT's position table is refused with the numbers
    hp2 = R.hparams(dim=128, enc_heads=2, graph_heads=2, enc_ffn=256, graph_ffn=128, text_layers=4, vit_layers=4, num_fusion_layers=1,
                    num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4, vocab_size=600, max_pos=64, image_size=32, patch=16)
    m2 = GraphormerModel.build_model(model_args(hp2), task=None).cuda()
    t2 = [synthetic.make_tree(3, rng, seq_len=100, vocab_size=600, image_frac=0.0, image_size=32)]
    with pytest.raises(ValueError, match="position table"):
        m2(pack_batch(t2, 5).batched_data)
