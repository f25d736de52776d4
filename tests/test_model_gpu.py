"""Model-level parity on the GPU: the HIP path (through the C ABI, via the drop-in modules)
against (1) the golden vectors produced by the real reference and (2) the oracle run on the
same seeded inputs.  fp32: north_star's 1e-3 gate on logits / gradients (observed ~1e-5);
integer counters and F1 exact.  bf16: the same pipeline against the fp32 oracle at a
bf16-sized tolerance (stated below)."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, hashinit
from oracle import mdt_ref_cpu as R
from oracle import structure as S
from tests.util_model import fill_hash_weights, model_args, named_canonical_params

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def hw(name, shape, dtype=torch.float32):
    return torch.from_numpy(hashinit.param(name, shape)).to(dtype).cuda()


def hu(name, shape, scale=1.0):
    return torch.from_numpy(hashinit.uniform(name, shape, scale))


# ------------------------------------------------------------------------------- modules
@pytest.mark.parametrize("D,H,Fg", [(128, 8, 128), (768, 12, 768)])
def test_graph_modules_vs_reference_golden(golden_dir, D, H, Fg):
    from multimodaldiscussiontransformer_amd.modules import GraphAttnBias, GraphNodeFeature, GraphormerGraphEncoderLayer
    g = _g(golden_dir, f"graph_modules_d{D}.npz")
    spatial = torch.from_numpy(g["spatial_pos"]).cuda()
    attn_bias = torch.from_numpy(g["attn_bias"]).cuda()
    deg = torch.from_numpy(g["in_degree"]).cuda()
    kpm = torch.from_numpy(g["key_padding_mask"]).cuda()
    B, N = deg.shape
    T = N + 1
    gab = GraphAttnBias(num_heads=H, num_atoms=16, num_edges=16, num_spatial=512, num_edge_dis=8, hidden_dim=D,
                        edge_type="", multi_hop_max_dist=5, n_layers=4).cuda()
    with torch.no_grad():
        for n, p in gab.named_parameters():
            p.copy_(hw("graph_attn_bias." + n, tuple(p.shape)))
    bias = gab(dict(attn_bias=attn_bias, spatial_pos=spatial, x=None))
    ref = g["gab/out"]
    got = bias.detach().cpu().numpy()
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    np.testing.assert_allclose(got[~np.isinf(ref)], ref[~np.isinf(ref)], atol=1e-5)
    cot = hu("gab/cot", tuple(bias.shape)).cuda()
    (torch.where(torch.isinf(bias), torch.zeros_like(bias), bias) * cot).sum().backward()
    np.testing.assert_allclose(gab.spatial_pos_encoder.weight.grad[:24].cpu().numpy(), g["gab/d_spatial"], atol=1e-4)
    np.testing.assert_allclose(gab.graph_token_virtual_distance.weight.grad.cpu().numpy(), g["gab/d_virtual"], atol=1e-4)

    gnf = GraphNodeFeature(num_heads=H, num_atoms=16, num_in_degree=512, num_out_degree=512, hidden_dim=D, n_layers=4).cuda()
    with torch.no_grad():
        for n, p in gnf.named_parameters():
            p.copy_(hw("graph_node_feature." + n, tuple(p.shape)))
    x = hu("gnf/x", (B, N, D)).cuda().requires_grad_(True)
    y = gnf(x, deg, deg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["gnf/out"], atol=1e-5)
    (y * hu("gnf/cot", tuple(y.shape)).cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gnf/dx"], atol=1e-5)
    np.testing.assert_allclose(gnf.in_degree_encoder.weight.grad[:8].cpu().numpy(), g["gnf/d_in"], atol=1e-5)
    np.testing.assert_allclose(gnf.out_degree_encoder.weight.grad[:8].cpu().numpy(), g["gnf/d_out"], atol=1e-5)
    np.testing.assert_allclose(gnf.graph_token.weight.grad.cpu().numpy(), g["gnf/d_tok"], atol=1e-5)

    dense_bias = torch.from_numpy(g["gab/out"]).cuda()
    for pre_ln in (False, True):
        tag = "pre" if pre_ln else "post"
        layer = GraphormerGraphEncoderLayer(embedding_dim=D, ffn_embedding_dim=Fg, num_attention_heads=H, dropout=0.0,
                                            attention_dropout=0.0, activation_dropout=0.0, activation_fn="gelu",
                                            pre_layernorm=pre_ln).cuda()
        sd = {k: hw("layers.0.layers.0." + k, tuple(v.shape)) for k, v in layer.state_dict().items()}
        layer.load_state_dict(sd)
        xin = hu("gl/x", (T, B, D)).cuda().requires_grad_(True)
        b2 = dense_bias.clone().requires_grad_(True)
        yo, attn = layer(xin, self_attn_bias=b2, self_attn_padding_mask=kpm)
        assert attn is None
        np.testing.assert_allclose(yo.detach().cpu().numpy(), g[f"layer_{tag}/out"], atol=1e-4)
        (yo * hu("gl/cot", (T, B, D)).cuda()).sum().backward()
        np.testing.assert_allclose(xin.grad.cpu().numpy(), g[f"layer_{tag}/dx"], atol=2e-4)
        np.testing.assert_allclose(b2.grad.cpu().numpy(), g[f"layer_{tag}/dbias"], atol=2e-4)
        grads = {"layers.0.layers.0." + k: p.grad for k, p in layer.named_parameters()}
        from tests.util_model import split_qkv_grad
        for key in [k for k in g.files if k.startswith(f"layer_{tag}/gnorm/")]:
            name = key.split("gnorm/")[1]
            gr = split_qkv_grad(name, grads)
            assert gr is not None, name
            gn = float(g[key])
            assert abs(float(gr.double().norm()) - gn) <= 1e-3 * max(1.0, gn), name
            np.testing.assert_allclose(gr.flatten()[:64].cpu().numpy(), g[f"layer_{tag}/gslice/{name}"], atol=2e-4)
        if not pre_ln:
            mha = layer.self_attn
            for p in layer.parameters():
                p.grad = None
            xq = hu("mha/x", (T, B, D)).cuda().requires_grad_(True)
            b3 = dense_bias.clone().requires_grad_(True)
            a, w = mha(xq, xq, xq, b3, key_padding_mask=kpm, need_weights=False)
            np.testing.assert_allclose(a.detach().cpu().numpy(), g["mha/out"], atol=1e-4)
            (a * hu("gl/cot", (T, B, D)).cuda()).sum().backward()
            np.testing.assert_allclose(xq.grad.cpu().numpy(), g["mha/dx"], atol=2e-4)
            np.testing.assert_allclose(b3.grad.cpu().numpy(), g["mha/dbias"], atol=2e-4)
            np.testing.assert_allclose(mha.qkv_weight.grad[:D].cpu().numpy(), g["mha/dWq"], atol=2e-4)
            np.testing.assert_allclose(mha.qkv_bias.grad[D:2 * D].cpu().numpy(), g["mha/dbk"], atol=2e-4)


def test_fusion_layer_vs_reference_golden(golden_dir):
    from multimodaldiscussiontransformer_amd.modules import GraphFusionLayer
    from multimodaldiscussiontransformer_amd.modules._fused import BertLayer, ViTLayer
    from tests.util_model import split_qkv_grad
    g = _g(golden_dir, "fusion_layer.npz")
    D, H, Fe, nb, L, P, M = 768, 12, 128, 4, 10, 5, 5
    fl = GraphFusionLayer(BertLayer(D, H, Fe), ViTLayer(D, H, Fe), nb, use_projection=True).cuda()
    pre = "fusion_layers.0.fusion_layers.0."
    fl.load_state_dict({k: hw(pre + k, tuple(v.shape)) for k, v in fl.state_dict().items()})
    img = torch.from_numpy(g["image_index"]).cuda()
    am = torch.from_numpy(g["attention_mask"]).cuda()
    ext = ((1.0 - am)[:, None, None, :].to(torch.half)) * torch.finfo(torch.half).min
    for with_img in (True, False):
        tag = "img" if with_img else "noimg"
        for p in fl.parameters():
            p.grad = None
        text = hu("fl/text", (M, L, D)).cuda().requires_grad_(True)
        vit = hu("fl/vit", (int(img.sum()), P, D)).cuda().requires_grad_(True)
        bn = hu("fl/bn", (M, nb, D)).cuda().requires_grad_(True)
        t, v, b = fl(text, vit if with_img else None, bn, ext, img)
        np.testing.assert_allclose(t.detach().cpu().numpy(), g[f"{tag}/text"], atol=2e-4)
        np.testing.assert_allclose(b.detach().cpu().numpy(), g[f"{tag}/bn"], atol=2e-4)
        loss = (t * hu("fl/ct", tuple(t.shape)).cuda()).sum() + (b * hu("fl/cb", tuple(b.shape)).cuda()).sum()
        if with_img:
            np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"{tag}/vit"], atol=2e-4)
            loss = loss + (v * hu("fl/cv", tuple(v.shape)).cuda()).sum()
        else:
            assert v is None
        loss.backward()
        np.testing.assert_allclose(text.grad.cpu().numpy(), g[f"{tag}/dtext"], atol=5e-4)
        np.testing.assert_allclose(bn.grad.cpu().numpy(), g[f"{tag}/dbn"], atol=5e-4)
        if with_img:
            np.testing.assert_allclose(vit.grad.cpu().numpy(), g[f"{tag}/dvit"], atol=5e-4)
        grads = {pre + k: p.grad for k, p in fl.named_parameters()}
        for key in [k for k in g.files if k.startswith(f"{tag}/gnorm/")]:
            name = key.split("gnorm/")[1]
            gn = float(g[key])
            gr = split_qkv_grad(name, grads)
            if gn < 0:
                assert gr is None or float(gr.abs().max()) == 0.0, name      # dead parameter in the reference too
                continue
            assert gr is not None, name
            assert abs(float(gr.double().norm()) - gn) <= 1e-3 * max(1.0, gn), (name, float(gr.norm()), gn)


# ------------------------------------------------------------------------------- full model
def _run_full(kind, dtype, use_main_grad=False, ragged=True):
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams(kind)
    trees = cases.tiny_trees(kind, hp)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda().to(dtype)
    model.train()
    model.encoder.graph_encoder.ragged_tokens = ragged      # valid-token packing vs the reference's padded layout
    if use_main_grad:
        model.prepare_main_grads()
    pb = pack_batch(trees, 5)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
    sample = {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}}
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    return hp, trees, model, pb, loss, sample_size, log


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("kind", ["A", "B"])
def test_full_model_fp32_vs_reference_golden_and_oracle(golden_dir, kind, ragged):
    """Both text layouts against the reference's golden vectors: the padded one reproduces everything, the ragged one
    (padded token positions never computed) everything except the hidden states AT padded positions, which it
    returns as zeros — logits, loss, counters and every parameter gradient are held to the same 1e-3 gate."""
    g = _g(golden_dir, f"full_tiny768_{kind}.npz")
    hp, trees, model, pb, loss, sample_size, log = _run_full(kind, torch.float32, ragged=ragged)
    # structural tensors handed to the kernels are the reference's, bit for bit
    ref_b = S.collate(trees, 5)
    for k in ("attn_bias", "spatial_pos", "in_degree", "x_token_mask", "x", "x_attention_mask", "x_image_indexes", "y_mask"):
        assert np.array_equal(pb.batched_data[k].cpu().numpy(), ref_b[k]), k
    with torch.no_grad():
        logits, glob = model(pb.batched_data)
        text, bn, glob2 = model.encoder.graph_encoder(pb.batched_data)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-3)
    np.testing.assert_allclose(glob.cpu().numpy(), g["enc/global"], atol=1e-3)
    valid = pb.text_mask[:, :3].bool().cpu().numpy()[:, :, None] if ragged else np.ones((pb.M, 3, 1), dtype=bool)
    np.testing.assert_allclose(text[:, :3, :64].cpu().numpy() * valid, g["enc/text_slice"] * valid, atol=1e-3)
    if ragged:
        assert float(text.cpu()[~pb.text_mask.bool().cpu()].abs().max() if (~pb.text_mask.bool()).any() else 0.0) == 0.0
    np.testing.assert_allclose(bn.cpu().numpy(), g["enc/bn"], atol=1e-3)
    assert abs(float(loss) - float(g["loss"])) <= 2e-2          # fp16 loss value: 1 ulp at ~8 is 7.8e-3
    assert sample_size == int(g["sample_size"])
    for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
        assert int(log[k]) == int(g["log/" + k]), k
    m = type(model).__mro__ and __import__("multimodaldiscussiontransformer_amd.criterions", fromlist=["x"]).GraphPredictionNodeCrossEntropy.compute_metrics([log])
    for k in ("accuracy", "recall", "precision", "f1"):
        assert abs(m[k] - float(g["metric/" + k])) < 1e-6, k
    # gradients: every parameter the reference gives a gradient to, at the 1e-3 gate
    from tests.util_model import split_qkv_grad
    grads = {n: p.grad for n, p in named_canonical_params(model).items()}
    n_checked = 0
    worst = 0.0
    for key in [k for k in g.files if k.startswith("gnorm/")]:
        name = key[len("gnorm/"):]
        gn = float(g[key])
        gr = split_qkv_grad(name, grads)
        if gn < 0:
            assert gr is None or float(gr.abs().max()) == 0.0, f"{name}: reference gives no gradient"
            continue
        assert gr is not None, name
        n_checked += 1
        err = abs(float(gr.double().norm()) - gn)
        assert err <= 1e-3 * max(1.0, gn), (name, float(gr.norm()), gn)
        sl = gr.flatten()[:64].float().cpu().numpy()
        d = float(np.abs(sl - g["gslice/" + name]).max())
        worst = max(worst, d)
        assert d <= 1e-3 * max(1.0, gn), (name, d)
    assert n_checked == int(g["n_trainable_with_grad"])
    print(f"[{kind}] checked {n_checked} parameter gradients, worst |slice diff| {worst:.2e}")
    # full-tensor comparison against the oracle (same seeded inputs)
    W = R.make_weights(hp)
    batch = R.to_torch_batch(ref_b)
    lo, _ = R.model_forward(W, hp, batch)
    ol, _ = R.node_cross_entropy(lo, batch["y"], batch["y_mask"], hp)
    ol.backward()
    for name in W:
        gr = split_qkv_grad(name, grads)
        if W[name].grad is None:
            continue
        assert gr is not None, name
        ref = W[name].grad
        tol = 1e-3 * max(1.0, float(ref.abs().max()))
        assert float((gr.float().cpu() - ref).abs().max()) <= tol, name


EDGE_CASES = ["single_comment_trees", "no_images", "every_comment_an_image", "one_token_texts", "chains_and_stars"]


def _edge_trees(case, hp):
    from multimodaldiscussiontransformer_amd import synthetic
    kw = dict(seq_len=24, vocab_size=hp.vocab_size, image_size=hp.image_size, min_len=2)
    rng = np.random.Generator(np.random.PCG64(99))
    mk = lambda n, **k: synthetic.make_tree(n, rng, **{**kw, **k})
    if case == "single_comment_trees":
        return [mk(1), mk(1, image_frac=1.0), mk(5, image_frac=0.4), mk(1)]
    if case == "no_images":
        return [mk(4), mk(7), mk(2)]
    if case == "every_comment_an_image":
        return [mk(3, image_frac=1.0), mk(5, image_frac=1.0)]
    if case == "one_token_texts":
        trees = [mk(6, image_frac=0.34), mk(3)]
        for t in trees:                                  # [CLS] alone on most comments, two tokens on one
            t["attention_mask"][:] = 0
            t["attention_mask"][:, 0] = 1
            t["attention_mask"][-1, 1] = 1
            t["input_ids"] = t["input_ids"] * t["attention_mask"]
        return trees
    chain = mk(9, image_frac=0.2)
    chain["parent"] = np.arange(9, dtype=np.int64) - 1           # a thread 9 deep: spatial positions beyond the clamp of 5
    star = mk(8, image_frac=0.25)
    star["parent"] = np.array([-1] + [0] * 7, dtype=np.int64)    # everybody answers the post
    return [chain, star, mk(2)]


@pytest.mark.parametrize("case", EDGE_CASES)
@pytest.mark.parametrize("mode", ["fp32-padded", "fp32-ragged", "bf16-ragged"])
def test_edge_case_batches_vs_oracle(case, mode):
    """The shapes the reference's collator has to cope with and the kernels' index arithmetic has corners for — trees of ONE
    comment (a graph of the graph token and one node), a batch without any image (the image branch never launches), a batch in
    which every comment carries one, texts of a single valid token ([CLS] alone: one-row attention, one-row LayerNorm statistics
    over a ragged batch), deep chains next to flat stars (spatial positions up to the clamp, degree 0 ... N - 1) — each against the
    oracle on the same seeded inputs.  fp32 (padded and ragged layouts): logits, loss and every parameter gradient at the 1e-3
    gate.  bf16 (the production kernels: MFMA GEMMs, register-resident attention, one-pass backward on one-token rows): finite,
    logits within 5e-2, all gradients together within 5e-2 relative L2 of the fp32 oracle."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from tests.util_model import split_qkv_grad
    hp = cases.tiny_hparams("A")
    trees = _edge_trees(case, hp)
    half = mode.startswith("bf16")
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda()
    model = (model.bfloat16() if half else model).eval()
    model.encoder.graph_encoder.ragged_tokens = mode.endswith("ragged")
    pb = pack_batch(trees, 5)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
    loss, sample_size, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
    loss.backward()
    with torch.no_grad():
        logits, _ = model(pb.batched_data)
    ref_b = S.collate(trees, 5)
    W = R.make_weights(hp)
    batch = R.to_torch_batch(ref_b)
    lo, _ = R.model_forward(W, hp, batch)
    ol, olog = R.node_cross_entropy(lo, batch["y"], batch["y_mask"], hp)
    ol.backward()
    assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(lo).all())
    np.testing.assert_allclose(logits.float().cpu().numpy(), lo.detach().numpy(), atol=5e-2 if half else 1e-3)
    assert abs(float(loss.detach()) - float(ol.detach())) <= (5e-2 if half else 2e-2) * max(1.0, abs(float(ol.detach())))
    grads = {n: q.grad for n, q in named_canonical_params(model).items()}
    checked, num, den = 0, 0.0, 0.0
    for name in W:
        if W[name].grad is None:
            continue
        gr = split_qkv_grad(name, grads)
        assert gr is not None, name
        ref = W[name].grad
        assert bool(torch.isfinite(gr).all()), name
        if half:
            num += float((gr.double().cpu() - ref.double()).pow(2).sum())
            den += float(ref.double().pow(2).sum())
        else:
            assert float((gr.float().cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max())), name
        checked += 1
    assert checked >= 90, checked            # without images the ViT side has no gradient (97 tensors), otherwise 150+
    if half:
        assert (num / den) ** 0.5 <= 5e-2, (num / den) ** 0.5


@pytest.mark.parametrize("half", [False, True])
def test_batch_without_a_labelled_comment(half):
    """criterions/hatespeech_loss.py:95-118 on an empty selection: the summed cross-entropy of no logits is 0, sample_size 0, every
    counter 0 — and backward runs (a data-parallel rank may draw such a batch while its peers do not): finite, all-zero gradients."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    trees = _edge_trees("chains_and_stars", hp)
    for t in trees:
        t["y_mask"][:] = False
        t["y"] = np.zeros(0, dtype=np.float32)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda()
    model = (model.bfloat16() if half else model).train()
    pb = pack_batch(trees, 5)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
    loss, sample_size, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
    assert sample_size == 0 and float(loss.detach()) == 0.0
    for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
        assert int(log[k]) == 0, k
    loss.backward()
    for n, q in model.named_parameters():
        if q.grad is not None:
            assert bool(torch.isfinite(q.grad).all()) and float(q.grad.abs().max()) == 0.0, n


def test_ragged_tokens_equal_padded_tokens_fp32():
    """Same weights, same batch (random comment lengths, one mask with a hole): logits, loss and every parameter
    gradient of the ragged layout equal the padded layout's to fp32 round-off."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from multimodaldiscussiontransformer_amd import synthetic
    hp = cases.tiny_hparams("A")
    trees = synthetic.make_trees(3, 6, seed=77, seq_len=24, vocab_size=hp.vocab_size, image_frac=0.34, image_size=hp.image_size, min_len=2)
    trees[0]["attention_mask"][1, 1] = 0                      # a hole: position 1 masked, later ones valid
    res = {}
    for ragged in (False, True):
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model)
        model = model.cuda().eval()                           # eval: no dropout, so the two layouts are comparable
        model.encoder.graph_encoder.ragged_tokens = ragged
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        with torch.no_grad():
            logits, glob = model(pb.batched_data)
        res[ragged] = (float(loss), logits.cpu(), glob.cpu(), {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None})
    (l0, lg0, gl0, g0), (l1, lg1, gl1, g1) = res[False], res[True]
    assert abs(l0 - l1) < 1e-5
    torch.testing.assert_close(lg1, lg0, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(gl1, gl0, atol=2e-5, rtol=1e-5)
    assert set(g0) == set(g1)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], atol=2e-5 * max(1.0, float(g0[n].abs().max())), rtol=1e-4, msg=n)


@pytest.mark.parametrize("half", [False, True])
def test_embeddings_with_their_layernorm_in_one_pass_change_no_bit(monkeypatch, half):
    """SURVEY K9 in the model (ragged token layout): the fused embedding + LayerNorm pass gives the loss and the logits of the two-launch route bit for bit, in fp32 and in bf16,
    and its gradients up to the order of fp32 atomics."""
    from multimodaldiscussiontransformer_amd import engine
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    trees = cases.tiny_trees("A", hp)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(engine, "EMBED_LN_FUSED", fused)
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model)
        model = model.cuda()
        model = (model.bfloat16() if half else model).eval()
        ge = model.encoder.graph_encoder
        ge.ragged_tokens = True
        ge.two_streams = False                    # one stream: the atomics of the embedding scatter in one order
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        with torch.no_grad():
            logits, _ = model(pb.batched_data)
        res[fused] = (float(loss.detach()), logits.float().cpu(), {n: q.grad.float().cpu() for n, q in model.named_parameters() if q.grad is not None})
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])
    assert set(res[True][2]) == set(res[False][2])
    for n, g0 in res[False][2].items():
        g1 = res[True][2][n]
        # fp32 atomics (embedding scatter, LayerNorm column sums, split-K slabs) add in whatever order they finish
        torch.testing.assert_close(g1, g0, atol=(2e-5 if not half else 2e-3) * max(1.0, float(g0.abs().max())), rtol=1e-4 if not half else 1e-2, msg=n)


def test_patch_embedding_takes_one_launch_when_nothing_needs_the_gathered_matrix(monkeypatch):
    """SURVEY K8 in the model: a bf16 model whose ViT patch projection is frozen (the shipped launch: --freeze_initial_encoders)
    or that runs without a tape (validation) embeds its patches with ops.vit_patch_embed; a trainable projection keeps the
    three-launch route (its weight gradient contracts over the gathered matrix).  Same logits either way up to the one bf16
    rounding the fused kernel saves; gradients of a frozen-projection step likewise."""
    from multimodaldiscussiontransformer_amd import engine, ops
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    trees = cases.tiny_trees("A", hp)
    calls = []
    real = ops.vit_patch_embed
    monkeypatch.setattr(ops, "vit_patch_embed", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(engine, "PATCH_EMBED_FUSED", fused)
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model)
        model = model.cuda().bfloat16().eval()
        ge = model.encoder.graph_encoder
        if ge.vit_model.embeddings.patch_embeddings.projection.weight.shape[0] % 128:
            pytest.skip("tiny ViT width is not a multiple of the 128-column tile")
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        n0 = len(calls)
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        trainable_calls = len(calls) - n0
        for q in ge.vit_model.embeddings.patch_embeddings.parameters():
            q.requires_grad_(False)
        model.zero_grad()
        n0 = len(calls)
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        frozen_calls = len(calls) - n0
        n0 = len(calls)
        with torch.no_grad():
            logits, _ = model(pb.batched_data)
        res[fused] = (float(loss.detach()), logits.float().cpu(), trainable_calls, frozen_calls, len(calls) - n0,
                      {n: q.grad.float().cpu() for n, q in model.named_parameters() if q.grad is not None})
    assert res[False][2:5] == (0, 0, 0) and res[True][2] == 0 and res[True][3] >= 1 and res[True][4] >= 1, (res[False][2:5], res[True][2:5])
    assert abs(res[True][0] - res[False][0]) < 2e-2 * max(1.0, abs(res[False][0]))
    torch.testing.assert_close(res[True][1], res[False][1], atol=3e-2, rtol=3e-2)
    g0, g1 = res[False][5], res[True][5]
    assert set(g0) == set(g1)
    num = sum(float((g1[n].double() - g0[n].double()).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].double().pow(2).sum()) for n in g0)
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5


@pytest.mark.parametrize("kind", ["A", "B"])
def test_pruned_last_fusion_layer_equals_full_fp32(kind):
    """The logits path computes only bottleneck token 0 and [CLS] of every comment in the LAST fusion layer (the
    other rows' outputs are dead values in the reference): logits, loss and every gradient equal the full computation."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams(kind)
    trees = cases.tiny_trees(kind, hp)
    res = {}
    for prune in (False, True):
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model)
        model = model.cuda().eval()
        model.encoder.graph_encoder.prune_last_layer = prune
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        with torch.no_grad():
            logits, glob = model(pb.batched_data)
        res[prune] = (float(loss.detach()), logits.cpu(), glob.cpu(), {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None})
    (l0, lg0, gl0, g0), (l1, lg1, gl1, g1) = res[False], res[True]
    assert abs(l0 - l1) < 1e-5
    torch.testing.assert_close(lg1, lg0, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(gl1, gl0, atol=2e-5, rtol=1e-5)
    assert set(g0) == set(g1)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], atol=2e-5 * max(1.0, float(g0[n].abs().max())), rtol=1e-4, msg=n)


@pytest.mark.parametrize("kind", ["A", "synthetic"])
def test_two_stream_tape_equals_single_stream(kind):
    """The image branch enqueued on a second HIP stream (engine.Tape fork / join / on_side) computes what the
    single-stream tape computes: same dropout sites and seeds, logits bit-identical, gradients equal up to the order
    of fp32 atomics — over several steps, so that blocks recycled by the caching allocator are exercised too."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from multimodaldiscussiontransformer_amd import synthetic
    hp = cases.tiny_hparams("A")
    if kind == "A":
        trees = cases.tiny_trees(kind, hp)
    else:
        trees = synthetic.make_trees(6, 9, seed=77, variable=True, seq_len=16, vocab_size=hp.vocab_size, image_frac=0.5,
                                     image_size=hp.image_size, min_len=3)
    res = {}
    for two in (False, True):
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model)
        model = model.cuda().train()
        ge = model.encoder.graph_encoder
        ge.two_streams = two
        pb = pack_batch(trees, 5)
        assert pb.I > 0
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        torch.manual_seed(1234)
        losses = []
        for _ in range(3):
            loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
            losses.append(float(loss.detach()))
        model.eval()
        with torch.no_grad():
            logits, glob = model(pb.batched_data)
        torch.cuda.synchronize()
        res[two] = (losses, logits.cpu(), glob.cpu(), {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None})
    (l0, lg0, gl0, g0), (l1, lg1, gl1, g1) = res[False], res[True]
    assert l0 == l1
    assert torch.equal(lg0, lg1) and torch.equal(gl0, gl1)
    assert set(g0) == set(g1)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], atol=2e-5 * max(1.0, float(g0[n].abs().max())), rtol=1e-4, msg=n)


def test_tree_permutation_and_node_padding_invariance_fp32():
    """SURVEY.md §4 property tests: (a) permuting the trees of a batch permutes the per-comment logits and leaves
    the loss and every parameter gradient unchanged; (b) adding a larger tree to the batch (more node padding N,
    other trees untouched) does not change the logits of the original comments."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from multimodaldiscussiontransformer_amd import synthetic
    hp = cases.tiny_hparams("A")
    trees = synthetic.make_trees(3, 7, seed=91, variable=True, seq_len=16, vocab_size=hp.vocab_size, image_frac=0.3,
                                 image_size=hp.image_size, min_len=3)
    big = synthetic.make_trees(1, 13, seed=92, seq_len=16, vocab_size=hp.vocab_size, image_frac=0.3,
                               image_size=hp.image_size, min_len=3)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda().eval()
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)

    def run(ts, grads=True):
        model.zero_grad(set_to_none=True)
        pb = pack_batch(ts, 5)
        if grads:
            loss, _, _ = crit(model, {"nsamples": len(ts), "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
        with torch.no_grad():
            logits, glob = model(pb.batched_data)
        g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None} if grads else {}
        return (float(loss) if grads else None), logits.cpu(), glob.cpu(), g

    l0, lg0, gl0, g0 = run(trees)
    perm = [2, 0, 1]
    l1, lg1, gl1, g1 = run([trees[i] for i in perm])
    sizes = [len(t["parent"]) for t in trees]
    starts = np.cumsum([0] + sizes)
    want = torch.cat([lg0[starts[i]:starts[i + 1]] for i in perm])
    torch.testing.assert_close(lg1, want, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(gl1, gl0[perm], atol=2e-5, rtol=1e-5)
    assert abs(l0 - l1) < 1e-5
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], atol=2e-5 * max(1.0, float(g0[n].abs().max())), rtol=1e-4, msg=n)
    _, lg2, gl2, _ = run(trees + big, grads=False)
    torch.testing.assert_close(lg2[: starts[-1]], lg0, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(gl2[:3], gl0, atol=2e-5, rtol=1e-5)


def test_full_model_main_grad_equals_autograd_grads():
    _, _, m1, _, _, _, _ = _run_full("A", torch.float32, use_main_grad=False)
    _, _, m2, _, _, _, _ = _run_full("A", torch.float32, use_main_grad=True)
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.grad is None:
            assert not hasattr(p2, "main_grad") or float(p2.main_grad.abs().max()) == 0.0, n
            continue
        assert p2.grad is None
        torch.testing.assert_close(p2.main_grad, p1.grad.float(), atol=1e-5, rtol=1e-4, msg=n)


@pytest.mark.parametrize("kind", ["A", "B"])
def test_full_model_bf16_vs_fp32_oracle(kind):
    """bf16 pipeline (bf16 MFMA GEMMs / attention, fp32 softmax, LN statistics and gradient
    accumulation): logits within 0.05 abs of the fp32 oracle, predictions / counters identical
    on this fixture, gradient direction cosine > 0.98 per large parameter."""
    hp, trees, model, pb, loss, sample_size, log = _run_full(kind, torch.bfloat16, use_main_grad=True)
    W = R.make_weights(hp)
    # the oracle sees the same bf16-rounded weights
    for n in W:
        W[n] = W[n].detach().bfloat16().float().requires_grad_(True)
    batch = R.to_torch_batch(S.collate(trees, 5))
    lo, _ = R.model_forward(W, hp, batch)
    ol, counters = R.node_cross_entropy(lo, batch["y"], batch["y_mask"], hp)
    ol.backward()
    with torch.no_grad():
        logits, _ = model(pb.batched_data)
    assert float((logits.float().cpu() - lo.detach()).abs().max()) < 0.05
    assert abs(float(loss) - float(ol)) < 0.1
    for k, i in (("ncorrect", 0), ("num_positive_correct", 1), ("total_positive", 2), ("num_pred_positive", 3)):
        assert int(log[k]) == counters[k]
    from tests.util_model import split_qkv_grad
    grads = {n: getattr(p, "main_grad", None) for n, p in named_canonical_params(model).items()}
    low = []
    for name in W:
        if W[name].grad is None or W[name].grad.numel() < 4096:
            continue
        gr = split_qkv_grad(name, grads)
        ref = W[name].grad.flatten()
        if float(ref.norm()) < 1e-6:
            continue
        cos = float(torch.dot(gr.float().cpu().flatten(), ref) / (gr.float().norm().cpu() * ref.norm() + 1e-20))
        if cos < 0.98:
            low.append((name, cos))
    assert not low, low[:8]


@pytest.mark.parametrize("kind,S,nseq", [("vit", 261, 3), ("bert", 104, 5)])
def test_large_shaped_encoder_block_bf16(kind, S, nseq):
    """mDT-large shapes (config 4: D 1024, 16 heads, ViT-L/14 → 4 + 257 tokens per image): one encoder
    block forward / backward in bf16 against the fp32 oracle block on bf16-rounded weights."""
    from multimodaldiscussiontransformer_amd import engine as E
    from multimodaldiscussiontransformer_amd.modules._fused import BertLayer, ViTLayer
    D, H, Fe = 1024, 16, 512
    layer = (ViTLayer if kind == "vit" else BertLayer)(D, H, Fe)
    pre = "L."
    sd = {k: torch.from_numpy(hashinit.param(pre + k, tuple(v.shape))) for k, v in layer.state_dict().items()}
    layer.load_state_dict(sd)
    layer = layer.cuda().bfloat16()
    W = {pre + k: v.bfloat16().float().requires_grad_(True) for k, v in sd.items()}
    x = hu("big/x", (nseq, S, D), 1.0).bfloat16()
    cot = hu("big/c", (nseq, S, D), 1.0).bfloat16()
    km = torch.ones(nseq, S, dtype=torch.uint8)
    if kind == "bert":
        km[1, S - 9:] = 0
    xr = x.float().requires_grad_(True)
    if kind == "vit":
        yr = R.vit_layer(xr, W, "L", H)
    else:
        add = (1.0 - km.float())[:, None, None, :] * float(torch.finfo(torch.half).min)
        yr = R.bert_layer(xr, W, "L", H, add)
    (yr * cot.float()).sum().backward()
    xg = x.cuda().view(nseq * S, D).requires_grad_(True)

    def run(tape, xv):
        spec = E.AttnSpec(nseq=nseq, S=S, H=H, key_mask=km.cuda() if kind == "bert" else None)
        return (E.transformer_block(tape, xv, layer.block_params(), spec, pre_ln=layer.pre_ln, eps=layer.eps),)

    (y,) = E.run_tape(run, [xg], list(layer.parameters()))
    (y.float() * cot.cuda().view(nseq * S, D).float()).sum().backward()
    assert float((y.float().cpu().view(nseq, S, D) - yr.detach()).abs().max()) < 0.12     # outputs are O(1..5), bf16 eps 0.8 %
    ref_g = xr.grad
    got_g = xg.grad.float().cpu().view(nseq, S, D)
    cos = float((ref_g * got_g).sum() / (ref_g.norm() * got_g.norm()))
    assert cos > 0.995, cos
    p = layer.intermediate.dense.weight
    rg = W[pre + "intermediate.dense.weight"].grad
    cos = float((rg * p.grad.float().cpu()).sum() / (rg.norm() * p.grad.float().norm().cpu()))
    assert cos > 0.99, cos


@pytest.mark.parametrize("pad", [True, False])
def test_patch_embedding_of_14_pixel_patches(monkeypatch, pad):
    """ViT-L/14: Conv2d(3, D, k = s = 14) is a GEMM with K = 588 — not a multiple of the MFMA tile kernels' k-step.  The gathered
    matrix and the weight are zero-padded to K = 640 (engine.vit_embeddings, engine.PATCH_K_PAD) so that the projection and its
    weight gradient run on the tile kernels (mDT-large step: -3 ms in one call); MDT_PATCH_K_PAD=0 keeps the any-shape kernel.
    Both against F.conv2d in fp32 on the bf16-rounded operands: tokens, and the gradients of weight, bias, [CLS] and positions."""
    import torch.nn.functional as F
    from multimodaldiscussiontransformer_amd import engine as E
    monkeypatch.setattr(E, "PATCH_K_PAD", pad)
    bf = torch.bfloat16
    I, HW, P, D = 20, 56, 14, 256                  # 16 patches per image: 320 rows
    npatch = (HW // P) ** 2
    img = hu("pe/img", (I, 3, HW, HW), 2.0)
    w = torch.nn.Parameter(hu("pe/w", (D, 3, P, P), 0.05).to(bf).cuda())
    b = torch.nn.Parameter(hu("pe/b", (D,), 0.3).to(bf).cuda())
    cls = torch.nn.Parameter(hu("pe/cls", (1, 1, D), 1.0).to(bf).cuda())
    pos = torch.nn.Parameter(hu("pe/pos", (1, npatch + 1, D), 0.5).to(bf).cuda())
    cot = hu("pe/cot", (I * (npatch + 1), D), 1.0).to(bf)

    def run(tape):
        return (E.vit_embeddings(tape, img.cuda(), w, b, cls, pos, P),)

    (tok,) = E.run_tape(run, [], [w, b, cls, pos])
    (tok.float() * cot.cuda().float()).sum().backward()
    wr, br = w.detach().float().cpu().requires_grad_(True), b.detach().float().cpu().requires_grad_(True)
    cr, pr = cls.detach().float().cpu().requires_grad_(True), pos.detach().float().cpu().requires_grad_(True)
    conv = F.conv2d(img.to(bf).float(), wr, br, stride=P).flatten(2).transpose(1, 2)
    ref = torch.cat([cr.expand(I, 1, D), conv], 1) + pr
    (ref.reshape(-1, D) * cot.float()).sum().backward()
    assert float((tok.detach().float().cpu() - ref.detach().reshape(-1, D)).abs().max()) <= 0.06          # values up to ~6, bf16 eps 0.4 % (two roundings)
    for got, want, name in ((w.grad, wr.grad, "weight"), (b.grad, br.grad, "bias"), (cls.grad, cr.grad, "cls"), (pos.grad, pr.grad, "pos")):
        rel = float((got.float().cpu() - want).norm() / want.norm())
        assert rel <= 2e-2, (name, rel)


def test_frozen_initial_encoders_stop_the_gradient_chain(monkeypatch):
    """--freeze_initial_encoders (run_train.sh:61): the reference's autograd never enters the frozen embeddings /
    pre-fusion layers.  Here too: no adjoint of the frozen prefix runs (counted by its GEMM launches), frozen parameters
    get no gradient, and every trainable parameter gets exactly the gradient of the unfrozen model."""
    from multimodaldiscussiontransformer_amd import engine
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    trees = cases.tiny_trees("A", hp)
    counts = {}
    raw = engine.ops.gemm
    phase = {"name": None}

    def counting(*a, **kw):
        if phase["name"]:
            counts[phase["name"]] = counts.get(phase["name"], 0) + 1
        return raw(*a, **kw)

    monkeypatch.setattr(engine.ops, "gemm", counting)
    res = {}
    for frozen in (False, True):
        model = GraphormerModel.build_model(model_args(hp, freeze_initial_encoders=frozen), task=None)
        fill_hash_weights(model)
        model = model.cuda().eval()
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
        tag = "frozen" if frozen else "live"
        phase["name"] = tag + "/fwd"
        loss, _, _ = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
        phase["name"] = tag + "/bwd"
        loss.backward()
        phase["name"] = None
        torch.cuda.synchronize()
        res[frozen] = (float(loss.detach()), {n: (None if p.grad is None else p.grad.cpu()) for n, p in model.named_parameters()},
                       {n: p.requires_grad for n, p in model.named_parameters()})
    (l0, g0, _), (l1, g1, rq) = res[False], res[True]
    assert l0 == l1 and counts["live/fwd"] == counts["frozen/fwd"]
    n_frozen = 0
    for n, g in g1.items():
        if not rq[n]:
            n_frozen += 1
            assert g is None, n
        elif g0[n] is not None:
            torch.testing.assert_close(g, g0[n], atol=2e-5 * max(1.0, float(g0[n].abs().max())), rtol=1e-4, msg=n)
    assert n_frozen > 30
    # hp "A": 2 + 2 pre-fusion blocks (5 backward GEMMs each with weight gradients skipped: 4 dgrads + ... ) and the two
    # embedding stages vanish from backward; what is left is the fusion / graph / head part
    assert counts["frozen/bwd"] < 0.7 * counts["live/bwd"], counts


def test_multihead_attention_need_weights_matches_oracle():
    """need_weights=True (the reference's default, modules/multihead_attention.py:205-214): head-averaged softmax
    probabilities [B, T, T], recomputed by mdt_attention_mean_probs; checked against the oracle's attention."""
    import math
    from multimodaldiscussiontransformer_amd.modules import MultiheadAttention
    T, B, D, H = 9, 3, 128, 8
    mha = MultiheadAttention(D, H, dropout=0.0, self_attention=True).cuda()
    with torch.no_grad():
        for n, p in mha.named_parameters():
            p.copy_(hw("mhaw/" + n, tuple(p.shape)))
    x = hu("mhaw/x", (T, B, D)).cuda()
    bias = hu("mhaw/b", (B, H, T, T), 2.0)
    bias[1, :, :, 7:] = float("-inf")
    kpm = torch.zeros(B, T, dtype=torch.bool)
    kpm[2, 5:] = True
    out, w = mha(x, x, x, bias.cuda(), key_padding_mask=kpm.cuda())          # need_weights defaults to True
    out2, none = mha(x, x, x, bias.cuda(), key_padding_mask=kpm.cuda(), need_weights=False)
    assert none is None and torch.equal(out, out2) and tuple(w.shape) == (B, T, T)
    hd = D // H
    wq, bq = mha.qkv_weight.detach().cpu(), mha.qkv_bias.detach().cpu()
    q, k, v = (torch.nn.functional.linear(x.cpu(), wq[i * D:(i + 1) * D], bq[i * D:(i + 1) * D]) for i in range(3))
    q = (q * hd ** -0.5).view(T, B * H, hd).transpose(0, 1)
    k = k.view(T, B * H, hd).transpose(0, 1)
    s_ = torch.bmm(q, k.transpose(1, 2)).view(B, H, T, T) + bias
    s_ = s_.masked_fill(kpm[:, None, None, :], float("-inf"))
    ref = torch.softmax(s_, dim=-1).mean(dim=1)
    torch.testing.assert_close(w.cpu(), ref, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(w.sum(-1).cpu(), torch.ones(B, T), atol=1e-5, rtol=1e-5)


def test_multihead_attention_head_weights_raw_scores_and_attn_mask():
    """The rest of the reference signature (modules/multihead_attention.py:91-102,176-178,189-190,205-214), none of it used by
    mDT: ``need_head_weights`` — per-head probabilities [H, B, T, T]; ``before_softmax`` — (scores [B * H, T, T], v
    [B * H, T, hd]) and no attention output; ``attn_mask`` — an additive [T, T] mask shared by all sequences and heads."""
    from multimodaldiscussiontransformer_amd.modules import MultiheadAttention
    T, B, D, H = 9, 3, 128, 2
    hd = D // H
    mha = MultiheadAttention(D, H, dropout=0.0, self_attention=True).cuda()
    with torch.no_grad():
        for n, p in mha.named_parameters():
            p.copy_(hw("mhah/" + n, tuple(p.shape)))
    x = hu("mhah/x", (T, B, D)).cuda()
    bias = hu("mhah/b", (B, H, T, T), 2.0)
    kpm = torch.zeros(B, T, dtype=torch.bool)
    kpm[2, 5:] = True
    amask = torch.zeros(T, T)
    amask[:, 0] = -1.5
    amask[3, 4] = float("-inf")
    wq, bq = mha.qkv_weight.detach().cpu(), mha.qkv_bias.detach().cpu()
    q, k, v = (torch.nn.functional.linear(x.cpu(), wq[i * D:(i + 1) * D], bq[i * D:(i + 1) * D]) for i in range(3))
    q = (q * hd ** -0.5).view(T, B * H, hd).transpose(0, 1)
    k = k.view(T, B * H, hd).transpose(0, 1)
    v = v.view(T, B * H, hd).transpose(0, 1)
    s_ = (torch.bmm(q, k.transpose(1, 2)).view(B, H, T, T) + bias + amask[None, None]).masked_fill(kpm[:, None, None, :], float("-inf"))
    probs = torch.softmax(s_, dim=-1)
    out, w = mha(x, x, x, bias.cuda(), key_padding_mask=kpm.cuda(), attn_mask=amask.cuda(), need_head_weights=True)
    assert tuple(w.shape) == (H, B, T, T)
    torch.testing.assert_close(w.cpu(), probs.transpose(0, 1), atol=1e-5, rtol=1e-4)
    ref_out = torch.nn.functional.linear(torch.bmm(probs.view(B * H, T, T), v).transpose(0, 1).reshape(T, B, D),
                                         mha.out_proj.weight.detach().cpu(), mha.out_proj.bias.detach().cpu())
    torch.testing.assert_close(out.detach().cpu(), ref_out, atol=2e-4, rtol=1e-4)
    sc, vv = mha(x, x, x, bias.cuda(), key_padding_mask=kpm.cuda(), attn_mask=amask.cuda(), before_softmax=True)
    assert tuple(sc.shape) == (B * H, T, T) and tuple(vv.shape) == (B * H, T, hd)
    got, want = sc.cpu().view(B, H, T, T), s_
    assert torch.equal(torch.isinf(got), torch.isinf(want))
    torch.testing.assert_close(torch.where(torch.isinf(got), torch.zeros_like(got), got),
                               torch.where(torch.isinf(want), torch.zeros_like(want), want), atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(vv.cpu(), v, atol=2e-4, rtol=1e-4)
    # attn_mask alone (no attn_bias), averaged weights
    _, wm = mha(x, x, x, None, attn_mask=amask.cuda())
    s2 = torch.bmm(q, k.transpose(1, 2)).view(B, H, T, T) + amask[None, None]
    torch.testing.assert_close(wm.cpu(), torch.softmax(s2, dim=-1).mean(dim=1), atol=1e-5, rtol=1e-4)
