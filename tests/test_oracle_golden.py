"""Pin the oracle (CPU restatement) against golden vectors produced by the real reference
(oracle/gen_golden.py, build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, hashinit
from oracle import mdt_ref_cpu as R
from oracle import structure as S

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_spatial_table(golden_dir):
    g = _load(golden_dir, "spatial_table.npz")["table"]
    assert np.array_equal(g, S.spatial_table())
    # the reference's >5 overflow bucket is the (5,5) bucket
    assert S.spatial_table()[5, 5] == 20


@pytest.mark.parametrize("spm", [5, 10])
@pytest.mark.parametrize("idx", range(4))
def test_structure_bit_exact(golden_dir, idx, spm):
    name, trees = cases.structure_specs()[idx]
    g = _load(golden_dir, f"structure_{name}_spm{spm}.npz")
    for i, t in enumerate(trees):
        assert np.array_equal(g[f"parent/{i}"], t["parent"])
        assert np.array_equal(g[f"updown/{i}"], S.updown_matrix(t["parent"]))
        sp, dist, deg = S.preprocess_tree(t["parent"])
        assert np.array_equal(g[f"spatial/{i}"], sp)
        assert np.array_equal(g[f"distance/{i}"], dist)
        assert np.array_equal(g[f"degree/{i}"], deg)
    b = S.collate(trees, spm)
    for k in ("attn_bias", "spatial_pos", "in_degree", "out_degree", "x_token_mask", "x", "x_token_type_ids",
              "x_attention_mask", "x_image_indexes", "y", "y_mask"):
        ref = g["batch/" + k]
        assert ref.dtype == b[k].dtype, (k, ref.dtype, b[k].dtype)
        assert np.array_equal(ref, b[k]), k
    assert bool(g["batch/has_images"]) == (b["x_images"] is not None)
    if b["x_images"] is not None:
        assert tuple(g["batch/x_images_shape"]) == b["x_images"].shape


def _graph_inputs(g):
    return (torch.from_numpy(g["spatial_pos"]), torch.from_numpy(g["attn_bias"]),
            torch.from_numpy(g["in_degree"]), torch.from_numpy(g["key_padding_mask"]))


@pytest.mark.parametrize("D,H,Fg", [(128, 8, 128), (768, 12, 768)])
def test_graph_modules(golden_dir, D, H, Fg):
    g = _load(golden_dir, f"graph_modules_d{D}.npz")
    spatial, attn_bias, deg, kpm = _graph_inputs(g)
    B, N = deg.shape
    T = N + 1

    def w(name, shape):
        return torch.from_numpy(hashinit.param(name, shape)).requires_grad_(True)

    W = {"graph_attn_bias.spatial_pos_encoder.weight": w("graph_attn_bias.spatial_pos_encoder.weight", (512, H)),
         "graph_attn_bias.graph_token_virtual_distance.weight":
             w("graph_attn_bias.graph_token_virtual_distance.weight", (1, H))}
    bias = R.graph_attn_bias(W, attn_bias, spatial, H)
    assert np.array_equal(np.isinf(g["gab/out"]), np.isinf(bias.detach().numpy()))
    fin = ~np.isinf(g["gab/out"])
    np.testing.assert_allclose(bias.detach().numpy()[fin], g["gab/out"][fin], atol=TOL)
    cot = torch.from_numpy(hashinit.uniform("gab/cot", tuple(bias.shape)))
    (torch.where(torch.isinf(bias), torch.zeros_like(bias), bias) * cot).sum().backward()
    np.testing.assert_allclose(W["graph_attn_bias.spatial_pos_encoder.weight"].grad[:24].numpy(),
                               g["gab/d_spatial"], atol=TOL)
    np.testing.assert_allclose(W["graph_attn_bias.graph_token_virtual_distance.weight"].grad.numpy(),
                               g["gab/d_virtual"], atol=TOL)

    Wn = {f"graph_node_feature.{n}.weight": w(f"graph_node_feature.{n}.weight", s) for n, s in
          (("in_degree_encoder", (512, D)), ("out_degree_encoder", (512, D)), ("graph_token", (1, D)))}
    x = torch.from_numpy(hashinit.uniform("gnf/x", (B, N, D))).requires_grad_(True)
    y = R.graph_node_feature(Wn, x, deg, deg)
    np.testing.assert_allclose(y.detach().numpy(), g["gnf/out"], atol=TOL)
    (y * torch.from_numpy(hashinit.uniform("gnf/cot", tuple(y.shape)))).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gnf/dx"], atol=TOL)
    np.testing.assert_allclose(Wn["graph_node_feature.in_degree_encoder.weight"].grad[:8].numpy(), g["gnf/d_in"],
                               atol=TOL)
    np.testing.assert_allclose(Wn["graph_node_feature.graph_token.weight"].grad.numpy(), g["gnf/d_tok"], atol=TOL)

    hp = R.hparams(dim=D, graph_heads=H, graph_ffn=Fg)
    p = "layers.0.layers.0"
    names = [n for n in R.param_shapes(R.hparams(dim=D, graph_heads=H, graph_ffn=Fg, text_layers=12))
             if n.startswith(p + ".")]
    shapes = R.param_shapes(hp)
    for pre_ln in (False, True):
        tag = "pre" if pre_ln else "post"
        Wl = {n: w(n, shapes[n]) for n in names}
        xin = torch.from_numpy(hashinit.uniform("gl/x", (T, B, D), 1.0)).requires_grad_(True)
        b2 = torch.from_numpy(g["gab/out"]).clone().requires_grad_(True)
        yo = R.graph_layer(xin, Wl, p, H, b2, kpm, pre_ln)
        np.testing.assert_allclose(yo.detach().numpy(), g[f"layer_{tag}/out"], atol=5e-5)
        cot = torch.from_numpy(hashinit.uniform("gl/cot", (T, B, D)))
        (yo * cot).sum().backward()
        np.testing.assert_allclose(xin.grad.numpy(), g[f"layer_{tag}/dx"], atol=5e-5)
        np.testing.assert_allclose(b2.grad.numpy(), g[f"layer_{tag}/dbias"], atol=5e-5)
        for n in names:
            gn = float(g[f"layer_{tag}/gnorm/{n}"])
            mine = Wl[n].grad.double().flatten()
            assert abs(float(mine.norm()) - gn) <= 1e-4 * max(1.0, gn), n
            np.testing.assert_allclose(mine[:64].float().numpy(), g[f"layer_{tag}/gslice/{n}"], atol=5e-5)
    # bare attention
    Wl = {n: w(n, shapes[n]) for n in names}
    xq = torch.from_numpy(hashinit.uniform("mha/x", (T, B, D), 1.0)).requires_grad_(True)
    b3 = torch.from_numpy(g["gab/out"]).clone().requires_grad_(True)
    a = R.graph_mha(xq, Wl, p + ".self_attn", H, b3, kpm)
    np.testing.assert_allclose(a.detach().numpy(), g["mha/out"], atol=TOL)
    (a * torch.from_numpy(hashinit.uniform("gl/cot", (T, B, D)))).sum().backward()
    np.testing.assert_allclose(xq.grad.numpy(), g["mha/dx"], atol=TOL)
    np.testing.assert_allclose(b3.grad.numpy(), g["mha/dbias"], atol=TOL)
    np.testing.assert_allclose(Wl[p + ".self_attn.q_proj.weight"].grad.numpy(), g["mha/dWq"], atol=TOL)


def test_fusion_layer(golden_dir):
    g = _load(golden_dir, "fusion_layer.npz")
    D, H, Fe, nb, L, P, M = 768, 12, 128, 4, 10, 5, 5
    hp = R.hparams(dim=D, enc_heads=H, enc_ffn=Fe, num_bottleneck=nb, text_layers=2, vit_layers=2,
                   num_fusion_layers=0)
    p = "fusion_layers.0.fusion_layers.0"
    shapes = {n: s for n, s in R.param_shapes(hp).items() if n.startswith(p + ".")}
    img = torch.from_numpy(g["image_index"])
    am = torch.from_numpy(g["attention_mask"])
    add_mask = (1.0 - am[:, None, None, :]) * float(torch.finfo(torch.half).min)
    for with_img in (True, False):
        tag = "img" if with_img else "noimg"
        W = {n: torch.from_numpy(hashinit.param(n, s)).requires_grad_(True) for n, s in shapes.items()}
        text = torch.from_numpy(hashinit.uniform("fl/text", (M, L, D), 1.0)).requires_grad_(True)
        vit = torch.from_numpy(hashinit.uniform("fl/vit", (int(img.sum()), P, D), 1.0)).requires_grad_(True)
        bn = torch.from_numpy(hashinit.uniform("fl/bn", (M, nb, D), 1.0)).requires_grad_(True)
        t, v, b = R.fusion_layer(text, vit if with_img else None, bn, W, p, hp, add_mask, img)
        np.testing.assert_allclose(t.detach().numpy(), g[f"{tag}/text"], atol=5e-5)
        np.testing.assert_allclose(b.detach().numpy(), g[f"{tag}/bn"], atol=5e-5)
        loss = (t * torch.from_numpy(hashinit.uniform("fl/ct", tuple(t.shape)))).sum() + \
               (b * torch.from_numpy(hashinit.uniform("fl/cb", tuple(b.shape)))).sum()
        if with_img:
            np.testing.assert_allclose(v.detach().numpy(), g[f"{tag}/vit"], atol=5e-5)
            loss = loss + (v * torch.from_numpy(hashinit.uniform("fl/cv", tuple(v.shape)))).sum()
        loss.backward()
        np.testing.assert_allclose(text.grad.numpy(), g[f"{tag}/dtext"], atol=1e-4)
        np.testing.assert_allclose(bn.grad.numpy(), g[f"{tag}/dbn"], atol=1e-4)
        if with_img:
            np.testing.assert_allclose(vit.grad.numpy(), g[f"{tag}/dvit"], atol=1e-4)
        for n in shapes:
            gn = float(g[f"{tag}/gnorm/{n}"])
            if gn < 0:                       # reference never produced a gradient (dead parameter)
                assert W[n].grad is None or float(W[n].grad.norm()) == 0.0, n
                continue
            mine = W[n].grad.double().flatten()
            assert abs(float(mine.norm()) - gn) <= 2e-4 * max(1.0, gn), (n, float(mine.norm()), gn)


def full_case(kind):
    """→ (golden file name or None, hparams, trees, weight overrides) of a full-model case (oracle/cases.py)."""
    if kind in ("A", "B"):
        hp = cases.tiny_hparams(kind)
        return f"full_tiny768_{kind}.npz", hp, cases.tiny_trees(kind, hp), {}
    hp = cases.real_hparams(kind)
    fname = {"M": "full_tiny768_M.npz", "C2": "full_c2_real.npz", "LAUNCH": "full_launch.npz"}.get(kind)        # C4, C4F, C1: the reference cannot run D != 768
    return fname, hp, cases.real_trees(kind, hp), cases.weight_overrides(kind)


@pytest.mark.parametrize("kind", ["A", "B", "M", "C2", "LAUNCH"])
def test_full_model(golden_dir, kind):
    """The oracle against the REAL reference's outputs.  "M": mixed predictions (TP, FP, FN all non-zero);
    "C2": BASELINE.json configs[1] at its true geometry (L 100, 224-px images, FFN 3072, 6 + 6 layers, 64-comment tree);
    "LAUNCH": the configuration the reference ships (sample_run.sh:3 = 8 4 5 2 2 0: split 3 + 9, fusion and graph stacks of 2,
    graph FFN 768, --freeze_initial_encoders: the frozen prefix gets no gradient), two trees of its batch."""
    fname, hp, trees, over = full_case(kind)
    g = _load(golden_dir, fname)
    if kind in ("M", "C2", "LAUNCH"):
        tp, predp, totp = int(g["log/num_positive_correct"]), int(g["log/num_pred_positive"]), int(g["log/total_positive"])
        assert tp > 0 and predp > tp and totp > tp and int(g["log/ncorrect"]) > tp      # TP, FP, FN, TN all present
    batch = R.to_torch_batch(S.collate(trees, 5))
    W = R.make_weights(hp, overrides=over)
    text, bn, glob = R.encoder_forward(W, hp, batch)
    np.testing.assert_allclose(text[:, :3, :64].detach().numpy(), g["enc/text_slice"], atol=1e-4)
    np.testing.assert_allclose(bn.detach().numpy(), g["enc/bn"], atol=1e-4)
    np.testing.assert_allclose(glob.detach().numpy(), g["enc/global"], atol=1e-4)
    logits, _ = R.model_forward(W, hp, batch)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=1e-4)
    loss, counters = R.node_cross_entropy(logits, batch["y"], batch["y_mask"], hp)
    assert abs(float(loss) - float(g["loss"])) <= 2e-2      # fp16 loss: 1 ulp at ~8 is 7.8e-3
    assert counters["sample_size"] == int(g["sample_size"])
    for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
        assert counters[k] == int(g["log/" + k]), k
    m = R.f1_metrics(counters)
    for k in ("accuracy", "recall", "precision", "f1"):
        assert abs(m[k] - float(g["metric/" + k])) < 1e-6, k
    loss.backward()
    n_with = 0
    for n in R.param_shapes(hp):
        gn = float(g["gnorm/" + n])
        if gn < 0:
            assert W[n].grad is None or float(W[n].grad.abs().max()) == 0.0, n
            continue
        n_with += 1
        assert W[n].grad is not None, n
        mine = W[n].grad.double().flatten()
        assert abs(float(mine.norm()) - gn) <= 1e-3 * max(1e-3, gn) + 1e-6, (n, float(mine.norm()), gn)
        np.testing.assert_allclose(mine[:64].float().numpy(), g["gslice/" + n], atol=1e-3 * max(1.0, gn))
    assert n_with == int(g["n_trainable_with_grad"])


def test_contrastive_loss(golden_dir):
    """oracle.contrastive_loss against the REAL reference criterion (criterions/contrastive_loss.py) on hash-generated
    embeddings: loss (a half value in the reference: equal), counters (equal) and d loss / d embeddings."""
    g = _load(golden_dir, "contrastive.npz")
    y, hard = torch.from_numpy(g["y"]), torch.from_numpy(g["hard_y"])
    for tag, kw in (("adaptive", dict(scale=20.0, soft_negative_weight=0.0, adaptive=True)),
                    ("fixed", dict(scale=20.0, soft_negative_weight=0.25, adaptive=False)),
                    ("strict", dict(scale=1.0, soft_negative_weight=0.0, adaptive=False))):
        emb = torch.from_numpy(g["emb"].copy()).requires_grad_(True)
        loss, c = R.contrastive_loss(emb, y, hard, **kw)
        loss.backward()
        assert float(loss) == float(g[f"{tag}/loss"]), tag
        assert c["sample_size"] == int(g[f"{tag}/sample_size"])
        for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive"):
            assert c[k] == int(g[f"{tag}/{k}"]), (tag, k)
        np.testing.assert_allclose(emb.grad.numpy(), g[f"{tag}/d_emb"], atol=1e-6, rtol=1e-5)
    assert int(g["adaptive/pred_positive"]) not in (0, 144) and int(g["adaptive/ncorrect"]) > 0


def test_contrastive_full_model(golden_dir):
    """End to end under the contrastive objective: the oracle's encoder → global embedding → contrastive loss against
    the reference run; the FINAL graph stack (gradient-free under node_cross_entropy, quirk 3) has a gradient here."""
    g = _load(golden_dir, "contrastive.npz")
    hp = cases.tiny_hparams("A")
    trees = cases.contrastive_trees(hp)
    batch = R.to_torch_batch(S.collate(trees, 5))
    W = R.make_weights(hp)
    _, _, glob = R.encoder_forward(W, hp, batch)
    np.testing.assert_allclose(glob.detach().numpy(), g["full/global"], atol=1e-4)
    loss, c = R.contrastive_loss(glob, batch["y"], batch["hard_y"])
    assert abs(float(loss) - float(g["full/loss"])) <= 0.5 + 1e-3 * abs(float(g["full/loss"]))      # a half value: ulp 0.5 at ~900
    for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive"):
        assert c[k] == int(g[f"full/{k}"]), k
    loss.backward()
    n_with = 0
    for n in R.param_shapes(hp):
        gn = float(g["full/gnorm/" + n])
        if gn < 0:
            assert W[n].grad is None or float(W[n].grad.abs().max()) == 0.0, n
            continue
        n_with += 1
        mine = W[n].grad.double().flatten()
        assert abs(float(mine.norm()) - gn) <= 1e-3 * max(1e-3, gn) + 1e-6, (n, float(mine.norm()), gn)
        np.testing.assert_allclose(mine[:64].float().numpy(), g["full/gslice/" + n], atol=1e-3 * max(1.0, gn))
    assert n_with == int(g["full/n_trainable_with_grad"])
    assert float(g["full/gnorm/layers.2.layers.0.fc1.weight"]) > 0          # the final graph stack trains
    assert float(g["full/gnorm/node_classifier.weight"]) < 0                # the classifier head does not
