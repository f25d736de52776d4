"""Float restatement (PyTorch CPU, fp32) of the mDT forward / loss.
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  Never imported by the product.

Functional style: ``W`` is a dict {canonical state-dict name (relative to
``encoder.graph_encoder.``) → torch tensor}, ``hp`` a namespace of hyper-parameters
(see ``hparams``), ``batch`` the collated dict of torch tensors.  Gradients come from
torch autograd over this code.

Reference lines followed (paths relative to /root/reference):
  forward orchestration    mDT/src/modules/multigraphormer_graph_encoder.py:310-464
  fusion layer             mDT/src/modules/multi_graphormer_fusion_layer.py:29-71
  graph layer / stack      mDT/src/modules/graphormer_graph_encoder_layer.py:103-142,186-195
  graph attention          mDT/src/modules/multihead_attention.py:91-214
  node feature / bias      mDT/src/modules/graphormer_layers.py:39-50, 86-110
  head                     mDT/src/models/multi_modal_discussion_transformer.py:256-276
  loss + counters + F1     mDT/src/criterions/hatespeech_loss.py:66-131,133-173
  BERT / ViT blocks        external HF transformers (call sites
                           multi_graphormer_fusion_layer.py:94-96,138-146 and
                           multigraphormer_graph_encoder.py:325-335); math restated from
                           the long-standing definitions (post-LN BERT, pre-LN ViT,
                           eps 1e-12, erf GELU, tanh pooler) — "parity unpinned" by the
                           reference's own tests, pinned here by tests/golden/*.npz made
                           with the installed transformers 5.15 eager modules.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def hparams(**kw) -> SimpleNamespace:
    hp = dict(
        dim=768, enc_heads=12, graph_heads=12, enc_ffn=3072, graph_ffn=768,
        text_layers=12, vit_layers=12, num_fusion_layers=5, num_fusion_stack=1,
        num_graph_stack=1, num_bottleneck=4, vocab_size=30522, max_pos=512,
        type_vocab=2, image_size=224, patch=16, num_in_degree=512, num_out_degree=512,
        num_spatial=512, num_atoms=512 * 9, num_edges=512 * 3, pre_layernorm=False,
        encoder_normalize_before=True, pos_weight=1.0, neg_weight=1.0, fp16_loss=True,
    )
    hp.update(kw)
    hp = SimpleNamespace(**hp)
    hp.n_fusion = hp.num_fusion_layers + 1          # multigraphormer_graph_encoder.py:140-142
    hp.n_pre_text = hp.text_layers - hp.n_fusion
    hp.n_pre_vit = hp.vit_layers - hp.n_fusion
    hp.n_fusion_stacks = -(-hp.n_fusion // hp.num_fusion_stack)   # :145-158
    hp.n_graph_stacks = hp.n_fusion_stacks + 1      # :189
    return hp


# --------------------------------------------------------------------------- names
def fusion_layer_names(hp):
    """[(stack, j)] in execution order."""
    out = []
    for k in range(hp.n_fusion):
        out.append((k // hp.num_fusion_stack, k % hp.num_fusion_stack))
    return out


def param_shapes(hp) -> dict:
    """Canonical parameter names (HF-4.x inner names, reference module attribute names)
    → shapes, relative to ``encoder.graph_encoder.``.  Includes the dead parameters the
    reference constructs so state-dict layouts line up."""
    D, H, Fe, Fg = hp.dim, hp.graph_heads, hp.enc_ffn, hp.graph_ffn
    s = {}

    def lin(p, o, i):
        s[p + ".weight"] = (o, i)
        s[p + ".bias"] = (o,)

    def ln(p):
        s[p + ".weight"] = (D,)
        s[p + ".bias"] = (D,)

    s["graph_node_feature.atom_encoder.weight"] = (hp.num_atoms + 1, D)
    s["graph_node_feature.in_degree_encoder.weight"] = (hp.num_in_degree, D)
    s["graph_node_feature.out_degree_encoder.weight"] = (hp.num_out_degree, D)
    s["graph_node_feature.graph_token.weight"] = (1, D)
    s["graph_attn_bias.edge_encoder.weight"] = (hp.num_edges + 1, H)
    s["graph_attn_bias.spatial_pos_encoder.weight"] = (hp.num_spatial, H)
    s["graph_attn_bias.graph_token_virtual_distance.weight"] = (1, H)
    if hp.encoder_normalize_before:
        ln("emb_layer_norm")
    for st in range(hp.n_graph_stacks):
        for j in range(hp.num_graph_stack):
            p = f"layers.{st}.layers.{j}"
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(f"{p}.self_attn.{n}", D, D)
            ln(f"{p}.self_attn_layer_norm")
            lin(f"{p}.fc1", Fg, D)
            lin(f"{p}.fc2", D, Fg)
            ln(f"{p}.final_layer_norm")

    def bert_layer(p):
        for n in ("query", "key", "value"):
            lin(f"{p}.attention.self.{n}", D, D)
        lin(f"{p}.attention.output.dense", D, D)
        ln(f"{p}.attention.output.LayerNorm")
        lin(f"{p}.intermediate.dense", Fe, D)
        lin(f"{p}.output.dense", D, Fe)
        ln(f"{p}.output.LayerNorm")

    def vit_layer(p):
        for n in ("query", "key", "value"):
            lin(f"{p}.attention.attention.{n}", D, D)
        lin(f"{p}.attention.output.dense", D, D)
        lin(f"{p}.intermediate.dense", Fe, D)
        lin(f"{p}.output.dense", D, Fe)
        ln(f"{p}.layernorm_before")
        ln(f"{p}.layernorm_after")

    s["text_model.embeddings.word_embeddings.weight"] = (hp.vocab_size, D)
    s["text_model.embeddings.position_embeddings.weight"] = (hp.max_pos, D)
    s["text_model.embeddings.token_type_embeddings.weight"] = (hp.type_vocab, D)
    ln("text_model.embeddings.LayerNorm")
    for i in range(hp.n_pre_text):
        bert_layer(f"text_model.encoder.layer.{i}")
    lin("text_model.pooler.dense", D, D)
    npatch = (hp.image_size // hp.patch) ** 2
    s["vit_model.embeddings.cls_token"] = (1, 1, D)
    s["vit_model.embeddings.position_embeddings"] = (1, npatch + 1, D)
    s["vit_model.embeddings.patch_embeddings.projection.weight"] = (D, 3, hp.patch, hp.patch)
    s["vit_model.embeddings.patch_embeddings.projection.bias"] = (D,)
    for i in range(hp.n_pre_vit):
        vit_layer(f"vit_model.encoder.layer.{i}")
    ln("vit_model.layernorm")
    lin("vit_model.pooler.dense", D, D)
    for st, j in fusion_layer_names(hp):
        p = f"fusion_layers.{st}.fusion_layers.{j}"
        bert_layer(p + ".bert_encoder")
        vit_layer(p + ".vit_encoder")
        lin(p + ".bert_projection", D, D)
        lin(p + ".vit_projection", D, D)
    lin("node_classifier", 2, D)
    s["bottle_neck.weight"] = (hp.num_bottleneck, D)
    return s


def make_weights(hp, dtype=torch.float32, requires_grad=True, overrides=None) -> dict:
    """``overrides``: {canonical name: array} replacing the hash fill (oracle/cases.py weight_overrides)."""
    from . import hashinit
    W = {}
    for name, shape in param_shapes(hp).items():
        if overrides and name in overrides:
            t = torch.from_numpy(overrides[name].reshape(shape).copy()).to(dtype)
        else:
            t = torch.from_numpy(hashinit.param(name, shape)).to(dtype)
        W[name] = t.requires_grad_(requires_grad and not is_frozen(hp, name))
    return W


def is_frozen(hp, name: str) -> bool:
    """--freeze_initial_encoders (multigraphormer_graph_encoder.py:223-228): every parameter of text_model and vit_model —
    embeddings, the pre-fusion layers left there after the fusion layers were sliced out, the ViT's final LayerNorm —
    except the two poolers, which are unfrozen again."""
    if not getattr(hp, "freeze_initial_encoders", False):
        return False
    return (name.startswith("text_model.") or name.startswith("vit_model.")) and ".pooler." not in name


# --------------------------------------------------------------------------- blocks
def _heads(x, h):
    b, s, d = x.shape
    return x.view(b, s, h, d // h).transpose(1, 2)


def _mha(x, W, p, names, nheads, add_mask):
    """x [b,s,D]; HF-style attention: scores / sqrt(d) + mask → softmax → PV."""
    q = F.linear(x, W[f"{p}.{names[0]}.weight"], W[f"{p}.{names[0]}.bias"])
    k = F.linear(x, W[f"{p}.{names[1]}.weight"], W[f"{p}.{names[1]}.bias"])
    v = F.linear(x, W[f"{p}.{names[2]}.weight"], W[f"{p}.{names[2]}.bias"])
    q, k, v = _heads(q, nheads), _heads(k, nheads), _heads(v, nheads)
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    pr = torch.softmax(s, dim=-1)
    o = (pr @ v).transpose(1, 2).reshape(x.shape)
    return o


def bert_layer(x, W, p, nheads, add_mask):
    a = _mha(x, W, p + ".attention.self", ("query", "key", "value"), nheads, add_mask)
    a = F.linear(a, W[p + ".attention.output.dense.weight"], W[p + ".attention.output.dense.bias"])
    a = F.layer_norm(a + x, (x.shape[-1],), W[p + ".attention.output.LayerNorm.weight"],
                     W[p + ".attention.output.LayerNorm.bias"], 1e-12)
    h = F.gelu(F.linear(a, W[p + ".intermediate.dense.weight"], W[p + ".intermediate.dense.bias"]))
    y = F.linear(h, W[p + ".output.dense.weight"], W[p + ".output.dense.bias"])
    return F.layer_norm(y + a, (x.shape[-1],), W[p + ".output.LayerNorm.weight"],
                        W[p + ".output.LayerNorm.bias"], 1e-12)


def vit_layer(x, W, p, nheads):
    D = x.shape[-1]
    n = F.layer_norm(x, (D,), W[p + ".layernorm_before.weight"], W[p + ".layernorm_before.bias"], 1e-12)
    a = _mha(n, W, p + ".attention.attention", ("query", "key", "value"), nheads, None)
    h = F.linear(a, W[p + ".attention.output.dense.weight"], W[p + ".attention.output.dense.bias"]) + x
    n = F.layer_norm(h, (D,), W[p + ".layernorm_after.weight"], W[p + ".layernorm_after.bias"], 1e-12)
    f = F.gelu(F.linear(n, W[p + ".intermediate.dense.weight"], W[p + ".intermediate.dense.bias"]))
    return F.linear(f, W[p + ".output.dense.weight"], W[p + ".output.dense.bias"]) + h


def bert_embeddings(ids, types, W):
    p = "text_model.embeddings"
    L = ids.shape[1]
    e = (W[p + ".word_embeddings.weight"][ids] + W[p + ".token_type_embeddings.weight"][types]
         + W[p + ".position_embeddings.weight"][:L][None])
    return F.layer_norm(e, (e.shape[-1],), W[p + ".LayerNorm.weight"], W[p + ".LayerNorm.bias"], 1e-12)


def vit_embeddings(images, W, hp):
    p = "vit_model.embeddings"
    x = F.conv2d(images, W[p + ".patch_embeddings.projection.weight"],
                 W[p + ".patch_embeddings.projection.bias"], stride=hp.patch)
    x = x.flatten(2).transpose(1, 2)
    cls = W[p + ".cls_token"].expand(x.shape[0], -1, -1)
    return torch.cat([cls, x], dim=1) + W[p + ".position_embeddings"]


def graph_attn_bias(W, attn_bias, spatial_pos, nheads):
    """graphormer_layers.py:86-110 — note attn_bias enters twice (clone :93, "reset" :108)."""
    B, T, _ = attn_bias.shape
    g = attn_bias.unsqueeze(1).repeat(1, nheads, 1, 1).clone()
    # nn.Embedding(padding_idx=0): row 0 is read in forward but never receives gradient
    sp = F.embedding(spatial_pos.long(), W["graph_attn_bias.spatial_pos_encoder.weight"],
                     padding_idx=0).permute(0, 3, 1, 2)
    g[:, :, 1:, 1:] = g[:, :, 1:, 1:] + sp
    t = W["graph_attn_bias.graph_token_virtual_distance.weight"].view(1, nheads, 1)
    g[:, :, 1:, 0] = g[:, :, 1:, 0] + t
    g[:, :, 0, :] = g[:, :, 0, :] + t
    return g + attn_bias.unsqueeze(1)


def graph_node_feature(W, x, in_degree, out_degree):
    """graphormer_layers.py:39-50."""
    p = "graph_node_feature"
    nf = (x + F.embedding(in_degree, W[p + ".in_degree_encoder.weight"], padding_idx=0)
          + F.embedding(out_degree, W[p + ".out_degree_encoder.weight"], padding_idx=0))
    tok = W[p + ".graph_token.weight"].unsqueeze(0).repeat(x.shape[0], 1, 1)
    return torch.cat([tok, nf], dim=1)


def graph_mha(x, W, p, nheads, bias, key_padding_mask):
    """multihead_attention.py:91-214; x is [T,B,D] (time-major)."""
    T, B, D = x.shape
    hd = D // nheads
    q = F.linear(x, W[p + ".q_proj.weight"], W[p + ".q_proj.bias"]) * hd ** -0.5
    k = F.linear(x, W[p + ".k_proj.weight"], W[p + ".k_proj.bias"])
    v = F.linear(x, W[p + ".v_proj.weight"], W[p + ".v_proj.bias"])
    q = q.contiguous().view(T, B * nheads, hd).transpose(0, 1)
    k = k.contiguous().view(T, B * nheads, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * nheads, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if bias is not None:
        s = s + bias.reshape(B * nheads, T, T)
    if key_padding_mask is not None:
        s = s.view(B, nheads, T, T).masked_fill(
            key_padding_mask[:, None, None, :].bool(), float("-inf")).view(B * nheads, T, T)
    pr = torch.softmax(s.float(), dim=-1).type_as(s)
    o = torch.bmm(pr, v).transpose(0, 1).contiguous().view(T, B, D)
    return F.linear(o, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def graph_layer(x, W, p, nheads, bias, key_padding_mask, pre_ln=False):
    """graphormer_graph_encoder_layer.py:103-142 (dropout = identity in parity runs)."""
    D = x.shape[-1]

    def ln(t, n):
        return F.layer_norm(t, (D,), W[f"{p}.{n}.weight"], W[f"{p}.{n}.bias"], 1e-5)

    r = x
    if pre_ln:
        x = ln(x, "self_attn_layer_norm")
    x = r + graph_mha(x, W, p + ".self_attn", nheads, bias, key_padding_mask)
    if not pre_ln:
        x = ln(x, "self_attn_layer_norm")
    r = x
    if pre_ln:
        x = ln(x, "final_layer_norm")
    x = F.gelu(F.linear(x, W[p + ".fc1.weight"], W[p + ".fc1.bias"]).float()).type_as(x)
    x = r + F.linear(x, W[p + ".fc2.weight"], W[p + ".fc2.bias"])
    if not pre_ln:
        x = ln(x, "final_layer_norm")
    return x


def graph_stack(x, W, st, hp, bias, kpm):
    for j in range(hp.num_graph_stack):
        x = graph_layer(x, W, f"layers.{st}.layers.{j}", hp.graph_heads, bias, kpm, hp.pre_layernorm)
    return x


def fusion_layer(text, vit, bn, W, p, hp, add_mask, img_idx):
    """multi_graphormer_fusion_layer.py:29-71.  The ViT branch reads the *input*
    bottleneck (:57), and image comments average the two bottleneck outputs (:63-66)."""
    nb = hp.num_bottleneck
    out = bert_layer(torch.cat([bn, text], dim=1), W, p + ".bert_encoder", hp.enc_heads, add_mask)
    text_out, bn_out = out[:, nb:], out[:, :nb]
    if vit is None:
        return text_out, None, bn_out
    vout = vit_layer(torch.cat([bn[img_idx], vit], dim=1), W, p + ".vit_encoder", hp.enc_heads)
    bn_new = bn_out.clone()
    bn_new[img_idx] = (vout[:, :nb] + bn_out[img_idx]) / 2
    return text_out, vout[:, nb:], bn_new


def fusion_stack(text, vit, bn, W, st, hp, add_mask, img_idx):
    n = min(hp.num_fusion_stack, hp.n_fusion - st * hp.num_fusion_stack)
    for j in range(n):
        text, vit, bn = fusion_layer(text, vit, bn, W, f"fusion_layers.{st}.fusion_layers.{j}",
                                     hp, add_mask, img_idx)
    return text, vit, bn


def encoder_forward(W, hp, batch):
    """→ (text [M,L,D], bottleneck [M,nb,D], global [B,D]);
    multigraphormer_graph_encoder.py:310-464."""
    mask = batch["x_token_mask"]
    ids = batch["x"][mask]
    types = batch["x_token_type_ids"][mask]
    am = batch["x_attention_mask"][mask]
    M = ids.shape[0]
    fmin = torch.finfo(torch.float32).min
    text = bert_embeddings(ids, types, W)
    pre_mask = (1.0 - am[:, None, None, :].float()) * fmin
    for i in range(hp.n_pre_text):
        text = bert_layer(text, W, f"text_model.encoder.layer.{i}", hp.enc_heads, pre_mask)
    vit = None
    if batch.get("x_images") is not None:
        vit = vit_embeddings(batch["x_images"], W, hp)
        for i in range(hp.n_pre_vit):
            vit = vit_layer(vit, W, f"vit_model.encoder.layer.{i}", hp.enc_heads)
        D = vit.shape[-1]
        # ViTModel applies its final LayerNorm after the kept layers (:332-335, quirk 5)
        vit = F.layer_norm(vit, (D,), W["vit_model.layernorm.weight"], W["vit_model.layernorm.bias"], 1e-12)
    nb = hp.num_bottleneck
    bn = W["bottle_neck.weight"].unsqueeze(0).repeat(M, 1, 1)
    full = torch.cat([torch.ones(M, nb, dtype=am.dtype), am], dim=1)
    half_min = float(torch.finfo(torch.half).min)          # :348-354 (quirk 9)
    add_mask = (1.0 - full[:, None, None, :].float()) * half_min
    img_idx = batch["x_image_indexes"]

    text, vit, bn = fusion_stack(text, vit, bn, W, 0, hp, add_mask, img_idx)
    B, N = batch["x"].shape[:2]
    D = hp.dim
    graph = torch.zeros(B, N, D, dtype=bn.dtype)
    graph = graph.masked_scatter(mask[..., None].expand(B, N, D), bn[:, 0])
    kpm = torch.cat([torch.zeros(B, 1, dtype=torch.bool), ~mask], dim=1)   # quirk 7
    tmask = torch.cat([torch.zeros(B, 1, dtype=torch.bool), mask], dim=1)
    x = graph_node_feature(W, graph, batch["in_degree"], batch["out_degree"])
    bias = graph_attn_bias(W, batch["attn_bias"], batch["spatial_pos"], hp.graph_heads)
    if hp.encoder_normalize_before:
        x = F.layer_norm(x, (D,), W["emb_layer_norm.weight"], W["emb_layer_norm.bias"], 1e-5)
    x = x.transpose(0, 1)
    Fs = hp.n_fusion_stacks
    for st in range(Fs - 1):                       # zip(layers, fusion_layers[1:]) :413
        x = graph_stack(x, W, st, hp, bias, kpm)
        xb = x.transpose(0, 1)
        bn = torch.cat([xb[tmask].unsqueeze(1), bn[:, 1:]], dim=1)           # :425
        text, vit, bn = fusion_stack(text, vit, bn, W, st + 1, hp, add_mask, img_idx)
        xb = xb.masked_scatter(tmask[..., None].expand_as(xb), bn[:, 0])       # :435
        x = xb.transpose(0, 1)
    x = graph_stack(x, W, hp.n_graph_stacks - 1, hp, bias, kpm)                # layers[-1] :441
    return text, bn, x[0]


def model_forward(W, hp, batch):
    """→ (logits [M,2], global [B,D]);  models/multi_modal_discussion_transformer.py:256-276."""
    text, bn, glob = encoder_forward(W, hp, batch)

    def head(t):
        pooled = torch.tanh(F.linear(t[:, 0], W["text_model.pooler.dense.weight"],
                                     W["text_model.pooler.dense.bias"]))
        return F.linear(pooled, W["node_classifier.weight"], W["node_classifier.bias"])

    # the reference applies pooler → dropout → classifier to the whole [M,S,D] tensors;
    # the pooler keeps token 0 only, so this is identical
    return (head(text) + head(bn)) / 2, glob


def node_cross_entropy(logits_all, y, y_mask, hp):
    """criterions/hatespeech_loss.py:66-131.  fp16 logits / weights (quirk 14)."""
    logits = logits_all[y_mask]
    w = torch.tensor([hp.neg_weight, hp.pos_weight])
    if hp.fp16_loss:
        logits, w = logits.half(), w.half()
    targets = y.flatten().long()
    with torch.no_grad():
        pred = torch.argmax(torch.softmax(logits.float(), dim=-1), dim=-1)
        counters = dict(
            ncorrect=int((pred == targets).sum()),
            num_positive_correct=int(((pred == targets) & (pred == 1)).sum()),
            total_positive=int((targets == 1).sum()),
            num_pred_positive=int((pred == 1).sum()),
            sample_size=int(logits.shape[0]),
        )
    loss = F.cross_entropy(logits, targets, reduction="sum", weight=w)
    return loss, counters


def contrastive_loss(embeddings, y, hard_y, scale=20.0, soft_negative_weight=0.0, adaptive=True):
    """criterions/contrastive_loss.py:76-180 on CPU tensors: → (loss, counters).  The broadcasts are the reference's:
    ``extra_weight`` [B] and ``targets`` [B] act along the LAST axis of the [B, B] matrices (:143-147, :155)."""
    nA = F.normalize(embeddings, p=2, dim=1)
    sim = (torch.mm(nA, nA.transpose(0, 1)) * scale).float()
    targets = y.float()
    # .half() as in the reference (:122, :126): binary_cross_entropy_with_logits then runs its in-place chain on a
    # HALF tensor (every step rounds to half, the loss value is a half); its backward formula works in fp32
    target_matrix = targets.unsqueeze(1).eq(targets).half()
    hard_matrix = hard_y.float().unsqueeze(1).eq(targets).half()
    soft_labels = torch.logical_and(target_matrix.eq(0), hard_matrix.eq(0))
    if adaptive:
        num_hard = torch.logical_or(target_matrix.eq(1), hard_matrix.eq(1)).sum(dim=1)
        extra = (num_hard / soft_labels.sum(dim=1)) * 2
    else:
        extra = soft_negative_weight
    soft_matrix = torch.where(soft_labels, extra, 1)
    soft_matrix = torch.where(torch.eye(soft_matrix.size(0)).eq(1), 0, soft_matrix)
    with torch.no_grad():
        pred = torch.sigmoid(sim).round()
        counters = dict(ncorrect=int((pred == targets).sum()),
                        positive_correct=int(torch.logical_and(pred == targets, pred == 1).sum()),
                        total_positive=int((targets == 1).sum()), pred_positive=int((pred == 1).sum()),
                        sample_size=int(sim.shape[0] * sim.shape[1]))
    loss = F.binary_cross_entropy_with_logits(sim, target_matrix, weight=soft_matrix, reduction="sum").float()
    return loss, counters


def f1_metrics(c: dict) -> dict:
    """hatespeech_loss.py:133-173 (zero guards included)."""
    tp, totp, predp, n = (c["num_positive_correct"], c["total_positive"],
                          c["num_pred_positive"], c["sample_size"])
    recall = 0 if totp == 0 else tp / totp
    precision = 0 if predp == 0 else tp / predp
    f1 = 0 if (precision == 0 and recall == 0) else 2 * (precision * recall) / (precision + recall)
    return dict(accuracy=c["ncorrect"] / n, recall=recall, precision=precision, f1=f1)


def to_torch_batch(np_batch: dict) -> dict:
    out = {}
    for k, v in np_batch.items():
        out[k] = None if v is None else torch.from_numpy(v)
    return out
