"""Generate golden vectors by running the REAL reference (build container only).

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.gen_golden   (from the repo root)

The reference Python is imported read-only from /root/reference with in-memory
stand-ins for the absent ``fairseq`` package (5 helper symbols + registration
decorators), ``Tensor.cuda`` neutralised, and ``build_vit_bert_encoders`` overridden to
build random-init HF models from configs (no network) behind adapters that expose the
transformers-4.x call convention the reference uses.  Parameters are filled through
``oracle.hashinit`` by canonical name, inputs come from the synthetic generator, so the
fixtures in tests/golden/ hold only outputs (plus the integer inputs of the structural
cases).  Nothing here travels to the GPU box except the .npz files it writes.
"""
from __future__ import annotations

import ast
import importlib
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import hashinit  # noqa: E402
from oracle import mdt_ref_cpu as R  # noqa: E402
from oracle import structure as S  # noqa: E402
from oracle import cases  # noqa: E402
from oracle.cases import tiny_hparams, tiny_trees  # noqa: E402
from multimodaldiscussiontransformer_amd import synthetic  # noqa: E402


# --------------------------------------------------------------------------- stand-ins
def install_fairseq_standins():
    fs = types.ModuleType("fairseq")
    utils = types.ModuleType("fairseq.utils")
    utils.softmax = lambda x, dim, onnx_trace=False: F.softmax(x, dim=dim, dtype=torch.float32)

    def get_activation_fn(name):
        if name == "gelu":
            return lambda x: F.gelu(x.float()).type_as(x)
        if name == "relu":
            return F.relu
        raise KeyError(name)

    utils.get_activation_fn = get_activation_fn
    utils.get_available_activation_fns = lambda: ["relu", "gelu"]
    utils.safe_hasattr = lambda o, k: getattr(o, k, None) is not None
    fs.utils = utils

    modules = types.ModuleType("fairseq.modules")
    modules.LayerNorm = lambda d, eps=1e-5, elementwise_affine=True, export=False: nn.LayerNorm(d, eps, elementwise_affine)

    class FairseqDropout(nn.Module):
        def __init__(self, p, module_name=None):
            super().__init__()
            self.p = p

        def forward(self, x, inplace=False):
            return F.dropout(x, p=self.p, training=self.training) if self.p > 0 else x

    modules.FairseqDropout = FairseqDropout
    modules.LayerDropModuleList = nn.ModuleList
    fd = types.ModuleType("fairseq.modules.fairseq_dropout")
    fd.FairseqDropout = FairseqDropout
    qn = types.ModuleType("fairseq.modules.quant_noise")
    qn.quant_noise = lambda m, p, bs: m
    models = types.ModuleType("fairseq.models")

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary=None):
            super().__init__()

    class FairseqEncoderModel(nn.Module):
        def __init__(self, encoder):
            super().__init__()
            self.encoder = encoder

    models.FairseqEncoder = FairseqEncoder
    models.FairseqEncoderModel = FairseqEncoderModel
    models.register_model = lambda name: (lambda c: c)
    models.register_model_architecture = lambda m, a: (lambda f: f)

    crit = types.ModuleType("fairseq.criterions")

    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task

    crit.FairseqCriterion = FairseqCriterion
    crit.register_criterion = lambda name, dataclass=None: (lambda c: c)
    metrics = types.ModuleType("fairseq.metrics")
    metrics.LOG = {}
    metrics.log_scalar = lambda k, v, w=1, round=None: metrics.LOG.__setitem__(k, float(v))
    fs.metrics = metrics
    dc = types.ModuleType("fairseq.dataclass")
    dcc = types.ModuleType("fairseq.dataclass.configs")

    class FairseqDataclass:
        pass

    dcc.FairseqDataclass = FairseqDataclass
    dc.configs = dcc
    dc.FairseqDataclass = FairseqDataclass
    for name, mod in {
        "fairseq": fs, "fairseq.utils": utils, "fairseq.modules": modules,
        "fairseq.modules.fairseq_dropout": fd, "fairseq.modules.quant_noise": qn,
        "fairseq.models": models, "fairseq.criterions": crit, "fairseq.metrics": metrics,
        "fairseq.dataclass": dc, "fairseq.dataclass.configs": dcc,
    }.items():
        sys.modules[name] = mod
    return metrics


def mount_reference():
    """Expose /root/reference/mDT/src as package ``src`` without running its __init__
    (which pulls fairseq.criterions registration for every criterion file)."""
    pkg = types.ModuleType("src")
    pkg.__path__ = [os.path.join(REF, "mDT", "src")]
    sys.modules["src"] = pkg
    torch.Tensor.cuda = lambda self, *a, **k: self
    mods = importlib.import_module("src.modules")
    models = importlib.import_module("src.models")
    collator = importlib.import_module("src.data.collator")
    spec = importlib.util.spec_from_file_location(
        "ref_pre_processing", os.path.join(REF, "mDT/src/data/pyg_datasets/pre_processing.py"))
    pre = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pre)
    loss = importlib.import_module("src.criterions.hatespeech_loss")
    global CONTRASTIVE_MOD
    CONTRASTIVE_MOD = importlib.import_module("src.criterions.contrastive_loss")
    return mods, models, collator, pre, loss


def load_updown_functions():
    """get_relative_depth / spread_downwards are plain-Python methods of a class whose
    module needs torch_geometric; compile just those two functions from the file text."""
    path = os.path.join(REF, "mDT/experiments/hateful_discussions/datasets/hateful_discussions.py")
    tree = ast.parse(open(path).read())
    fns = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("get_relative_depth", "spread_downwards"):
            fns.append(node)
    module = ast.Module(body=fns, type_ignores=[])
    ns = {"copy": __import__("copy")}
    exec(compile(module, path, "exec"), ns)
    holder = type("Holder", (), {"get_relative_depth": ns["get_relative_depth"],
                                 "spread_downwards": ns["spread_downwards"]})
    return holder()


# --------------------------------------------------------------------------- HF adapters
class BertLayer4x(nn.Module):
    """transformers-4.x BertLayer call convention over the installed 5.x layer."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, hidden, attention_mask=None, head_mask=None, enc_h=None, enc_m=None,
                past=None, output_attentions=False):
        return (self.inner(hidden, attention_mask),)


class ViTLayer4x(nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, hidden, head_mask=None, output_attentions=False):
        return (self.inner(hidden),)


def make_builder(hp):
    from transformers import BertConfig, BertForSequenceClassification, ViTConfig, ViTModel

    def build(self, num_fusion_layers, attention_dropout, activation_dropout):
        vc = ViTConfig(hidden_size=hp.dim, num_hidden_layers=hp.vit_layers, num_attention_heads=hp.enc_heads,
                       intermediate_size=hp.enc_ffn, image_size=hp.image_size, patch_size=hp.patch,
                       hidden_dropout_prob=activation_dropout, attention_probs_dropout_prob=attention_dropout)
        vc._attn_implementation = "eager"
        bc = BertConfig(hidden_size=hp.dim, num_hidden_layers=hp.text_layers, num_attention_heads=hp.enc_heads,
                        intermediate_size=hp.enc_ffn, vocab_size=hp.vocab_size, max_position_embeddings=hp.max_pos,
                        type_vocab_size=hp.type_vocab, hidden_dropout_prob=activation_dropout,
                        attention_probs_dropout_prob=attention_dropout, num_labels=2)
        bc._attn_implementation = "eager"
        vit = ViTModel(vc)
        bert = BertForSequenceClassification(bc)
        bm = bert.bert
        n = num_fusion_layers
        vit_other = [ViTLayer4x(l) for l in vit.layers[-n:]]
        vit.layers = vit.layers[:-n]
        bert_other = [BertLayer4x(l) for l in bm.encoder.layer[-n:]]
        bm.encoder.layer = bm.encoder.layer[:-n]
        return (vit, vit.pooler, vit_other, bm, bm.pooler, bert_other, bert.classifier, bert.dropout)

    return build


_VIT5_TO_4 = [
    (".attention.q_proj.", ".attention.attention.query."),
    (".attention.k_proj.", ".attention.attention.key."),
    (".attention.v_proj.", ".attention.attention.value."),
    (".attention.o_proj.", ".attention.output.dense."),
    (".mlp.fc1.", ".intermediate.dense."),
    (".mlp.fc2.", ".output.dense."),
]


def canonical(name: str) -> str:
    """reference-side parameter name (installed transformers 5.x, adapters) → canonical
    transformers-4.x name used by checkpoints of the reference."""
    name = name.replace(".inner.", ".")
    if "vit_encoder" in name or name.startswith("vit_model."):
        for a, b in _VIT5_TO_4:
            name = name.replace(a, b)
        name = name.replace("vit_model.layers.", "vit_model.encoder.layer.")
    return name


def fill_params(module: nn.Module, prefix: str = "", overrides=None):
    names = {}
    with torch.no_grad():
        for n, p in module.named_parameters():
            c = prefix + canonical(n)
            if overrides and c in overrides:
                p.copy_(torch.from_numpy(np.asarray(overrides[c]).reshape(tuple(p.shape)).copy()))
            else:
                p.copy_(torch.from_numpy(hashinit.param(c, tuple(p.shape))))
            names[c] = p
    return names


def build_reference_encoder(mods, hp):
    cls = mods.MultiGraphormerGraphEncoder
    cls.build_vit_bert_encoders = make_builder(hp)
    enc = cls(
        num_atoms=hp.num_atoms, num_in_degree=hp.num_in_degree, num_out_degree=hp.num_out_degree,
        num_edges=hp.num_edges, num_spatial=hp.num_spatial, num_edge_dis=128,
        num_bottle_neck=hp.num_bottleneck, num_fusion_layers=hp.num_fusion_layers, edge_type="",
        multi_hop_max_dist=5, num_fusion_stack=hp.num_fusion_stack, num_graph_stack=hp.num_graph_stack,
        embedding_dim=hp.dim, ffn_embedding_dim=hp.graph_ffn, num_attention_heads=hp.graph_heads,
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        encoder_normalize_before=hp.encoder_normalize_before, pre_layernorm=hp.pre_layernorm,
        activation_fn="gelu", freeze_initial_encoders=bool(getattr(hp, "freeze_initial_encoders", False)),
    )
    return enc


# --------------------------------------------------------------------------- cases
def np_(t):
    return t.detach().cpu().numpy()


def grad_summary(named, out):
    for n, p in named.items():
        g = p.grad
        if g is None:
            out["gnorm/" + n] = np.float64(-1.0)      # parameter never received a gradient
            continue
        gf = g.detach().double().flatten()
        out["gnorm/" + n] = gf.norm().numpy()
        out["gslice/" + n] = gf[:64].float().numpy()


def ref_items_from_trees(trees, pre, updown):
    """Build reference ``item`` objects → preprocess_item → collator 9-tuples."""
    items = []
    for i, t in enumerate(trees):
        parent = t["parent"]
        n = len(parent)
        # nested comment dicts for the reference's recursive distance passes
        nodes = [{"id": k, "tree": []} for k in range(n)]
        for k in range(1, n):
            nodes[int(parent[k])]["tree"].append(nodes[k])
        updown.get_relative_depth(nodes[0])
        updown.spread_downwards(nodes[0])
        dm = [[nodes[a]["distances"][b] for b in range(n)] for a in range(n)]
        src = [k for k in range(1, n)] + [int(parent[k]) for k in range(1, n)]
        dst = [int(parent[k]) for k in range(1, n)] + [k for k in range(1, n)]
        item = SimpleNamespace(
            edge_attr=None,
            edge_index=torch.tensor([src, dst], dtype=torch.long).reshape(2, -1),
            x={"input_ids": torch.from_numpy(t["input_ids"]),
               "token_type_ids": torch.from_numpy(t["token_type_ids"]),
               "attention_mask": torch.from_numpy(t["attention_mask"])},
            distance_matrix=dm,
        )
        item = pre.preprocess_item(item)
        imgs = torch.from_numpy(t["images"]) if t["images"] is not None else torch.zeros(1, 3, 8, 8)
        items.append(SimpleNamespace(
            idx=i, attn_bias=item.attn_bias, spatial_pos=item.spatial_pos, in_degree=item.in_degree,
            x=item.x, x_image_index=torch.from_numpy(t["image_index"].astype(np.float32)),
            x_images=imgs, distance=item.distance, y=torch.from_numpy(t["y"]),
            y_mask=torch.from_numpy(t["y_mask"]), dm=dm))
    return items


def ref_collate(items, collator_mod, spatial_pos_max):
    tup = [(it.idx, it.attn_bias, it.spatial_pos, it.in_degree, it.x, it.x_image_index, it.x_images,
            it.distance, it.y) for it in items]
    out = collator_mod.collator(tup, spatial_pos_max)
    out["y_mask"] = torch.cat([it.y_mask for it in items]).bool()      # dataset.py:210-213
    return out


def case_structure(pre, collator_mod, updown):
    """Integer tensors of preprocess_item + collator (bit-exact targets)."""
    specs = cases.structure_specs()
    for name, trees in specs:
        for spm in (5, 10):
            items = ref_items_from_trees(trees, pre, updown)
            out = {}
            for i, it in enumerate(items):
                out[f"parent/{i}"] = trees[i]["parent"]
                out[f"updown/{i}"] = np.asarray(it.dm, dtype=np.int64)
                out[f"spatial/{i}"] = np_(it.spatial_pos).astype(np.int64)
                out[f"distance/{i}"] = np_(it.distance).astype(np.int64)
                out[f"degree/{i}"] = np_(it.in_degree)
            b = ref_collate(items, collator_mod, spm)
            for k in ("attn_bias", "spatial_pos", "in_degree", "out_degree", "x_token_mask", "x",
                      "x_token_type_ids", "x_attention_mask", "x_image_indexes", "y", "y_mask"):
                out["batch/" + k] = np_(b[k])
            out["batch/has_images"] = np.asarray(b["x_images"] is not None)
            if b["x_images"] is not None:
                out["batch/x_images_shape"] = np.asarray(b["x_images"].shape)
            out["seed"] = np.asarray(7)
            np.savez_compressed(os.path.join(OUT, f"structure_{name}_spm{spm}.npz"), **out)
    # the 21-bucket table itself
    tbl = np.zeros((6, 6), dtype=np.int64)
    for u in range(6):
        for d in range(6):
            it = SimpleNamespace(edge_attr=None, edge_index=torch.zeros(2, 0, dtype=torch.long),
                                 x={"input_ids": torch.zeros(1, 1)}, distance_matrix=[[[u, d]]])
            tbl[u, d] = int(pre.preprocess_item(it).spatial_pos[0, 0])
    np.savez_compressed(os.path.join(OUT, "spatial_table.npz"), table=tbl)


def case_graph_modules(mods):
    """GraphAttnBias / GraphNodeFeature / MultiheadAttention / GraphormerGraphEncoderLayer."""
    for D, H, Fg in ((128, 8, 128), (768, 12, 768)):
        B, N = 3, 7
        T = N + 1
        rng = np.random.Generator(np.random.PCG64(11))
        nreal = [7, 4, 1]
        spatial = np.zeros((B, N, N), dtype=np.int32)
        attn_bias = np.full((B, T, T), -np.inf, dtype=np.float32)
        deg = np.zeros((B, N), dtype=np.int64)
        for b, n in enumerate(nreal):
            spatial[b, :n, :n] = rng.integers(1, 22, size=(n, n))
            ab = np.zeros((n + 1, n + 1), dtype=np.float32)
            ab[1:, 1:][rng.random((n, n)) < 0.2] = -np.inf
            ab[np.arange(1, n + 1), np.arange(1, n + 1)] = 0
            attn_bias[b, : n + 1, : n + 1] = ab
            attn_bias[b, n + 1:, : n + 1] = 0
            deg[b, :n] = rng.integers(1, 6, size=n)
        kpm = np.zeros((B, T), dtype=bool)
        for b, n in enumerate(nreal):
            kpm[b, n + 1:] = True
        out = dict(spatial_pos=spatial, attn_bias=attn_bias, in_degree=deg, key_padding_mask=kpm)
        bd = dict(attn_bias=torch.from_numpy(attn_bias), spatial_pos=torch.from_numpy(spatial),
                  x=torch.zeros(B, N, 1))
        gab = mods.GraphAttnBias(num_heads=H, num_atoms=16, num_edges=16, num_spatial=512, num_edge_dis=8,
                                 hidden_dim=D, edge_type="", multi_hop_max_dist=5, n_layers=4)
        fill_params(gab, "graph_attn_bias.")
        bias = gab(bd)
        out["gab/out"] = np_(bias)
        wgt = torch.from_numpy(hashinit.uniform("gab/cot", tuple(bias.shape)))
        (torch.where(torch.isinf(bias), torch.zeros_like(bias), bias) * wgt).sum().backward()
        out["gab/d_spatial"] = np_(gab.spatial_pos_encoder.weight.grad[:24])
        out["gab/d_virtual"] = np_(gab.graph_token_virtual_distance.weight.grad)

        gnf = mods.GraphNodeFeature(num_heads=H, num_atoms=16, num_in_degree=512, num_out_degree=512,
                                    hidden_dim=D, n_layers=4)
        fill_params(gnf, "graph_node_feature.")
        x = torch.from_numpy(hashinit.uniform("gnf/x", (B, N, D))).requires_grad_(True)
        y = gnf(x, torch.from_numpy(deg), torch.from_numpy(deg))
        out["gnf/out"] = np_(y)
        (y * torch.from_numpy(hashinit.uniform("gnf/cot", tuple(y.shape)))).sum().backward()
        out["gnf/dx"] = np_(x.grad)
        out["gnf/d_in"] = np_(gnf.in_degree_encoder.weight.grad[:8])
        out["gnf/d_out"] = np_(gnf.out_degree_encoder.weight.grad[:8])
        out["gnf/d_tok"] = np_(gnf.graph_token.weight.grad)

        for pre_ln in (False, True):
            tag = "pre" if pre_ln else "post"
            layer = mods.GraphormerGraphEncoderLayer(embedding_dim=D, ffn_embedding_dim=Fg, num_attention_heads=H,
                                                     dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                                                     activation_fn="gelu", pre_layernorm=pre_ln)
            named = fill_params(layer, "layers.0.layers.0.")
            xin = torch.from_numpy(hashinit.uniform("gl/x", (T, B, D), 1.0)).requires_grad_(True)
            b2 = bias.detach().clone().requires_grad_(True)
            yo, _ = layer(xin, self_attn_bias=b2, self_attn_padding_mask=torch.from_numpy(kpm))
            out[f"layer_{tag}/out"] = np_(yo)
            cot = torch.from_numpy(hashinit.uniform("gl/cot", (T, B, D)))
            (yo * cot).sum().backward()
            out[f"layer_{tag}/dx"] = np_(xin.grad)
            dbias = b2.grad.clone()
            out[f"layer_{tag}/dbias"] = np_(dbias)
            grad_summary(named, out_pref := {})
            for k, v in out_pref.items():
                out[f"layer_{tag}/{k}"] = v
            if not pre_ln:
                mha = layer.self_attn
                for p in layer.parameters():
                    p.grad = None
                xq = torch.from_numpy(hashinit.uniform("mha/x", (T, B, D), 1.0)).requires_grad_(True)
                b3 = bias.detach().clone().requires_grad_(True)
                a, _ = mha(xq, xq, xq, b3, key_padding_mask=torch.from_numpy(kpm), need_weights=False)
                out["mha/out"] = np_(a)
                (a * cot).sum().backward()
                out["mha/dx"] = np_(xq.grad)
                out["mha/dbias"] = np_(b3.grad)
                out["mha/dWq"] = np_(mha.q_proj.weight.grad)
                out["mha/dbk"] = np_(mha.k_proj.bias.grad)
        np.savez_compressed(os.path.join(OUT, f"graph_modules_d{D}.npz"), **out)


def case_full_model(mods, models, collator_mod, pre, updown, loss_mod, metrics, kinds=("A", "B", "M", "C2")):
    for kind in kinds:
        if kind in ("A", "B"):
            hp, over, fname = tiny_hparams(kind), {}, f"full_tiny768_{kind}.npz"
            trees = tiny_trees(kind, hp)
        else:       # "M": tiny shapes, mixed predictions; "C2": BASELINE.json configs[1] at its true geometry
            hp, over = cases.real_hparams(kind), cases.weight_overrides(kind)
            trees = cases.real_trees(kind, hp)
            fname = {"M": "full_tiny768_M.npz", "C2": "full_c2_real.npz", "LAUNCH": "full_launch.npz"}[kind]
        items = ref_items_from_trees(trees, pre, updown)
        batch = ref_collate(items, collator_mod, 5)
        enc = build_reference_encoder(mods, hp)
        args = SimpleNamespace(max_nodes=512, share_encoder_input_output_embed=False, encoder_embed_dim=hp.dim,
                               activation_fn="gelu", num_classes=1)
        genc = models.GraphormerEncoder.__new__(models.GraphormerEncoder)
        nn.Module.__init__(genc)
        genc.graph_encoder = enc
        genc.node_encoder_stack = nn.ModuleList([enc.text_pooler, enc.text_dropout, enc.node_classifier])
        model = models.GraphormerModel.__new__(models.GraphormerModel)
        nn.Module.__init__(model)
        model.encoder = genc
        model.train()                       # dropout p = 0 everywhere; train mode like the launch
        named = fill_params(enc, overrides=over)
        # oracle shapes and reference shapes must agree name by name
        shapes = R.param_shapes(hp)
        ref_shapes = {n: tuple(p.shape) for n, p in named.items()}
        missing = set(ref_shapes) - set(shapes)
        extra = set(shapes) - set(ref_shapes)
        assert not missing, sorted(missing)[:10]
        extra = {e for e in extra if "edge_dis_encoder" not in e}
        assert not extra, sorted(extra)[:10]
        for n in ref_shapes:
            assert ref_shapes[n] == tuple(shapes[n]), (n, ref_shapes[n], shapes[n])

        crit = loss_mod.GraphPredictionNodeCrossEntropy(task=None, positive_weight=hp.pos_weight,
                                                        negative_weight=hp.neg_weight)
        sample = {"nsamples": len(trees), "net_input": {"batched_data": batch}}
        text, bn, glob = enc(batch)
        out = {"enc/text_slice": np_(text[:, :3, :64]), "enc/bn": np_(bn), "enc/global": np_(glob)}
        for p in enc.parameters():
            p.grad = None
        lossv, sample_size, log = crit(model, sample)
        lossv.backward()
        logits, glob2 = model(batch)
        out["logits"] = np_(logits)
        out["loss"] = np_(lossv.float())
        out["sample_size"] = np.asarray(sample_size)
        for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"):
            out["log/" + k] = np.asarray(int(log[k]))
        metrics.LOG.clear()
        crit.reduce_metrics([log])
        for k, v in metrics.LOG.items():
            out["metric/" + k] = np.asarray(v)
        grad_summary(named, out)
        out["n_trainable_with_grad"] = np.asarray(sum(1 for p in named.values() if p.grad is not None))
        np.savez_compressed(os.path.join(OUT, fname), **out)
        print(kind, "loss", float(lossv), "logits", logits[:2].tolist(), "F1", metrics.LOG.get("f1"),
              {k: int(log[k]) for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive")},
              "n", sample_size, flush=True)


def case_contrastive(mods, models, collator_mod, pre, updown):
    """criterions/contrastive_loss.py (the REAL criterion) — (a) on hash-generated embeddings through a stub model, all
    three weighting modes; (b) end to end: the reference encoder on 5 small trees, which gives the final graph stack
    (gradient-free under node_cross_entropy) a gradient."""
    cl = CONTRASTIVE_MOD
    out = {}
    B, D = 12, 768
    y, hard = cases.contrastive_labels(B, "A")
    emb0 = hashinit.uniform("cl/emb", (B, D), 1.0)
    emb0[:, :64] += np.asarray(y)[:, None] * 0.35            # communities are partly separable: mixed predictions
    for tag, kw in (("adaptive", dict(soft_negative_weight=0.0, multiplication_scale=20.0, adaptive_soft_negative_weight=True)),
                    ("fixed", dict(soft_negative_weight=0.25, multiplication_scale=20.0, adaptive_soft_negative_weight=False)),
                    ("strict", dict(soft_negative_weight=0.0, multiplication_scale=1.0, adaptive_soft_negative_weight=False))):
        emb = torch.from_numpy(emb0.copy()).requires_grad_(True)
        crit = cl.GraphContrastiveLoss(task=None, **kw)
        stub = lambda batched_data, e=emb: (None, e)
        sample = {"net_input": {"batched_data": {"x": torch.zeros(B, 7, 3), "y": torch.from_numpy(y), "hard_y": torch.from_numpy(hard)}}}
        lossv, n, log = crit(stub, sample)
        lossv.backward()
        out[f"{tag}/loss"] = np_(lossv)
        out[f"{tag}/sample_size"] = np.asarray(n)
        out[f"{tag}/d_emb"] = np_(emb.grad)
        for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive"):
            out[f"{tag}/{k}"] = np.asarray(int(log[k]))
        print("contrastive", tag, float(lossv), {k: int(log[k]) for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive")})
    out["emb"] = emb0
    out["y"], out["hard_y"] = y, hard
    # (b) full model
    hp = tiny_hparams("A")
    trees = cases.contrastive_trees(hp)
    for t in trees:
        t["y_mask"] = np.zeros(len(t["parent"]), dtype=bool)      # ref_items_from_trees reads it; unused by this task
    items = ref_items_from_trees(trees, pre, updown)
    batch = ref_collate(items, collator_mod, 5)
    del batch["y_mask"]
    batch["hard_y"] = torch.cat([torch.from_numpy(t["hard_y"]) for t in trees])
    enc = build_reference_encoder(mods, hp)
    genc = models.GraphormerEncoder.__new__(models.GraphormerEncoder)
    nn.Module.__init__(genc)
    genc.graph_encoder = enc
    genc.node_encoder_stack = nn.ModuleList([enc.text_pooler, enc.text_dropout, enc.node_classifier])
    model = models.GraphormerModel.__new__(models.GraphormerModel)
    nn.Module.__init__(model)
    model.encoder = genc
    model.train()
    named = fill_params(enc)
    crit = cl.GraphContrastiveLoss(task=None, soft_negative_weight=0.0, multiplication_scale=20.0, adaptive_soft_negative_weight=True)
    lossv, n, log = crit(model, {"net_input": {"batched_data": batch}})
    lossv.backward()
    _, glob = model(batch)
    out["full/loss"] = np_(lossv)
    out["full/sample_size"] = np.asarray(n)
    out["full/global"] = np_(glob)
    for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive"):
        out[f"full/{k}"] = np.asarray(int(log[k]))
    gs = {}
    grad_summary(named, gs)
    for k, v in gs.items():
        out["full/" + k] = v
    out["full/n_trainable_with_grad"] = np.asarray(sum(1 for p in named.values() if p.grad is not None))
    print("contrastive full", float(lossv), int(out["full/n_trainable_with_grad"]),
          "final stack |g|", float(gs["gnorm/layers.2.layers.0.fc1.weight"]))
    np.savez_compressed(os.path.join(OUT, "contrastive.npz"), **out)


def case_state_dict_keys(mods, models):
    """Names and shapes of every state-dict entry of the reference model at the SHIPPED launch
    (mDT/experiments/hateful_discussions/sample_run.sh:3 = `run_train.sh 8 4 5 2 2 0`, flags run_train.sh:28-65) built as
    NodePredictionTask.build_model builds it (tasks/node_prediction.py:34-55) — the checkpoint-compatibility contract.
    Inner names of the installed transformers 5.x are mapped to the 4.x names real checkpoints of the reference carry."""
    import json
    hp = R.hparams(dim=768, enc_heads=12, graph_heads=12, enc_ffn=3072, graph_ffn=768, text_layers=12, vit_layers=12,
                   num_fusion_layers=8, num_fusion_stack=2, num_graph_stack=2, num_bottleneck=4)
    mods.MultiGraphormerGraphEncoder.build_vit_bert_encoders = make_builder(hp)
    args = SimpleNamespace(
        max_nodes=10000, num_atoms=512 * 9, num_in_degree=512, num_out_degree=512, num_edges=512 * 3, num_spatial=512,
        num_edge_dis=128, edge_type="multi_hop", multi_hop_max_dist=5, num_bottleneck_tokens=4, num_fusion_layers=8,
        num_fusion_stack=2, num_graph_stack=2, encoder_layers=4, encoder_embed_dim=768, encoder_ffn_embed_dim=768,
        encoder_attention_heads=12, dropout=0.4, attention_dropout=0.3, act_dropout=0.3, encoder_normalize_before=True,
        pre_layernorm=False, apply_graphormer_init=False, activation_fn="gelu", freeze_initial_encoders=True,
        share_encoder_input_output_embed=False, num_classes=1, remove_head=False)
    enc = models.GraphormerEncoder(args)
    model = models.GraphormerModel(args, enc)
    model.node_encoder_stack = nn.ModuleList([enc.graph_encoder.text_pooler, enc.graph_encoder.text_dropout, nn.Linear(768, 2)])
    GE = "encoder.graph_encoder."

    def canon(k):
        return GE + canonical(k[len(GE):]) if k.startswith(GE) else canonical(k)

    keys = {}
    for k, v in model.state_dict().items():
        keys[canon(k)] = list(v.shape)
    trainable = sorted(canon(k) for k, p in model.named_parameters() if p.requires_grad)
    n_params = sum(p.numel() for p in model.parameters())
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    with open(os.path.join(OUT, "state_dict_keys_launch.json"), "w") as f:
        json.dump(dict(launch="sample_run.sh:3 (8 4 5 2 2 0), --freeze_initial_encoders", n_keys=len(keys), n_params=n_params,
                       n_trainable_params=n_train, keys=keys, trainable=trainable), f, indent=0, sort_keys=True)
    print("state dict keys", len(keys), "params", n_params, "trainable", n_train)


def case_discussions():
    """Dataset-builder semantics of mDT/experiments/hateful_discussions/datasets/hateful_discussions.py on a synthetic
    JSON-lines sample (tests/golden/discussions/sample.jsonl): the module needs torch_geometric, so the plain-Python
    pieces are compiled from its text — the methods get_relative_depth / spread_downwards / collapse_tree and, out of
    ``process``, the two URL regexes with the nested clean_urls / extract_text — and run exactly as ``process`` runs them
    (:103-107).  Stored: node order, text handed to the tokenizer, labels and the (up, down) matrix of every discussion."""
    import json
    path = os.path.join(REF, "mDT/experiments/hateful_discussions/datasets/hateful_discussions.py")
    tree = ast.parse(open(path).read())
    methods, nested = [], []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("get_relative_depth", "spread_downwards", "collapse_tree"):
            methods.append(node)
        if isinstance(node, ast.FunctionDef) and node.name == "process":
            for sub in node.body:
                if isinstance(sub, ast.Assign) and getattr(sub.targets[0], "id", "") in ("markdown_regex", "all_url_regex"):
                    nested.append(sub)
                if isinstance(sub, ast.FunctionDef) and sub.name in ("clean_urls", "extract_text"):
                    nested.append(sub)
    ns = {"copy": __import__("copy"), "re": __import__("re"), "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=methods + nested, type_ignores=[]), path, "exec"), ns)
    H = type("Holder", (), {k: ns[k] for k in ("get_relative_depth", "spread_downwards", "collapse_tree")})()
    src = os.path.join(OUT, "discussions", "sample.jsonl")
    out = []
    for line in open(src):
        raw = json.loads(line)
        H.get_relative_depth(raw)
        H.spread_downwards(raw)
        data = {}
        H.collapse_tree(raw, data, [])
        order = list(data.keys())
        out.append(dict(
            order=order,
            texts=[ns["extract_text"](data[k]) for k in order],
            labels=[data[k][3] for k in order],
            images=[data[k][1] for k in order],
            updown=[[data[a][2][b] for b in order] for a in order]))
    with open(os.path.join(OUT, "discussions", "expected.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("discussions", [len(o["order"]) for o in out])


def case_pixel_values():
    """The reference's image path (experiments/hateful_discussions/datasets/hateful_discussions.py:47-49,168-184):
    ``extractor(images, return_tensors="pt")["pixel_values"]`` with the ViT image processor's defaults (the checkpoint name it
    passes, google/vit-base-patch16-224, only supplies those defaults: 224 x 224 bilinear, 1 / 255, mean = std = 0.5) on
    ``Image.open(..).convert(mode="RGB")`` inputs.  Stored (tests/golden/discussions/pixel_values.npz): the resized bytes of
    every image and the 3 x 256 table of values the rescale / normalise arithmetic can produce; the generator asserts that
    pixel_values == table[resized bytes] for every image, so the two together ARE the processor's output."""
    from PIL import Image
    from transformers import ViTImageProcessor
    extractor = ViTImageProcessor()
    from . import cases
    imgs = cases.pixel_value_inputs(os.path.join(OUT, "discussions"))
    pv = extractor([Image.fromarray(a) for a in imgs], return_tensors="pt")["pixel_values"].numpy()
    ramp = np.arange(224 * 224, dtype=np.int64).reshape(224, 224, 1) % 256
    ramp = np.repeat(ramp, 3, axis=2).astype(np.uint8)
    lut_img = extractor([Image.fromarray(ramp)], return_tensors="pt", do_resize=False)["pixel_values"].numpy()[0]      # [3, 224, 224]
    lut = np.stack([lut_img[c].reshape(-1)[:256] for c in range(3)])
    resized = np.stack([np.asarray(Image.fromarray(a).resize((224, 224), resample=Image.BILINEAR)) for a in imgs])
    for i in range(len(imgs)):
        for c in range(3):
            assert np.array_equal(pv[i, c], lut[c][resized[i, :, :, c]]), (i, c)
    np.savez_compressed(os.path.join(OUT, "discussions", "pixel_values.npz"), resized=resized, lut=lut.astype(np.float32),
                        sizes=np.asarray([a.shape[:2] for a in imgs], dtype=np.int64), checksum=np.asarray([float(pv.astype(np.float64).sum())]))
    print("pixel_values", pv.shape, float(pv.min()), float(pv.max()))


def case_fusion_layer(mods):
    from transformers import BertConfig, ViTConfig
    from transformers.models.bert.modeling_bert import BertLayer
    from transformers.models.vit.modeling_vit import ViTLayer
    D, H, Fe, nb, L, P = 768, 12, 128, 4, 10, 5
    bc = BertConfig(hidden_size=D, num_attention_heads=H, intermediate_size=Fe, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    bc._attn_implementation = "eager"
    vc = ViTConfig(hidden_size=D, num_attention_heads=H, intermediate_size=Fe, hidden_dropout_prob=0.0,
                   attention_probs_dropout_prob=0.0)
    vc._attn_implementation = "eager"
    fl = mods.multi_graphormer_fusion_layer.GraphFusionLayer(
        BertLayer4x(BertLayer(bc)), ViTLayer4x(ViTLayer(vc)), nb, use_projection=True)
    named = fill_params(fl, "fusion_layers.0.fusion_layers.0.")
    M = 5
    img = np.array([False, True, False, True, True])
    am = np.ones((M, nb + L), dtype=np.float32)
    am[0, nb + 6:] = 0
    am[3, nb + 2:] = 0
    ext = ((1.0 - torch.from_numpy(am))[:, None, None, :].to(torch.half)) * torch.finfo(torch.half).min
    out = dict(image_index=img, attention_mask=am)
    for with_img in (True, False):
        tag = "img" if with_img else "noimg"
        for p in fl.parameters():
            p.grad = None
        text = torch.from_numpy(hashinit.uniform("fl/text", (M, L, D), 1.0)).requires_grad_(True)
        vit = torch.from_numpy(hashinit.uniform("fl/vit", (int(img.sum()), P, D), 1.0)).requires_grad_(True)
        bn = torch.from_numpy(hashinit.uniform("fl/bn", (M, nb, D), 1.0)).requires_grad_(True)
        t, v, b = fl(text, vit if with_img else None, bn, ext, torch.from_numpy(img))
        out[f"{tag}/text"] = np_(t)
        out[f"{tag}/bn"] = np_(b)
        lossv = (t * torch.from_numpy(hashinit.uniform("fl/ct", tuple(t.shape)))).sum() + \
                (b * torch.from_numpy(hashinit.uniform("fl/cb", tuple(b.shape)))).sum()
        if with_img:
            out[f"{tag}/vit"] = np_(v)
            lossv = lossv + (v * torch.from_numpy(hashinit.uniform("fl/cv", tuple(v.shape)))).sum()
        lossv.backward()
        out[f"{tag}/dtext"] = np_(text.grad)
        out[f"{tag}/dbn"] = np_(bn.grad)
        if with_img:
            out[f"{tag}/dvit"] = np_(vit.grad)
        gs = {}
        grad_summary(named, gs)
        for k, val in gs.items():
            out[f"{tag}/{k}"] = val
    np.savez_compressed(os.path.join(OUT, "fusion_layer.npz"), **out)


def main():
    only = sys.argv[1:]          # e.g. `python -m oracle.gen_golden full:M,C2` regenerates just those cases
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    metrics = install_fairseq_standins()
    mods, models, collator_mod, pre, loss_mod = mount_reference()
    updown = load_updown_functions()
    if only and only[0] == "discussions":
        case_discussions()
        case_pixel_values()
        return
    if only and only[0] == "keys":
        case_state_dict_keys(mods, models)
        return
    if only and only[0] == "contrastive":
        case_contrastive(mods, models, collator_mod, pre, updown)
        return
    if only and only[0].startswith("full:"):
        case_full_model(mods, models, collator_mod, pre, updown, loss_mod, metrics, kinds=tuple(only[0][5:].split(",")))
        return
    case_structure(pre, collator_mod, updown)
    print("structure done")
    case_graph_modules(mods)
    print("graph modules done")
    case_fusion_layer(mods)
    print("fusion layer done")
    case_full_model(mods, models, collator_mod, pre, updown, loss_mod, metrics)
    print("full model done")
    case_contrastive(mods, models, collator_mod, pre, updown)
    print("contrastive done")
    case_state_dict_keys(mods, models)
    case_discussions()
    case_pixel_values()


if __name__ == "__main__":
    main()
