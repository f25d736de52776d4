"""Helpers shared by the model-level tests: build the product model from oracle hyper-parameters
and fill it with the hash-generated parity weights through its (reference-named) state dict."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import hashinit


def model_args(hp, **over):
    a = SimpleNamespace(
        num_atoms=hp.num_atoms, num_in_degree=hp.num_in_degree, num_out_degree=hp.num_out_degree, num_edges=hp.num_edges,
        num_spatial=hp.num_spatial, num_edge_dis=128, edge_type="", multi_hop_max_dist=5,
        num_bottleneck_tokens=hp.num_bottleneck, num_fusion_layers=hp.num_fusion_layers,
        num_fusion_stack=hp.num_fusion_stack, num_graph_stack=hp.num_graph_stack, encoder_layers=4,
        encoder_embed_dim=hp.dim, encoder_ffn_embed_dim=hp.graph_ffn, encoder_attention_heads=hp.graph_heads,
        dropout=0.0, attention_dropout=0.0, act_dropout=0.0, encoder_normalize_before=hp.encoder_normalize_before,
        pre_layernorm=hp.pre_layernorm, apply_graphormer_init=False, activation_fn="gelu",
        freeze_initial_encoders=bool(getattr(hp, "freeze_initial_encoders", False)), share_encoder_input_output_embed=False, max_nodes=512, num_classes=1,
        bert_config=dict(dim=hp.dim, layers=hp.text_layers, heads=hp.enc_heads, intermediate=hp.enc_ffn,
                         vocab=hp.vocab_size, max_pos=hp.max_pos, type_vocab=hp.type_vocab),
        vit_config=dict(dim=hp.dim, layers=hp.vit_layers, heads=hp.enc_heads, intermediate=hp.enc_ffn,
                        image_size=hp.image_size, patch=hp.patch),
    )
    for k, v in over.items():
        setattr(a, k, v)
    return a


ENC = "encoder.graph_encoder."


def canonical_to_oracle(name: str):
    """state-dict key of GraphormerModel → oracle / reference canonical name (or None for keys
    outside ``encoder.graph_encoder.`` and for alias entries)."""
    if not name.startswith(ENC):
        return None
    n = name[len(ENC):]
    if n.startswith("text_pooler.") or n.startswith("vit_pooler."):
        return None                      # aliases of text_model.pooler / vit_model.pooler
    return n


def fill_hash_weights(model, dtype=torch.float32, overrides=None):
    """Load hash-generated parity weights through ``load_state_dict`` (exercises the q/k/v
    split-merge and the alias keys).  ``overrides``: {canonical name: array} (oracle/cases.py weight_overrides)."""
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        n = k[len(ENC):] if k.startswith(ENC) else k
        if k.startswith("encoder.node_encoder_stack.0."):
            n = "text_model.pooler." + k.split("node_encoder_stack.0.")[1]
        elif k.startswith("encoder.node_encoder_stack.2."):
            n = "node_classifier." + k.split("node_encoder_stack.2.")[1]
        elif n.startswith("text_pooler."):
            n = "text_model.pooler." + n[len("text_pooler."):]
        elif n.startswith("vit_pooler."):
            n = "vit_model.pooler." + n[len("vit_pooler."):]
        if overrides and n in overrides:
            new[k] = torch.from_numpy(np.asarray(overrides[n]).reshape(tuple(v.shape)).copy()).to(v.dtype)
        else:
            new[k] = torch.from_numpy(hashinit.param(n, tuple(v.shape))).to(v.dtype)
    model.load_state_dict(new)
    return model


def named_canonical_params(model):
    out = {}
    for k, p in model.named_parameters():
        n = canonical_to_oracle(k)
        if n is not None:
            out[n] = p
    return out


def split_qkv_grad(name: str, grads: dict):
    """oracle parameter name → gradient tensor from a dict keyed by product parameter names
    (fused qkv tensors are sliced)."""
    for i, part in enumerate(("query", "key", "value")):
        for pat in (f".attention.self.{part}.", f".attention.attention.{part}."):
            if pat in name:
                base = name.replace(pat, pat.replace(f".{part}.", ".qkv_"))
                base = base.replace("qkv_weight", "qkv_weight").replace("qkv_bias", "qkv_bias")
                g = grads[base]
                if g is None:
                    return None
                d = g.shape[0] // 3
                return g[i * d:(i + 1) * d]
    for i, part in enumerate(("q_proj", "k_proj", "v_proj")):
        pat = f".self_attn.{part}."
        if pat in name:
            g = grads[name.replace(pat, ".self_attn.qkv_")]
            if g is None:
                return None
            d = g.shape[0] // 3
            return g[i * d:(i + 1) * d]
    return grads.get(name)
