"""Shared building blocks of the drop-in modules: fused Q/K/V parameter storage with
reference-compatible state-dict names, HF-style encoder layers, parameter containers.

The kernels want one [3D, D] projection; checkpoints of the reference hold three D x D
matrices (``q_proj/k_proj/v_proj`` in modules/multihead_attention.py:54-62,
``attention.self.{query,key,value}`` in HF BertLayer, ``attention.attention.{...}`` in HF
ViTLayer, transformers 4.x).  ``QKVFusedMixin`` keeps the fused tensor as the live
``nn.Parameter`` and splits / merges on ``state_dict`` / ``load_state_dict``.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..engine import BlockParams


class QKVFusedMixin:
    """Mixin for an ``nn.Module`` that owns ``qkv_weight [3D, D]`` and ``qkv_bias [3D]``.
    ``_qkv_names`` are the three sub-prefixes (relative to this module) used in checkpoints."""
    _qkv_names = ("q_proj", "k_proj", "v_proj")

    def _init_qkv(self, dim: int, bias: bool = True):
        self.qkv_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.qkv_bias = nn.Parameter(torch.zeros(3 * dim)) if bias else None
        self._qkv_dim = dim

    def _qkv_view(self, i: int):
        d = self._qkv_dim
        return SimpleNamespace(weight=self.qkv_weight[i * d:(i + 1) * d],
                               bias=None if self.qkv_bias is None else self.qkv_bias[i * d:(i + 1) * d])

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        d = self._qkv_dim
        w = destination.pop(prefix + "qkv_weight")
        b = destination.pop(prefix + "qkv_bias", None)
        for i, n in enumerate(self._qkv_names):
            destination[f"{prefix}{n}.weight"] = w[i * d:(i + 1) * d]
            if b is not None:
                destination[f"{prefix}{n}.bias"] = b[i * d:(i + 1) * d]

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        names = [f"{prefix}{n}.weight" for n in self._qkv_names]
        if all(k in state_dict for k in names):
            state_dict[prefix + "qkv_weight"] = torch.cat([state_dict.pop(k) for k in names], dim=0)
            bnames = [f"{prefix}{n}.bias" for n in self._qkv_names]
            if all(k in state_dict for k in bnames):
                state_dict[prefix + "qkv_bias"] = torch.cat([state_dict.pop(k) for k in bnames], dim=0)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def _hf_init_linear(lin: nn.Linear, std: float = 0.02):
    nn.init.normal_(lin.weight, 0.0, std)
    if lin.bias is not None:
        nn.init.zeros_(lin.bias)


class _SelfAttentionParams(QKVFusedMixin, nn.Module):
    def __init__(self, dim, names):
        super().__init__()
        self._qkv_names = names
        self._init_qkv(dim)
        nn.init.normal_(self.qkv_weight, 0.0, 0.02)


class _DenseLN(nn.Module):
    def __init__(self, d_in, d_out, eps, with_ln=True):
        super().__init__()
        self.dense = nn.Linear(d_in, d_out)
        _hf_init_linear(self.dense)
        if with_ln:
            self.LayerNorm = nn.LayerNorm(d_out, eps=eps)


class _Dense(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.dense = nn.Linear(d_in, d_out)
        _hf_init_linear(self.dense)


class _BertAttention(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.self = _SelfAttentionParams(dim, ("query", "key", "value"))
        self.output = _DenseLN(dim, dim, eps)


class BertLayer(nn.Module):
    """Parameter layout of HF ``BertLayer`` (transformers 4.x names); post-LN block, eps 1e-12."""
    pre_ln = False

    def __init__(self, dim=768, heads=12, intermediate=3072, eps=1e-12):
        super().__init__()
        self.dim, self.heads, self.eps = dim, heads, eps
        self.hidden_dropout_p, self.attention_dropout_p = 0.0, 0.0     # hidden_dropout_prob / attention_probs_dropout_prob
        self.attention = _BertAttention(dim, eps)
        self.intermediate = _Dense(dim, intermediate)
        self.output = _DenseLN(intermediate, dim, eps)

    def drop_kwargs(self):
        t = self.training
        return dict(p_hidden=self.hidden_dropout_p if t else 0.0, p_attn=self.attention_dropout_p if t else 0.0)

    def block_params(self) -> BlockParams:
        a, o = self.attention, self.output
        return BlockParams(a.self.qkv_weight, a.self.qkv_bias, a.output.dense.weight, a.output.dense.bias,
                           a.output.LayerNorm.weight, a.output.LayerNorm.bias, self.intermediate.dense.weight,
                           self.intermediate.dense.bias, o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)


class _ViTAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.attention = _SelfAttentionParams(dim, ("query", "key", "value"))
        self.output = _DenseLN(dim, dim, 0.0, with_ln=False)


class ViTLayer(nn.Module):
    """Parameter layout of HF ``ViTLayer`` (transformers 4.x names); pre-LN block, eps 1e-12."""
    pre_ln = True

    def __init__(self, dim=768, heads=12, intermediate=3072, eps=1e-12):
        super().__init__()
        self.dim, self.heads, self.eps = dim, heads, eps
        self.hidden_dropout_p, self.attention_dropout_p = 0.0, 0.0
        self.attention = _ViTAttention(dim)
        self.intermediate = _Dense(dim, intermediate)
        self.output = _Dense(intermediate, dim)
        self.layernorm_before = nn.LayerNorm(dim, eps=eps)
        self.layernorm_after = nn.LayerNorm(dim, eps=eps)

    def drop_kwargs(self):
        t = self.training
        return dict(p_hidden=self.hidden_dropout_p if t else 0.0, p_attn=self.attention_dropout_p if t else 0.0)

    def block_params(self) -> BlockParams:
        a = self.attention
        return BlockParams(a.attention.qkv_weight, a.attention.qkv_bias, a.output.dense.weight, a.output.dense.bias,
                           self.layernorm_before.weight, self.layernorm_before.bias, self.intermediate.dense.weight,
                           self.intermediate.dense.bias, self.output.dense.weight, self.output.dense.bias,
                           self.layernorm_after.weight, self.layernorm_after.bias)


class _Pooler(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)
        _hf_init_linear(self.dense)


class _LayerList(nn.Module):
    """``encoder.layer.{i}`` container."""
    def __init__(self, layers):
        super().__init__()
        self.layer = nn.ModuleList(layers)


class _BertEmbeddings(nn.Module):
    def __init__(self, vocab, max_pos, type_vocab, dim, eps):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, dim)
        self.position_embeddings = nn.Embedding(max_pos, dim)
        self.token_type_embeddings = nn.Embedding(type_vocab, dim)
        self.LayerNorm = nn.LayerNorm(dim, eps=eps)
        for e in (self.word_embeddings, self.position_embeddings, self.token_type_embeddings):
            nn.init.normal_(e.weight, 0.0, 0.02)


class BertModel(nn.Module):
    """``text_model``: embeddings + the first layers of BERT + pooler (parameters only — the
    compute is scheduled by MultiGraphormerGraphEncoder)."""
    def __init__(self, dim=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512, type_vocab=2, eps=1e-12):
        super().__init__()
        self.dim, self.heads, self.eps = dim, heads, eps
        self.embeddings = _BertEmbeddings(vocab, max_pos, type_vocab, dim, eps)
        self.encoder = _LayerList([BertLayer(dim, heads, intermediate, eps) for _ in range(layers)])
        self.pooler = _Pooler(dim)


class _PatchEmbeddings(nn.Module):
    def __init__(self, dim, patch, channels=3):
        super().__init__()
        self.projection = nn.Conv2d(channels, dim, kernel_size=patch, stride=patch)
        nn.init.trunc_normal_(self.projection.weight, std=0.02)
        nn.init.zeros_(self.projection.bias)


class _ViTEmbeddings(nn.Module):
    def __init__(self, dim, image_size, patch):
        super().__init__()
        n = (image_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.empty(1, 1, dim))
        self.position_embeddings = nn.Parameter(torch.empty(1, n + 1, dim))
        self.patch_embeddings = _PatchEmbeddings(dim, patch)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.position_embeddings, std=0.02)


class ViTModel(nn.Module):
    """``vit_model``: patch embeddings + first layers + the final LayerNorm (applied mid-network
    by the reference, quirk 5) + pooler."""
    def __init__(self, dim=768, layers=12, heads=12, intermediate=3072, image_size=224, patch=16, eps=1e-12):
        super().__init__()
        self.dim, self.heads, self.eps, self.image_size, self.patch = dim, heads, eps, image_size, patch
        self.embeddings = _ViTEmbeddings(dim, image_size, patch)
        self.encoder = _LayerList([ViTLayer(dim, heads, intermediate, eps) for _ in range(layers)])
        self.layernorm = nn.LayerNorm(dim, eps=eps)
        self.pooler = _Pooler(dim)


def xavier_uniform_(t: torch.Tensor, gain: float = 1.0):
    fan_out, fan_in = t.shape[0], t.shape[1]
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        t.uniform_(-a, a)
