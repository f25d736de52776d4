"""Drop-in for mDT/src/modules/graphormer_graph_encoder_layer.py
(``GraphormerGraphEncoderLayer``, ``GraphEncoderStack``): one Graphormer block over the
graph tokens of each discussion tree, post-LN by default (pre-LN with ``pre_layernorm``),
LayerNorm eps 1e-5, erf-GELU computed in fp32 — executed as one tape op
(``engine.transformer_block``).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import engine as E
from .multihead_attention import MultiheadAttention


class GraphormerGraphEncoderLayer(nn.Module):
    def __init__(self, embedding_dim: int = 768, ffn_embedding_dim: int = 3072, num_attention_heads: int = 8,
                 dropout: float = 0.1, attention_dropout: float = 0.1, activation_dropout: float = 0.1,
                 activation_fn: str = "relu", export: bool = False, q_noise: float = 0.0, qn_block_size: int = 8,
                 init_fn: Callable = None, pre_layernorm: bool = False) -> None:
        super().__init__()
        if init_fn is not None:
            init_fn()
        if activation_fn != "gelu":
            raise NotImplementedError("the HIP path implements the gelu activation the reference launches with")
        self.embedding_dim = embedding_dim
        self.num_attention_heads = num_attention_heads
        self.attention_dropout = attention_dropout
        self.q_noise = q_noise
        self.qn_block_size = qn_block_size
        self.pre_layernorm = pre_layernorm
        self.dropout_p = dropout
        self.activation_dropout_p = activation_dropout
        self.self_attn = self.build_self_attention(embedding_dim, num_attention_heads, dropout=attention_dropout,
                                                   self_attention=True, q_noise=q_noise, qn_block_size=qn_block_size)
        self.self_attn_layer_norm = nn.LayerNorm(embedding_dim, eps=1e-5)
        self.fc1 = self.build_fc1(embedding_dim, ffn_embedding_dim, q_noise, qn_block_size)
        self.fc2 = self.build_fc2(ffn_embedding_dim, embedding_dim, q_noise, qn_block_size)
        self.final_layer_norm = nn.LayerNorm(embedding_dim, eps=1e-5)

    def build_fc1(self, input_dim, output_dim, q_noise, qn_block_size):
        return nn.Linear(input_dim, output_dim)

    def build_fc2(self, input_dim, output_dim, q_noise, qn_block_size):
        return nn.Linear(input_dim, output_dim)

    def build_self_attention(self, embed_dim, num_attention_heads, dropout, self_attention, q_noise, qn_block_size):
        return MultiheadAttention(embed_dim, num_attention_heads, dropout=dropout, self_attention=True, q_noise=q_noise,
                                  qn_block_size=qn_block_size)

    def block_params(self) -> E.BlockParams:
        a = self.self_attn
        return E.BlockParams(a.qkv_weight, a.qkv_bias, a.out_proj.weight, a.out_proj.bias,
                             self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias, self.fc1.weight,
                             self.fc1.bias, self.fc2.weight, self.fc2.bias, self.final_layer_norm.weight,
                             self.final_layer_norm.bias)

    def _fwd(self, tape, x: E.Var, spec: E.AttnSpec) -> E.Var:
        t = self.training
        return E.transformer_block(tape, x, self.block_params(), spec, pre_ln=self.pre_layernorm, eps=1e-5,
                                   p_hidden=self.dropout_p if t else 0.0, p_attn=self.attention_dropout if t else 0.0,
                                   p_act=self.activation_dropout_p if t else 0.0)

    def forward(self, x: torch.Tensor, self_attn_bias: Optional[torch.Tensor] = None,
                self_attn_mask: Optional[torch.Tensor] = None, self_attn_padding_mask: Optional[torch.Tensor] = None):
        """x: T x B x C (time-major, as in the reference); returns (x, None)."""
        if self_attn_mask is not None:
            raise NotImplementedError("attn_mask is never used by mDT and is unsupported")
        T, B, C = x.shape
        H = self.num_attention_heads
        kpad = None if self_attn_padding_mask is None else self_attn_padding_mask.to(torch.uint8).contiguous()
        inputs = [x.contiguous().view(T * B, C)]
        if self_attn_bias is not None:
            inputs.append(self_attn_bias.reshape(B, H, T, T).float().contiguous())

        def run(tape, xv, bv=None):
            spec = E.AttnSpec(nseq=B, S=T, H=H, seq_stride=1, pos_stride=B, scale=self.self_attn.scaling,
                              dense_bias=None if bv is None else bv.data, dense_bias_var=bv, key_pad=kpad)
            return (self._fwd(tape, xv, spec),)

        (out,) = E.run_tape(run, inputs, list(self.parameters()))
        return out.view(T, B, C), None


class GraphEncoderStack(nn.Module):
    """``num_layers`` Graphormer layers applied in sequence (graphormer_graph_encoder_layer.py:145-195)."""

    def __init__(self, num_layers, embedding_dim: int = 768, ffn_embedding_dim: int = 3072, num_attention_heads: int = 8,
                 dropout: float = 0.1, attention_dropout: float = 0.1, activation_dropout: float = 0.1,
                 activation_fn: str = "relu", export: bool = False, q_noise: float = 0.0, qn_block_size: int = 8,
                 init_fn: Callable = None, pre_layernorm: bool = False):
        super().__init__()
        self.layers = nn.ModuleList([
            GraphormerGraphEncoderLayer(embedding_dim, ffn_embedding_dim, num_attention_heads, dropout, attention_dropout,
                                        activation_dropout, activation_fn, export, q_noise, qn_block_size, init_fn,
                                        pre_layernorm)
            for _ in range(num_layers)])

    def _fwd(self, tape, x: E.Var, spec: E.AttnSpec) -> E.Var:
        for layer in self.layers:
            x = layer._fwd(tape, x, spec)
        return x

    def forward(self, x: torch.Tensor, self_attn_bias: Optional[torch.Tensor] = None,
                self_attn_mask: Optional[torch.Tensor] = None, self_attn_padding_mask: Optional[torch.Tensor] = None):
        attn = None
        for layer in self.layers:
            x, attn = layer(x, self_attn_bias, self_attn_mask, self_attn_padding_mask)
        return x, attn
