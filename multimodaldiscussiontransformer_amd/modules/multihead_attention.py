"""Drop-in for mDT/src/modules/multihead_attention.py (``MultiheadAttention``).

Same constructor, attributes, state-dict keys (``q_proj / k_proj / v_proj / out_proj``) and
``forward`` signature; the body is one fused QKV GEMM, the fused bias-softmax-attention
kernel and the output GEMM behind the C ABI.  fairseq's time-major ``[T, B, C]`` layout is
consumed in place through the kernel's (seq_stride, pos_stride) addressing.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import engine as E
from ._fused import QKVFusedMixin, xavier_uniform_


class MultiheadAttention(QKVFusedMixin, nn.Module):
    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, self_attention=False,
                 q_noise=0.0, qn_block_size=8):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout_p = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        assert self.self_attention, "Only support self attention"
        assert self.qkv_same_dim, "Self-attention requires query, key and value to be of the same size"
        if q_noise > 0:
            raise NotImplementedError("quant_noise is not part of the HIP path (the reference runs with q_noise=0)")
        self._init_qkv(embed_dim, bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.reset_parameters()
        self.onnx_trace = False

    # reference attribute surface -----------------------------------------------------
    @property
    def q_proj(self):
        return self._qkv_view(0)

    @property
    def k_proj(self):
        return self._qkv_view(1)

    @property
    def v_proj(self):
        return self._qkv_view(2)

    def prepare_for_onnx_export_(self):
        raise NotImplementedError

    def reset_parameters(self):
        """modules/multihead_attention.py:75-89: xavier_uniform with gain 1/sqrt(2) on q, k, v;
        default nn.Linear bias init; xavier_uniform out_proj with zero bias."""
        d = self.embed_dim
        for i in range(3):
            xavier_uniform_(self.qkv_weight.data[i * d:(i + 1) * d], gain=1 / math.sqrt(2))
        if self.qkv_bias is not None:
            bound = 1 / math.sqrt(d)
            nn.init.uniform_(self.qkv_bias, -bound, bound)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    # tape-level ------------------------------------------------------------------------
    def _fwd(self, tape, x: E.Var, spec: E.AttnSpec, stash=None) -> E.Var:
        p = self.dropout_p if self.training else 0.0
        return E.attention_layer(tape, x, self.qkv_weight, self.qkv_bias, self.out_proj.weight, self.out_proj.bias, spec,
                                 p_attn=p, stash=stash)

    # public (reference signature) ------------------------------------------------------
    def forward(self, query, key: Optional[Tensor], value: Optional[Tensor], attn_bias: Optional[Tensor],
                key_padding_mask: Optional[Tensor] = None, need_weights: bool = True, attn_mask: Optional[Tensor] = None,
                before_softmax: bool = False, need_head_weights: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
        """Input shape: Time x Batch x Channel.  ``key`` / ``value`` are ignored exactly as in the
        reference (:134-136 project ``query`` three times)."""
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim, f"query dim {embed_dim} != {self.embed_dim}"
        if need_head_weights:
            need_weights = True                      # :91-102
        if key_padding_mask is not None and key_padding_mask.dim() == 0:
            key_padding_mask = None
        if key_padding_mask is not None:
            assert key_padding_mask.size(0) == bsz and key_padding_mask.size(1) == tgt_len
        x2 = query.contiguous().view(tgt_len * bsz, embed_dim)
        kpad = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
        if attn_mask is not None:
            # an additive [tgt, src] mask shared by every sequence and head (:176-178): folded into the dense bias
            am = attn_mask.float().reshape(1, 1, tgt_len, tgt_len)
            attn_bias = am.expand(bsz, self.num_heads, tgt_len, tgt_len) if attn_bias is None else \
                attn_bias.reshape(bsz, self.num_heads, tgt_len, tgt_len).float() + am
        inputs = [x2]
        if attn_bias is not None:
            inputs.append(attn_bias.reshape(bsz, self.num_heads, tgt_len, tgt_len).float().contiguous())
        from .. import ops
        if before_softmax:
            # (:189-190) the raw scores and v, no attention output: q k^T * scaling + bias, masked keys at -inf,
            # [bsz * heads, tgt, src] and [bsz * heads, src, head_dim].  Detached (nothing in mDT asks for them).
            with torch.no_grad():
                qkv = ops.gemm(x2, self.qkv_weight.data, bias=None if self.qkv_bias is None else self.qkv_bias.data)
                sc = ops.attention_head_weights(qkv, None, bsz, tgt_len, self.num_heads, raw_scores=True, seq_stride=1, pos_stride=bsz,
                                                scale=self.scaling, dense_bias=inputs[1] if len(inputs) > 1 else None, key_pad=kpad)
                v = qkv[:, 2 * embed_dim:].reshape(tgt_len, bsz * self.num_heads, self.head_dim).transpose(0, 1)
            return sc.view(bsz * self.num_heads, tgt_len, tgt_len).to(query.dtype), v

        stash = {} if need_weights else None

        def run(tape, xv, bv=None):
            spec = E.AttnSpec(nseq=bsz, S=tgt_len, H=self.num_heads, seq_stride=1, pos_stride=bsz, scale=self.scaling,
                              dense_bias=None if bv is None else bv.data, dense_bias_var=bv, key_pad=kpad)
            return (self._fwd(tape, xv, spec, stash=stash),)

        params = [p for p in self.parameters()]
        (out,) = E.run_tape(run, inputs, params)
        weights = None
        if need_weights:
            # softmax probabilities BEFORE dropout (:205-214): head-averaged [bsz, tgt_len, src_len] (the reference's default) or,
            # with need_head_weights, per head [heads, bsz, tgt_len, src_len]; recomputed from q, k and the forward's
            # log-sum-exp — the fused kernel never stores them.  Returned detached: nothing in mDT differentiates through them.
            kw = {k: v for k, v in stash["kw"].items() if k not in ("seq_offsets", "q_limit", "bins")}
            if need_head_weights:
                weights = ops.attention_head_weights(stash["qkv"], stash["lse"], bsz, tgt_len, self.num_heads, **kw).transpose(0, 1)
            else:
                weights = ops.attention_mean_probs(stash["qkv"], stash["lse"], bsz, tgt_len, self.num_heads, **kw)
        return out.view(tgt_len, bsz, embed_dim), weights

    def apply_sparse_mask(self, attn_weights, tgt_len: int, src_len: int, bsz: int):
        return attn_weights

    def upgrade_state_dict_named(self, state_dict, name):
        """Legacy checkpoints carry ``in_proj_weight`` / ``in_proj_bias`` (q;k;v stacked):
        rename them to the three projections (modules/multihead_attention.py:219-248)."""
        prefix = name + "." if name != "" else ""
        for k in list(state_dict.keys()):
            if k.endswith(prefix + "in_proj_weight"):
                w = state_dict.pop(k)
                d = w.shape[0] // 3
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    state_dict[f"{prefix}{n}.weight"] = w[i * d:(i + 1) * d]
                kb = prefix + "in_proj_bias"
                if kb in state_dict:
                    b = state_dict.pop(kb)
                    for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                        state_dict[f"{prefix}{n}.bias"] = b[i * d:(i + 1) * d]
