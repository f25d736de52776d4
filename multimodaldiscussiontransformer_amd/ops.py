"""Tensor-level wrappers over the C ABI (one per entry point of include/mdt_hip.h).

These allocate outputs with torch, pass raw pointers + the current HIP stream, and never
touch tensor contents on the host.  Autograd lives one level up (autograd.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from ._lib import (EPI_ACCUM, EPI_ASUM, EPI_ATOMIC, EPI_AUX_GRAD, EPI_BIAS, EPI_COLSUM, EPI_DGELU, EPI_DROPOUT, EPI_GELU, EPI_MULAUX, EPI_RESIDUAL, check, dt, lib,
                   ptr, stream)
from ._lib import MdtError

__all__ = [
    "bert_embed_rows",
    "gemm", "colsum", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd", "attention_mean_probs", "graph_attn_bias",
    "row_axpby", "row_scatter_add", "bert_embed_sum", "bert_embed_ln_rows", "vit_patchify", "vit_assemble", "vit_patch_embed", "graph_node_feature",
    "tanh_fwd", "tanh_bwd", "node_ce", "contrastive_loss", "fp8_quantize", "fp8_scale_update", "gemm_fp8", "cast", "transpose2d", "dropout", "dropout_mask",
    "EPI_BIAS", "EPI_GELU", "EPI_RESIDUAL", "EPI_DGELU", "EPI_ACCUM", "EPI_ATOMIC", "EPI_DROPOUT", "EPI_AUX_GRAD", "EPI_MULAUX", "EPI_ASUM",
]


def _2d(t: torch.Tensor):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor (unit inner stride)"
    return t.stride(0)


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a=False, trans_b=False, out: Optional[torch.Tensor] = None,
         out_dtype=None, bias=None, residual=None, aux=None, epilogue=0, alpha=1.0, split_k=1, drop_p=0.0,
         drop_seed=0, colsum=None, asum=None) -> torch.Tensor:
    """out[M,N] = epilogue(alpha * op(a) @ op(b)); b is [N,K] unless trans_b (then [K,N]).
    ``asum`` (fp32[M], with trans_a and EPI_ATOMIC): += the column sums of the stored ``a`` — a bias gradient riding
    on the weight-gradient GEMM (MDT_EPI_ASUM)."""
    lda, ldb = _2d(a), _2d(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[1] if trans_b else b.shape[0]
    kb = b.shape[0] if trans_b else b.shape[1]
    assert kb == K, f"gemm: inner dimensions differ ({K} vs {kb})"
    assert a.dtype == b.dtype
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
    assert out.shape == (M, N)
    if bias is not None:
        epilogue |= EPI_BIAS
        assert bias.dtype == a.dtype and bias.numel() == N and bias.is_contiguous()
    if residual is not None:
        epilogue |= EPI_RESIDUAL
        assert residual.dtype == a.dtype and residual.shape == (M, N)
    if aux is not None:
        assert aux.dtype == a.dtype and aux.shape == (M, N)
    if drop_p > 0.0:
        epilogue |= EPI_DROPOUT
    if colsum is not None:
        epilogue |= EPI_COLSUM
        assert colsum.dtype == torch.float32 and colsum.numel() == N
    if asum is not None:
        assert colsum is None and trans_a and asum.dtype == torch.float32 and asum.numel() == M and asum.is_contiguous()
        epilogue |= EPI_ASUM
        colsum = asum
    check(lib.mdt_gemm(stream(), dt(a), dt(out), int(trans_a), int(trans_b), M, N, K, ptr(a), lda, ptr(b), ldb,
                       ptr(out), _2d(out), epilogue, float(alpha), ptr(bias), ptr(residual),
                       _2d(residual) if residual is not None else 0, ptr(aux), _2d(aux) if aux is not None else 0,
                       int(split_k), float(drop_p), int(drop_seed), ptr(colsum)), "mdt_gemm")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, row_weight=None) -> torch.Tensor:
    """out[n] += sum_m w[m] * x[m, n] (fp32, atomics); w int32 or None."""
    if out is None:
        out = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    assert row_weight is None or (row_weight.dtype == torch.int32 and row_weight.numel() == x.shape[0])
    check(lib.mdt_colsum(stream(), dt(x), x.shape[0], x.shape[1], ptr(x), _2d(x), ptr(out), ptr(row_weight)),
          "mdt_colsum")
    return out


def layernorm_fwd(x, gamma, beta, eps, out=None, q8=None):
    """→ (y, mean, rstd).  ``q8`` = (u8[rows, D] out, format, scale, amax): y also leaves as fp8 for the 8-bit GEMM that consumes
    it (= fp8_quantize(y, format, scale, amax) without its pass)."""
    rows, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if q8 is not None:
        q8_out, q8_format, q8_scale, q8_amax = q8
        assert q8_out.dtype == torch.uint8 and q8_out.shape == (rows, D)
        check(lib.mdt_layernorm_fwd_q8(stream(), dt(x), rows, D, ptr(x), _2d(x), ptr(gamma), ptr(beta), float(eps), ptr(out), _2d(out),
                                       ptr(mean), ptr(rstd), ptr(q8_out), _2d(q8_out), int(q8_format), ptr(q8_scale), ptr(q8_amax)),
              "mdt_layernorm_fwd_q8")
        return out, mean, rstd
    check(lib.mdt_layernorm_fwd(stream(), dt(x), rows, D, ptr(x), _2d(x), ptr(gamma), ptr(beta), float(eps), ptr(out),
                                _2d(out), ptr(mean), ptr(rstd)), "mdt_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, add=None, dgamma=None, dbeta=None, dx=None, drop_p=0.0, drop_seed=0,
                  colsum=None, want_dropped=False):
    """→ dx, or (dx, dxd) with ``want_dropped`` (dxd = dx through the hidden-dropout mask); ``colsum`` (fp32[D])
    accumulates the column sums of dxd (dx when no dropout) — the upstream dense layer's bias gradient."""
    rows, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if (want_dropped and drop_p > 0.0) else None
    check(lib.mdt_layernorm_bwd(stream(), dt(x), rows, D, ptr(dy), _2d(dy), ptr(x), _2d(x), ptr(gamma), ptr(mean),
                                ptr(rstd), ptr(add), _2d(add) if add is not None else 0, ptr(dx), _2d(dx),
                                ptr(dgamma), ptr(dbeta), ptr(dxd), _2d(dxd) if dxd is not None else 0, float(drop_p),
                                int(drop_seed), ptr(colsum)), "mdt_layernorm_bwd")
    if want_dropped:
        return dx, (dxd if dxd is not None else dx)
    return dx


def _attn_args(qkv, out, lse, nseq, S, H, hd, seq_stride, pos_stride, scale, key_mask, dense_bias, attn_bias,
               spatial_pos, sp_table, virt, key_pad, drop_p=0.0, drop_seed=0, seq_offsets=None, q_limit=0, seq_ids=None, s_cap=0):
    a = L.AttnFwdArgs()
    a.q_limit = int(q_limit)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed)
    assert seq_offsets is None or (seq_offsets.dtype == torch.int32 and seq_offsets.is_contiguous() and
                                   seq_offsets.numel() == nseq + 1), "attention: seq_offsets must be i32[nseq + 1]"
    a.seq_offsets = ptr(seq_offsets)
    a.dtype = dt(qkv)
    a.nseq, a.S, a.H, a.hd = nseq, S, H, hd
    a.s_cap = int(s_cap)
    if seq_ids is not None:        # one length bin of the ragged set: the launch covers these sequences only
        assert seq_offsets is not None and seq_ids.dtype == torch.int32 and seq_ids.is_contiguous()
        a.seq_ids, a.nseq, a.nseq_total = ptr(seq_ids), int(seq_ids.numel()), nseq
    a.seq_stride, a.pos_stride, a.scale = seq_stride, pos_stride, float(scale)
    a.qkv, a.ld_qkv = ptr(qkv), _2d(qkv)
    a.out, a.ld_out = ptr(out), _2d(out)
    a.lse = ptr(lse)
    a.key_mask = ptr(key_mask)
    a.dense_bias = ptr(dense_bias)
    a.attn_bias = ptr(attn_bias)
    a.spatial_pos = ptr(spatial_pos)
    a.sp_table = ptr(sp_table)
    a.virt = ptr(virt)
    a.key_pad = ptr(key_pad)
    a.num_spatial = sp_table.shape[0] if sp_table is not None else 0
    for t, want in ((key_mask, torch.uint8), (key_pad, torch.uint8), (dense_bias, torch.float32),
                    (attn_bias, torch.float32), (spatial_pos, torch.int32)):
        assert t is None or (t.dtype == want and t.is_contiguous()), f"attention: expected contiguous {want}"
    return a


def attention_fwd(qkv, nseq, S, H, *, seq_stride=None, pos_stride=1, scale=None, key_mask=None, dense_bias=None,
                  attn_bias=None, spatial_pos=None, sp_table=None, virt=None, key_pad=None, drop_p=0.0, drop_seed=0,
                  seq_offsets=None, q_limit=0, bins=None):
    """qkv [rows, 3*D] → (out [rows, D], lse f32[nseq, H, S]).  ``q_limit`` > 0: only the first q_limit rows of every
    sequence are needed as queries (other rows of out / lse unspecified).  ``seq_offsets`` i32[nseq+1]: ragged sequences
    (sequence s = rows off[s]..off[s+1], at most S of them), see include/mdt_hip.h.  ``bins``: [(seq_ids i32[n], cap), ...]
    — a partition of the ragged set by length (no sequence of a bin longer than its cap): one launch per bin, each with
    the kernels that fit its cap; results are identical to the single launch."""
    D = qkv.shape[1] // 3
    hd = D // H
    # No fill, also with q_limit: the kernels that honour it (bf16, head dim 64) write whole 16-row query tiles, and the backward
    # kernels dispatched for such a launch read exactly those rows of out / lse and nothing past them
    # (tests/test_kernels_gpu.py::test_attention_backward_never_reads_what_forward_did_not_write); every other kernel family ignores
    # q_limit and writes all rows.  Positions of lse past a ragged sequence's length are never read either.
    out = torch.empty(qkv.shape[0], D, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(nseq, H, S, dtype=torch.float32, device=qkv.device)
    if bins is not None and (qkv.dtype != torch.bfloat16 or hd != 64 or S > 272):
        bins = None                      # the binned launches exist for the bf16 head-dim-64 kernels only
    for ids, cap in (bins or [(None, 0)]):
        if ids is not None and ids.numel() == 0:
            continue
        a = _attn_args(qkv, out, lse, nseq, S, H, hd, S if seq_stride is None else seq_stride, pos_stride,
                       hd ** -0.5 if scale is None else scale, key_mask, dense_bias, attn_bias, spatial_pos, sp_table,
                       virt, key_pad, drop_p, drop_seed, seq_offsets, q_limit, ids, cap)
        check(lib.mdt_attention_fwd(stream(), C.byref(a)), "mdt_attention_fwd")
    return out, lse


def attention_mean_probs(qkv, lse, nseq, S, H, *, seq_stride=None, pos_stride=1, scale=None, key_mask=None, dense_bias=None,
                         attn_bias=None, spatial_pos=None, sp_table=None, virt=None, key_pad=None):
    """Head-averaged attention probabilities f32[nseq, S, S] from the qkv buffer and the forward's lse."""
    D = qkv.shape[1] // 3
    hd = D // H
    out = torch.empty(nseq, S, S, dtype=torch.float32, device=qkv.device)
    a = _attn_args(qkv, qkv, lse, nseq, S, H, hd, S if seq_stride is None else seq_stride, pos_stride,
                   hd ** -0.5 if scale is None else scale, key_mask, dense_bias, attn_bias, spatial_pos, sp_table, virt, key_pad)
    check(lib.mdt_attention_mean_probs(stream(), C.byref(a), ptr(out)), "mdt_attention_mean_probs")
    return out


def attention_head_weights(qkv, lse, nseq, S, H, *, raw_scores=False, seq_stride=None, pos_stride=1, scale=None, key_mask=None,
                           dense_bias=None, attn_bias=None, spatial_pos=None, sp_table=None, virt=None, key_pad=None):
    """Per-head attention weights f32[nseq, H, S, S]: the softmax probabilities before dropout (needs the forward's ``lse``), or
    with ``raw_scores`` the scores q k^T * scale + bias with masked keys at -inf (``lse`` may be None)."""
    D = qkv.shape[1] // 3
    hd = D // H
    out = torch.empty(nseq, H, S, S, dtype=torch.float32, device=qkv.device)
    if lse is None:            # raw scores do not read it; the argument check wants a buffer
        assert raw_scores
        lse = torch.empty(nseq * H * S, dtype=torch.float32, device=qkv.device)
    a = _attn_args(qkv, qkv, lse, nseq, S, H, hd, S if seq_stride is None else seq_stride, pos_stride,
                   hd ** -0.5 if scale is None else scale, key_mask, dense_bias, attn_bias, spatial_pos, sp_table, virt, key_pad)
    check(lib.mdt_attention_head_weights(stream(), C.byref(a), 1 if raw_scores else 0, ptr(out)), "mdt_attention_head_weights")
    return out


def attention_bwd(dout, qkv, out, lse, nseq, S, H, *, seq_stride=None, pos_stride=1, scale=None, key_mask=None,
                  dense_bias=None, attn_bias=None, spatial_pos=None, sp_table=None, virt=None, key_pad=None,
                  want_dense_dbias=False, d_sp_table=None, d_virt=None, drop_p=0.0, drop_seed=0, seq_offsets=None, q_limit=0,
                  bins=None):
    D = qkv.shape[1] // 3
    hd = D // H
    # with q_limit the kernels skip the query rows beyond it: their dQ must read as zero (dK / dV are written for every row)
    dqkv = torch.empty_like(qkv)
    if q_limit:
        dqkv[:, :D].zero_()
    dbias = None
    if want_dense_dbias:
        dbias = torch.zeros(nseq, H, S, S, dtype=torch.float32, device=qkv.device)
    if bins is not None and (qkv.dtype != torch.bfloat16 or hd != 64 or S > 272):
        bins = None
    for ids, cap in (bins or [(None, 0)]):
        if ids is not None and ids.numel() == 0:
            continue
        b = L.AttnBwdArgs()
        b.f = _attn_args(qkv, out, lse, nseq, S, H, hd, S if seq_stride is None else seq_stride, pos_stride,
                         hd ** -0.5 if scale is None else scale, key_mask, dense_bias, attn_bias, spatial_pos, sp_table,
                         virt, key_pad, drop_p, drop_seed, seq_offsets, q_limit, ids, cap)
        b.dout, b.ld_dout = ptr(dout), _2d(dout)
        b.dqkv, b.ld_dqkv = ptr(dqkv), _2d(dqkv)
        b.d_dense_bias = ptr(dbias)
        b.d_sp_table = ptr(d_sp_table)
        b.d_virt = ptr(d_virt)
        check(lib.mdt_attention_bwd(stream(), C.byref(b)), "mdt_attention_bwd")
    return dqkv, dbias


def graph_attn_bias(attn_bias, spatial_pos, sp_table, virt):
    nseq, S, _ = attn_bias.shape
    H = sp_table.shape[1]
    out = torch.empty(nseq, H, S, S, dtype=torch.float32, device=attn_bias.device)
    check(lib.mdt_graph_attn_bias(stream(), dt(sp_table), nseq, S, H, ptr(attn_bias), ptr(spatial_pos), ptr(sp_table),
                                  ptr(virt), ptr(out)), "mdt_graph_attn_bias")
    return out


def row_axpby(dst, nrows, *, di=None, d_inner=1, d_stride=1, d_off=0, a=None, ai=None, a_inner=1, a_stride=1, a_off=0,
              alpha=1.0, b=None, bi=None, b_inner=1, b_stride=1, b_off=0, beta=1.0, accumulate=False):
    """dst[di(r)] = alpha*a[ai(r)] + beta*b[bi(r)] (+ dst); 2-D row-major tensors, int32 index vectors."""
    D = dst.shape[1]
    for t in (di, ai, bi):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous())
    check(lib.mdt_row_axpby(stream(), dt(dst), nrows, D, ptr(dst), _2d(dst), ptr(di), d_inner, d_stride, d_off,
                            ptr(a), _2d(a) if a is not None else 0, ptr(ai), a_inner, a_stride, a_off, float(alpha),
                            ptr(b), _2d(b) if b is not None else 0, ptr(bi), b_inner, b_stride, b_off, float(beta),
                            int(accumulate)), "mdt_row_axpby")
    return dst


def row_scatter_add(table_f32, idx, src, nrows, *, s_stride=1, s_off=0):
    assert table_f32.dtype == torch.float32 and idx.dtype == torch.int32
    check(lib.mdt_row_scatter_add_f32(stream(), dt(src), nrows, table_f32.shape[1], ptr(table_f32), _2d(table_f32),
                                      ptr(idx), ptr(src), _2d(src), s_stride, s_off), "mdt_row_scatter_add_f32")
    return table_f32


def bert_embed_sum(ids, types, word, pos, type_emb, out, *, out_seq_stride, out_off):
    M, Lq = ids.shape
    assert ids.dtype == torch.int32 and types.dtype == torch.int32 and ids.is_contiguous() and types.is_contiguous()
    check(lib.mdt_bert_embed_sum(stream(), dt(word), M, Lq, ptr(ids), ptr(types), ptr(word), ptr(pos), ptr(type_emb),
                                 word.shape[1], ptr(out), _2d(out), out_seq_stride, out_off), "mdt_bert_embed_sum")
    return out


def bert_embed_rows(ids, types, pos_ids, word, pos, type_emb, out):
    rows = ids.numel()
    for t in (ids, types, pos_ids):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == rows
    check(lib.mdt_bert_embed_rows(stream(), dt(word), rows, ptr(ids), ptr(types), ptr(pos_ids), ptr(word), ptr(pos),
                                  ptr(type_emb), word.shape[1], ptr(out), _2d(out)), "mdt_bert_embed_rows")
    return out


def bert_embed_ln_rows(ids, types, pos_ids, word, pos, type_emb, gamma, beta, eps, *, keep_sum=True):
    """word + type + position rows and their LayerNorm in one pass → (y, mean, rstd, summed rows or None)."""
    rows = ids.numel()
    for t in (ids, types, pos_ids):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == rows
    D = word.shape[1]
    y = torch.empty(rows, D, dtype=word.dtype, device=word.device)
    xs = torch.empty(rows, D, dtype=word.dtype, device=word.device) if keep_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=word.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=word.device)
    check(lib.mdt_bert_embed_ln_rows(stream(), dt(word), rows, ptr(ids), ptr(types), ptr(pos_ids), ptr(word), ptr(pos), ptr(type_emb), D,
                                     ptr(gamma), ptr(beta), float(eps), ptr(xs), _2d(xs) if xs is not None else 0, ptr(y), _2d(y),
                                     ptr(mean), ptr(rstd)), "mdt_bert_embed_ln_rows")
    return y, mean, rstd, xs


def vit_patchify(images, patch, dtype, k_pad=0):
    """``k_pad`` > C * patch^2: the gathered matrix gets zero columns up to that width (the MFMA tile GEMMs want K a multiple of
    their k-step: ViT-L/14 has K = 588)."""
    I, Cc, HW, _ = images.shape
    assert images.dtype == torch.float32 and images.is_contiguous()
    g = HW // patch
    k = Cc * patch * patch
    if k_pad > k:
        cols = torch.empty(I * g * g, k_pad, dtype=dtype, device=images.device)
        cols[:, k:].zero_()
    else:
        cols = torch.empty(I * g * g, k, dtype=dtype, device=images.device)
    check(lib.mdt_vit_patchify(stream(), dt(cols), I, Cc, HW, patch, ptr(images), ptr(cols), _2d(cols)),
          "mdt_vit_patchify")
    return cols


def vit_assemble(patches, cls, pos, tokens, I, npatch, *, seq_stride, off):
    check(lib.mdt_vit_assemble(stream(), dt(patches), I, npatch, patches.shape[1], ptr(patches), _2d(patches), ptr(cls),
                               ptr(pos), ptr(tokens), _2d(tokens), seq_stride, off), "mdt_vit_assemble")
    return tokens


def vit_patch_embed(images, patch, wmat, bias, cls, pos, tokens, *, seq_stride, off):
    """Patch gather + projection + bias + [CLS] + position add in one launch (bf16 weights, 16 x 16 patches); raises
    ``MdtUnsupported`` for other shapes — the caller then takes vit_patchify + gemm + vit_assemble."""
    I, Cc, HW, _ = images.shape
    assert images.dtype == torch.float32 and images.is_contiguous() and wmat.dtype == torch.bfloat16 and wmat.stride(1) == 1
    check(lib.mdt_vit_patch_embed(stream(), I, Cc, HW, patch, ptr(images), ptr(wmat), wmat.stride(0), ptr(bias), ptr(cls), ptr(pos),
                                  wmat.shape[0], ptr(tokens), _2d(tokens), seq_stride, off), "mdt_vit_patch_embed")
    return tokens


def graph_node_feature(src, node_row, in_degree, out_degree, in_emb, out_emb, graph_token, B, T):
    D = in_emb.shape[1]
    x = torch.empty(B * T, D, dtype=in_emb.dtype, device=in_emb.device)
    assert node_row.dtype == torch.int32 and in_degree.dtype == torch.int32 and out_degree.dtype == torch.int32
    check(lib.mdt_graph_node_feature(stream(), dt(x), B, T, D, ptr(src), _2d(src) if src is not None else 0,
                                     ptr(node_row), ptr(in_degree), ptr(out_degree), ptr(in_emb), ptr(out_emb),
                                     ptr(graph_token), ptr(x), _2d(x)), "mdt_graph_node_feature")
    return x


def tanh_fwd(x):
    y = torch.empty_like(x)
    check(lib.mdt_tanh_fwd(stream(), dt(x), x.numel(), ptr(x), ptr(y)), "mdt_tanh_fwd")
    return y


def tanh_bwd(y, dy):
    dx = torch.empty_like(y)
    check(lib.mdt_tanh_bwd(stream(), dt(y), y.numel(), ptr(y), ptr(dy), ptr(dx)), "mdt_tanh_bwd")
    return dx


def node_ce(logits, rows, targets, w_neg, w_pos, *, fp16_loss=True, grad_scale=1.0, want_grad=True):
    """→ (loss f32[1], counters i32[4], dlogits or None)."""
    M = logits.shape[0]
    assert logits.shape[1] == 2 and logits.is_contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    counters = torch.empty(4, dtype=torch.int32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    check(lib.mdt_node_ce(stream(), dt(logits), M, rows.numel(), ptr(logits), ptr(rows), ptr(targets), float(w_neg),
                          float(w_pos), int(fp16_loss), float(grad_scale), ptr(loss), ptr(counters), ptr(dlogits)),
          "mdt_node_ce")
    return loss, counters, dlogits


def contrastive_loss(emb, y, hard_y, scale, soft_negative_weight, adaptive, *, grad_scale=1.0, want_grad=True):
    """emb [B, D] (row-major), y / hard_y f32[B] → (loss f32[1], counters i32[4], d_emb or None); include/mdt_hip.h."""
    B, D = emb.shape
    assert y.dtype == torch.float32 and hard_y.dtype == torch.float32 and y.numel() == B and hard_y.numel() == B
    ws = torch.empty(lib.mdt_contrastive_loss_workspace_bytes(B, D), dtype=torch.uint8, device=emb.device)
    loss = torch.empty(1, dtype=torch.float32, device=emb.device)
    counters = torch.empty(4, dtype=torch.int32, device=emb.device)
    d_emb = torch.empty_like(emb) if want_grad else None
    check(lib.mdt_contrastive_loss(stream(), dt(emb), B, D, ptr(emb), _2d(emb), ptr(y), ptr(hard_y), float(scale),
                                   float(soft_negative_weight), int(bool(adaptive)), ptr(ws), float(grad_scale), ptr(loss),
                                   ptr(counters), ptr(d_emb), _2d(d_emb) if d_emb is not None else 0), "mdt_contrastive_loss")
    return loss, counters, d_emb


FP8_E4M3, FP8_E5M2 = 0, 1
FP8_MAX = {FP8_E4M3: 448.0, FP8_E5M2: 57344.0}


def fp8_quantize(src: torch.Tensor, fmt: int = FP8_E4M3, scale: Optional[torch.Tensor] = None, amax: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """u8[rows, cols] = saturate_fp8(src * scale); ``scale`` / ``amax``: fp32 device scalars (1-element tensors or views)."""
    rows, cols = src.shape
    if out is None:
        out = torch.empty(rows, cols, dtype=torch.uint8, device=src.device)
    check(lib.mdt_fp8_quantize(stream(), dt(src), int(fmt), rows, cols, ptr(src), _2d(src), ptr(out), _2d(out), ptr(scale), ptr(amax)),
          "mdt_fp8_quantize")
    return out


def fp8_scale_update(amax, scale, inv_scale, fmt_max, margin=1.0):
    check(lib.mdt_fp8_scale_update(stream(), amax.numel(), ptr(amax), ptr(scale), ptr(inv_scale), ptr(fmt_max), float(margin)),
          "mdt_fp8_scale_update")


def gemm_fp8(a8: torch.Tensor, b8: torch.Tensor, inv_scale_a: torch.Tensor, inv_scale_b: torch.Tensor, *, a_format: int = FP8_E4M3,
             out: Optional[torch.Tensor] = None, bias=None, residual=None, aux=None, epilogue=0, drop_p=0.0, drop_seed=0, colsum=None,
             q8_out: Optional[torch.Tensor] = None, q8_format: int = FP8_E4M3, q8_scale=None, q8_amax=None):
    """out[M, N] (bf16) = epilogue(inv_scale_a * inv_scale_b * a8 @ b8^T); a8 u8[M, K] (e4m3 / e5m2), b8 u8[N, K] (e4m3).
    ``q8_out`` u8[M, N]: also the fp8 copy of ``out`` (= fp8_quantize(out, q8_format, q8_scale, q8_amax)), written by the GEMM's
    own epilogue; raises MdtError (unsupported) where no kernel does that."""
    M, K = a8.shape
    N = b8.shape[0]
    assert b8.shape[1] == K and a8.dtype == torch.uint8 and b8.dtype == torch.uint8
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device)
    if bias is not None:
        epilogue |= EPI_BIAS
    if residual is not None:
        epilogue |= EPI_RESIDUAL
    if drop_p > 0.0:
        epilogue |= EPI_DROPOUT
    if colsum is not None:
        epilogue |= EPI_COLSUM
    if q8_out is not None:
        assert q8_out.dtype == torch.uint8 and q8_out.shape == (M, N)
        check(lib.mdt_gemm_fp8_q8(stream(), int(a_format), M, N, K, ptr(a8), _2d(a8), ptr(b8), _2d(b8), ptr(out), _2d(out), epilogue,
                                  ptr(inv_scale_a), ptr(inv_scale_b), ptr(bias), ptr(residual), _2d(residual) if residual is not None else 0,
                                  ptr(aux), _2d(aux) if aux is not None else 0, float(drop_p), int(drop_seed), ptr(colsum),
                                  ptr(q8_out), _2d(q8_out), int(q8_format), ptr(q8_scale), ptr(q8_amax)), "mdt_gemm_fp8_q8")
        return out
    check(lib.mdt_gemm_fp8(stream(), int(a_format), M, N, K, ptr(a8), _2d(a8), ptr(b8), _2d(b8), ptr(out), _2d(out), epilogue,
                           ptr(inv_scale_a), ptr(inv_scale_b), ptr(bias), ptr(residual), _2d(residual) if residual is not None else 0,
                           ptr(aux), _2d(aux) if aux is not None else 0, float(drop_p), int(drop_seed), ptr(colsum)), "mdt_gemm_fp8")
    return out


def cast(src, dtype):
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    assert src.is_contiguous()
    check(lib.mdt_cast(stream(), dt(src), dt(dst), src.numel(), ptr(src), ptr(dst)), "mdt_cast")
    return dst


def transpose2d(src, dtype=None):
    dst = torch.empty(src.shape[1], src.shape[0], dtype=dtype or src.dtype, device=src.device)
    check(lib.mdt_transpose2d(stream(), dt(src), dt(dst), src.shape[0], src.shape[1], ptr(src), _2d(src), ptr(dst),
                              _2d(dst)), "mdt_transpose2d")
    return dst


def dropout(x, p, seed, out=None):
    """y = x * keep(seed, row*D + col) / (1 - p); the same call on a gradient is the backward."""
    if out is None:
        out = torch.empty_like(x)
    check(lib.mdt_dropout(stream(), dt(x), x.shape[0], x.shape[1], ptr(x), _2d(x), ptr(out), _2d(out), float(p), int(seed)),
          "mdt_dropout")
    return out


def dropout_mask(n, p, seed, device="cuda"):
    m = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib.mdt_dropout_mask(stream(), n, float(p), int(seed), ptr(m)), "mdt_dropout_mask")
    return m


# ------------------------------------------------------------------------------------------ image front end
_PLAN_CACHE: dict = {}


def resize_plan(in_size: int, out_size: int):
    """PIL's bilinear taps of one axis (``mdt_resize_plan``, host): (bounds i32[out, 2], coeffs i32[out, k])."""
    import numpy as np
    key = (int(in_size), int(out_size))
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        k = lib.mdt_resize_plan_ksize(*key)
        bounds = np.zeros((key[1], 2), dtype=np.int32)
        coeffs = np.zeros((key[1], k), dtype=np.int32)
        check(lib.mdt_resize_plan(key[0], key[1], bounds.ctypes.data, coeffs.ctypes.data, k), "mdt_resize_plan")
        hit = _PLAN_CACHE[key] = (bounds, coeffs)
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    return hit


def image_norm_lut(rescale=1.0 / 255.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    import numpy as np
    m, s_ = np.asarray(mean, dtype=np.float32), np.asarray(std, dtype=np.float32)
    lut = np.zeros(3 * 256, dtype=np.float32)
    check(lib.mdt_image_norm_lut(float(rescale), m.ctypes.data, s_.ctypes.data, lut.ctypes.data), "mdt_image_norm_lut")
    return lut


class PackedImages:
    """Host side of ``mdt_image_preprocess``: decoded RGB images (uint8 HWC, any sizes) laid back to back in ONE pinned byte
    buffer with the descriptor / tap tables the kernel walks.  ``upload()`` → device copies (non-blocking)."""

    def __init__(self, images, out_size: int = 224, pin: bool = True):
        import numpy as np
        n = len(images)
        self.n, self.out_size = n, out_size
        desc = np.zeros((n, 8), dtype=np.int64)
        plans, plan_off, off_of = [], 0, {}
        pix_off = tmp_off = 0
        for i, im in enumerate(images):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise MdtError("image_preprocess takes decoded RGB images: uint8 [H, W, 3]")
            H, W = int(im.shape[0]), int(im.shape[1])
            offs = []
            for size in (W, H):
                if size not in off_of:
                    b, c = resize_plan(size, out_size)
                    off_of[size] = (plan_off, c.shape[1])
                    plans += [b.reshape(-1), c.reshape(-1)]
                    plan_off += b.size + c.size
                offs.append(off_of[size])
            desc[i] = (pix_off, tmp_off, H, W, offs[0][0], offs[0][1], offs[1][0], offs[1][1])
            pix_off += H * W * 3
            tmp_off += H * out_size * 3
        self.max_h = int(desc[:, 2].max()) if n else 0
        self.tmp_bytes = int(tmp_off)
        self.pixels = torch.empty(max(pix_off, 1), dtype=torch.uint8, pin_memory=pin and torch.cuda.is_available())
        view = self.pixels.numpy()
        for i, im in enumerate(images):
            o = int(desc[i, 0])
            view[o:o + im.size] = np.ascontiguousarray(im).reshape(-1)
        self.desc = torch.from_numpy(desc)
        self.plan = torch.from_numpy(np.concatenate(plans) if plans else np.zeros(1, dtype=np.int32))

    def upload(self, device="cuda"):
        return (self.pixels.to(device, non_blocking=True), self.desc.to(device, non_blocking=True), self.plan.to(device, non_blocking=True))


_LUT_DEV: dict = {}


def image_preprocess(packed: "PackedImages", *, dtype=torch.float32, device="cuda", rescale=1.0 / 255.0, mean=(0.5, 0.5, 0.5),
                     std=(0.5, 0.5, 0.5), return_bytes: bool = False):
    """Decoded RGB bytes → ViT pixel tensors [n, 3, out, out] on the device (``mdt_image_preprocess``): PIL-exact bilinear resize,
    x / 255, (x - mean) / std — the reference's ViTImageProcessor call (hateful_discussions.py:168-184) for a whole batch.
    ``return_bytes``: also the resized uint8 images [n, out, out, 3] (tests)."""
    n, S = packed.n, packed.out_size
    out = torch.empty(n, 3, S, S, dtype=dtype, device=device)
    if n == 0:
        return (out, torch.empty(0, S, S, 3, dtype=torch.uint8, device=device)) if return_bytes else out
    key = (float(rescale), tuple(mean), tuple(std), str(device))
    lut = _LUT_DEV.get(key)
    if lut is None:
        lut = _LUT_DEV[key] = torch.from_numpy(image_norm_lut(rescale, mean, std)).to(device)
    pix, desc, plan = packed.upload(device)
    tmp = torch.empty(packed.tmp_bytes, dtype=torch.uint8, device=device)
    u8 = torch.empty(n, S, S, 3, dtype=torch.uint8, device=device) if return_bytes else None
    check(lib.mdt_image_preprocess(stream(), n, packed.max_h, ptr(pix), ptr(desc), ptr(plan), ptr(tmp), ptr(lut), dt(out), ptr(out),
                                   ptr(u8), S), "mdt_image_preprocess")
    return (out, u8) if return_bytes else out
