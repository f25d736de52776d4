"""Drop-in for mDT/src/modules/multi_graphormer_fusion_layer.py (``GraphFusionLayer``,
``GraphFusionStack``): one BERT layer over ``[bottleneck ‖ text]`` and one ViT layer over
``[bottleneck[image comments] ‖ patches]`` sharing ``nb`` bottleneck tokens.

Token buffers are persistent ``[M, nb+L, D]`` / ``[I, nb+P, D]`` tensors: the reference's
``torch.cat`` of the full hidden state on every layer (:37-39, :56-58) is gone, the
bottleneck exchange is three indexed row kernels:
  1. ViT bottleneck rows ← the layer's *input* text bottleneck (the ViT branch reads the
     pre-BERT bottleneck, :57, quirk 6),
  2. BERT layer, ViT layer,
  3. text bottleneck rows of image comments ← ½·(ViT bottleneck + BERT bottleneck) (:63-66).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import engine as E
from ._fused import BertLayer, ViTLayer


class GraphFusionLayer(nn.Module):
    def __init__(self, bert_layer: BertLayer, vit_layer: ViTLayer, num_bottle_neck_tokens: int,
                 use_projection: bool = False) -> None:
        super().__init__()
        self.bert_encoder = bert_layer
        self.vit_encoder = vit_layer
        self.gradient_checkpointing = False
        self.num_bottle_neck_tokens = num_bottle_neck_tokens
        if use_projection:
            # constructed and saved by the reference, never called (:22-24); kept for checkpoint parity
            self.bert_projection = nn.Linear(bert_layer.dim, bert_layer.dim)
            self.vit_projection = nn.Linear(bert_layer.dim, bert_layer.dim)
        else:
            self.bert_projection = nn.Identity()
            self.vit_projection = nn.Identity()

    def live_parameters(self):
        return list(self.bert_encoder.parameters()) + list(self.vit_encoder.parameters())

    def _fwd(self, tape, text: E.Var, vit: Optional[E.Var], M: int, text_spec: dict, I: int, Sv: int,
             img_text_bn_rows, vit_bn_rows, prune: Optional[dict] = None):
        """text [rows, D] (padded: M*(nb+L) rows + key mask; ragged: valid tokens only + sequence offsets — both
        described by ``text_spec`` = AttnSpec keywords), vit [I*Sv, D] or None.  ``img_text_bn_rows`` /
        ``vit_bn_rows``: i32[I*nb] row indices of the bottleneck tokens of image comments in the text / image buffers.
        ``prune`` (the LAST fusion layer of the logits path): dict(text_keep i32[2M] = rows of bottleneck token 0 and
        [CLS] of every comment, vit_keep i32[I] = bottleneck-0 row of every image, img_bn0_compact i32[I] = row of
        that comment's bottleneck token 0 in the compact text output).  Both blocks then run their output projection,
        LayerNorms and FFN on the kept rows only and return compact [2M, D] / [I, D] tensors."""
        nb = self.num_bottle_neck_tokens
        be, ve = self.bert_encoder, self.vit_encoder
        if vit is not None:
            E.rows_mix(tape, vit, text, I * nb, alpha=1.0, beta=0.0, d_idx=vit_bn_rows, s_idx=img_text_bn_rows)
        # pruned layer: the kept rows (bottleneck 0, [CLS]) are among the first nb + 1 rows of a comment, row 0 of an image
        # the two blocks are independent: on a two-stream tape the image block runs beside the text block
        vit_out = None
        if vit is not None:
            tape.fork()
            with tape.on_side():
                spec_v = E.AttnSpec(nseq=I, S=Sv, H=ve.heads, q_limit=0 if prune is None else 1)
                vit_out = E.transformer_block(tape, vit, ve.block_params(), spec_v, pre_ln=True, eps=ve.eps, **ve.drop_kwargs(),
                                              keep_rows=None if prune is None else prune["vit_keep"])
        spec_t = E.AttnSpec(nseq=M, H=be.heads, q_limit=0 if prune is None else nb + 1, **text_spec)
        text_out = E.transformer_block(tape, text, be.block_params(), spec_t, pre_ln=False, eps=be.eps, **be.drop_kwargs(),
                                       keep_rows=None if prune is None else prune["text_keep"])
        if vit is not None:
            tape.join()
            if prune is None:
                E.rows_mix(tape, text_out, vit_out, I * nb, alpha=0.5, beta=0.5, d_idx=img_text_bn_rows, s_idx=vit_bn_rows)
            else:   # only bottleneck token 0 is read after the last fusion layer
                E.rows_mix(tape, text_out, vit_out, I, alpha=0.5, beta=0.5, d_idx=prune["img_bn0_compact"])
        return text_out, vit_out

    def forward(self, bert_hidden_states: torch.Tensor, vit_hidden_states: torch.Tensor, bottle_neck: torch.Tensor,
                bert_attention_mask: Optional[torch.FloatTensor] = None, x_image_indexes: Optional[torch.Tensor] = None):
        """Reference signature: text [M,L,D], vit [I,P,D] | None, bottle_neck [M,nb,D], additive mask
        [M,1,1,nb+L] (0 = keep, large negative = drop), x_image_indexes bool[M].
        Returns (text [M,L,D], vit [I,P,D] | None, bottle_neck [M,nb,D])."""
        M, Lq, D = bert_hidden_states.shape
        nb = self.num_bottle_neck_tokens
        St = nb + Lq
        dev = bert_hidden_states.device
        if bert_attention_mask is None:
            mask = torch.ones(M, St, dtype=torch.uint8, device=dev)
        else:
            mask = (bert_attention_mask.reshape(M, St).float() > -1.0).to(torch.uint8).contiguous()
        have_img = vit_hidden_states is not None
        I, P = (vit_hidden_states.shape[0], vit_hidden_states.shape[1]) if have_img else (0, 0)
        Sv = nb + P
        if have_img:
            # positions of the image comments WITHOUT torch.nonzero (its output size is data-dependent: a D2H sync): I is known
            # from the ViT tensor, and a stable sort of the negated mask lists the True positions first, in order
            img = torch.sort((~x_image_indexes.bool()).to(torch.uint8), stable=True).indices[:I]
            j = torch.arange(nb, device=dev)
            img_text_rows = (img[:, None] * St + j[None]).to(torch.int32).reshape(-1).contiguous()
            vit_rows = (torch.arange(I, device=dev)[:, None] * Sv + j[None]).to(torch.int32).reshape(-1).contiguous()
        else:
            img_text_rows = vit_rows = None
        inputs = [bert_hidden_states.contiguous().view(M * Lq, D), bottle_neck.contiguous().view(M * nb, D)]
        if have_img:
            inputs.append(vit_hidden_states.contiguous().view(I * P, D))

        def run(tape, tv, bv, vv=None):
            text = E.expand_sequences(tape, tv, M, Lq, nb, None)
            E.rows_mix(tape, text, bv, M * nb, alpha=1.0, beta=0.0, d_map=(nb, St, 0))
            vit = E.expand_sequences(tape, vv, I, P, nb, None) if vv is not None else None
            t_out, v_out = self._fwd(tape, text, vit, M, dict(S=St, key_mask=mask), I, Sv, img_text_rows, vit_rows)
            outs = [E.take_rows(tape, t_out, M * Lq, s_map=(Lq, St, nb)), E.take_rows(tape, t_out, M * nb, s_map=(nb, St, 0))]
            if v_out is not None:
                outs.append(E.take_rows(tape, v_out, I * P, s_map=(P, Sv, nb)))
            return tuple(outs)

        outs = E.run_tape(run, inputs, self.live_parameters())
        text_o = outs[0].view(M, Lq, D)
        bn_o = outs[1].view(M, nb, D)
        vit_o = outs[2].view(I, P, D) if have_img else None
        return text_o, vit_o, bn_o


class GraphFusionStack(nn.Module):
    def __init__(self, bert_layers, vit_layers, num_bottle_neck_tokens, use_projection=False) -> None:
        super().__init__()
        self.fusion_layers = nn.ModuleList([
            GraphFusionLayer(b, v, num_bottle_neck_tokens, use_projection) for b, v in zip(bert_layers, vit_layers)])

    def _fwd(self, tape, text, vit, *a, prune=None):
        n = len(self.fusion_layers)
        for k, f in enumerate(self.fusion_layers):
            text, vit = f._fwd(tape, text, vit, *a, prune=prune if k == n - 1 else None)
        return text, vit

    def forward(self, bert_hidden_states, vit_hidden_states, bottle_neck, bert_attention_mask=None, x_image_indexes=None):
        for f in self.fusion_layers:
            bert_hidden_states, vit_hidden_states, bottle_neck = f(bert_hidden_states, vit_hidden_states, bottle_neck,
                                                                   bert_attention_mask, x_image_indexes)
        return bert_hidden_states, vit_hidden_states, bottle_neck
