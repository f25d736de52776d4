"""Drop-in for mDT/src/modules/multigraphormer_graph_encoder.py
(``MultiGraphormerGraphEncoder``): the whole fused multimodal graph-attention forward /
backward of mDT scheduled as one tape of HIP kernels.

What is kept from the reference: constructor arguments, sub-module / parameter names (656
state-dict keys at the shipped configuration, HF transformers-4.x inner names), the layer
count rules (``num_fusion_layers + 1`` fusion layers sliced off the *end* of BERT / ViT,
``ceil(Lf / num_fusion_stack)`` fusion stacks, one more graph stack than fusion stacks of
which stack ``F-1`` is never executed), and every load-bearing quirk of SURVEY.md §8.

What changed: ragged comments are addressed through CSR index vectors computed by the
packer (no boolean-mask indexing, no host sync, no ``.cuda()`` uploads inside forward),
token sequences live in persistent ``[M, nb+L, D]`` / ``[I, nb+P, D]`` buffers, the
structural attention bias is never materialised, graph tokens stay batch-major.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import engine as E
from ..data.packer import get_ragged, PackedBatch, packed_from_batched_data
from ._fused import BertModel, ViTModel
from .graphormer_graph_encoder_layer import GraphEncoderStack, GraphormerGraphEncoderLayer  # noqa: F401
from .graphormer_layers import GraphAttnBias, GraphNodeFeature
from .multi_graphormer_fusion_layer import GraphFusionLayer, GraphFusionStack  # noqa: F401
from .multihead_attention import MultiheadAttention


def init_graphormer_params(module):
    """multigraphormer_graph_encoder.py:18-39 (only with --apply-graphormer-init)."""
    def normal_(data):
        data.copy_(data.cpu().normal_(mean=0.0, std=0.02).to(data.device))

    if isinstance(module, nn.Linear):
        normal_(module.weight.data)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        normal_(module.weight.data)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    if isinstance(module, MultiheadAttention):
        normal_(module.qkv_weight.data)


BERT_BASE = dict(dim=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512, type_vocab=2)
VIT_BASE = dict(dim=768, layers=12, heads=12, intermediate=3072, image_size=224, patch=16)


class MultiGraphormerGraphEncoder(nn.Module):
    # Shapes of the two pre-trained encoders the reference downloads ("bert-base-uncased",
    # "google/vit-base-patch16-224", :236-245).  There is no network here: the same
    # architectures are built with random initialisation; override for other sizes.
    bert_config = BERT_BASE
    vit_config = VIT_BASE

    def __init__(self, num_atoms: int, num_in_degree: int, num_out_degree: int, num_edges: int, num_spatial: int,
                 num_edge_dis: int, num_bottle_neck: int, num_fusion_layers: int, edge_type: str, multi_hop_max_dist: int,
                 num_fusion_stack: int = 1, num_graph_stack: int = 1, num_encoder_layers: int = 12, embedding_dim: int = 768,
                 ffn_embedding_dim: int = 768, num_attention_heads: int = 32, dropout: float = 0.1,
                 attention_dropout: float = 0.1, activation_dropout: float = 0.1, layerdrop: float = 0.0,
                 encoder_normalize_before: bool = False, pre_layernorm: bool = False, apply_graphormer_init: bool = False,
                 activation_fn: str = "gelu", embed_scale: float = None, freeze_embeddings: bool = False,
                 n_trans_layers_to_freeze: int = 0, export: bool = False, traceable: bool = False, q_noise: float = 0.0,
                 qn_block_size: int = 8, freeze_initial_encoders: bool = False, bert_config: Optional[dict] = None,
                 vit_config: Optional[dict] = None) -> None:
        super().__init__()
        if bert_config is not None:
            self.bert_config = dict(BERT_BASE, **bert_config)
        if vit_config is not None:
            self.vit_config = dict(VIT_BASE, **vit_config)
        if layerdrop > 0 or q_noise > 0 or embed_scale is not None:
            raise NotImplementedError("layerdrop / quant_noise / embed_scale are unused by mDT and unsupported here")
        self.dropout_p = dropout
        self.attention_dropout_p = attention_dropout
        self.activation_dropout_p = activation_dropout
        self.layerdrop = layerdrop
        self.embedding_dim = embedding_dim
        self.apply_graphormer_init = apply_graphormer_init
        self.traceable = traceable
        self.pre_layernorm = pre_layernorm
        self.num_graph_heads = num_attention_heads
        num_encoder_layers = num_fusion_layers        # :86 — --encoder-layers is overwritten
        if self.bert_config["dim"] != embedding_dim or self.vit_config["dim"] != embedding_dim:
            raise ValueError("graph, BERT and ViT hidden sizes must agree (the reference hard-codes 768 for all three)")
        self.graph_node_feature = GraphNodeFeature(num_heads=num_attention_heads, num_atoms=num_atoms,
                                                   num_in_degree=num_in_degree, num_out_degree=num_out_degree,
                                                   hidden_dim=embedding_dim, n_layers=num_encoder_layers)
        self.graph_attn_bias = GraphAttnBias(num_heads=num_attention_heads, num_atoms=num_atoms, num_edges=num_edges,
                                             num_spatial=num_spatial, num_edge_dis=num_edge_dis, edge_type=edge_type,
                                             multi_hop_max_dist=multi_hop_max_dist, hidden_dim=embedding_dim,
                                             n_layers=num_encoder_layers)
        self.embed_scale = embed_scale
        self.quant_noise = None
        self.emb_layer_norm = nn.LayerNorm(embedding_dim, eps=1e-5) if encoder_normalize_before else None
        if pre_layernorm:
            self.final_layer_norm = nn.LayerNorm(embedding_dim, eps=1e-5)      # constructed, never applied (:124-125)
        self.layers = nn.ModuleList([])
        (self.vit_model, self.vit_pooler, vit_other_layers, self.text_model, self.text_pooler, text_other_layers,
         self.node_classifier, self.text_dropout) = self.build_vit_bert_encoders(num_fusion_layers + 1, attention_dropout,
                                                                                activation_dropout)
        self.fusion_layers = nn.ModuleList([])
        nfs = num_fusion_stack
        text_groups = [text_other_layers[i * nfs:(i + 1) * nfs] for i in range((len(text_other_layers) + nfs - 1) // nfs)]
        vit_groups = [vit_other_layers[i * nfs:(i + 1) * nfs] for i in range((len(vit_other_layers) + nfs - 1) // nfs)]
        self.fusion_layers.extend([GraphFusionStack(t, v, num_bottle_neck, use_projection=True)
                                   for t, v in zip(text_groups, vit_groups)])
        self.layers.extend([self.build_graphormer_graph_encoder_layer(
            num_layers=num_graph_stack, embedding_dim=embedding_dim, ffn_embedding_dim=ffn_embedding_dim,
            num_attention_heads=num_attention_heads, dropout=dropout, attention_dropout=attention_dropout,
            activation_dropout=activation_dropout, activation_fn=activation_fn, export=export, q_noise=q_noise,
            qn_block_size=qn_block_size, pre_layernorm=pre_layernorm) for _ in range(len(self.fusion_layers) + 1)])
        self.num_bottle_neck = num_bottle_neck
        self.bottle_neck = nn.Embedding(num_bottle_neck, embedding_dim)
        # valid-token packing of the text side (see _indices); MDT_DENSE_TOKENS=1 or ``ragged_tokens = False`` runs
        # the padded layout, which also reproduces the reference's hidden states at padded positions
        self.ragged_tokens = os.environ.get("MDT_DENSE_TOKENS", "0") != "1"
        # ragged text attention as one launch per length bin (data/packer.py RaggedText.length_bins); MDT_LENGTH_BINS=0: one launch
        self.length_bins = os.environ.get("MDT_LENGTH_BINS", "1") != "0"
        # logits path only (GraphormerModel.forward): the last fusion layer computes just the rows that are read
        # afterwards; MDT_FULL_LAST_LAYER=1 or ``prune_last_layer = False`` computes every row as the reference does
        self.prune_last_layer = os.environ.get("MDT_FULL_LAST_LAYER", "0") != "1"
        # image branch on a second HIP stream beside the text branch (engine.Tape: fork / join / on_side);
        # MDT_TWO_STREAMS=0 or ``two_streams = False`` enqueues everything on one stream
        self.two_streams = os.environ.get("MDT_TWO_STREAMS", "1") != "0"

        def set_grad(m, flag):
            if m is not None:
                for p in m.parameters():
                    p.requires_grad = flag

        if freeze_embeddings:
            raise NotImplementedError("Freezing embeddings is not implemented yet.")
        if freeze_initial_encoders:      # :223-228 — after the fusion layers were sliced out
            set_grad(self.text_model, False)
            set_grad(self.vit_model, False)
            set_grad(self.node_classifier, True)
            set_grad(self.text_pooler, True)
            set_grad(self.vit_pooler, True)
        for layer in range(n_trans_layers_to_freeze):
            set_grad(self.layers[layer], False)
        self.use_main_grad = False
        self.grad_ready_hook = None

    # ------------------------------------------------------------------ construction
    def build_vit_bert_encoders(self, num_fusion_layers, attention_dropout, activation_dropout):
        """Same return tuple as the reference (:233-278).  The last ``num_fusion_layers`` blocks of
        each encoder become fusion layers; the truncated models keep embeddings, the first
        blocks, ViT's final LayerNorm and the poolers."""
        bc, vc = self.bert_config, self.vit_config
        vit_model = ViTModel(vc["dim"], vc["layers"], vc["heads"], vc["intermediate"], vc["image_size"], vc["patch"])
        bert_model = BertModel(bc["dim"], bc["layers"], bc["heads"], bc["intermediate"], bc["vocab"], bc["max_pos"],
                               bc["type_vocab"])
        if num_fusion_layers == 0:
            vit_other, bert_other = [], []
        else:
            vit_other = list(vit_model.encoder.layer[-num_fusion_layers:])
            vit_model.encoder.layer = vit_model.encoder.layer[:-num_fusion_layers]
            bert_other = list(bert_model.encoder.layer[-num_fusion_layers:])
            bert_model.encoder.layer = bert_model.encoder.layer[:-num_fusion_layers]
        for lyr in list(vit_model.encoder.layer) + vit_other + list(bert_model.encoder.layer) + bert_other:
            lyr.hidden_dropout_p = activation_dropout            # hidden_dropout_prob=activation_dropout (:238,:243)
            lyr.attention_dropout_p = attention_dropout          # attention_probs_dropout_prob=attention_dropout
        node_classifier = nn.Linear(bc["dim"], 2)       # BertForSequenceClassification.classifier, num_labels = 2
        nn.init.normal_(node_classifier.weight, 0.0, 0.02)
        nn.init.zeros_(node_classifier.bias)
        bert_dropout = nn.Dropout(activation_dropout)
        return (vit_model, vit_model.pooler, vit_other, bert_model, bert_model.pooler, bert_other, node_classifier,
                bert_dropout)

    def build_graphormer_graph_encoder_layer(self, embedding_dim, ffn_embedding_dim, num_attention_heads, dropout,
                                             attention_dropout, activation_dropout, activation_fn, export, q_noise,
                                             qn_block_size, pre_layernorm, num_layers=1):
        return GraphEncoderStack(num_layers=num_layers, embedding_dim=embedding_dim, ffn_embedding_dim=ffn_embedding_dim,
                                 num_attention_heads=num_attention_heads, dropout=dropout,
                                 attention_dropout=attention_dropout, activation_dropout=activation_dropout,
                                 activation_fn=activation_fn, export=export, q_noise=q_noise, qn_block_size=qn_block_size,
                                 pre_layernorm=pre_layernorm)

    # ------------------------------------------------------------------ index helpers
    def _indices(self, pb: PackedBatch):
        """Row geometry of the text side, as index vectors, for the two layouts the tape can run in:
        padded (every comment owns nb + L rows, attention masks the padding — bit-for-bit the reference, including
        the hidden states of padded positions) and ragged (``ragged_tokens``, default: only valid tokens own rows;
        logits and gradients are unchanged, see data/packer.py RaggedText)."""
        nb = self.num_bottle_neck
        ragged = bool(self.ragged_tokens) and not self.needs_long_attention(pb)
        key = ("enc_idx", nb, ragged)
        if key in pb.extras:
            return pb.extras[key]
        dev = pb.ids.device
        M, Lq = pb.M, pb.L
        np_ = (self.vit_config["image_size"] // self.vit_config["patch"]) ** 2
        Sv = nb + np_ + 1
        if pb.host is not None:
            idx = self._indices_host(pb, nb, ragged, np_, Sv)
            pb.extras[key] = idx
            return idx
        i32 = dict(device=dev, dtype=torch.int32)
        j = torch.arange(nb, **i32)
        m_ar = torch.arange(M, **i32)
        if ragged:
            rt = get_ragged(pb)
            off_pre = rt.offsets                                   # [M+1]
            off_fus = (off_pre + torch.arange(M + 1, **i32) * nb).contiguous()
            bn0 = off_fus[:M].contiguous()
            pre2fus = (torch.arange(rt.rows, **i32) + (rt.comment + 1) * nb).contiguous()
            rows_pre, rows_fus = rt.rows, rt.rows + M * nb
            S_pre, S_fus = rt.max_len, rt.max_len + nb
            # comments of at most 64 rows (four key tiles) run the small attention kernels in a launch of their own
            spec_pre = dict(S=S_pre, seq_offsets=off_pre, bins=rt.length_bins(0) if self.length_bins else None)
            spec_fus = dict(S=S_fus, seq_offsets=off_fus, bins=rt.length_bins(nb) if self.length_bins else None)
        else:
            St = nb + Lq
            bn0 = (m_ar * St).contiguous()
            r = torch.arange(M * Lq, **i32)
            pre2fus = (torch.div(r, Lq, rounding_mode="floor") * St + nb + r % Lq).contiguous()
            rows_pre, rows_fus = M * Lq, M * St
            ones = torch.ones(M, nb, dtype=torch.uint8, device=dev)
            spec_pre = dict(S=Lq, key_mask=pb.text_mask)
            spec_fus = dict(S=St, key_mask=torch.cat([ones, pb.text_mask], dim=1).contiguous())
        idx = dict(
            ragged=ragged, Sv=Sv, P=np_ + 1, rows_pre=rows_pre, rows_fus=rows_fus, spec_pre=spec_pre, spec_fus=spec_fus,
            pre2fus=pre2fus, bn0_rows=bn0, cls_rows=(bn0 + nb).contiguous(),
            bn_rows_all=(bn0[:, None] + j[None]).reshape(-1).contiguous(),
            text_row_of_node=torch.where(pb.node_row >= 0, bn0[pb.node_row.clamp(min=0).long()], pb.node_row).contiguous(),
            img_text_bn_rows=(bn0[pb.img_comment.long()][:, None] + j[None]).reshape(-1).contiguous(),
            vit_bn_rows=(torch.arange(pb.I, **i32)[:, None] * Sv + j[None]).reshape(-1).contiguous(),
        )
        pb.extras[key] = idx
        return idx

    def _indices_host(self, pb: PackedBatch, nb: int, ragged: bool, np_: int, Sv: int):
        """The same vectors as the device arithmetic of ``_indices``, computed with numpy from the packer's host copies and
        uploaded (pinned, non-blocking, on whatever stream is current — the prefetch thread's copy stream): a batch from
        ``pack_batch`` costs the step no eager index kernels at all."""
        import numpy as np
        h = pb.host
        dev = pb.ids.device
        M, Lq, I = pb.M, pb.L, pb.I
        pin = bool(h["pin"]) and dev.type == "cuda"

        def up(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
            if pin:
                t = t.pin_memory()
            return t.to(dev, non_blocking=bool(h["non_blocking"]))

        j = np.arange(nb, dtype=np.int64)
        m_ar = np.arange(M, dtype=np.int64)
        if ragged:
            rt = get_ragged(pb)
            off_pre = h["offsets"]
            off_fus = off_pre + np.arange(M + 1, dtype=np.int64) * nb
            bn0 = off_fus[:M]
            pre2fus = np.arange(rt.rows, dtype=np.int64) + (h["comment"] + 1) * nb
            rows_pre, rows_fus = rt.rows, rt.rows + M * nb
            S_pre, S_fus = rt.max_len, rt.max_len + nb
            spec_pre = dict(S=S_pre, seq_offsets=rt.offsets, bins=rt.length_bins(0) if self.length_bins else None)
            spec_fus = dict(S=S_fus, seq_offsets=up(off_fus), bins=rt.length_bins(nb) if self.length_bins else None)
        else:
            St = nb + Lq
            bn0 = m_ar * St
            r = np.arange(M * Lq, dtype=np.int64)
            pre2fus = (r // Lq) * St + nb + r % Lq
            rows_pre, rows_fus = M * Lq, M * St
            km = np.concatenate([np.ones((M, nb), dtype=np.uint8), h["text_mask"].reshape(M, Lq).astype(np.uint8)], axis=1)
            kmt = torch.from_numpy(np.ascontiguousarray(km))
            if pin:
                kmt = kmt.pin_memory()
            spec_pre = dict(S=Lq, key_mask=pb.text_mask)
            spec_fus = dict(S=St, key_mask=kmt.to(dev, non_blocking=bool(h["non_blocking"])))
        node_row = h["node_row"]
        bn0_dev = up(bn0)
        return dict(
            ragged=ragged, Sv=Sv, P=np_ + 1, rows_pre=rows_pre, rows_fus=rows_fus, spec_pre=spec_pre, spec_fus=spec_fus,
            pre2fus=up(pre2fus), bn0_rows=bn0_dev, cls_rows=up(bn0 + nb),
            bn_rows_all=up((bn0[:, None] + j[None]).reshape(-1)),
            text_row_of_node=up(np.where(node_row >= 0, bn0[np.clip(node_row, 0, None)] if M else node_row, node_row)),
            img_text_bn_rows=up((bn0[h["img_comment"]][:, None] + j[None]).reshape(-1)),
            vit_bn_rows=up((np.arange(I, dtype=np.int64)[:, None] * Sv + j[None]).reshape(-1)),
            _host=dict(bn0=bn0, up=up),
        )

    # ------------------------------------------------------------------ tape-level forward
    def _fwd(self, tape, pb: PackedBatch, prune_last: bool = False):
        """→ (text buffer Var, global embedding Var [B, D], rows) where ``rows`` = dict(bn0_rows, cls_rows) locates
        bottleneck token 0 and [CLS] of every comment in the returned text buffer: the [rows_fus, D] layout of
        ``_indices``, or — ``prune_last``, the logits path — the compact [2M, D] output of a last fusion layer that
        computed only those two rows per comment (engine.transformer_block ``keep_rows``)."""
        tr = self.training
        self.check_sequence_limits(pb)                       # before any kernel is enqueued
        p_emb = self.activation_dropout_p if tr else 0.0     # HF hidden_dropout_prob := act_dropout (:238,:243)
        nb = self.num_bottle_neck
        ix = self._indices(pb)
        Sv, P = ix["Sv"], ix["P"]
        M, I, B, T = pb.M, pb.I, pb.B, pb.T
        tm, vm = self.text_model, self.vit_model
        if self.two_streams and I > 0:
            tape.enable_side(pb.ids.device)
        # the 6 + 6 pre-fusion layers of the two modalities are independent: image branch first (on the side stream of
        # a two-stream tape), text branch beside it
        # --freeze_initial_encoders (the reference launch, run_train.sh:61): autograd stops at the frozen stacks there.
        # A prefix without any trainable parameter records no adjoint (nothing of it is kept for backward) and its
        # output is a constant of the tape, so the gradient chain ends at the first fusion layer.
        def trainable(*mods):
            return any(p.requires_grad for m in mods for p in m.parameters())

        vit_live = trainable(vm.embeddings, vm.encoder, vm.layernorm)
        text_live = trainable(tm.embeddings, tm.encoder)
        was_inference = tape.inference
        vit = None
        if I > 0:
            tape.fork()
            tape.inference = was_inference or not vit_live
            with tape.on_side():
                ve = vm.embeddings
                v = E.vit_embeddings(tape, pb.images, ve.patch_embeddings.projection.weight, ve.patch_embeddings.projection.bias,
                                     ve.cls_token, ve.position_embeddings, vm.patch)
                v = E.dropout(tape, v, p_emb)
                specv = E.AttnSpec(nseq=I, S=P, H=vm.heads)
                for layer in vm.encoder.layer:
                    v = E.transformer_block(tape, v, layer.block_params(), specv, pre_ln=True, eps=vm.eps,
                                            **layer.drop_kwargs())
                v = E.layernorm(tape, v, vm.layernorm.weight, vm.layernorm.bias, vm.eps)      # quirk 5: final LN mid-network
                v.needs_grad = vit_live
                vit = E.expand_sequences(tape, v, I, P, nb, None)     # zero bottleneck rows in front: no parameter involved
        tape.inference = was_inference or not text_live
        e = tm.embeddings
        if ix["ragged"]:
            rt = get_ragged(pb)
            if E.EMBED_LN_FUSED:
                normed = E.bert_embeddings_ln_rows(tape, rt.ids, rt.types, rt.pos, e.word_embeddings.weight, e.position_embeddings.weight,
                                                   e.token_type_embeddings.weight, e.LayerNorm.weight, e.LayerNorm.bias, tm.eps)
            else:
                emb = E.bert_embeddings_rows(tape, rt.ids, rt.types, rt.pos, e.word_embeddings.weight,
                                             e.position_embeddings.weight, e.token_type_embeddings.weight)
                normed = E.layernorm(tape, emb, e.LayerNorm.weight, e.LayerNorm.bias, tm.eps)
        else:
            emb = E.bert_embeddings(tape, pb.ids, pb.types, e.word_embeddings.weight, e.position_embeddings.weight,
                                    e.token_type_embeddings.weight)
            normed = E.layernorm(tape, emb, e.LayerNorm.weight, e.LayerNorm.bias, tm.eps)
        text = E.dropout(tape, normed, p_emb)
        spec0 = E.AttnSpec(nseq=M, H=tm.heads, **ix["spec_pre"])
        for layer in tm.encoder.layer:
            text = E.transformer_block(tape, text, layer.block_params(), spec0, pre_ln=False, eps=tm.eps,
                                       **layer.drop_kwargs())
        text.needs_grad = text_live
        tape.inference = was_inference
        if I > 0:
            tape.join()
        text = E.expand_rows(tape, text, ix["rows_fus"], ix["pre2fus"], ix["bn_rows_all"], nb, self.bottle_neck.weight)
        fargs = (M, ix["spec_fus"], I, Sv, ix["img_text_bn_rows"], ix["vit_bn_rows"])
        text, vit = self.fusion_layers[0]._fwd(tape, text, vit, *fargs)
        gnf = self.graph_node_feature
        x = E.graph_node_features(tape, text, ix["text_row_of_node"], pb.degree, pb.degree, gnf.in_degree_encoder.weight,
                                  gnf.out_degree_encoder.weight, gnf.graph_token.weight, B, T, ix["bn0_rows"], pb.graph_row,
                                  M, pb.deg_scatter, pb.deg_scatter)
        if self.emb_layer_norm is not None:
            x = E.layernorm(tape, x, self.emb_layer_norm.weight, self.emb_layer_norm.bias, 1e-5)
        x = E.dropout(tape, x, self.dropout_p if tr else 0.0)          # self.dropout_module(x)  (:403)
        gab = self.graph_attn_bias
        hd = self.embedding_dim // self.num_graph_heads
        gspec = E.AttnSpec(nseq=B, S=T, H=self.num_graph_heads, scale=hd ** -0.5, attn_bias=pb.attn_bias,
                           spatial_pos=pb.spatial_pos, sp_table=gab.spatial_pos_encoder.weight,
                           virt=gab.graph_token_virtual_distance.weight, key_pad=pb.key_pad)
        F = len(self.fusion_layers)
        rows = dict(bn0_rows=ix["bn0_rows"], cls_rows=ix["cls_rows"])
        for st in range(F - 1):                       # zip(self.layers, self.fusion_layers[1:])  (:413)
            x = self.layers[st]._fwd(tape, x, gspec)
            # bottle_neck[:, 0, :] = x[mask]  (:425)
            E.rows_mix(tape, text, x, M, alpha=1.0, beta=0.0, d_idx=ix["bn0_rows"], s_idx=pb.graph_row)
            prune = None
            if prune_last and st == F - 2 and not self.needs_long_attention(pb):
                # after the last fusion layer only bottleneck token 0 (graph copy-back, head) and [CLS] (head) of each
                # comment are ever read: compute just those rows there
                prune, rows = self._prune_indices(pb, ix)
            text, vit = self.fusion_layers[st + 1]._fwd(tape, text, vit, *fargs, prune=prune)
            # x[mask] = bottle_neck[:, 0, :]  (:435)
            E.rows_mix(tape, x, text, M, alpha=1.0, beta=0.0, d_idx=pb.graph_row, s_idx=rows["bn0_rows"])
        x = self.layers[-1]._fwd(tape, x, gspec)       # layers[F]; layers[F-1] is never executed (quirk 3)
        glob = E.take_rows(tape, x, B, s_map=(1, T, 0))
        if I == 0 and tape.on_params_ready is not None and not tape.inference:
            # a batch without any image comment: the ViT side never runs, so its parameters would never be reported to the
            # gradient exchange and every bucket behind them would wait for the end of backward — on ALL ranks, since
            # collectives are matched.  Their gradients are the zeros already in the arena: report them first thing in
            # backward (the last entry recorded is the first one the backward walk reaches).
            image_side = [p for p in self.vit_model.parameters()]
            for st in self.fusion_layers:
                for fl in st.fusion_layers:
                    image_side += list(fl.vit_encoder.parameters())
            tape.record(lambda: tape.on_params_ready(image_side))
        return text, glob, rows

    MAX_ATTENTION_TOKENS = 272      # one workgroup holds a whole (sequence, head) on chip (csrc/attention*.hip)

    def needs_long_attention(self, pb: PackedBatch) -> bool:
        """Text (nb + L tokens) and image (nb + P tokens) sequences run on the single-pass MFMA attention kernels up to 272
        tokens — every configuration of BASELINE.json (L = 100; ViT-B/16: 201, ViT-L/14: 261).  The reference's encoders take
        more (BERT up to max_position_embeddings = 512 tokens, multigraphormer_graph_encoder.py:236-245; a 384-px ViT has 577):
        such a batch runs in the reference's padded layout with every row of the last fusion layer computed (the key-chunked
        kernels of csrc/attention_long.hip take key masks, not ragged offsets or a query limit) — same numbers, plain fp32
        FMAs instead of MFMA.  Discussion TREES may be of any size either way (--max-nodes 10000 is declared and never
        enforced, tasks/task.py:41-44): graph attention over more than 271 comments takes the same kernels."""
        lim = self.MAX_ATTENTION_TOKENS
        nb = self.num_bottle_neck
        npatch = (self.vit_config["image_size"] // self.vit_config["patch"]) ** 2 + 1
        return nb + pb.L > lim or (pb.I > 0 and nb + npatch > lim)

    def check_sequence_limits(self, pb: PackedBatch):
        """What the encoders cannot take at all is refused HERE, with the numbers, before anything is enqueued: comments longer
        than BERT's position table (HF raises an index error there)."""
        max_pos = self.text_model.embeddings.position_embeddings.weight.shape[0]
        if pb.L > max_pos:
            raise ValueError(f"{pb.L} text tokens per comment exceed the BERT position table ({max_pos})")

    def _prune_indices(self, pb: PackedBatch, ix):
        key = ("prune_idx", self.num_bottle_neck, bool(self.ragged_tokens))
        if key not in pb.extras and "_host" in ix:
            import numpy as np
            hb, up = ix["_host"]["bn0"], ix["_host"]["up"]
            m2 = np.arange(pb.M, dtype=np.int64) * 2
            prune = dict(text_keep=up(np.stack([hb, hb + self.num_bottle_neck], axis=1).reshape(-1)),
                         vit_keep=up(np.arange(pb.I, dtype=np.int64) * ix["Sv"]),
                         img_bn0_compact=up(pb.host["img_comment"] * 2))
            pb.extras[key] = (prune, dict(bn0_rows=up(m2), cls_rows=up(m2 + 1)))
        if key not in pb.extras:
            dev = pb.ids.device
            i32 = dict(device=dev, dtype=torch.int32)
            m_ar = torch.arange(pb.M, **i32)
            text_keep = torch.stack([ix["bn0_rows"], ix["cls_rows"]], dim=1).reshape(-1).contiguous()     # 2m, 2m+1
            prune = dict(text_keep=text_keep,
                         vit_keep=(torch.arange(pb.I, **i32) * ix["Sv"]).contiguous(),
                         img_bn0_compact=(pb.img_comment * 2).contiguous())
            rows = dict(bn0_rows=(m_ar * 2).contiguous(), cls_rows=(m_ar * 2 + 1).contiguous())
            pb.extras[key] = (prune, rows)
        return pb.extras[key]

    def live_parameters(self):
        seen, out = set(), []
        for p in self.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def forward(self, batched_data, last_state_only: bool = False, token_embeddings: Optional[torch.Tensor] = None,
                attn_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """→ (text [M, L, D], bottle_neck [M, nb, D], global_embedding [B, D]) as the reference (:464)."""
        if attn_mask is not None or token_embeddings is not None:
            raise NotImplementedError("attn_mask / token_embeddings are unused by mDT and unsupported")
        pb = packed_from_batched_data(batched_data)
        nb = self.num_bottle_neck

        def run(tape):
            text, glob, _ = self._fwd(tape, pb)
            return text, glob

        ix = self._indices(pb)
        if not ix["ragged"]:
            buf, glob = E.run_tape(run, [], self.live_parameters(), use_main_grad=self.use_main_grad, hook=self.grad_ready_hook)
            D = buf.shape[1]
            buf = buf.view(pb.M, nb + pb.L, D)
            return buf[:, nb:], buf[:, :nb], glob

        # ragged tape: give the caller the reference's padded shapes; rows of padded positions (which the
        # reference fills with the hidden states of padding tokens, never read by mDT) are zero
        def run_padded(tape):
            text, glob, _ = self._fwd(tape, pb)
            rt = get_ragged(pb)
            dense_rows = (rt.comment * pb.L + rt.pos).contiguous()
            fus_rows = ix["pre2fus"]
            txt = E.scatter_rows(tape, text, pb.M * pb.L, dense_rows, fus_rows)
            bn = E.take_rows(tape, text, pb.M * nb, s_idx=ix["bn_rows_all"])
            return txt, bn, glob

        txt, bn, glob = E.run_tape(run_padded, [], self.live_parameters(), use_main_grad=self.use_main_grad,
                                   hook=self.grad_ready_hook)
        D = txt.shape[1]
        return txt.view(pb.M, pb.L, D), bn.view(pb.M, nb, D), glob
