"""Ragged-batch packer: discussion trees → the tensors the HIP path consumes.

Replaces ``preprocess_item`` + ``collator`` of the reference
(mDT/src/data/pyg_datasets/pre_processing.py:18-69, mDT/src/data/collator.py:69-179,
``y_mask`` from mDT/src/data/dataset.py:210-213).  The structural integer tensors come
from the native C++ packer (``mdt_pack_structure``) and are bit-exact with the
reference; on top of the reference's padded dict this packer emits the CSR index vectors
(comment ↔ graph row, image ↔ comment, labelled rows) that let every bottleneck-token
exchange on the device run without ``nonzero`` / host synchronisation.  Host buffers are
pinned when a GPU is present so the H2D copies are asynchronous.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from .. import _lib as L


def pack_structure(parents: List[np.ndarray], spatial_pos_max: int, nmax: Optional[int] = None, updown=None):
    """→ (attn_bias f32[B,T,T], spatial_pos i32[B,N,N], in_degree i64[B,N]) as numpy arrays.  ``updown``: optional list
    with, per tree, None or an explicit i64[n, n, 2] (hops up, hops down) matrix that replaces the one derived from the
    parent array (include/mdt_hip.h, mdt_pack_structure_ud)."""
    B = len(parents)
    parents = [np.ascontiguousarray(p, dtype=np.int64) for p in parents]
    n_nodes = np.asarray([len(p) for p in parents], dtype=np.int64)
    nmax = int(n_nodes.max()) if nmax is None else nmax
    T = nmax + 1
    attn_bias = np.empty((B, T, T), dtype=np.float32)
    spatial_pos = np.empty((B, nmax, nmax), dtype=np.int32)
    in_degree = np.empty((B, nmax), dtype=np.int64)
    ptrs = (C.c_void_p * B)(*[p.ctypes.data for p in parents])
    ud_arg = None
    if updown is not None and any(u is not None for u in updown):
        uds = [None if u is None else np.ascontiguousarray(u, dtype=np.int64) for u in updown]
        for u, p in zip(uds, parents):
            if u is not None and u.shape != (len(p), len(p), 2):
                raise ValueError(f"updown matrix of shape {u.shape} for a tree of {len(p)} comments")
        ud_ptrs = (C.c_void_p * B)(*[None if u is None else u.ctypes.data for u in uds])
        ud_arg = C.cast(ud_ptrs, C.c_void_p)
    L.check(L.lib.mdt_pack_structure_ud(B, n_nodes.ctypes.data, C.cast(ptrs, C.c_void_p), ud_arg, nmax, int(spatial_pos_max),
                                        attn_bias.ctypes.data, spatial_pos.ctypes.data, in_degree.ctypes.data),
            "mdt_pack_structure")
    return attn_bias, spatial_pos, in_degree


@dataclass
class PackedBatch:
    """Device-resident batch.  ``batched_data`` is the reference's dict (same keys, dtypes and
    padding); the remaining fields are the ragged index vectors of the fused path."""
    B: int
    N: int                      # padded nodes per tree
    M: int                      # real comments in the batch
    I: int                      # image-bearing comments
    L: int                      # text tokens per comment
    batched_data: dict
    ids: torch.Tensor           # i32[M, L]
    types: torch.Tensor         # i32[M, L]
    text_mask: torch.Tensor     # u8[M, L]   1 = real token
    node_row: torch.Tensor      # i32[B*N]   comment index of (b, n) or -1
    graph_row: torch.Tensor     # i32[M]     b*T + 1 + n of comment m
    degree: torch.Tensor        # i32[B*N]   shifted degree (0 = padding)
    deg_scatter: torch.Tensor   # i32[B*T]   embedding row fed by graph row r in backward (-1: graph token / padding_idx)
    key_pad: torch.Tensor       # u8[B, T]   1 = padded graph key
    attn_bias: torch.Tensor     # f32[B, T, T]
    spatial_pos: torch.Tensor   # i32[B, N, N]
    img_comment: torch.Tensor   # i32[I]     comment index of image i
    images: Optional[torch.Tensor]   # f32[I, 3, H, W]
    label_rows: torch.Tensor    # i32[#lab]  comment index of each label
    targets: torch.Tensor       # i32[#lab]
    n_labels: int = 0
    extras: dict = field(default_factory=dict)
    ragged: Optional["RaggedText"] = None      # valid-token packing (built on the host by pack_batch; lazily otherwise)
    host: Optional[dict] = None                # pack_batch only: numpy copies of what the encoder's index vectors are derived from
    #                                            (token offsets, comment of every token row, node_row, img_comment, token mask),
    #                                            so that those vectors are built on the HOST in the packer thread, not by eager kernels

    @property
    def T(self):
        return self.N + 1


@dataclass
class RaggedText:
    """Valid-token packing of the text side: only positions with attention_mask == 1 get a row.  The reference
    pads every comment to L tokens and computes all of them (HF BERT with an additive mask); padded positions are
    never attended to and never read by the head, so logits and every gradient are unchanged when they are simply
    not computed (SURVEY.md §8 item 8 makes the same point for padded graph nodes)."""
    rows: int                   # T0 = number of valid tokens in the batch
    max_len: int                # longest comment (tokens)
    offsets: torch.Tensor       # i32[M + 1]  first packed row of every comment
    ids: torch.Tensor           # i32[T0]
    types: torch.Tensor         # i32[T0]
    pos: torch.Tensor           # i32[T0]     original position of the token in its comment (position-embedding index)
    comment: torch.Tensor       # i32[T0]     comment index of the row
    order: Optional[torch.Tensor] = None      # i32[M]   comment indices by increasing token count (stable)
    sorted_lens: Optional["np.ndarray"] = None   # host int array [M]: the token counts in that order — lets the host cut
    #                                             length bins (``length_bins``) without reading anything back from the device

    def length_bins(self, extra: int, caps=(64,)):
        """[(comment ids i32[n] (device), cap), ...]: the comments whose ``tokens + extra`` fit each cap, shortest bin
        first, the rest under the overall maximum — attention then runs one launch per bin with the kernels that fit it
        (ops.attention_fwd ``bins``).  None when there is nothing to cut (one bin would hold everything)."""
        if self.order is None or self.sorted_lens is None:
            return None
        top = self.max_len + extra
        cuts = [c for c in sorted(caps) if c < top]
        if not cuts:
            return None
        bins, lo = [], 0
        for c in cuts + [top]:
            hi = int(np.searchsorted(self.sorted_lens, c - extra, side="right")) if c != top else int(self.sorted_lens.shape[0])
            if hi > lo:
                bins.append((self.order[lo:hi], int(c)))
            lo = hi
        return bins if len(bins) > 1 else None


def ragged_text(ids: torch.Tensor, types: torch.Tensor, mask: torch.Tensor, device=None, non_blocking=True) -> RaggedText:
    """Build the packing from [M, L] id / type / mask matrices (host tensors in ``pack_batch``: no device sync;
    device tensors in the compatibility path: one ``nonzero``)."""
    M, Lq = ids.shape
    valid = mask.reshape(-1) != 0
    src = torch.nonzero(valid).flatten()
    lens = (mask != 0).sum(1)
    if int(lens.min()) < 1:
        raise ValueError("a comment without any valid token cannot be packed")
    off = torch.zeros(M + 1, dtype=torch.int64, device=ids.device)
    off[1:] = torch.cumsum(lens, 0)
    comment = torch.div(src, Lq, rounding_mode="floor")

    def to(t):
        t = t.to(torch.int32).contiguous()
        return t if device is None else t.to(device, non_blocking=non_blocking)

    sl, order = torch.sort(lens, stable=True)
    return RaggedText(rows=int(src.numel()), max_len=int(lens.max()), offsets=to(off), ids=to(ids.reshape(-1)[src]),
                      types=to(types.reshape(-1)[src]), pos=to(src - comment * Lq), comment=to(comment), order=to(order),
                      sorted_lens=sl.cpu().numpy().astype(np.int64))


def pack_batch(trees: List[dict], spatial_pos_max: int = 10, device="cuda", non_blocking=True) -> PackedBatch:
    B = len(trees)
    n_nodes = [len(t["parent"]) for t in trees]
    N = max(n_nodes)
    Lq = trees[0]["input_ids"].shape[1]
    M = int(sum(n_nodes))
    attn_bias, spatial_pos, in_degree = pack_structure([t["parent"] for t in trees], spatial_pos_max, N,
                                                       updown=[t.get("updown") for t in trees])
    pin = torch.cuda.is_available() and str(device).startswith("cuda")

    def host(shape, dtype):
        return torch.zeros(shape, dtype=dtype, pin_memory=pin)

    x = host((B, N, Lq), torch.int64)
    tt = host((B, N, Lq), torch.int64)
    am = host((B, N, Lq), torch.int64)
    ids = host((M, Lq), torch.int32)
    types = host((M, Lq), torch.int32)
    tmask = host((M, Lq), torch.uint8)
    node_row = torch.full((B * N,), -1, dtype=torch.int32, pin_memory=pin)
    graph_row = host((M,), torch.int32)
    key_pad = host((B, N + 1), torch.uint8)
    deg_scatter = torch.full((B, N + 1), -1, dtype=torch.int32, pin_memory=pin)
    img_index, images, ys, y_masks = [], [], [], []
    images_u8, u8_size = [], 224     # decoded RGB bytes at the images' own sizes (device front end, data/discussions.py)
    m = 0
    for b, t in enumerate(trees):
        n = n_nodes[b]
        ii, ty, at = (torch.from_numpy(np.ascontiguousarray(t[k])) for k in ("input_ids", "token_type_ids", "attention_mask"))
        x[b, :n], tt[b, :n], am[b, :n] = ii, ty, at
        ids[m:m + n], types[m:m + n], tmask[m:m + n] = ii.int(), ty.int(), at.to(torch.uint8)
        node_row[b * N:b * N + n] = torch.arange(m, m + n, dtype=torch.int32)
        graph_row[m:m + n] = torch.arange(b * (N + 1) + 1, b * (N + 1) + 1 + n, dtype=torch.int32)
        key_pad[b, n + 1:] = 1
        img_index.append(np.asarray(t["image_index"], dtype=bool))
        if t.get("images") is not None and len(t["images"]):
            images.append(t["images"])
        if t.get("images_u8"):
            images_u8 += list(t["images_u8"])
            u8_size = int(t.get("image_size", 224))
        ys.append(np.asarray(t["y"], dtype=np.float32).reshape(-1))
        # node task: which comments carry a label; graph-level (contrastive) trees have one y per tree and no mask
        y_masks.append(np.asarray(t["y_mask"], dtype=bool) if "y_mask" in t else np.zeros(n, dtype=bool))
        m += n
    token_mask = ~(x == 0).all(dim=2)                      # collator.py:141
    if not bool(torch.equal(token_mask.sum(1), torch.tensor(n_nodes))):
        raise ValueError("a comment with an all-zero input_ids row cannot be told from padding (collator.py:141)")
    img_index = np.concatenate(img_index)
    y = np.concatenate(ys)
    y_mask = np.concatenate(y_masks)
    node_task = "y_mask" in trees[0]
    if node_task and int(y_mask.sum()) != y.shape[0]:
        raise ValueError("y must hold exactly one entry per True in y_mask")
    img_comment = torch.from_numpy(np.nonzero(img_index)[0].astype(np.int32))
    label_rows = torch.from_numpy(np.nonzero(y_mask)[0].astype(np.int32))
    images_t = None
    if images:
        # one pinned buffer, every tree's pixels copied into it once (308 MB per C2 batch: a torch.cat followed by
        # pin_memory() would move them twice and keep the packer thread behind the training step)
        n_img = sum(int(im.shape[0]) for im in images)
        if n_img != img_comment.numel():
            raise ValueError("number of image tensors differs from the number of image-bearing comments")
        images_t = torch.empty((n_img,) + tuple(images[0].shape[1:]), dtype=torch.float32, pin_memory=pin)
        o = 0
        for im in images:
            k = int(im.shape[0])
            src = im if torch.is_tensor(im) else torch.from_numpy(np.asarray(im))
            images_t[o:o + k].copy_(src)           # converts a non-fp32 source on the way
            o += k

    def to(t):
        return None if t is None else t.to(device, non_blocking=non_blocking)

    images_dev = None
    if images_u8:
        # device image front end: bytes up, mdt_image_preprocess (PIL-exact resize, 1 / 255, (x - 0.5) / 0.5) on the current stream
        if images:
            raise ValueError("a batch mixes host-preprocessed images (fp32) with decoded bytes (images_u8)")
        if len(images_u8) != img_comment.numel():
            raise ValueError("number of decoded images differs from the number of image-bearing comments")
        if not str(device).startswith("cuda"):
            raise ValueError("images_u8 needs the device front end (device='cuda'); on the host use image_preprocess='host'")
        from .. import ops as _ops
        images_dev = _ops.image_preprocess(_ops.PackedImages(images_u8, u8_size, pin=pin), device=device)

    in_deg_t = torch.from_numpy(in_degree)
    d32 = in_deg_t.int()
    deg_scatter[:, 1:] = torch.where(d32 > 0, d32, torch.full_like(d32, -1))
    bd = dict(
        idx=torch.arange(B, dtype=torch.int64),
        attn_bias=to(torch.from_numpy(attn_bias)),
        spatial_pos=to(torch.from_numpy(spatial_pos)),
        in_degree=to(in_deg_t),
        x_token_mask=to(token_mask),
        x=to(x), x_token_type_ids=to(tt), x_attention_mask=to(am),
        x_images=images_dev if images_dev is not None else to(images_t),
        x_image_indexes=to(torch.from_numpy(img_index)),
        y=to(torch.from_numpy(y)),
    )
    if node_task:
        bd["y_mask"] = to(torch.from_numpy(y_mask))
    if "hard_y" in trees[0]:
        bd["hard_y"] = to(torch.from_numpy(np.concatenate([np.asarray(t["hard_y"], dtype=np.float32).reshape(-1) for t in trees])))
    bd["out_degree"] = bd["in_degree"]            # collator.py:171 — the same tensor object
    pb = PackedBatch(
        B=B, N=N, M=M, I=int(img_comment.numel()), L=Lq, batched_data=bd,
        ids=to(ids), types=to(types), text_mask=to(tmask), node_row=to(node_row), graph_row=to(graph_row),
        degree=to(in_deg_t.reshape(-1).int()), deg_scatter=to(deg_scatter.view(-1)), key_pad=to(key_pad), attn_bias=bd["attn_bias"],
        spatial_pos=bd["spatial_pos"], img_comment=to(img_comment), images=bd["x_images"],
        label_rows=to(label_rows), targets=to(torch.from_numpy((y if node_task else y[:0]).astype(np.int32))),
        n_labels=int(label_rows.numel()),
    )
    pb.ragged = ragged_text(ids, types, tmask, device=device, non_blocking=non_blocking)   # host-side: no device sync
    lens_np = tmask.numpy().astype(np.int64).sum(1)
    off_np = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(lens_np, out=off_np[1:])
    pb.host = dict(offsets=off_np, comment=np.repeat(np.arange(M, dtype=np.int64), lens_np), node_row=node_row.numpy().astype(np.int64),
                   img_comment=img_comment.numpy().astype(np.int64), text_mask=tmask.numpy(), pin=pin, non_blocking=non_blocking)
    bd["_packed"] = pb                             # lets model(**net_input) find the CSR view
    # the same view as plain tensors / ints: a sample that is moved between devices by a generic "apply to every tensor
    # of the nested dict" (FairSeq's utils.move_to_cuda) carries these along, and packed_from_batched_data rebuilds the
    # PackedBatch from them on the other side without any nonzero / host sync
    rt = pb.ragged
    bd["_csr"] = dict(
        B=B, N=N, M=M, I=pb.I, L=Lq, n_labels=pb.n_labels, ids=pb.ids, types=pb.types, text_mask=pb.text_mask,
        node_row=pb.node_row, graph_row=pb.graph_row, degree=pb.degree, deg_scatter=pb.deg_scatter, key_pad=pb.key_pad,
        img_comment=pb.img_comment, label_rows=pb.label_rows, targets=pb.targets,
        rt_rows=rt.rows, rt_max_len=rt.max_len, rt_offsets=rt.offsets, rt_ids=rt.ids, rt_types=rt.types, rt_pos=rt.pos,
        rt_comment=rt.comment, rt_order=rt.order, rt_sorted_lens=rt.sorted_lens)
    return pb


def get_ragged(pb: PackedBatch) -> RaggedText:
    if pb.ragged is None:
        pb.ragged = ragged_text(pb.ids, pb.types, pb.text_mask)
    return pb.ragged


def packed_from_batched_data(bd: dict) -> PackedBatch:
    """Compatibility path: derive the CSR view from a reference-style collated dict (as produced by
    ``collator``) that is already on the device.  Uses boolean-mask ``nonzero`` (one host sync);
    ``pack_batch`` avoids it by building the indices on the host."""
    dev_now = bd["x_token_mask"].device
    if "_packed" in bd and bd["_packed"].ids.device == dev_now:
        return bd["_packed"]
    if "_csr" in bd and bd["_csr"]["ids"].device == dev_now:
        c = bd["_csr"]
        images = bd.get("x_images")
        pb = PackedBatch(
            B=c["B"], N=c["N"], M=c["M"], I=c["I"], L=c["L"], batched_data=bd, ids=c["ids"], types=c["types"],
            text_mask=c["text_mask"], node_row=c["node_row"], graph_row=c["graph_row"], degree=c["degree"],
            deg_scatter=c["deg_scatter"], key_pad=c["key_pad"], attn_bias=bd["attn_bias"], spatial_pos=bd["spatial_pos"],
            img_comment=c["img_comment"], images=images, label_rows=c["label_rows"], targets=c["targets"],
            n_labels=c["n_labels"])
        pb.ragged = RaggedText(rows=c["rt_rows"], max_len=c["rt_max_len"], offsets=c["rt_offsets"], ids=c["rt_ids"],
                               types=c["rt_types"], pos=c["rt_pos"], comment=c["rt_comment"], order=c.get("rt_order"),
                               sorted_lens=c.get("rt_sorted_lens"))
        bd["_packed"] = pb
        return pb
    mask = bd["x_token_mask"]
    dev = mask.device
    B, N = mask.shape
    Lq = bd["x"].shape[2]
    T = N + 1
    flat = mask.reshape(-1)
    real = torch.nonzero(flat).flatten()
    M = int(real.numel())
    node_row = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    node_row[real] = torch.arange(M, dtype=torch.int32, device=dev)
    b_of = torch.div(real, N, rounding_mode="floor")
    graph_row = (b_of * T + 1 + (real - b_of * N)).to(torch.int32)
    deg = bd["in_degree"].reshape(B, N).to(torch.int32)
    deg_scatter = torch.full((B, T), -1, dtype=torch.int32, device=dev)
    deg_scatter[:, 1:] = torch.where(deg > 0, deg, torch.full_like(deg, -1))
    key_pad = torch.zeros(B, T, dtype=torch.uint8, device=dev)
    key_pad[:, 1:] = (~mask).to(torch.uint8)
    img_idx = bd.get("x_image_indexes")
    images = bd.get("x_images")
    img_comment = (torch.nonzero(img_idx).flatten().to(torch.int32) if images is not None
                   else torch.zeros(0, dtype=torch.int32, device=dev))
    y_mask = bd.get("y_mask")
    label_rows = (torch.nonzero(y_mask).flatten().to(torch.int32) if y_mask is not None
                  else torch.zeros(0, dtype=torch.int32, device=dev))
    y = bd.get("y")
    pb = PackedBatch(
        B=B, N=N, M=M, I=int(img_comment.numel()), L=Lq, batched_data=bd,
        ids=bd["x"][mask].to(torch.int32).contiguous(), types=bd["x_token_type_ids"][mask].to(torch.int32).contiguous(),
        text_mask=bd["x_attention_mask"][mask].to(torch.uint8).contiguous(), node_row=node_row, graph_row=graph_row,
        degree=deg.reshape(-1).contiguous(), deg_scatter=deg_scatter.view(-1), key_pad=key_pad,
        attn_bias=bd["attn_bias"].float().contiguous(), spatial_pos=bd["spatial_pos"].to(torch.int32).contiguous(),
        img_comment=img_comment, images=None if images is None else images.float().contiguous(),
        label_rows=label_rows, targets=(y.flatten().to(torch.int32) if y is not None else label_rows),
        n_labels=int(label_rows.numel()),
    )
    bd["_packed"] = pb
    return pb
