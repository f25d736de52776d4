"""``collator(items, spatial_pos_max)`` with the reference's call signature and output dict
(mDT/src/data/collator.py:69-179) for callers that already hold per-tree tensors
(``idx, attn_bias, spatial_pos, in_degree, x, x_image_index, x_images, distance, y``).

Every output is allocated once at its padded size and filled by slice assignment (the
reference concatenates per-tree padded copies); values, dtypes and padding conventions are
identical: +1 shifts with 0 padding for ``spatial_pos`` / ``in_degree``, −inf pad columns and
0 pad rows for ``attn_bias``, distance clipping on the node×node block only, unpadded
``x_image_indexes``.  The fused training path uses ``packer.pack_batch`` instead, which
starts from parent arrays and also emits the CSR index vectors.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch


def collator(items: Sequence[Sequence], spatial_pos_max: int = 10) -> Dict[str, torch.Tensor]:
    B = len(items)
    sizes = [it[4]["input_ids"].size(0) for it in items]
    N = max(sizes)
    T = N + 1
    Lq = items[0][4]["input_ids"].size(1)
    ab0 = items[0][1]
    attn_bias = ab0.new_full((B, T, T), float("-inf"))
    spatial_pos = torch.zeros(B, N, N, dtype=torch.int32)
    in_degree = torch.zeros(B, N, dtype=items[0][3].dtype)
    text = {k: torch.zeros(B, N, Lq, dtype=items[0][4][k].dtype) for k in ("input_ids", "token_type_ids", "attention_mask")}
    images: List[torch.Tensor] = []
    img_index: List[torch.Tensor] = []
    ys: List[torch.Tensor] = []
    for b, (idx, ab, sp, deg, x, x_img_idx, x_img, dist, y) in enumerate(items):
        n = sizes[b]
        blk = ab.clone()
        blk[1:, 1:][dist >= spatial_pos_max] = float("-inf")
        attn_bias[b, : n + 1, : n + 1] = blk
        attn_bias[b, n + 1:, : n + 1] = 0
        spatial_pos[b, :n, :n] = (sp + 1).to(torch.int32)
        in_degree[b, :n] = deg + 1
        for k in text:
            text[k][b, :n] = x[k]
        if not bool(torch.all(x_img.eq(0))):          # placeholder images are all-zero (collator.py:144)
            images.append(x_img)
        img_index.append(x_img_idx.reshape(-1))
        ys.append(y)
    return dict(
        idx=torch.LongTensor([it[0] for it in items]),
        attn_bias=attn_bias,
        spatial_pos=spatial_pos,
        in_degree=in_degree,
        out_degree=in_degree,
        x_token_mask=~text["input_ids"].eq(0).all(dim=2),
        x=text["input_ids"],
        x_token_type_ids=text["token_type_ids"],
        x_attention_mask=text["attention_mask"],
        x_images=torch.cat(images) if images else None,
        x_image_indexes=torch.cat(img_index).bool(),
        y=torch.cat(ys),
    )
