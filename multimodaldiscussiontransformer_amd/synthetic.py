"""Synthetic discussion trees: the inputs of every bench / parity run.

A *tree* is a dict of numpy arrays:
    parent          i64[N]      parent index, -1 for the root (post)
    input_ids       i64[N, L]   BERT token ids, 0-padded; position 0 = [CLS] (101)
    token_type_ids  i64[N, L]
    attention_mask  i64[N, L]
    image_index     bool[N]     comment carries an image
    images          f32[I, 3, H, W] or None
    y               f32[#labelled]   1 = hateful
    y_mask          bool[N]     which comments are labelled

Shapes mirror what the reference dataset builder stores per graph
(mDT/experiments/hateful_discussions/datasets/hateful_discussions.py:148-232):
one labelled comment per graph, text padded to max_length, ViT pixel_values per image.
"""
from __future__ import annotations

import numpy as np


def bushy_parents(n: int, fanout: int = 3) -> np.ndarray:
    """Heap-ordered tree: parent(k) = (k-1)//fanout."""
    p = (np.arange(n, dtype=np.int64) - 1) // fanout
    p[0] = -1
    return p


def deep_thread_parents(n: int, rng: np.random.Generator, p_chain: float = 0.8) -> np.ndarray:
    """Mostly a chain: parent(k) = k-1 w.p. ``p_chain`` else a uniform earlier comment."""
    p = np.full(n, -1, dtype=np.int64)
    for k in range(1, n):
        p[k] = k - 1 if rng.random() < p_chain else int(rng.integers(0, k))
    return p


def make_tree(
    n_nodes: int,
    rng: np.random.Generator,
    *,
    seq_len: int = 100,
    vocab_size: int = 30522,
    image_frac: float = 0.0,
    image_size: int = 224,
    shape: str = "bushy",
    min_len: int = 8,
    image_pool: "np.ndarray | None" = None,
) -> dict:
    parent = bushy_parents(n_nodes) if shape == "bushy" else deep_thread_parents(n_nodes, rng)
    lo = min(1000, max(1, vocab_size // 8))
    ids = rng.integers(lo, vocab_size, size=(n_nodes, seq_len), dtype=np.int64)
    ids[:, 0] = min(101, vocab_size - 1)
    lens = rng.integers(min(min_len, seq_len), seq_len + 1, size=n_nodes)
    pos = np.arange(seq_len)[None, :]
    am = (pos < lens[:, None]).astype(np.int64)
    ids = ids * am
    n_img = int(round(image_frac * n_nodes))
    image_index = np.zeros(n_nodes, dtype=bool)
    images = None
    if n_img > 0:
        image_index[rng.choice(n_nodes, size=n_img, replace=False)] = True
        if image_pool is None:
            images = rng.standard_normal((n_img, 3, image_size, image_size), dtype=np.float32)
        else:       # a window of a shared pool of random images at a random offset: a VIEW (no host copy), different per tree
            o = int(rng.integers(0, image_pool.shape[0] - n_img + 1))
            images = image_pool[o:o + n_img]
    y_mask = np.zeros(n_nodes, dtype=bool)
    y_mask[int(rng.integers(0, n_nodes))] = True
    y = np.asarray([1.0 if rng.random() < 0.3 else 0.0], dtype=np.float32)
    return dict(
        parent=parent,
        input_ids=ids,
        token_type_ids=np.zeros_like(ids),
        attention_mask=am,
        image_index=image_index,
        images=images,
        y=y,
        y_mask=y_mask,
    )


def make_trees(
    n_trees: int,
    n_nodes: int,
    *,
    seed: int = 1234,
    variable: bool = False,
    **kw,
) -> list:
    """``n_trees`` trees of ``n_nodes`` comments (U{n/2..n} when ``variable``).

    seed convention: 1234 + rank*1000 + step.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n_trees):
        n = int(rng.integers(max(1, n_nodes // 2), n_nodes + 1)) if variable else n_nodes
        out.append(make_tree(n, rng, **kw))
    return out
