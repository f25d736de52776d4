"""Shared definitions of the golden cases (inputs are regenerated, never stored).
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
from __future__ import annotations

import os

import numpy as np

from multimodaldiscussiontransformer_amd import synthetic
from . import mdt_ref_cpu as R


def structure_specs():
    rng = np.random.Generator(np.random.PCG64(7))
    specs = [
        ("bushy16", [synthetic.make_tree(16, rng, seq_len=12, vocab_size=500)]),
        ("single", [synthetic.make_tree(1, rng, seq_len=12, vocab_size=500)]),
        ("deep40", [synthetic.make_tree(40, rng, seq_len=12, vocab_size=500, shape="deep")]),
        ("ragged_img", [synthetic.make_tree(n, rng, seq_len=12, vocab_size=500, image_frac=f, image_size=32,
                                            shape=s)
                        for n, f, s in ((9, 0.34, "bushy"), (3, 0.0, "deep"), (14, 1.0, "deep"), (6, 0.0, "bushy"))]),
    ]
    return specs


def tiny_hparams(kind):
    if kind == "A":      # text + image, stacks 1/1
        return R.hparams(dim=768, enc_heads=12, graph_heads=12, enc_ffn=256, graph_ffn=768, text_layers=4,
                         vit_layers=4, num_fusion_layers=1, num_fusion_stack=1, num_graph_stack=1,
                         num_bottleneck=4, vocab_size=600, max_pos=64, image_size=32, patch=16,
                         pos_weight=1.5, neg_weight=1.0)
    if kind == "B":      # text only, stacks 2/2, uneven last fusion stack
        return R.hparams(dim=768, enc_heads=12, graph_heads=12, enc_ffn=192, graph_ffn=768, text_layers=4,
                         vit_layers=4, num_fusion_layers=2, num_fusion_stack=2, num_graph_stack=2,
                         num_bottleneck=4, vocab_size=600, max_pos=64, image_size=32, patch=16,
                         pos_weight=1.5, neg_weight=1.0)
    raise KeyError(kind)


def tiny_trees(kind, hp):
    rng = np.random.Generator(np.random.PCG64(99))
    if kind == "A":
        spec = ((8, 0.25, "bushy"), (5, 0.0, "deep"), (3, 0.34, "bushy"))
    else:
        spec = ((7, 0.0, "bushy"), (8, 0.0, "deep"), (2, 0.0, "bushy"), (1, 0.0, "bushy"))
    trees = [synthetic.make_tree(n, rng, seq_len=16, vocab_size=hp.vocab_size, image_frac=f,
                                 image_size=hp.image_size, shape=s, min_len=3) for n, f, s in spec]
    # more than one labelled comment per tree, both classes present
    for i, t in enumerate(trees):
        n = len(t["parent"])
        t["y_mask"][:] = False
        lab = list(range(0, n, 3))
        t["y_mask"][lab] = True
        t["y"] = np.asarray([(i + k) % 2 for k in range(len(lab))], dtype=np.float32)
    return trees




# ----------------------------------------------------------------------------- real-geometry cases
# "C2": BASELINE.json configs[1] at its TRUE geometry — BERT-base + ViT-B/16 split 6 + 6, 6 executed graph layers,
#       FFN 3072, L = 100 tokens (S = 104 with the bottleneck tokens), 224-px images (P = 197, S = 201), one bushy
#       64-comment tree (T = 65) with 25 % image comments.  The reference itself runs this case (D = 768).
# "C4": BASELINE.json configs[3] mDT-large SHAPES — D 1024, 16 heads, FFN 4096, ViT-L/14 (P = 257, S = 261), one
#       128-comment deep-thread tree (T = 129, banded -inf mask at spatial_pos_max = 5); the layer count is cut to
#       2 + 2 (1 executed graph layer pair) so that the CPU oracle finishes in seconds.  The reference cannot run it
#       (768 is a literal there): oracle only.
# "M":  the tiny "A" shapes with many labelled comments and a classifier bias that centres the logit margins, so that
#       predictions are MIXED (TP, FP, FN, TN all non-zero) and a sign flip in one logit column cannot pass.
def real_hparams(kind):
    if kind == "C2":
        return R.hparams(dim=768, enc_heads=12, graph_heads=12, enc_ffn=3072, graph_ffn=768, text_layers=12,
                         vit_layers=12, num_fusion_layers=5, num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4,
                         vocab_size=30522, max_pos=512, image_size=224, patch=16, pos_weight=1.5, neg_weight=1.0)
    if kind == "C4":
        return R.hparams(dim=1024, enc_heads=16, graph_heads=16, enc_ffn=4096, graph_ffn=1024, text_layers=4,
                         vit_layers=4, num_fusion_layers=1, num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4,
                         vocab_size=30522, max_pos=512, image_size=224, patch=14, pos_weight=1.5, neg_weight=1.0)
    if kind == "C4F":    # configs[3] at its FULL depth: BERT-large / ViT-L/14 split 12 + 12, 12 executed graph layers
        return R.hparams(dim=1024, enc_heads=16, graph_heads=16, enc_ffn=4096, graph_ffn=1024, text_layers=24,
                         vit_layers=24, num_fusion_layers=11, num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4,
                         vocab_size=30522, max_pos=512, image_size=224, patch=14, pos_weight=1.5, neg_weight=1.0)
    if kind == "C1":     # configs[0] exactly: Tiny mDT — 128-d, BERT-mini 2 + 2 (2 heads, FFN 512), 2 executed graph layers
        return R.hparams(dim=128, enc_heads=2, graph_heads=8, enc_ffn=512, graph_ffn=128, text_layers=4,
                         vit_layers=4, num_fusion_layers=1, num_fusion_stack=1, num_graph_stack=1, num_bottleneck=4,
                         vocab_size=30522, max_pos=512, image_size=32, patch=16, pos_weight=1.5, neg_weight=1.0)
    if kind == "LAUNCH":  # the configuration the reference SHIPS: sample_run.sh:3 = `run_train.sh 8 4 5 2 2 0` — 8 fusion layers
        # (BERT-base / ViT-B/16 split 3 + 9), 4 bottleneck tokens, spatial_pos_max 5, graph stacks of 2 (10 executed graph layers),
        # fusion stacks of 2 (uneven last stack: 9 = 2+2+2+2+1), graph FFN 768 (run_train.sh:57), --freeze_initial_encoders (:61)
        return R.hparams(dim=768, enc_heads=12, graph_heads=12, enc_ffn=3072, graph_ffn=768, text_layers=12, vit_layers=12,
                         num_fusion_layers=8, num_fusion_stack=2, num_graph_stack=2, num_bottleneck=4, vocab_size=30522,
                         max_pos=512, image_size=224, patch=16, pos_weight=1.5, neg_weight=1.0, freeze_initial_encoders=True)
    if kind == "M":
        return tiny_hparams("A")
    raise KeyError(kind)


def _label_many(trees, every=2):
    for i, t in enumerate(trees):
        n = len(t["parent"])
        t["y_mask"][:] = False
        lab = list(range(i % every, n, every))
        t["y_mask"][lab] = True
        t["y"] = np.asarray([((k * 7 + i) % 3 == 0) for k in range(len(lab))], dtype=np.float32)
    return trees


def real_trees(kind, hp):
    if kind == "C2":
        rng = np.random.Generator(np.random.PCG64(2024))
        trees = [synthetic.make_tree(64, rng, seq_len=100, vocab_size=hp.vocab_size, image_frac=0.25,
                                     image_size=hp.image_size, shape="bushy", min_len=8)]
        return _label_many(trees, every=2)
    if kind == "C4":
        rng = np.random.Generator(np.random.PCG64(4096))
        trees = [synthetic.make_tree(128, rng, seq_len=100, vocab_size=hp.vocab_size, image_frac=0.125,
                                     image_size=hp.image_size, shape="deep", min_len=8)]
        return _label_many(trees, every=2)
    if kind == "C4F":    # one 128-comment deep thread; comments of at most 48 tokens and 8 image comments keep the fp32 CPU oracle's
        rng = np.random.Generator(np.random.PCG64(4097))     # autograd graph (24 + 24 blocks at D 1024) near 10 GB
        trees = [synthetic.make_tree(128, rng, seq_len=48, vocab_size=hp.vocab_size, image_frac=0.0625,
                                     image_size=hp.image_size, shape="deep", min_len=8)]
        return _label_many(trees, every=2)
    if kind == "C1":     # configs[0]'s workload: 8 bushy 16-comment trees, text only, L = 100
        rng = np.random.Generator(np.random.PCG64(128))
        trees = [synthetic.make_tree(16, rng, seq_len=100, vocab_size=hp.vocab_size, image_frac=0.0,
                                     image_size=hp.image_size, shape="bushy", min_len=8) for _ in range(8)]
        return _label_many(trees, every=2)
    if kind == "LAUNCH":  # two trees of the launch's 12-tree batch (what the CPU reference runs in about a minute): L = 100, 224-px images
        rng = np.random.Generator(np.random.PCG64(8452))
        trees = [synthetic.make_tree(n, rng, seq_len=100, vocab_size=hp.vocab_size, image_frac=f, image_size=hp.image_size,
                                     shape=s, min_len=8) for n, f, s in ((13, 0.25, "bushy"), (9, 0.34, "deep"))]
        return _label_many(trees, every=1)
    if kind == "M":
        rng = np.random.Generator(np.random.PCG64(515))
        spec = ((8, 0.25, "bushy"), (7, 0.0, "deep"), (5, 0.4, "bushy"), (4, 0.0, "deep"))
        trees = [synthetic.make_tree(n, rng, seq_len=16, vocab_size=hp.vocab_size, image_frac=f, image_size=hp.image_size,
                                     shape=s, min_len=3) for n, f, s in spec]
        return _label_many(trees, every=1)
    raise KeyError(kind)


# node_classifier.bias per case: minus / plus half the median logit margin of the hash-weight model (measured once with
# the oracle, tools/margin_probe.py), so that about half of the comments are predicted positive.  Applied identically
# to the reference (gen_golden), the oracle (make_weights) and the product (tests.util_model.fill_hash_weights).
_BIAS_SHIFT = {"C2": 0.68, "C4": 0.0874, "M": 1.0912, "C1": -0.2122039, "LAUNCH": 0.229}   # C1: the midpoint of the widest gap (1.3e-3) between the sorted zero-bias margins around their median: no labelled comment within 6e-4 of a tie


def weight_overrides(kind):
    s = _BIAS_SHIFT.get(kind)
    if s is None:
        return {}
    return {"node_classifier.bias": np.asarray([s / 2.0, -s / 2.0], dtype=np.float32)}


# ----------------------------------------------------------------------------- contrastive cases
def contrastive_labels(n_trees, kind="A"):
    """(y, hard_y) community labels per tree.  Community 1 exists so that ``total_positive`` and the reference's
    ``pred == targets`` counters are non-trivial; every tree has at least one soft-negative partner (no 0 / 0)."""
    if kind == "A":          # 12 trees, 4 communities
        y = np.asarray([0, 1, 2, 3, 0, 1, 2, 3, 1, 0, 3, 2][:n_trees], dtype=np.float32)
        hard = np.asarray([1, 0, 3, 2, 1, 0, 3, 2, 0, 1, 2, 3][:n_trees], dtype=np.float32)
    else:                    # 5 trees, 3 communities (full-model case)
        y = np.asarray([1, 0, 2, 1, 0][:n_trees], dtype=np.float32)
        hard = np.asarray([0, 1, 0, 2, 2][:n_trees], dtype=np.float32)
    return y, hard


def contrastive_trees(hp):
    """5 small trees (tiny "A" shapes, one with images) labelled per TREE for the contrastive task."""
    rng = np.random.Generator(np.random.PCG64(808))
    spec = ((6, 0.34, "bushy"), (4, 0.0, "deep"), (7, 0.0, "bushy"), (3, 0.34, "deep"), (5, 0.0, "bushy"))
    trees = [synthetic.make_tree(n, rng, seq_len=16, vocab_size=hp.vocab_size, image_frac=f, image_size=hp.image_size,
                                 shape=s, min_len=3) for n, f, s in spec]
    y, hard = contrastive_labels(len(trees), "B")
    for i, t in enumerate(trees):
        t.pop("y_mask")
        t["y"] = np.asarray([y[i]], dtype=np.float32)
        t["hard_y"] = np.asarray([hard[i]], dtype=np.float32)
    return trees


# ----------------------------------------------------------------------------- image front end
def pixel_value_inputs(sample_dir):
    """Decoded RGB images of the pixel-value fixture: the three sample PNGs of tests/golden/discussions plus seeded synthetic
    images at sizes that make PIL's resize enlarge, reduce (antialiasing: up to 11 taps) and skip an axis."""
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(224))
    imgs = [np.asarray(Image.open(os.path.join(sample_dir, n + ".png")).convert(mode="RGB")) for n in "abc"]
    for (H, W) in ((300, 500), (731, 1000), (224, 224), (97, 1023), (224, 301)):
        base = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2, 3)).astype(np.float64)      # blocky structure + noise
        up = np.kron(base, np.ones((8, 8, 1)))[:H, :W]
        imgs.append(np.clip(up + rng.normal(0, 20, (H, W, 3)), 0, 255).astype(np.uint8))
    return imgs
