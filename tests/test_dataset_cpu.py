"""Task.load_dataset / the dataset wrappers (mDT/src/tasks/task.py:116-204, mDT/src/data/dataset.py) on the host: a
``--user-data-dir`` module registers a dataset, the task builds the three splits, ``collater`` batches with the native
packer (bit-exact integer tensors, checked against the oracle), and the CSR view survives a generic
move-every-tensor-of-the-sample step such as FairSeq's ``move_to_cuda``."""
import os
import sys
import textwrap

import numpy as np
import torch

from multimodaldiscussiontransformer_amd import synthetic
from oracle import structure as S


def _apply(f, x):
    if torch.is_tensor(x):
        return f(x)
    if isinstance(x, dict):
        return {k: _apply(f, v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_apply(f, v) for v in x)
    return x


def test_user_data_dir_dataset_to_batches(tmp_path):
    from multimodaldiscussiontransformer_amd.data.packer import PackedBatch, packed_from_batched_data
    from multimodaldiscussiontransformer_amd.tasks import NodePredictionConfig, NodePredictionTask
    pkg = tmp_path / "my_datasets"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "toy.py").write_text(textwrap.dedent("""
        import numpy as np
        from multimodaldiscussiontransformer_amd.data import register_dataset
        from multimodaldiscussiontransformer_amd import synthetic

        @register_dataset("toy_discussions")
        def create():
            trees = synthetic.make_trees(20, 9, seed=5, variable=True, seq_len=12, vocab_size=400, image_frac=0.3, image_size=32, min_len=2)
            return {"dataset": trees, "train_idx": np.arange(0, 14), "valid_idx": np.arange(14, 17), "test_idx": np.arange(17, 20), "source": "pyg"}
    """))
    cfg = NodePredictionConfig(dataset_name="toy_discussions", user_data_dir=str(pkg), spatial_pos_max=5, max_nodes=32, seed=3)
    task = NodePredictionTask.setup_task(cfg)
    ds = task.load_dataset("train")
    assert len(ds) == 14 and len(task.load_dataset("valid")) == 3 and len(task.load_dataset("test")) == 3
    assert task.dataset("train") is ds and ds.size(0) == 32 and list(ds.ordered_indices()) == list(range(14))
    # the split is the registered index list shuffled with RandomState(seed) (pyg_dataset.py:52-56)
    want = np.arange(0, 14)
    np.random.RandomState(3).shuffle(want)
    all_trees = synthetic.make_trees(20, 9, seed=5, variable=True, seq_len=12, vocab_size=400, image_frac=0.3, image_size=32, min_len=2)
    for i in range(14):
        assert np.array_equal(ds[i]["parent"], all_trees[want[i]]["parent"]) and ds[i]["idx"] == i
    samples = [ds[i] for i in (0, 3, 5, 6)]
    batch = ds.collater(samples)
    assert batch["nsamples"] == 4 and set(batch["net_input"]) == {"batched_data"}
    bd = batch["net_input"]["batched_data"]
    ref = S.collate([all_trees[want[i]] for i in (0, 3, 5, 6)], 5)
    for k in ("attn_bias", "spatial_pos", "in_degree", "out_degree", "x_token_mask", "x", "x_token_type_ids", "x_attention_mask",
              "x_image_indexes", "y", "y_mask"):
        assert np.array_equal(bd[k].numpy(), ref[k]), k
    assert (bd["x_images"] is None) == (ref["x_images"] is None)
    if ref["x_images"] is not None:
        assert np.array_equal(bd["x_images"].numpy(), ref["x_images"])
    assert bd["in_degree"] is bd["out_degree"]                                        # collator.py:171
    # a generic tensor-by-tensor move (fairseq.utils.move_to_cuda does exactly this) keeps the CSR view usable
    pb0 = bd["_packed"]
    moved = _apply(lambda t: t.clone(), {k: v for k, v in bd.items() if k != "_packed"})
    pb1 = packed_from_batched_data(moved)
    assert isinstance(pb1, PackedBatch) and pb1 is not pb0
    for f in ("ids", "types", "text_mask", "node_row", "graph_row", "degree", "deg_scatter", "key_pad", "img_comment", "label_rows", "targets"):
        assert torch.equal(getattr(pb1, f), getattr(pb0, f)), f
    assert (pb1.B, pb1.N, pb1.M, pb1.I, pb1.L, pb1.n_labels) == (pb0.B, pb0.N, pb0.M, pb0.I, pb0.L, pb0.n_labels)
    assert torch.equal(pb1.ragged.offsets, pb0.ragged.offsets) and pb1.ragged.rows == pb0.ragged.rows
    # the compatibility path (a dict straight from ``collator``, no CSR view) derives the same indices
    bare = {k: v for k, v in moved.items() if k != "_csr"}
    pb2 = packed_from_batched_data(bare)
    for f in ("ids", "node_row", "graph_row", "deg_scatter", "key_pad", "img_comment", "label_rows"):
        assert torch.equal(getattr(pb2, f), getattr(pb0, f)), f
    sys.path.remove(str(tmp_path)) if str(tmp_path) in sys.path else None


def test_epoch_shuffle_and_default_split():
    from multimodaldiscussiontransformer_amd.data.dataset import (DiscussionDataset, EpochShuffleDataset, GraphormerDataset,
                                                                   NodeBatchedDataDataset, SampleEnvelopeDataset)
    trees = synthetic.make_trees(30, 5, seed=9, seq_len=8, vocab_size=300, min_len=2)
    dm = GraphormerDataset(dataset=DiscussionDataset(trees), dataset_source="pyg", seed=1)       # no index lists: sklearn split
    n = (len(dm.dataset_train), len(dm.dataset_val), len(dm.dataset_test))
    assert sum(n) == 30 and n == (24, 3, 3)                                                      # test_size n // 5, then n // 10
    env = SampleEnvelopeDataset(NodeBatchedDataDataset(dm.dataset_train, spatial_pos_max=5), np.full(24, 8))
    sh = EpochShuffleDataset(env, num_samples=24, seed=7)
    o1 = sh.ordered_indices().copy()
    sh.set_epoch(2)
    o2 = sh.ordered_indices().copy()
    sh.set_epoch(1)
    assert sorted(o1) == list(range(24)) and not np.array_equal(o1, o2) and np.array_equal(o1, sh.ordered_indices())
    assert not sh.can_reuse_epoch_itr_across_epochs


def test_contrastive_collater_carries_hard_y():
    from multimodaldiscussiontransformer_amd.data.dataset import ContrastiveBatchedDataDataset, DiscussionDataset
    from oracle import cases
    hp = cases.tiny_hparams("A")
    trees = cases.contrastive_trees(hp)
    bd = ContrastiveBatchedDataDataset(DiscussionDataset(trees), spatial_pos_max=5).collater(trees)
    ref = S.collate(trees, 5)
    assert np.array_equal(bd["y"].numpy(), ref["y"]) and np.array_equal(bd["hard_y"].numpy(), ref["hard_y"])
    assert "y_mask" not in bd and bd["y"].shape[0] == len(trees)
