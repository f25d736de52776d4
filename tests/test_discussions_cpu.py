"""SURVEY.md §8f-3 — the real-data front end (data/discussions.py) against what the reference's dataset builder does
with the same JSON lines.  tests/golden/discussions/expected.json was produced by oracle/gen_golden.py from the
reference's own ``collapse_tree`` / ``get_relative_depth`` / ``spread_downwards`` / ``extract_text`` / ``clean_urls``
(mDT/experiments/hateful_discussions/datasets/hateful_discussions.py) on tests/golden/discussions/sample.jsonl."""
import json
import os

import numpy as np
import pytest

from oracle import structure as S

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "discussions")


@pytest.fixture(scope="module")
def expected():
    return json.load(open(os.path.join(D, "expected.json")))


@pytest.fixture(scope="module")
def raws():
    return [json.loads(l) for l in open(os.path.join(D, "sample.jsonl"))]


def test_flattening_text_and_distances_equal_reference(expected, raws):
    from multimodaldiscussiontransformer_amd.data.discussions import (extract_text, flatten_discussion, has_repeated_ids,
                                                                     updown_with_repeated_ids)
    for raw, exp in zip(raws, expected):
        nodes, parent = flatten_discussion(json.loads(json.dumps(raw)))
        assert [n["id"] for n in nodes] == exp["order"]                        # depth-first order, first occurrence of an id
        assert [extract_text(n["data"]) for n in nodes] == exp["texts"]        # markdown links, stripped URLs, title + body
        assert [n["label"] for n in nodes] == exp["labels"]
        assert [n["images"] for n in nodes] == exp["images"]
        assert parent[0] == -1 and all(0 <= parent[i] < i for i in range(1, len(parent)))
        # the (hops up, hops down) matrix the reference stores per graph is a function of the parent array alone —
        # unless a comment id occurs twice: its id-keyed hop tables then depend on the visiting order (reproduced)
        if has_repeated_ids(raw):
            assert updown_with_repeated_ids(json.loads(json.dumps(raw)), exp["order"]).tolist() == exp["updown"]
            assert S.updown_matrix(parent).tolist() != exp["updown"]
        else:
            assert S.updown_matrix(parent).tolist() == exp["updown"]
            assert updown_with_repeated_ids(json.loads(json.dumps(raw)), exp["order"]).tolist() == exp["updown"]
    # the duplicate rule: "[deleted]" bodies are replaced by a later copy, anything else keeps the first copy
    t2 = expected[1]["texts"]
    assert "the real text of d1" in t2 and "first version stays" in t2 and "second version is ignored" not in t2


def test_label_variants_follow_the_reference_loop(expected):
    from multimodaldiscussiontransformer_amd.data.discussions import label_variants
    labs = expected[1]["labels"]          # ['NA', 'lti_hate', 'NA', 'NA', 'HOM', 'SomethingElse', 'NDG']
    v = label_variants(labs)
    # three labels of a known class -> three graphs; the i-th graph marks the i-th label that is not "NA" — which makes
    # the THIRD graph land on the unknown class (y = 0) and never on 'NDG': reproduced, not repaired (:196-232)
    assert len(v) == 3
    assert [int(np.nonzero(m)[0][0]) for m, _ in v] == [1, 4, 5]
    assert [float(y[0]) for _, y in v] == [1.0, 0.0, 0.0]
    assert label_variants(expected[3]["labels"]) == []                          # no known label: the discussion yields no graph
    v5 = label_variants(expected[4]["labels"])
    assert len(v5) == 3 and float(v5[0][1][0]) == 0.0 and float(v5[1][1][0]) == 1.0


def test_dataset_builds_trees_the_packer_accepts(expected, tmp_path):
    from transformers import BertTokenizerFast
    from multimodaldiscussiontransformer_amd.data.discussions import HatefulDiscussions
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    tok = BertTokenizerFast(vocab_file=os.path.join(D, "vocab.txt"), do_lower_case=True)
    ds = HatefulDiscussions(os.path.join(D, "sample.jsonl"), tokenizer=tok, max_length=24, image_root=D, image_size=32)
    assert len(ds) == 2 + 3 + 1 + 0 + 3
    assert ds.graph_of_line == {0: (0, 2), 1: (2, 3), 2: (5, 1), 3: (6, 0), 4: (6, 3)}
    t = ds[0]
    exp = expected[0]
    assert t["ids"] == exp["order"] and t["input_ids"].shape == (11, 24) and t["input_ids"].dtype == np.int64
    ref = tok(exp["texts"], padding="max_length", truncation=True, max_length=24, return_tensors="np")
    assert np.array_equal(t["input_ids"], ref["input_ids"]) and np.array_equal(t["attention_mask"], ref["attention_mask"])
    assert t["input_ids"][0, 0] == tok.cls_token_id and (t["input_ids"][:, -1] == 0).any()
    assert t["image_index"].tolist() == [len(x) != 0 for x in exp["images"]] and t["images"].shape == (2, 3, 32, 32)
    assert -1.0 <= float(t["images"].min()) and float(t["images"].max()) <= 1.0
    assert t["y_mask"].sum() == 1 and t["y"].tolist() == [1.0]                  # c7 carries DEG: the first labelled comment in order
    assert ds[1]["y"].tolist() == [0.0] and ds[1]["ids"] == exp["order"]        # second graph of the same discussion: 'Neutral'
    # image preprocessing = ViTImageProcessor defaults (resize bilinear, 1/255, mean = std = 0.5)
    from transformers import ViTImageProcessor
    from PIL import Image
    proc = ViTImageProcessor(size={"height": 32, "width": 32})
    want = proc([Image.open(os.path.join(D, "a.png")).convert("RGB")], return_tensors="np")["pixel_values"]
    np.testing.assert_allclose(t["images"][0], want[0], atol=1e-6)
    # trees go straight into the native packer; the structural tensors equal the oracle's collation of the same parents
    assert "updown" in ds[2] and "updown" not in ds[0]                         # only the discussion with a repeated id
    assert ds[2]["updown"].tolist() == expected[1]["updown"]
    trees = [ds[i] for i in (0, 2, 5, 8)]
    pb = pack_batch(trees, 5, device="cpu")
    ref_b = S.collate(trees, 5)
    for k in ("attn_bias", "spatial_pos", "in_degree", "x_token_mask", "x", "x_attention_mask", "x_image_indexes", "y", "y_mask"):
        assert np.array_equal(pb.batched_data[k].numpy(), ref_b[k]), k
    assert pb.M == sum(len(t_["parent"]) for t_ in trees) and pb.n_labels == 4


def test_registered_dataset_feeds_the_task(tmp_path, monkeypatch):
    from transformers import BertTokenizerFast
    from multimodaldiscussiontransformer_amd.data.discussions import register_hateful_discussions
    from multimodaldiscussiontransformer_amd.registry import DATASET_REGISTRY
    from multimodaldiscussiontransformer_amd.tasks import NodePredictionConfig, NodePredictionTask
    (tmp_path / "train-idx.txt").write_text("0\n1\n4\n")
    (tmp_path / "test-idx.txt").write_text("2\n")
    tok = BertTokenizerFast(vocab_file=os.path.join(D, "vocab.txt"), do_lower_case=True)
    register_hateful_discussions("hateful_discussions_sample", json_path=os.path.join(D, "sample.jsonl"),
                                 train_idx_file=str(tmp_path / "train-idx.txt"), test_idx_file=str(tmp_path / "test-idx.txt"),
                                 tokenizer=tok, max_length=16, image_root=D, image_size=32)
    try:
        task = NodePredictionTask.setup_task(NodePredictionConfig(dataset_name="hateful_discussions_sample", spatial_pos_max=5, max_nodes=32))
        assert len(task.load_dataset("train")) == 8 and len(task.load_dataset("valid")) == 1 and len(task.load_dataset("test")) == 1
        ds = task.dataset("train")
        batch = ds.collater([ds[i] for i in range(4)])
        bd = batch["net_input"]["batched_data"]
        assert batch["nsamples"] == 4 and bd["x"].shape[0] == 4 and bd["x"].shape[2] == 16 and int(bd["y_mask"].sum()) == 4
    finally:
        DATASET_REGISTRY.pop("hateful_discussions_sample", None)
