"""Real-data front end: HatefulDiscussions JSON lines → discussion-tree dicts for the packer, without
torch_geometric / networkx.

Replaces the dataset builder of the reference (mDT/experiments/hateful_discussions/datasets/hateful_discussions.py:46-304):
``process`` tokenises every comment with the BERT tokenizer (``padding="max_length"``, ``max_length=100``, :160-166),
runs the ViT image processor on the first image of every comment (:168-184), derives the (hops-up, hops-down) distance
matrix with two recursive passes (``get_relative_depth`` :242-255, ``spread_downwards`` :257-264), flattens the nested
comment tree in depth-first order keeping the first occurrence of an id — or its second one when the first body was
"[deleted]" (``collapse_tree`` :266-298) — and writes one graph per labelled comment (:186-232), each pickled to its own
``graph-<k>.pt`` and ``torch.load``-ed TWICE per access (``get`` :300-304).

Here a discussion is read straight from its JSON line into the arrays the native packer consumes:

    parent          i64[N]   parent index in depth-first order, -1 for the post — the (up, down) matrix, the 21-bucket
                             spatial index, distance and degree are derived from it in C++ (csrc/host.cpp), bit-exact
                             with ``preprocess_item`` on the reference's ``distance_matrix`` (tests/golden/structure_*)
    input_ids / token_type_ids / attention_mask   i64[N, max_length]
    image_index     bool[N], images f32[I, 3, S, S]
    y f32[1], y_mask bool[N]      ONE labelled comment per tree: a discussion with k labels yields k trees (:196-232)

The JSON schema (one discussion per line, as the reference's Pre-Processing scripts write it): a node is
``{"id": str, "data": {...reddit fields..., "label": str}, "images": [paths], "tree": [child nodes]}``; the post
carries ``title`` (+ ``selftext`` or ``body``), comments carry ``body``.
"""
from __future__ import annotations

import json
import os
import re
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np

HATE_LABELS = ("DEG", "lti_hate", "IdentityDirectedAbuse", "AffiliationDirectedAbuse")      # hateful_discussions.py:187-192
GOOD_LABELS = ("Neutral", "lti_normal", "NDG", "HOM")                                        # :193

# a comment that is nothing but one markdown link "[title](url)" keeps its title between [LINK1] / [LINK2]; every other
# URL is removed (:52-66)
_MARKDOWN_LINK = re.compile(r"^\[([\w\s\d]+)\]\(((?:\/|https?:\/\/)[\w\d./?=#]+)\)$")
_ANY_URL = re.compile(r"https?:\/\/(?:www\.)?[-a-zA-Z0-9@:%._\+~#=]{1,256}\.[a-zA-Z0-9()]{1,6}\b(?:[-a-zA-Z0-9()@:%_\+.~#?&\/=]*)")


def clean_urls(text: str) -> str:
    return _ANY_URL.sub("", _MARKDOWN_LINK.sub(r"[LINK1] \g<1> [LINK2]", text))


def extract_text(data: dict) -> str:
    """Text of one node as the reference feeds it to the tokenizer (:68-87)."""
    if "title" in data:
        if "selftext" in data:
            body = "\n" + clean_urls(data["selftext"]) if data["selftext"] != "" else ""
        else:
            body = "\n" + clean_urls(data["body"]) if data["body"] != "NA" else ""
        return data["title"] + body
    return clean_urls(data["body"])


def flatten_discussion(root: dict):
    """Depth-first flattening with the reference's duplicate rule (``collapse_tree``) → (nodes, parent) where
    ``nodes[i]`` = dict(id, data, images, label) and ``parent[i]`` the index of the enclosing comment (-1: the post).
    A repeated id keeps its first position; its record is replaced by the later occurrence only when the bodies differ
    and the stored one is "[deleted]".  Children of a repeated comment hang under the first occurrence."""
    nodes: List[dict] = []
    index = {}
    parent: List[int] = []
    stack = [(root, -1)]
    while stack:
        c, par = stack.pop()
        d = c["data"]
        cid = c["id"]
        d["id"] = cid
        rec = dict(id=cid, data=d, images=list(c.get("images") or []), label=d.get("label", "NA"))
        if cid in index:
            i = index[cid]
            old = nodes[i]["data"]
            if d.get("body") != old.get("body") and old.get("body") == "[deleted]":
                nodes[i] = rec
        else:
            i = len(nodes)
            index[cid] = i
            nodes.append(rec)
            parent.append(par)
        for child in reversed(c.get("tree") or []):
            stack.append((child, i))
    return nodes, np.asarray(parent, dtype=np.int64)


def has_repeated_ids(root: dict) -> bool:
    seen, stack = set(), [root]
    while stack:
        c = stack.pop()
        if c["id"] in seen:
            return True
        seen.add(c["id"])
        stack.extend(c.get("tree") or [])
    return False


def updown_with_repeated_ids(root: dict, order: Sequence) -> np.ndarray:
    """(hops up, hops down) matrix of a discussion in which a comment id occurs more than once, exactly as the
    reference's two passes produce it (hateful_discussions.py:242-264): hop tables are keyed by comment ID, so the
    occurrences of an id share one entry and the table a node ends up with depends on the visiting order — it is not
    the matrix of any tree.  Well-formed discussions never come here (their matrix follows from the parent array).

    Pass 1 (down the tree, children in order): a node starts from its parent's table as it stands when the node is
    reached — ancestors and the subtrees of earlier siblings — with every "up" count raised by one, sets itself to
    (0, 0), then takes over each child's finished table, raising "down" by one for ids it does not know yet.
    Pass 2: whatever the parent knows and the node does not (subtrees of later siblings) is added with "up" raised."""
    tables = {}

    def first_pass(node, inherited):
        table = {k: [v[0] + 1, v[1]] for k, v in inherited.items()}
        table[node["id"]] = [0, 0]
        for child in node.get("tree") or []:
            for k, v in first_pass(child, table).items():
                if k not in table:
                    table[k] = [v[0], v[1] + 1]
        tables[id(node)] = table
        return {k: list(v) for k, v in table.items()}

    def second_pass(node, from_parent):
        table = tables[id(node)]
        for k, v in from_parent.items():
            if k not in table:
                table[k] = [v[0] + 1, v[1]]
        for child in node.get("tree") or []:
            second_pass(child, {k: list(v) for k, v in table.items()})

    first_pass(root, {})
    second_pass(root, {})
    # the record kept for an id (collapse_tree): its first occurrence, or the later one that replaced a "[deleted]" body
    kept, stack = {}, [root]
    while stack:
        c = stack.pop()
        cid = c["id"]
        if cid not in kept:
            kept[cid] = c
        elif c["data"].get("body") != kept[cid]["data"].get("body") and kept[cid]["data"].get("body") == "[deleted]":
            kept[cid] = c
        for child in reversed(c.get("tree") or []):
            stack.append(child)
    n = len(order)
    ud = np.zeros((n, n, 2), dtype=np.int64)
    for a, ka in enumerate(order):
        t = tables[id(kept[ka])]
        for b, kb in enumerate(order):
            ud[a, b] = t[kb]
    return ud


def label_variants(labels: Sequence[str]):
    """One (y_mask, y) per labelled comment (:196-232): the i-th variant marks the i-th comment whose label is not "NA";
    there are as many variants as labels of a KNOWN class, y = 1 for a hate label, 0 otherwise."""
    n_true = sum(1 for x in labels if x in HATE_LABELS or x in GOOD_LABELS)
    out = []
    for i in range(n_true):
        z = 0
        hit = None
        for k, lab in enumerate(labels):
            if lab != "NA":
                if z == i:
                    hit = k
                    break
                z += 1
        if hit is None:
            continue                                   # the reference prints "missing label!!" and skips
        mask = np.zeros(len(labels), dtype=bool)
        mask[hit] = True
        out.append((mask, np.asarray([1.0 if labels[hit] in HATE_LABELS else 0.0], dtype=np.float32)))
    return out


def vit_pixel_values(paths: Sequence[str], size: int = 224) -> np.ndarray:
    """ViTImageProcessor defaults of "google/vit-base-patch16-224" (:47-49, :175-180) on the host: RGB, PIL bilinear resize to
    size x size, then the processor's arithmetic — x * (1 / 255) in double rounded once to float, (x - 0.5) / 0.5 in float —
    through the 3 x 256 table of its possible results (``mdt_image_norm_lut``; ``a * float32(1 / 255)`` is one ulp off for
    some bytes) → f32[n, 3, size, size], bit-identical to the processor and to the device front end."""
    from PIL import Image
    from .. import ops
    lut = ops.image_norm_lut().reshape(3, 256)
    out = np.empty((len(paths), 3, size, size), dtype=np.float32)
    for i, p in enumerate(paths):
        a = np.asarray(Image.open(p).convert(mode="RGB").resize((size, size), resample=Image.BILINEAR))
        for c in range(3):
            out[i, c] = lut[c][a[:, :, c]]
    return out


def decoded_rgb(paths: Sequence[str]) -> list:
    """``Image.open(p).convert(mode="RGB")`` (:175-177) as uint8 [H, W, 3] arrays at the images' OWN sizes: what the device front
    end takes (``image_preprocess="device"``) — resize, rescale and normalise then run as ``mdt_image_preprocess`` on the GPU,
    and the pixels cross PCIe as bytes."""
    from PIL import Image
    return [np.asarray(Image.open(p).convert(mode="RGB")) for p in paths]


def default_tokenizer(name_or_vocab: str = "bert-base-uncased"):
    """The reference's ``AutoTokenizer.from_pretrained("bert-base-uncased")`` (:46).  There is no network here: a local
    cache entry, a directory or a ``vocab.txt`` path must exist."""
    if os.path.isfile(name_or_vocab):
        from transformers import BertTokenizerFast
        return BertTokenizerFast(vocab_file=name_or_vocab, do_lower_case=True)
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(name_or_vocab, local_files_only=True)


class HatefulDiscussions:
    """Lazy, random-access list of trees over a JSON-lines file (line offsets are indexed once; a tree is built when it
    is asked for).  ``indices``: which discussions (line numbers) to use — the reference keeps the lines listed in
    train-idx.txt / test-idx.txt (:93-100)."""

    def __init__(self, json_path: str, tokenizer: Optional[Callable] = None, max_length: int = 100, image_root: str = "",
                 image_size: int = 224, image_loader: Optional[Callable] = None, indices: Optional[Iterable[int]] = None,
                 image_preprocess: str = "host"):
        """``image_preprocess``: "host" — PIL resize + numpy normalise per image on the CPU, fp32 pixels in the tree (the
        reference's way); "device" — the tree carries the decoded bytes (``images_u8``) and the packer hands them to
        ``mdt_image_preprocess`` (bit-identical pixel values, a quarter or less of the PCIe bytes)."""
        if image_preprocess not in ("host", "device"):
            raise ValueError(f"image_preprocess = {image_preprocess!r} (host | device)")
        self.path = json_path
        self.tok = tokenizer if tokenizer is not None else default_tokenizer()
        self.max_length, self.image_root, self.image_size = max_length, image_root, image_size
        self.image_preprocess = image_preprocess
        self.load_images = image_loader or ((lambda paths: vit_pixel_values(paths, image_size)) if image_preprocess == "host" else decoded_rgb)
        offs, off = [], 0
        with open(json_path, "rb") as f:
            for line in f:
                offs.append(off)
                off += len(line)
        keep = set(int(i) for i in indices) if indices is not None else None
        # (line, variant) pairs: one entry per labelled comment, found with a cheap pass over the labels only
        self.entries = []
        self.graph_of_line = {}
        with open(json_path, "rb") as f:
            for ln, o in enumerate(offs):
                if keep is not None and ln not in keep:
                    continue
                f.seek(o)
                nodes, _ = flatten_discussion(json.loads(f.readline()))
                nv = len(label_variants([n["label"] for n in nodes]))
                self.graph_of_line[ln] = (len(self.entries), nv)
                self.entries += [(o, v) for v in range(nv)]

    def __len__(self):
        return len(self.entries)

    def build(self, raw: dict, variant: int) -> dict:
        nodes, parent = flatten_discussion(raw)
        texts = [extract_text(n["data"]) for n in nodes]
        enc = self.tok(texts, padding="max_length", truncation=True, max_length=self.max_length, return_tensors="np")
        ids = np.asarray(enc["input_ids"], dtype=np.int64)
        tt = np.asarray(enc["token_type_ids"], dtype=np.int64) if "token_type_ids" in enc else np.zeros_like(ids)
        am = np.asarray(enc["attention_mask"], dtype=np.int64)
        has_img = np.asarray([len(n["images"]) != 0 for n in nodes], dtype=bool)
        paths = [os.path.join(self.image_root, n["images"][0]) for n in nodes if n["images"]]
        images = self.load_images(paths) if paths else None
        mask, y = label_variants([n["label"] for n in nodes])[variant]
        tree = dict(parent=parent, input_ids=ids, token_type_ids=tt, attention_mask=am, image_index=has_img, images=images,
                    y=y, y_mask=mask, ids=[n["id"] for n in nodes])
        if self.image_preprocess == "device":
            tree["images"], tree["images_u8"], tree["image_size"] = None, (images or []), self.image_size
        if has_repeated_ids(raw):        # rare: the reference's id-keyed hop tables are then not those of the tree
            tree["updown"] = updown_with_repeated_ids(raw, tree["ids"])
        return tree

    def __getitem__(self, i: int) -> dict:
        off, variant = self.entries[int(i)]
        with open(self.path, "rb") as f:
            f.seek(off)
            raw = json.loads(f.readline())
        return self.build(raw, variant)


def register_hateful_discussions(name: str = "hateful_discussions", json_path: Optional[str] = None, train_idx_file: Optional[str] = None,
                                 test_idx_file: Optional[str] = None, **reader_kw):
    """The dataset-registry entry of the reference (mDT/experiments/hateful_discussions/datasets/dataset.py:7-28):
    ``$SLURM_TMPDIR/pruned-with-images-fixed-big.json`` with train-idx.txt / test-idx.txt beside it; validation and test
    splits are both the test list, as there."""
    from ..registry import register_dataset

    @register_dataset(name)
    def create():
        root = os.path.expandvars("$SLURM_TMPDIR")
        jp = json_path or os.path.join(root, "pruned-with-images-fixed-big.json")
        tr = [int(x) for x in open(train_idx_file or os.path.join(root, "train-idx.txt")) if x.strip()]
        te = [int(x) for x in open(test_idx_file or os.path.join(root, "test-idx.txt")) if x.strip()]
        ds = HatefulDiscussions(jp, indices=sorted(set(tr) | set(te)), image_root=reader_kw.pop("image_root", root), **reader_kw)

        def graphs(lines):
            out = []
            for ln in lines:
                first, n = ds.graph_of_line.get(ln, (0, 0))
                out += list(range(first, first + n))
            return np.asarray(out, dtype=np.int64)

        return {"dataset": ds, "train_idx": graphs(tr), "valid_idx": graphs(te), "test_idx": graphs(te), "source": "pyg"}

    return create
