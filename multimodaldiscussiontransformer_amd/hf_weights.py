"""Pretrained HuggingFace weights for the BERT / ViT stacks.

The reference builds its encoders with ``AutoModelForSequenceClassification.from_pretrained("bert-base-uncased")`` and
``AutoModel.from_pretrained("google/vit-base-patch16-224")`` and then moves the last ``num_fusion_layers + 1`` blocks
of each into the fusion stacks (mDT/src/modules/multigraphormer_graph_encoder.py:233-278).  This module maps such a
HuggingFace state dict onto the product's (reference-named) state dict — embeddings, pre-fusion blocks, fusion blocks,
ViT's final LayerNorm, the poolers with their alias entries, the classifier head — and loads it through
``load_state_dict`` (which fuses query / key / value into the [3D, D] projection).

There is no network on the boxes this runs on: names are resolved from a local directory or the local HuggingFace
cache only (``local_files_only=True``); nothing is downloaded and a missing model is an error, not a silent random init.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

BERT_NAME = "bert-base-uncased"                    # multigraphormer_graph_encoder.py:241-245
VIT_NAME = "google/vit-base-patch16-224"           # :236-240


def _fusion_slots(graph_encoder) -> List[Tuple[int, int]]:
    """(stack, index in stack) of every fusion layer, in execution order"""
    return [(s, j) for s, stack in enumerate(graph_encoder.fusion_layers) for j in range(len(stack.fusion_layers))]


def map_bert_state_dict(sd: Dict[str, torch.Tensor], graph_encoder) -> Dict[str, torch.Tensor]:
    """``BertForSequenceClassification`` (or ``BertModel``) state dict → graph-encoder keys"""
    sd = {(k[len("bert."):] if k.startswith("bert.") else k): v for k, v in sd.items()}
    n_pre = len(graph_encoder.text_model.encoder.layer)
    slots = _fusion_slots(graph_encoder)
    out = {}
    for k, v in sd.items():
        if k.startswith("embeddings.position_ids") or k.startswith("dropout."):
            continue                                            # a buffer of newer transformers versions / no parameters
        if k.startswith("embeddings."):
            out["text_model." + k] = v
        elif k.startswith("encoder.layer."):
            rest = k[len("encoder.layer."):]
            i, tail = rest.split(".", 1)
            i = int(i)
            if i < n_pre:
                out[f"text_model.encoder.layer.{i}.{tail}"] = v
            else:
                if i - n_pre >= len(slots):
                    raise ValueError(f"BERT checkpoint has layer {i}; the model holds {n_pre} + {len(slots)} text blocks")
                s, j = slots[i - n_pre]
                out[f"fusion_layers.{s}.fusion_layers.{j}.bert_encoder.{tail}"] = v
        elif k.startswith("pooler."):
            out["text_model." + k] = v
            out["text_pooler." + k[len("pooler."):]] = v          # the reference registers the pooler twice (:246-247)
        elif k.startswith("classifier."):
            out["node_classifier." + k[len("classifier."):]] = v  # bert.classifier IS the node classifier (:268)
        else:
            raise ValueError(f"unexpected key in the BERT state dict: {k}")
    return out


_VIT_5X = (("attention.q_proj.", "attention.attention.query."), ("attention.k_proj.", "attention.attention.key."),
           ("attention.v_proj.", "attention.attention.value."), ("attention.o_proj.", "attention.output.dense."),
           ("mlp.fc1.", "intermediate.dense."), ("mlp.fc2.", "output.dense."))


def normalize_vit_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers 5.x names of an in-memory ViT (``layers.N.attention.q_proj`` ...) → the 4.x names the reference was
    written against and every published checkpoint uses (``encoder.layer.N.attention.attention.query`` ...)"""
    out = {}
    for k, v in sd.items():
        k = k[len("vit."):] if k.startswith("vit.") else k
        if k.startswith("layers."):
            k = "encoder.layer." + k[len("layers."):]
            for new_, old_ in _VIT_5X:
                k = k.replace("." + new_, "." + old_)
        out[k] = v
    return out


def map_vit_state_dict(sd: Dict[str, torch.Tensor], graph_encoder) -> Dict[str, torch.Tensor]:
    """``ViTModel`` state dict → graph-encoder keys"""
    sd = normalize_vit_keys(sd)
    n_pre = len(graph_encoder.vit_model.encoder.layer)
    slots = _fusion_slots(graph_encoder)
    out = {}
    for k, v in sd.items():
        if k.startswith("embeddings.") or k.startswith("layernorm."):
            out["vit_model." + k] = v
        elif k.startswith("encoder.layer."):
            rest = k[len("encoder.layer."):]
            i, tail = rest.split(".", 1)
            i = int(i)
            if i < n_pre:
                out[f"vit_model.encoder.layer.{i}.{tail}"] = v
            else:
                if i - n_pre >= len(slots):
                    raise ValueError(f"ViT checkpoint has layer {i}; the model holds {n_pre} + {len(slots)} image blocks")
                s, j = slots[i - n_pre]
                out[f"fusion_layers.{s}.fusion_layers.{j}.vit_encoder.{tail}"] = v
        elif k.startswith("pooler."):
            out["vit_model." + k] = v
            out["vit_pooler." + k[len("pooler."):]] = v
        elif k.startswith("classifier."):
            continue                                             # ViTForImageClassification head: not part of mDT
        else:
            raise ValueError(f"unexpected key in the ViT state dict: {k}")
    return out


def _resolve(source, kind: str) -> Dict[str, torch.Tensor]:
    """a state dict, a local directory / file, or a hub name found in the LOCAL HuggingFace cache"""
    if isinstance(source, dict):
        return source
    import os
    if os.path.isfile(source):
        if source.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(source)
        return torch.load(source, map_location="cpu")
    try:
        import transformers
        cls = transformers.AutoModelForSequenceClassification if kind == "bert" else transformers.AutoModel
        model = cls.from_pretrained(source, local_files_only=True)
    except Exception as e:          # noqa: BLE001 — whatever transformers raises, the message below is what the user needs
        raise FileNotFoundError(
            f"pretrained {kind} weights {source!r} are not available locally ({type(e).__name__}: {e}); this build never "
            f"downloads — pass a local directory / file, or ask for random-init encoders explicitly") from e
    return model.state_dict()


def load_pretrained_encoders(graph_encoder, bert=None, vit=None) -> dict:
    """Load HuggingFace weights into the text and / or image stacks of ``graph_encoder`` (a MultiGraphormerGraphEncoder).
    ``bert`` / ``vit``: None (leave that stack as it is), a state dict, a local path or a name in the local HF cache.
    Every mapped tensor must find a parameter of its shape and every parameter of a loaded stack must be covered."""
    mapped: Dict[str, torch.Tensor] = {}
    if bert is not None:
        mapped.update(map_bert_state_dict(_resolve(bert, "bert"), graph_encoder))
    if vit is not None:
        mapped.update(map_vit_state_dict(_resolve(vit, "vit"), graph_encoder))
    own = graph_encoder.state_dict()
    shapes = {k: tuple(v.shape) for k, v in own.items()}
    bad = [k for k, v in mapped.items() if k not in shapes or shapes[k] != tuple(v.shape)]
    if bad:
        k = bad[0]
        raise ValueError(f"pretrained tensor {k} {tuple(mapped[k].shape)} does not fit the model "
                         f"({shapes.get(k, 'no such parameter')}); {len(bad)} mismatches in all")
    covered_prefixes = []
    if bert is not None:
        covered_prefixes += ["text_model.", "text_pooler."]
    if vit is not None:
        covered_prefixes += ["vit_model.", "vit_pooler."]
    for s, j in _fusion_slots(graph_encoder):
        if bert is not None:
            covered_prefixes.append(f"fusion_layers.{s}.fusion_layers.{j}.bert_encoder.")
        if vit is not None:
            covered_prefixes.append(f"fusion_layers.{s}.fusion_layers.{j}.vit_encoder.")
    missing = [k for k in own if any(k.startswith(p) for p in covered_prefixes) and k not in mapped]
    if missing:
        raise ValueError(f"the pretrained weights do not cover {missing[0]} ({len(missing)} parameters in all)")
    full = dict(own)
    full.update({k: v.to(own[k].dtype) for k, v in mapped.items()})
    graph_encoder.load_state_dict(full)
    return dict(loaded=len(mapped), bert=bert is not None, vit=vit is not None)
