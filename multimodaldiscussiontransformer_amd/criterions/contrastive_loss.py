"""Drop-in for mDT/src/criterions/contrastive_loss.py (criterion ``contrastive_loss``): community-contrastive
pre-training on the global discussion embedding — discussions of the same community (``y``) are positive pairs,
discussions of the polar-opposite community (``hard_y``) hard negatives, everything else soft negatives whose weight is
either fixed or adapted to the hard / soft pair counts.

Normalisation, the B x B similarity, the weighted BCE, the counters and the gradient w.r.t. the embeddings are one C-ABI
call (``mdt_contrastive_loss``, csrc/contrastive.hip); the reference's quirks are kept: the per-row adaptive weight and
the labels in ``pred == targets`` are broadcast along the LAST axis (contrastive_loss.py:143-147, :155), the diagonal is
excluded from the loss but not from the counters, ``sample_size`` is B * B.  This criterion is the only consumer of the
encoder's ``global_embedding``: under it the final graph stack (which ``node_cross_entropy`` leaves gradient-free,
SURVEY.md §8 quirk 3) trains.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List

import torch

from .. import ops
from ..registry import FairseqCriterion, FairseqDataclass, register_criterion


@dataclass
class GraphContrastiveLossConfig(FairseqDataclass):
    soft_negative_weight: float = field(default=0.0, metadata={"help": "Weight to associate to soft negative pairs in the contrastive loss. Flag is exclusive against adaptive_soft_negative_weight"})
    adaptive_soft_negative_weight: bool = field(default=True, metadata={"help": "Whether to adapt the soft negative weight based on the number of positive pairs and negative pairs. Flag is exclusive against soft_negative_weight"})
    multiplication_scale: float = field(default=20.0, metadata={"help": "Multiplcation factor to scale the similarity matrix (1 = strict match, 20 = less strict)"})


class _ContrastiveLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, y, hard_y, scale, soft_w, adaptive):
        loss, counters, d_emb = ops.contrastive_loss(emb.contiguous(), y, hard_y, scale, soft_w, adaptive)
        ctx.save_for_backward(d_emb)
        ctx.mark_non_differentiable(counters)
        return loss, counters

    @staticmethod
    def backward(ctx, gloss, _gc):
        (d_emb,) = ctx.saved_tensors
        return d_emb * gloss.to(d_emb.dtype), None, None, None, None, None


@register_criterion("contrastive_loss", dataclass=GraphContrastiveLossConfig)
class GraphContrastiveLoss(FairseqCriterion):
    """Contrastive loss for discussion embeddings."""

    def __init__(self, task, soft_negative_weight: float = 0.0, multiplication_scale: float = 20.0,
                 adaptive_soft_negative_weight: bool = True) -> None:
        super().__init__(task)
        self.soft_negative_weight = soft_negative_weight
        self.multiplication_scale = multiplication_scale
        self.adaptive_soft_negative_weight = adaptive_soft_negative_weight
        if self.adaptive_soft_negative_weight and self.soft_negative_weight != 0:
            raise ValueError("adaptive_soft_negative_weight and soft_negative_weight are mutually exclusive")

    def forward(self, model, sample: Dict[str, Any], reduce=True):
        """→ (loss, sample_size = B * B, logging_output) as the reference (:76-180)."""
        if not reduce:
            raise NotImplementedError("reduce=False is never used by the reference trainer")
        if "batched_data" not in sample["net_input"]:
            raise ValueError(f"Invalid sample, missing batched_data: {sample['net_input']}")
        bd = sample["net_input"]["batched_data"]
        num_comments = bd["x"].shape[1]
        _, embeddings = model(**sample["net_input"])
        y = bd["y"].to(device=embeddings.device, dtype=torch.float32).reshape(-1)
        hard_y = bd["hard_y"].to(device=embeddings.device, dtype=torch.float32).reshape(-1)
        if y.numel() != embeddings.shape[0] or hard_y.numel() != embeddings.shape[0]:
            raise ValueError("contrastive_loss needs one y / hard_y label per discussion tree")
        loss, counters = _ContrastiveLoss.apply(embeddings, y, hard_y, self.multiplication_scale,
                                                self.soft_negative_weight, self.adaptive_soft_negative_weight)
        loss = loss.squeeze(0)
        sim_count = embeddings.shape[0] * embeddings.shape[0]
        logging_output = {
            "loss": loss.detach(), "sample_size": sim_count, "nsentences": sim_count, "ntokens": num_comments,
            "ncorrect": counters[0], "positive_correct": counters[1], "total_positive": counters[2],
            "pred_positive": counters[3],
        }
        return loss, sim_count, logging_output

    @staticmethod
    def compute_metrics(logging_outputs: List[Dict[str, Any]]) -> Dict[str, float]:
        """The values the reference logs (:182-218); precision / recall are left out when their denominator is 0
        (the reference would raise ZeroDivisionError there)."""
        def tot(k):
            return sum(float(log.get(k, 0)) for log in logging_outputs)

        sample_size = tot("sample_size")
        out = {"loss": tot("loss") / sample_size if sample_size else 0.0}
        if len(logging_outputs) > 0 and "ncorrect" in logging_outputs[0]:
            out["accuracy"] = 100.0 * tot("ncorrect") / sample_size
            if tot("pred_positive"):
                out["precision"] = 100.0 * tot("positive_correct") / tot("pred_positive")
            if tot("total_positive"):
                out["recall"] = 100.0 * tot("positive_correct") / tot("total_positive")
        return out

    @staticmethod
    def reduce_metrics(logging_outputs) -> None:
        m = GraphContrastiveLoss.compute_metrics(logging_outputs)
        GraphContrastiveLoss.last_metrics = m
        try:
            from fairseq import metrics
        except ImportError:
            return
        sample_size = sum(float(log.get("sample_size", 0)) for log in logging_outputs)
        metrics.log_scalar("loss", m["loss"], sample_size, round=3)
        for k in ("accuracy", "precision", "recall"):
            if k in m:
                metrics.log_scalar(k, m[k], sample_size, round=2)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True
