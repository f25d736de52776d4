"""Drop-in for mDT/src/criterions/hatespeech_loss.py (criterion ``node_cross_entropy``).

The weighted two-class cross entropy, the fp16 rounding of logits / class weights the
reference applies whatever the model dtype (``.type(torch.HalfTensor)``, :58-64, :95) and the
TP / FP / FN counters are one HIP kernel (``mdt_node_ce``); labelled rows come from the
packer's CSR index instead of a boolean-mask gather.  ``reduce_metrics`` reproduces the
reference's accuracy / precision / recall / F1 with its zero guards (:133-173).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List

import torch

from .. import ops
from ..data.packer import packed_from_batched_data
from ..registry import FairseqCriterion, FairseqDataclass, register_criterion


@dataclass
class GraphPredictionNodeCrossEntropyConfig(FairseqDataclass):
    positive_weight: float = field(default=1.0, metadata={"help": "Weight to associate to the positive class"})
    negative_weight: float = field(default=1.0, metadata={"help": "Weight to associate to the negative class"})


class _NodeCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, rows, targets, w_neg, w_pos, fp16_loss):
        loss, counters, dlogits = ops.node_ce(logits.contiguous(), rows, targets, w_neg, w_pos, fp16_loss=fp16_loss)
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(counters)
        return loss, counters

    @staticmethod
    def backward(ctx, gloss, _gc):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss.to(dlogits.dtype), None, None, None, None, None


@register_criterion("node_cross_entropy", dataclass=GraphPredictionNodeCrossEntropyConfig)
class GraphPredictionNodeCrossEntropy(FairseqCriterion):
    """Node cross-entropy loss for graph node classification."""

    def __init__(self, task, positive_weight: float = 1.0, negative_weight: float = 1.0, fp16_loss: bool = True) -> None:
        super().__init__(task)
        self.positive_weight = float(positive_weight)
        self.negative_weight = float(negative_weight)
        self.fp16_loss = fp16_loss
        # the reference keeps the weights as a half tensor [negative, positive]
        self.weight = torch.tensor([negative_weight, positive_weight], dtype=torch.half)

    def forward(self, model, sample: Dict[str, Any], reduce=True):
        """→ (loss, sample_size, logging_output) exactly as the reference (:66-131)."""
        if not reduce:
            raise NotImplementedError("reduce=False is never used by the reference trainer")
        bd = sample["net_input"]["batched_data"]
        num_comments = bd["x"].shape[1]
        comment_logits, _ = model(**sample["net_input"])
        pb = packed_from_batched_data(bd)
        loss, counters = _NodeCE.apply(comment_logits, pb.label_rows, pb.targets, self.negative_weight,
                                       self.positive_weight, self.fp16_loss)
        loss = loss.squeeze(0)
        sample_size = pb.n_labels
        logging_output = {
            "loss": loss.data, "sample_size": sample_size, "nsentences": sample_size, "ntokens": num_comments,
            "ncorrect": counters[0], "num_positive_correct": counters[1], "total_positive": counters[2],
            "num_pred_positive": counters[3],
        }
        return loss, sample_size, logging_output

    @staticmethod
    def compute_metrics(logging_outputs: List[Dict[str, Any]]) -> Dict[str, float]:
        def tot(k):
            return sum(float(log.get(k, 0)) for log in logging_outputs)

        loss_sum, sample_size = tot("loss"), tot("sample_size")
        out = {"loss": loss_sum / sample_size if sample_size else 0.0}
        if len(logging_outputs) > 0 and "ncorrect" in logging_outputs[0]:
            ncorrect, tp, total_pos, pred_pos = tot("ncorrect"), tot("num_positive_correct"), tot("total_positive"), tot("num_pred_positive")
            recall = 0 if total_pos == 0 else tp / total_pos
            precision = 0 if pred_pos == 0 else tp / pred_pos
            f1 = 0 if (precision == 0 and recall == 0) else 2 * ((precision * recall) / (precision + recall))
            out.update(accuracy=ncorrect / sample_size, recall=recall, precision=precision, f1=f1)
        return out

    @staticmethod
    def reduce_metrics(logging_outputs) -> None:
        """Aggregate logging outputs from data parallel training; logs through fairseq.metrics when
        fairseq is installed, otherwise keeps the values in ``last_metrics``."""
        m = GraphPredictionNodeCrossEntropy.compute_metrics(logging_outputs)
        GraphPredictionNodeCrossEntropy.last_metrics = m
        try:
            from fairseq import metrics
        except ImportError:
            return
        sample_size = sum(float(log.get("sample_size", 0)) for log in logging_outputs)
        for k, v in m.items():
            metrics.log_scalar(k, v, sample_size, round=3)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True
