"""Task ``node_prediction`` (mDT/src/tasks/node_prediction.py:22-55)."""
from dataclasses import dataclass

import torch.nn as nn

from ..registry import register_task
from .task import Task, TaskConfig


@dataclass
class NodePredictionConfig(TaskConfig):
    ...


@register_task("node_prediction", dataclass=NodePredictionConfig)
class NodePredictionTask(Task):
    """Node prediction (classification) task: one labelled comment per discussion tree."""

    def get_batched_dataset(self, dataset):
        from ..data.dataset import NodeBatchedDataDataset
        return NodeBatchedDataDataset(dataset, spatial_pos_max=self.cfg.spatial_pos_max, device=self.collate_device)

    def build_model(self, args):
        model = super().build_model(args)
        # node_prediction.py:44-53: a fresh classifier list is attached to the MODEL (not to the encoder the forward
        # uses), so it is a dead entry of the state dict — kept so that checkpoints carry the same keys
        ge = model.encoder.graph_encoder
        model.node_encoder_stack = nn.ModuleList([ge.text_pooler, ge.text_dropout, nn.Linear(ge.embedding_dim, 2)])
        for p in model.node_encoder_stack[2].parameters():
            p.requires_grad = False      # never used by forward: keep it out of the gradient arena / optimizer
        return model
