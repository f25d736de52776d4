"""Task ``contrastive_learning`` (mDT/src/tasks/contrastive.py:23-48): community-contrastive pre-training on the
global discussion embedding; batches carry ``hard_y`` beside ``y`` (data/dataset.py ContrastiveBatchedDataDataset),
the criterion is ``contrastive_loss`` (criterions/contrastive_loss.py)."""
from dataclasses import dataclass

from ..registry import register_task
from .task import Task, TaskConfig


@dataclass
class ContrastiveLearningConfig(TaskConfig):
    ...


@register_task("contrastive_learning", dataclass=ContrastiveLearningConfig)
class ContrastiveLearningTask(Task):
    """Graph-level task: one community label (and one hard-negative community label) per discussion tree."""

    def get_batched_dataset(self, dataset):
        from ..data.dataset import ContrastiveBatchedDataDataset
        return ContrastiveBatchedDataDataset(dataset, spatial_pos_max=self.cfg.spatial_pos_max, device=self.collate_device)
