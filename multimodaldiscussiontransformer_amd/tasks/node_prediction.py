"""Task ``node_prediction`` (mDT/src/tasks/node_prediction.py:22-55)."""
from dataclasses import dataclass

from ..registry import register_task
from .task import Task, TaskConfig


@dataclass
class NodePredictionConfig(TaskConfig):
    ...


@register_task("node_prediction", dataclass=NodePredictionConfig)
class NodePredictionTask(Task):
    """Node prediction (classification) task: one labelled comment per discussion tree."""
