"""Task name ``contrastive_learning`` (mDT/src/tasks/contrastive.py:23): registered so launch
scripts resolve; its criterion (community-contrastive BCE on the global embedding) is outside
the accelerated path — the model already returns the global embedding it needs."""
from dataclasses import dataclass

from ..registry import register_task
from .task import Task, TaskConfig


@dataclass
class ContrastiveLearningConfig(TaskConfig):
    ...


@register_task("contrastive_learning", dataclass=ContrastiveLearningConfig)
class ContrastiveLearningTask(Task):
    pass
