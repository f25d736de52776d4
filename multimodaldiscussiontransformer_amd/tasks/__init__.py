from .task import Task, TaskConfig  # noqa: F401
from .node_prediction import NodePredictionTask, NodePredictionConfig  # noqa: F401
from .contrastive import ContrastiveLearningTask  # noqa: F401
