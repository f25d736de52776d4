from .hatespeech_loss import GraphPredictionNodeCrossEntropy, GraphPredictionNodeCrossEntropyConfig  # noqa: F401
from .contrastive_loss import GraphContrastiveLoss, GraphContrastiveLossConfig  # noqa: F401
