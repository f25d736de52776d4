from .hatespeech_loss import GraphPredictionNodeCrossEntropy, GraphPredictionNodeCrossEntropyConfig  # noqa: F401
