from .multihead_attention import MultiheadAttention  # noqa: F401
from .graphormer_layers import GraphNodeFeature, GraphAttnBias  # noqa: F401
from .graphormer_graph_encoder_layer import GraphormerGraphEncoderLayer, GraphEncoderStack  # noqa: F401
from .multi_graphormer_fusion_layer import GraphFusionLayer, GraphFusionStack  # noqa: F401
from .multigraphormer_graph_encoder import MultiGraphormerGraphEncoder, init_graphormer_params  # noqa: F401
