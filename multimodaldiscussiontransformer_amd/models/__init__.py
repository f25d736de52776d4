from .multi_modal_discussion_transformer import *  # noqa: F401,F403
from .multi_modal_discussion_transformer import GraphormerEncoder, GraphormerModel, base_architecture, graphormer_base_architecture  # noqa: F401
