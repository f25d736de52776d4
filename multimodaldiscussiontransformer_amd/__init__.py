"""MI355X-native (gfx950) implementation of the mDT fused multimodal graph-attention
forward / backward path behind the reference's module / model / criterion surface."""
__version__ = "0.1.0"
