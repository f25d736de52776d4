from ..registry import DATASET_REGISTRY, register_dataset  # noqa: F401  (mDT/src/data/__init__.py:1-8)
from .packer import PackedBatch, pack_batch, pack_structure  # noqa: F401
from .collator import collator  # noqa: F401
