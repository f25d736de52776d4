from .packer import PackedBatch, pack_batch, pack_structure  # noqa: F401
from .collator import collator  # noqa: F401
