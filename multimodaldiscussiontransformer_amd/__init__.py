"""MI355X-native (gfx950) implementation of the mDT fused multimodal graph-attention
forward / backward path behind the reference's module / model / criterion surface."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A training process here owns the default stream,
# the image-branch stream (engine.side_stream), the gradient-exchange stream (ddp.py) and RCCL's own stream, besides
# torch's pools: with 4 queues the image branch ends up behind another stream's event wait in the same hardware queue and
# the two branches serialise again (measured: 161 ms per step against 154 with 8 queues, one GPU, RCCL path forced on).
# Read by the HIP runtime when it initialises, i.e. at the first device call — import this package before that.
# One process per GPU is assumed: with several processes on ONE card (a rehearsal on a single-GPU box) 2 x 8 queues
# oversubscribe the card's queue slots and cross-queue event waits can stall for good — set GPU_MAX_HW_QUEUES=4 there.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
