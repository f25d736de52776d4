"""MI355X-native (gfx950) implementation of the mDT fused multimodal graph-attention
forward / backward path behind the reference's module / model / criterion surface."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A training process here owns the default stream,
# the image-branch stream (engine.side_stream), the packer's copy stream, the gradient-exchange stream (ddp.py) and RCCL's
# own stream, besides torch's pools: with 4 queues the image branch ends up behind another stream's event wait in the same
# hardware queue and the two branches serialise (in-call matrix, tools/ddp_force_ab.sh, ms per step without / with the
# exchange machinery — RCCL path forced at world size 1: 2 queues 138.1 / 139.6, 4: 136.9 / 136.8, 6: 133.4 / 135.9,
# 7: 131.4-132.1 / 132.1, 8: 131.7-133.4 / 143.9-145.3).  Eight was round 2's setting and is as good as seven while no
# process group exists, but as soon as one does the step loses 9 % there — every N > 1 run would have paid that — so the
# default is SEVEN.  Read by the HIP runtime when it initialises, i.e. at the first device call — import this package before that.
# One process per GPU is assumed: with several processes on ONE card (a rehearsal on a single-GPU box) the queues of all
# of them oversubscribe the card's queue slots and cross-queue event waits can stall for good — set GPU_MAX_HW_QUEUES=4 there.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")

__version__ = "0.1.0"
