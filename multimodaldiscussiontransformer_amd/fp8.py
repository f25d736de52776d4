"""fp8 operands for the encoder blocks' big GEMMs (BASELINE.json configs[4] "mDT-base fp8 MFMA weights/activations").

Which GEMMs: those where halving the operand bytes pays for the extra quantisation pass over the activation — the
K = D contractions into a wide output, measured on MI355X at C2 shapes (tools/fp8_bench.py): the fused QKV projection
(1.26 x including the pass), fc1 (1.23 x) and, in backward, fc2's input gradient dY W2 (same shape class).  The output
projection and fc2 forward (narrow N, or a 4 D-wide activation to quantise) lose to bf16 with a stand-alone pass and stay
in bf16, as do all weight gradients (fp32 accumulation into the gradient arena) — fusing the quantisation into the
producing kernels (LayerNorm, attention, GELU epilogue) is what would bring those in.

Scaling: per tensor.  Activations / gradients: DELAYED — a step quantises with the scale derived from the |x| maximum
the previous step measured at the same site (fmax / (amax * margin), margin 2 = one binade of headroom), the quantiser
records this step's maximum, and one launch at the end of backward turns all maxima into the next scales.  Nothing ever
leaves the device.  A site seen for the first time measures its maximum first (one extra reduction, once).  Weights:
current scaling, re-quantised (and transposed where the input gradient needs the k-contiguous copy) whenever the
optimiser has stepped.  Formats: OCP e4m3 for activations and weights, e5m2 for gradients; fp32 accumulation.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

import os

MARGIN = 2.0
# which GEMMs take the 8-bit kernel: any of "qkv", "fc1", "d_fc2" (MDT_FP8_SITES=fc1,d_fc2 leaves the QKV projection in bf16:
# the softmax amplifies errors of q and k, see DESIGN.md for the measured trade)
SITES = tuple(x for x in os.environ.get("MDT_FP8_SITES", "qkv,fc1,d_fc2").split(",") if x)


class Fp8State:
    def __init__(self, device, capacity: int = 8192):
        self.device = torch.device(device)
        self.scale = torch.ones(capacity, dtype=torch.float32, device=self.device)
        self.inv = torch.ones(capacity, dtype=torch.float32, device=self.device)
        self.amax = torch.zeros(capacity, dtype=torch.float32, device=self.device)
        self.fmax = torch.full((capacity,), 448.0, dtype=torch.float32, device=self.device)
        self.sites: Dict[tuple, int] = {}
        self.weights: Dict[tuple, Tuple[torch.Tensor, int, int]] = {}
        self.weights_version = 0
        self.gemms = 0                      # launches that took the 8-bit kernel (diagnostics / tests)

    # -- sites -------------------------------------------------------------------------
    def _site(self, key, fmt) -> Tuple[int, bool]:
        i = self.sites.get(key)
        if i is not None:
            return i, False
        i = len(self.sites)
        if i >= self.scale.numel():
            raise RuntimeError("Fp8State: more quantisation sites than capacity")
        self.sites[key] = i
        self.fmax[i] = ops.FP8_MAX[fmt]
        return i, True

    def quantize(self, x: torch.Tensor, key, fmt=ops.FP8_E4M3):
        """→ (u8 tensor, inv_scale view).  Delayed scaling; a new site derives its first scale from the tensor itself."""
        i, new = self._site(key, fmt)
        if new:
            a = x.detach().abs().max().float().clamp_(min=1e-30)          # device-side, no sync
            self.scale[i] = ops.FP8_MAX[fmt] / (a * MARGIN)
            self.inv[i] = (a * MARGIN) / ops.FP8_MAX[fmt]
        return ops.fp8_quantize(x, fmt, scale=self.scale[i:i + 1], amax=self.amax[i:i + 1]), self.inv[i:i + 1]

    def weight(self, w: torch.nn.Parameter, transposed: bool = False):
        """e4m3 copy of a weight ([N, K], or its transpose for dX = dY W), cached until the optimiser steps."""
        key = (id(w), transposed)
        hit = self.weights.get(key)
        if hit is not None and hit[2] == self.weights_version:
            return hit[0], self.inv[hit[1]:hit[1] + 1]
        i, _ = self._site(("w", id(w), transposed), ops.FP8_E4M3)
        src = w.data
        if transposed:
            src = ops.transpose2d(src)
        a = src.detach().abs().max().float().clamp_(min=1e-30)
        self.scale[i] = 448.0 / a
        self.inv[i] = a / 448.0
        q = ops.fp8_quantize(src.contiguous(), ops.FP8_E4M3, scale=self.scale[i:i + 1])
        self.weights[key] = (q, i, self.weights_version)
        return q, self.inv[i:i + 1]

    def optimizer_stepped(self):
        self.weights_version += 1

    def end_of_step(self):
        """All activation / gradient maxima of this step → next step's scales (one launch); weights keep theirs."""
        n = len(self.sites)
        if n:
            ops.fp8_scale_update(self.amax[:n], self.scale[:n], self.inv[:n], self.fmax[:n], MARGIN)

    # -- the GEMM ----------------------------------------------------------------------
    @staticmethod
    def eligible(rows: int, n_out: int, k: int) -> bool:
        return n_out % 256 == 0 and k % 64 == 0 and k >= 256 and rows >= 256

    def linear(self, x: torch.Tensor, w: torch.nn.Parameter, site, *, transposed_weight=False, grad=False, **kw) -> Optional[torch.Tensor]:
        """x[M, K] @ W^T (W [N, K]; ``transposed_weight``: x[M, N] @ W, the input gradient) through the 8-bit kernel,
        or None when the shape is not one it is built for (the caller then runs the bf16 GEMM)."""
        n_out = w.shape[1] if transposed_weight else w.shape[0]
        k = w.shape[0] if transposed_weight else w.shape[1]
        if site[0] not in SITES or x.dtype != torch.bfloat16 or not self.eligible(x.shape[0], n_out, k):
            return None
        fmt = ops.FP8_E5M2 if grad else ops.FP8_E4M3
        x8, inv_x = self.quantize(x, site, fmt)
        w8, inv_w = self.weight(w, transposed_weight)
        self.gemms += 1
        return ops.gemm_fp8(x8, w8, inv_x, inv_w, a_format=fmt, **kw)


# the active state (None: bf16 everywhere).  Set by GraphormerModel.enable_fp8(); read by engine.transformer_block.
ACTIVE: Optional[Fp8State] = None
