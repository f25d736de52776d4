"""fp8 operands for the encoder blocks' big GEMMs (BASELINE.json configs[4] "mDT-base fp8 MFMA weights/activations").

Which GEMMs: those where halving the operand bytes pays for the extra quantisation pass over the activation — the
K = D contractions into a wide output, measured on MI355X at C2 shapes (tools/fp8_bench.py): the fused QKV projection
(1.26 x including the pass), fc1 (1.23 x) and, in backward, fc2's input gradient dY W2 (same shape class).  The output
projection and fc2 forward (narrow N, or a 4 D-wide activation to quantise) lose to bf16 with a stand-alone pass and stay
in bf16, as do all weight gradients (fp32 accumulation into the gradient arena) — fusing the quantisation into the
producing kernels (LayerNorm, attention, GELU epilogue) is what would bring those in (docs/experiment_log.md §9).

Since round 3 the 8-bit GEMMs with K a multiple of 128 run on the block MFMA v_mfma_f32_16x16x128_f8f6f4 (csrc/gemm_f8.hip: twice
the bf16 MFMA rate), and two of them hand their OUTPUT to the next GEMM already quantised, from their own epilogue
(``q8_site``): the GELU forward writes fc2's e4m3 operand, fc2's input gradient writes fc1's e5m2 operand.  That makes "fc2"
(forward, K = 4 D) and "d_fc1" (input gradient, K = 4 D) 8-bit sites without a quantisation pass of their own; they are taken only
when the producer delivered the copy (otherwise bf16, as before).

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
# "fc2" / "d_fc1": only with the operand quantised by the producing GEMM's epilogue (fc1 forward / fc2's input gradient)
# Presets (Fp8State(sites=...), model.enable_fp8(sites=...), MDT_FP8_SITES): measured on MI355X, mDT-base step / the C2 and C4
# fixtures of tests/test_fp8_gpu.py (hash weights, logits spanning +-0.6) — speed over bf16 in one call, logits |err|, worst
# parameter-gradient relative L2:
#   "all"    qkv,fc1,fc2,d_fc2,d_fc1   +9...10 %   0.14 / 0.20   0.48     every big GEMM of the blocks but the output projection
#   "fast4"  qkv,fc1,d_fc2,d_fc1       +7 %        0.13 / 0.12   0.43
#   "grads"  d_fc2,d_fc1               +3 %        0.010 (the bf16 level)  0.16   8-bit only where the loss cannot see it
PRESETS = {"all": "qkv,fc1,fc2,d_fc2,d_fc1", "fast4": "qkv,fc1,d_fc2,d_fc1", "grads": "d_fc2,d_fc1"}


def parse_sites(spec) -> tuple:
    if spec is None:
        spec = os.environ.get("MDT_FP8_SITES", "all")
    if not isinstance(spec, str):
        return tuple(spec)
    return tuple(x for x in PRESETS.get(spec, spec).split(",") if x)


SITES = parse_sites(None)
FED_BY_PRODUCER = ("fc2", "d_fc1")
# MDT_FP8_FUSED_Q=0: the producers do not write the copies and those two sites quantise their operand with a pass of their own —
# the same arithmetic in two launches (tests: both routes give the same bytes; tools: what the fusion is worth)
FUSED_Q = os.environ.get("MDT_FP8_FUSED_Q", "1") != "0"


class Fp8State:
    def __init__(self, device, capacity: int = 8192, sites=None):
        self.device = torch.device(device)
        self.site_names = SITES if sites is None else parse_sites(sites)      # which GEMMs take the 8-bit kernel
        self.scale = torch.ones(capacity, dtype=torch.float32, device=self.device)
        self.inv = torch.ones(capacity, dtype=torch.float32, device=self.device)
        self.amax = torch.zeros(capacity, dtype=torch.float32, device=self.device)
        self.fmax = torch.full((capacity,), 448.0, dtype=torch.float32, device=self.device)
        self.sites: Dict[tuple, int] = {}
        self.weights: Dict[tuple, Tuple[torch.Tensor, int, int]] = {}
        self.weights_version = 0
        self.gemms = 0                      # launches that took the 8-bit kernel (diagnostics / tests)
        self.fused_outputs = 0              # ... of which also wrote the next GEMM's operand
        self.no_q8: set = set()             # consumer sites whose producer has no kernel that writes the copy (asked once)

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

    def observe(self, x: torch.Tensor, key, fmt=ops.FP8_E4M3) -> bool:
        """Make ``key`` a site whose first scale comes from ``x`` (no quantisation): the producer of x writes the fp8 copy from
        the next step on.  → True when the site is new."""
        i, new = self._site(key, fmt)
        if new:
            a = x.detach().abs().max().float().clamp_(min=1e-30)
            self.scale[i] = ops.FP8_MAX[fmt] / (a * MARGIN)
            self.inv[i] = (a * MARGIN) / ops.FP8_MAX[fmt]
        return new

    def producer_slot(self, rows: int, w: torch.nn.Parameter, site, fmt=ops.FP8_E4M3, transposed_weight: bool = False):
        """For the kernel that PRODUCES the operand of the 8-bit GEMM ``site`` (a LayerNorm in front of the QKV / fc1
        projection): (u8[rows, K] buffer, format, scale view, amax view, inv-scale view) to write the fp8 copy and its maximum
        into — or None when that GEMM will not take the 8-bit kernel, the site has no scale yet (first sight: the consumer
        measures the tensor itself), or producers are told not to (MDT_FP8_FUSED_Q=0).  Hand ``(buffer, inv view)`` to
        ``linear(..., x8=...)``."""
        n_out = w.shape[1] if transposed_weight else w.shape[0]
        k = w.shape[0] if transposed_weight else w.shape[1]
        i = self.sites.get(site)
        if i is None or not FUSED_Q or site[0] not in self.site_names or w.dtype != torch.bfloat16 or not self.eligible(rows, n_out, k):
            return None
        self.fused_outputs += 1
        return torch.empty(rows, k, dtype=torch.uint8, device=self.device), fmt, self.scale[i:i + 1], self.amax[i:i + 1], self.inv[i:i + 1]

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

    def linear(self, x: torch.Tensor, w: torch.nn.Parameter, site, *, transposed_weight=False, grad=False, x8=None, q8_site=None,
               q8_grad=False, **kw):
        """x[M, K] @ W^T (W [N, K]; ``transposed_weight``: x[M, N] @ W, the input gradient) through the 8-bit kernel,
        or None when the shape is not one it is built for (the caller then runs the bf16 GEMM).
        ``x8``: (u8 tensor, inv-scale view) — x already quantised by the kernel that produced it.
        ``q8_site``: the site of the GEMM that will consume the OUTPUT: the result is then (out, (out8, inv-scale view)), the
        second part None when the copy could not be made here (consumer site not known yet, no kernel for it)."""
        n_out = w.shape[1] if transposed_weight else w.shape[0]
        k = w.shape[0] if transposed_weight else w.shape[1]
        if site[0] not in self.site_names or x.dtype != torch.bfloat16 or not self.eligible(x.shape[0], n_out, k):
            return None
        fmt = ops.FP8_E5M2 if grad else ops.FP8_E4M3
        if x8 is not None:
            x8, inv_x = x8
        elif site[0] in FED_BY_PRODUCER and (self.observe(x, site, fmt) or FUSED_Q):
            return None                          # first sight (known from now on: the producer writes the copy next time), or no copy came
        else:
            x8, inv_x = self.quantize(x, site, fmt)
        w8, inv_w = self.weight(w, transposed_weight)
        self.gemms += 1
        if q8_site is None:
            return ops.gemm_fp8(x8, w8, inv_x, inv_w, a_format=fmt, **kw)
        j = self.sites.get(q8_site)
        if j is not None and FUSED_Q and q8_site[0] in self.site_names and q8_site not in self.no_q8:
            from ._lib import MdtUnsupported
            out8 = torch.empty(x.shape[0], n_out, dtype=torch.uint8, device=x.device)
            try:
                out = ops.gemm_fp8(x8, w8, inv_x, inv_w, a_format=fmt, q8_out=out8, q8_format=ops.FP8_E5M2 if q8_grad else ops.FP8_E4M3,
                                   q8_scale=self.scale[j:j + 1], q8_amax=self.amax[j:j + 1], **kw)
                self.fused_outputs += 1
                return out, (out8, self.inv[j:j + 1])
            except MdtUnsupported:               # only "no kernel writes the copy": a launch error or a contract violation propagates
                self.no_q8.add(q8_site)          # e.g. MDT_GEMM_F8W=0: nothing writes the copy; do not ask again
        return ops.gemm_fp8(x8, w8, inv_x, inv_w, a_format=fmt, **kw), None


# the active state (None: bf16 everywhere).  Set by GraphormerModel.enable_fp8(); read by engine.transformer_block.
ACTIVE: Optional[Fp8State] = None
