"""Tape engine: the host-side scheduler of the fused forward / backward.

The reference runs eager PyTorch autograd over ~1000 small ops per step.  Here a step is a
short *tape* of coarse ops (one per transformer block, exchange, embedding, head), each
of which launches hand-written HIP kernels through the C ABI and knows its own adjoint.
The tape owns activation gradients explicitly, which is what lets the bottleneck-token
exchanges (in-place row writes in the reference: modules/multigraphormer_graph_encoder.py:
371,425,435; modules/multi_graphormer_fusion_layer.py:63-66) stay sparse row updates in
backward too, instead of dense zero-filled gradient tensors.

``TapeFunction`` is the single bridge to ``torch.autograd``: a module's public
``forward`` (same signature as the reference module) wraps its tape-level ``_fwd`` in
one autograd node.  Parameter gradients are accumulated in fp32 — into ``param.main_grad``
(persistent, what the RCCL data-parallel hook all-reduces) when present, otherwise into
temporaries that are handed back to autograd as ordinary ``.grad``s.
"""
from __future__ import annotations

import os

import contextlib
import threading
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from . import fp8 as F8
from . import ops


class _StreamState(threading.local):
    """Which branch the calling thread is enqueuing for (autograd runs a tape's backward on its own thread)."""
    side_active = False


_SS = _StreamState()
_SIDE_STREAMS = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The second HIP stream of ``device`` (one per process and device): the image branch of a two-stream tape."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class Var:
    """An activation on the tape: data + lazily created gradient buffer."""
    __slots__ = ("data", "grad", "needs_grad", "side")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True):
        self.data = data
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self.side = _SS.side_active            # produced (and consumed) by the image branch of a two-stream tape

    @property
    def rows(self):
        return self.data.shape[0]


class Tape:
    def __init__(self, use_main_grad: bool = False, on_params_ready: Optional[Callable] = None, inference: bool = False):
        self.ops: List[Callable[[], None]] = []
        self.inference = inference                 # forward only (torch.no_grad): adjoints and what they hold are dropped
        self.use_main_grad = use_main_grad
        self.tmp_grads = {}
        self.on_params_ready = on_params_ready     # DDP hook: called with the params an op just finished
        self._seed_base = None
        self._seed_ctr = 0
        self.side: Optional[torch.cuda.Stream] = None
        self._main: Optional[torch.cuda.Stream] = None

    # -- two HIP streams ------------------------------------------------------------------
    # The text and image branches of mDT are independent between bottleneck exchanges (the 6 + 6 pre-fusion layers,
    # and the two blocks of every fusion layer).  A two-stream tape enqueues the image branch on a second stream, so
    # the last partially filled wave of one branch's GEMM tiles and its small kernels run beside the other branch's
    # work.  The discipline that keeps this race-free:
    #   * ``fork()``  — side waits for everything main has enqueued; ``join()`` — main waits for side.  In backward
    #     the tape is walked in reverse, so a fork acts as a join and vice versa (both are recorded on the tape).
    #   * only ops inside ``on_side()`` run on the side stream, and they touch nothing but image-branch tensors
    #     (allocated there, by the stream-aware caching allocator), parameters, parameter gradients and index
    #     vectors (all persistent).  Exchange ops run on main strictly between a join and the next fork.
    #   * the one tensor class that crosses the other way — a gradient buffer that a main-stream adjoint creates for
    #     an image-branch activation — is allocated from the side stream's pool (``grad_buf``), so the allocator
    #     recycles it in side-stream order; a dense gradient handed over by an op (``add_grad``) falls back to
    #     ``record_stream``.
    def enable_side(self, device):
        if self.side is None:
            self.side = side_stream(device)

    def fork(self):
        if self.side is None:
            return
        self.side.wait_stream(torch.cuda.current_stream())
        self.record(lambda: torch.cuda.current_stream().wait_stream(self.side))

    def join(self):
        if self.side is None:
            return
        torch.cuda.current_stream().wait_stream(self.side)
        self.record(lambda: self.side.wait_stream(torch.cuda.current_stream()))

    def _enter_side(self):
        self._main = torch.cuda.current_stream()
        torch.cuda.set_stream(self.side)
        _SS.side_active = True

    def _leave_side(self):
        if _SS.side_active:
            torch.cuda.set_stream(self._main)
            _SS.side_active = False

    @contextlib.contextmanager
    def on_side(self):
        """Ops issued inside run on the side stream, forward and backward (no-op on a single-stream tape)."""
        if self.side is None:
            yield
            return
        self.record(self._leave_side)             # backward reaches the region's first op last
        self._enter_side()
        try:
            yield
        finally:
            self._leave_side()
            self.record(self._enter_side)

    def _crossing(self, v: "Var", t: torch.Tensor):
        """``t`` was just allocated on the current stream as a gradient of ``v``: tell the allocator when the other
        branch is the one that will read it."""
        if self.side is not None and v.side != _SS.side_active:
            t.record_stream(self.side if v.side else self._main)

    def next_seed(self) -> int:
        """A fresh 63-bit dropout-site seed (CPU generator: follows torch.manual_seed, no device sync)."""
        if self._seed_base is None:
            self._seed_base = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._seed_ctr += 1
        return (self._seed_base + 0x9E3779B97F4A7C15 * self._seed_ctr) & 0x7FFFFFFFFFFFFFFF

    # -- gradient plumbing ------------------------------------------------------------
    def record(self, bwd: Callable[[], None]):
        if not self.inference:                     # the closure keeps every saved activation alive
            self.ops.append(bwd)

    def pgrad(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """fp32 accumulation buffer of a parameter, or None when it is frozen."""
        if p is None or not p.requires_grad:
            return None
        if self.use_main_grad:
            g = getattr(p, "main_grad", None)
            if g is None:
                raise RuntimeError("use_main_grad=True but a trainable parameter has no .main_grad buffer")
            return g
        g = self.tmp_grads.get(id(p))
        if g is None:
            g = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
            self.tmp_grads[id(p)] = g
        return g

    def grad_buf(self, v: Var) -> torch.Tensor:
        """Gradient buffer of ``v`` for sparse (row-wise) accumulation; zero-created on first use."""
        if v.grad is None:
            if self.side is not None and v.side and not _SS.side_active:
                # an exchange adjoint (main stream) creates the gradient of an image-branch activation: take the
                # buffer from the side stream's pool, where its reader will run and where it is freed.  The side
                # stream is idle here (between a join and the next fork), so waiting for its zero-fill costs nothing;
                # record_stream instead would park the block behind an event on every free, and a host that runs
                # steps ahead of the GPU then falls back to hipMalloc
                with torch.cuda.stream(self.side):
                    v.grad = torch.zeros_like(v.data)
                torch.cuda.current_stream().wait_stream(self.side)
            else:
                v.grad = torch.zeros_like(v.data)
                self._crossing(v, v.grad)
        return v.grad

    def add_grad(self, v: Var, g: torch.Tensor):
        """Dense contribution: take ownership of ``g`` or accumulate it."""
        if not v.needs_grad:
            return
        if v.grad is None:
            v.grad = g
            self._crossing(v, g)
        else:
            ops.row_axpby(v.grad, v.grad.shape[0], a=g, accumulate=True)

    def backward(self):
        try:
            while self.ops:
                self.ops.pop()()
        finally:
            self._leave_side()
        self.ops = []
        self.sync_streams()
        if F8.ACTIVE is not None:
            F8.ACTIVE.end_of_step()          # this step's |x| maxima become the next step's scales (one launch)

    def sync_streams(self):
        """End of a pass: whatever follows on the main stream sees the side stream's work."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


# ------------------------------------------------------------------------------------------
# parameter bundles
@dataclass
class BlockParams:
    """One transformer block.  ``qkv_w`` is the row-concatenation [Wq; Wk; Wv] (3D x D)."""
    qkv_w: torch.nn.Parameter
    qkv_b: torch.nn.Parameter
    o_w: torch.nn.Parameter
    o_b: torch.nn.Parameter
    ln1_w: torch.nn.Parameter
    ln1_b: torch.nn.Parameter
    fc1_w: torch.nn.Parameter
    fc1_b: torch.nn.Parameter
    fc2_w: torch.nn.Parameter
    fc2_b: torch.nn.Parameter
    ln2_w: torch.nn.Parameter
    ln2_b: torch.nn.Parameter

    def all(self):
        return [self.qkv_w, self.qkv_b, self.o_w, self.o_b, self.ln1_w, self.ln1_b, self.fc1_w, self.fc1_b,
                self.fc2_w, self.fc2_b, self.ln2_w, self.ln2_b]


@dataclass
class AttnSpec:
    nseq: int
    S: int
    H: int
    seq_stride: Optional[int] = None
    pos_stride: int = 1
    scale: Optional[float] = None
    key_mask: Optional[torch.Tensor] = None        # u8[nseq,S]
    dense_bias: Optional[torch.Tensor] = None      # f32[nseq,H,S,S]
    dense_bias_var: Optional[Var] = None           # gradient sink for dense_bias (module-level API)
    attn_bias: Optional[torch.Tensor] = None       # f32[nseq,S,S]   structural: {0,-inf} mask
    spatial_pos: Optional[torch.Tensor] = None     # i32[nseq,S-1,S-1]
    sp_table: Optional[torch.nn.Parameter] = None  # [num_spatial,H]
    virt: Optional[torch.nn.Parameter] = None      # [1,H]
    key_pad: Optional[torch.Tensor] = None         # u8[nseq,S]
    seq_offsets: Optional[torch.Tensor] = None     # i32[nseq+1]: ragged sequences of at most S rows (no masks / biases)
    q_limit: int = 0                               # > 0: only the first q_limit rows of every sequence are needed as queries
    bins: Optional[list] = None                    # ragged: [(sequence ids i32[n], cap), ...] — one launch per length bin

    def kwargs(self):
        return dict(seq_stride=self.seq_stride, pos_stride=self.pos_stride, scale=self.scale, key_mask=self.key_mask,
                    seq_offsets=self.seq_offsets, q_limit=self.q_limit, bins=self.bins,
                    dense_bias=self.dense_bias, attn_bias=self.attn_bias, spatial_pos=self.spatial_pos,
                    sp_table=None if self.sp_table is None else self.sp_table.data,
                    virt=None if self.virt is None else self.virt.data.view(-1), key_pad=self.key_pad)


# ------------------------------------------------------------------------------------------
# helpers
def _split_k(n_out: int, k_out: int, red: int) -> int:
    tiles = ((n_out + 127) // 128) * ((k_out + 127) // 128)
    s = max(1, 1024 // max(1, tiles))
    return int(max(1, min(s, red // 1024 if red >= 1024 else 1)))


_WGRAD_ASUM = os.environ.get("MDT_WGRAD_ASUM", "1") != "0"     # 0: separate column-sum launches (A/B runs)


def wgrad(tape: Tape, dy: torch.Tensor, x: torch.Tensor, w: torch.nn.Parameter, b: Optional[torch.nn.Parameter]):
    """dW[N,K] += dY^T X (fp32, split-K atomics) and db[N] += colsum(dY)."""
    gw = tape.pgrad(w)
    gb = tape.pgrad(b) if b is not None else None
    if gw is not None:
        # the bias gradient rides on the weight-gradient GEMM, which streams dY anyway (MDT_EPI_ASUM; bf16 operands)
        ride = gb is not None and dy.dtype == torch.bfloat16 and _WGRAD_ASUM
        ops.gemm(dy, x, trans_a=True, trans_b=True, out=gw.view(dy.shape[1], x.shape[1]), epilogue=ops.EPI_ATOMIC,
                 split_k=_split_k(dy.shape[1], x.shape[1], dy.shape[0]), asum=gb.view(-1) if ride else None)
        if ride:
            return
    if gb is not None:
        ops.colsum(dy, out=gb.view(-1))


_LN_BIAS_RIDE = os.environ.get("MDT_LN_BIAS_RIDE", "1") != "0"    # 0: those sums stay in the LayerNorm backward (A/B runs)


# ------------------------------------------------------------------------------------------
# input gradients against a transposed weight copy
_NN_DGRAD = os.environ.get("MDT_NN_DGRAD", "0") == "1"     # 1: the big launches read W through a transposed copy (A/B runs)
EMBED_LN_FUSED = os.environ.get("MDT_EMBED_LN_FUSED", "1") != "0"         # 0: embedding sum and its LayerNorm as two launches (A/B runs, tests)
PATCH_K_PAD = os.environ.get("MDT_PATCH_K_PAD", "1") != "0"               # 0: ViT-L/14's K = 588 patch GEMMs stay on the any-shape kernel (A/B runs)
PATCH_EMBED_FUSED = os.environ.get("MDT_PATCH_EMBED_FUSED", "1") != "0"   # 0: patch gather, GEMM and assembly as three launches (A/B runs, tests)
_WT_CACHE: dict = {}
WEIGHT_EPOCH = 0            # bumped by whoever rewrites weights behind torch's back (optim.FusedAdam.step)


def weights_changed():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def _transposed(w: torch.nn.Parameter) -> torch.Tensor:
    """W^T ([in, out], contiguous) of a Linear weight [out, in], cached until the weight changes (torch-side writes bump
    ``_version``, the fused optimiser bumps WEIGHT_EPOCH)."""
    key = id(w)
    ver = (w._version, WEIGHT_EPOCH, w.data_ptr())
    hit = _WT_CACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == ver:
        return hit[2]
    wt = ops.transpose2d(w.data)
    _WT_CACHE[key] = (weakref.ref(w), ver, wt)
    return wt


_KP_CACHE: dict = {}


def _k_padded(w: torch.nn.Parameter, wmat: torch.Tensor, kp: int) -> torch.Tensor:
    """[out, kp] copy of a weight viewed as [out, k] with zero columns behind k, cached until the weight changes."""
    key = (id(w), kp)
    ver = (w._version, WEIGHT_EPOCH, w.data_ptr())
    hit = _KP_CACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == ver:
        return hit[2]
    wp = torch.zeros(wmat.shape[0], kp, dtype=wmat.dtype, device=wmat.device)
    wp[:, :wmat.shape[1]].copy_(wmat)
    _KP_CACHE[key] = (weakref.ref(w), ver, wp)
    return wp


def dgrad(dy: torch.Tensor, w: torch.nn.Parameter, **kw) -> torch.Tensor:
    """dX[M, in] = dY[M, out] @ W[out, in] (+ epilogue).  W is read in place as the k-major operand: since the 4-wave
    kernel reads k-major fragments with asm ds_read_b64_tr_b16 pairs (gemm.hip, w4_frag) that form runs as fast as the
    k-contiguous one (tools/gemm_ab.py --nn-dgrad; whole step 137.9 / 138.0 ms either way in one call).  MDT_NN_DGRAD=1
    keeps the earlier route for A/B runs: the big launches read a transposed bf16 copy of W, cached until the weight
    changes (2 bytes per block parameter, one transpose per optimiser step)."""
    if _NN_DGRAD and dy.dtype == torch.bfloat16 and dy.shape[0] >= 8192 and w.shape[0] * w.shape[1] >= 1_500_000:
        return ops.gemm(dy, _transposed(w), **kw)
    return ops.gemm(dy, w.data, trans_b=True, **kw)


def dyd_rides(g: torch.Tensor) -> bool:
    """bias gradients of the dense layers behind a LayerNorm ride on their weight-gradient GEMM (MDT_EPI_ASUM) when that
    GEMM takes the 256 x 256 kernel (bf16, enough rows); small problems keep the sums inside the LayerNorm backward"""
    return _LN_BIAS_RIDE and g.dtype == torch.bfloat16 and g.shape[0] >= 8192


def _ln_bwd(tape, dy, x, w, b, mean, rstd, add=None):
    return ops.layernorm_bwd(dy, x, w.data, mean, rstd, add=add, dgamma=tape.pgrad(w), dbeta=tape.pgrad(b))


def _ln_bwd_dense(tape, dy, x, w, b, mean, rstd, bias_param, p_drop, seed, add=None):
    """LayerNorm backward + the fused tail for the dense layer that feeds it: returns (dx, dxd) where dxd is dx
    through that layer's hidden-dropout mask, and accumulates the layer's bias gradient (column sums of dxd)."""
    gb = tape.pgrad(bias_param) if bias_param is not None else None
    return ops.layernorm_bwd(dy, x, w.data, mean, rstd, add=add, dgamma=tape.pgrad(w), dbeta=tape.pgrad(b),
                             drop_p=p_drop, drop_seed=seed, colsum=None if gb is None else gb.view(-1),
                             want_dropped=True)


# ------------------------------------------------------------------------------------------
# ops
def transformer_block(tape: Tape, x: Var, P: BlockParams, spec: AttnSpec, *, pre_ln: bool, eps: float,
                      p_hidden: float = 0.0, p_attn: float = 0.0, p_act: float = 0.0, keep_rows=None) -> Var:
    """One encoder block (post-LN: HF BertLayer / Graphormer layer; pre-LN: HF ViTLayer or
    Graphormer with --pre-layernorm).  7 GEMM-class launches + attention + 2 LayerNorms
    forward; the adjoint below mirrors it with the residual adds folded into epilogues.
    Dropout (training): ``p_attn`` on attention probabilities (inside the attention kernel),
    ``p_hidden`` on the two dense outputs before their residual adds and ``p_act`` after GELU
    (both in the GEMM epilogue); masks are regenerated in backward from per-site seeds.  The FFN
    saves ``u`` = d h / d pre-activation (GELU' times the activation-dropout scale) rather than
    the pre-activation itself, so the backward epilogue is a single multiply.

    ``keep_rows`` (i32[R]): only these token rows of the block's OUTPUT are needed downstream (the last fusion
    layer feeds nothing but bottleneck token 0 and [CLS] of every comment to the graph / the head).  Keys and
    values still come from every row, but the output projection, both LayerNorms and the FFN — three quarters of
    the block's GEMM work — run on the R kept rows only, and the block returns [R, D] in ``keep_rows`` order.
    Exact: the dropped rows' outputs are dead values in the reference too."""
    xd = x.data
    kw = spec.kwargs()
    s_attn, s_o, s_act, s_f2 = (tape.next_seed() for _ in range(4))
    akw = dict(drop_p=p_attn, drop_seed=s_attn)
    R = None if keep_rows is None else int(keep_rows.numel())

    def gather(src):
        """rows ``keep_rows`` of src (identity when the block keeps everything)"""
        if keep_rows is None:
            return src
        out_ = torch.empty(R, src.shape[1], dtype=src.dtype, device=src.device)
        ops.row_axpby(out_, R, a=src, ai=keep_rows)
        return out_

    def spread(g_, rows):
        """adjoint of gather: a [rows, D] tensor that is zero except g_ at ``keep_rows``"""
        if keep_rows is None:
            return g_
        full = torch.zeros(rows, g_.shape[1], dtype=g_.dtype, device=g_.device)
        ops.row_axpby(full, R, di=keep_rows, a=g_)
        return full

    f8 = F8.ACTIVE

    def lin8(x_, w_, tag, **kw_):
        """the 8-bit kernel where it pays (fp8.py) and the shape allows, else None"""
        return None if f8 is None else f8.linear(x_, w_, (tag, id(w_)), **kw_)

    def ln8(x_, lw, lb, w_, tag):
        """LayerNorm whose output feeds the 8-bit GEMM (tag, w_): the fp8 copy leaves the LayerNorm kernel itself when that site
        has a scale (fp8.py producer_slot).  → (y, mean, rstd, (y8, inv scale) or None)"""
        slot = None if (f8 is None or x_.dtype != torch.bfloat16) else f8.producer_slot(x_.shape[0], w_, (tag, id(w_)))
        if slot is None:
            return (*ops.layernorm_fwd(x_, lw.data, lb.data, eps), None)
        return (*ops.layernorm_fwd(x_, lw.data, lb.data, eps, q8=slot[:4]), (slot[0], slot[4]))

    if not pre_ln:
        qkv = lin8(xd, P.qkv_w, "qkv", bias=P.qkv_b.data)
        if qkv is None:
            qkv = ops.gemm(xd, P.qkv_w.data, bias=P.qkv_b.data)
        ctx_full, lse = ops.attention_fwd(qkv, spec.nseq, spec.S, spec.H, **kw, **akw)
        ctx, xk = gather(ctx_full), gather(xd)
        t = ops.gemm(ctx, P.o_w.data, bias=P.o_b.data, residual=xk, drop_p=p_hidden, drop_seed=s_o)
        a, m1, r1, a8 = ln8(t, P.ln1_w, P.ln1_b, P.fc1_w, "fc1")
        u = torch.empty(a.shape[0], P.fc1_w.shape[0], dtype=a.dtype, device=a.device)
        h, h8 = lin8(a, P.fc1_w, "fc1", x8=a8, bias=P.fc1_b.data, aux=u, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=p_act, drop_seed=s_act,
                     q8_site=("fc2", id(P.fc2_w))) or (None, None)
        if h is None:
            h = ops.gemm(a, P.fc1_w.data, bias=P.fc1_b.data, aux=u, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=p_act, drop_seed=s_act)
        y = lin8(h, P.fc2_w, "fc2", x8=h8, bias=P.fc2_b.data, residual=a, drop_p=p_hidden, drop_seed=s_f2)   # 8-bit when fc1 handed over the quantised h
        if y is None:
            y = ops.gemm(h, P.fc2_w.data, bias=P.fc2_b.data, residual=a, drop_p=p_hidden, drop_seed=s_f2)
        out, m2, r2 = ops.layernorm_fwd(y, P.ln2_w.data, P.ln2_b.data, eps)
    else:
        n1, m1, r1, n18 = ln8(xd, P.ln1_w, P.ln1_b, P.qkv_w, "qkv")
        qkv = lin8(n1, P.qkv_w, "qkv", x8=n18, bias=P.qkv_b.data)
        if qkv is None:
            qkv = ops.gemm(n1, P.qkv_w.data, bias=P.qkv_b.data)
        ctx_full, lse = ops.attention_fwd(qkv, spec.nseq, spec.S, spec.H, **kw, **akw)
        ctx, xk = gather(ctx_full), gather(xd)
        hmid = ops.gemm(ctx, P.o_w.data, bias=P.o_b.data, residual=xk, drop_p=p_hidden, drop_seed=s_o)
        n2, m2, r2, n28 = ln8(hmid, P.ln2_w, P.ln2_b, P.fc1_w, "fc1")
        u = torch.empty(n2.shape[0], P.fc1_w.shape[0], dtype=n2.dtype, device=n2.device)
        f, f8q = lin8(n2, P.fc1_w, "fc1", x8=n28, bias=P.fc1_b.data, aux=u, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=p_act, drop_seed=s_act,
                      q8_site=("fc2", id(P.fc2_w))) or (None, None)
        if f is None:
            f = ops.gemm(n2, P.fc1_w.data, bias=P.fc1_b.data, aux=u, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=p_act, drop_seed=s_act)
        out = lin8(f, P.fc2_w, "fc2", x8=f8q, bias=P.fc2_b.data, residual=hmid, drop_p=p_hidden, drop_seed=s_f2)
        if out is None:
            out = ops.gemm(f, P.fc2_w.data, bias=P.fc2_b.data, residual=hmid, drop_p=p_hidden, drop_seed=s_f2)
    o = Var(out)

    def hdrop(g, seed):
        """gradient w.r.t. a dense output that went through hidden dropout"""
        return ops.dropout(g, p_hidden, seed) if p_hidden > 0 else g

    def attn_bwd(dctx):
        extra = {}
        if spec.sp_table is not None:
            extra = dict(d_sp_table=tape.pgrad(spec.sp_table),
                         d_virt=None if tape.pgrad(spec.virt) is None else tape.pgrad(spec.virt).view(-1))
        want_dense = spec.dense_bias_var is not None and spec.dense_bias_var.needs_grad
        dqkv, dbias = ops.attention_bwd(spread(dctx, xd.shape[0]), qkv, ctx_full, lse, spec.nseq, spec.S, spec.H, **kw,
                                        want_dense_dbias=want_dense, **extra, **akw)
        if want_dense:
            tape.add_grad(spec.dense_bias_var, dbias)
        return dqkv

    def bwd_post():
        g = o.grad
        o.grad = None
        if g is None:
            return
        ride = _WGRAD_ASUM and dyd_rides(g)     # bias gradients ride on the weight-gradient GEMMs (wgrad) where they can
        dy, dyd = _ln_bwd_dense(tape, g, y, P.ln2_w, P.ln2_b, m2, r2, None if ride else P.fc2_b, p_hidden, s_f2)
        wgrad(tape, dyd, h, P.fc2_w, P.fc2_b if ride else None)
        gb1 = tape.pgrad(P.fc1_b)           # fc1 bias gradient = colsum(du): fused into the GEMM epilogue
        du, du8 = lin8(dyd, P.fc2_w, "d_fc2", transposed_weight=True, grad=True, aux=u, epilogue=ops.EPI_MULAUX,
                       colsum=None if gb1 is None else gb1.view(-1), q8_site=("d_fc1", id(P.fc1_w)), q8_grad=True) or (None, None)
        if du is None:
            du = dgrad(dyd, P.fc2_w, aux=u, epilogue=ops.EPI_MULAUX,
                          colsum=None if gb1 is None else gb1.view(-1))
        wgrad(tape, du, a, P.fc1_w, None)
        da = lin8(du, P.fc1_w, "d_fc1", transposed_weight=True, grad=True, x8=du8, residual=dy)      # 8-bit when the GEMM above handed over the quantised du
        if da is None:
            da = dgrad(du, P.fc1_w, residual=dy)
        dt_, dtd = _ln_bwd_dense(tape, da, t, P.ln1_w, P.ln1_b, m1, r1, None if ride else P.o_b, p_hidden, s_o)
        wgrad(tape, dtd, ctx, P.o_w, P.o_b if ride else None)
        dctx = ops.gemm(dtd, P.o_w.data, trans_b=True)
        dqkv = attn_bwd(dctx)
        wgrad(tape, dqkv, xd, P.qkv_w, P.qkv_b)
        if x.needs_grad:
            tape.add_grad(x, dgrad(dqkv, P.qkv_w, residual=spread(dt_, xd.shape[0])))
        if tape.on_params_ready:
            tape.on_params_ready(P.all())

    def bwd_pre():
        g = o.grad
        o.grad = None
        if g is None:
            return
        gd = hdrop(g, s_f2)
        wgrad(tape, gd, f, P.fc2_w, P.fc2_b)
        gb1 = tape.pgrad(P.fc1_b)
        du, du8 = lin8(gd, P.fc2_w, "d_fc2", transposed_weight=True, grad=True, aux=u, epilogue=ops.EPI_MULAUX,
                       colsum=None if gb1 is None else gb1.view(-1), q8_site=("d_fc1", id(P.fc1_w)), q8_grad=True) or (None, None)
        if du is None:
            du = dgrad(gd, P.fc2_w, aux=u, epilogue=ops.EPI_MULAUX,
                          colsum=None if gb1 is None else gb1.view(-1))
        wgrad(tape, du, n2, P.fc1_w, None)
        dn2 = lin8(du, P.fc1_w, "d_fc1", transposed_weight=True, grad=True, x8=du8)
        if dn2 is None:
            dn2 = dgrad(du, P.fc1_w)
        ride = _WGRAD_ASUM and dyd_rides(g)
        dh, dhd = _ln_bwd_dense(tape, dn2, hmid, P.ln2_w, P.ln2_b, m2, r2, None if ride else P.o_b, p_hidden, s_o, add=g)
        wgrad(tape, dhd, ctx, P.o_w, P.o_b if ride else None)
        dctx = ops.gemm(dhd, P.o_w.data, trans_b=True)
        dqkv = attn_bwd(dctx)
        wgrad(tape, dqkv, n1, P.qkv_w, P.qkv_b)
        if x.needs_grad:
            dn1 = dgrad(dqkv, P.qkv_w)
            tape.add_grad(x, _ln_bwd(tape, dn1, xd, P.ln1_w, P.ln1_b, m1, r1, add=spread(dh, xd.shape[0])))
        else:   # LayerNorm parameters still need their gradients
            dn1 = dgrad(dqkv, P.qkv_w)
            _ln_bwd(tape, dn1, xd, P.ln1_w, P.ln1_b, m1, r1)
        if tape.on_params_ready:
            tape.on_params_ready(P.all())

    tape.record(bwd_pre if pre_ln else bwd_post)
    return o


def attention_layer(tape: Tape, x: Var, qkv_w, qkv_b, o_w, o_b, spec: AttnSpec, p_attn: float = 0.0, stash: Optional[dict] = None) -> Var:
    """Bare multi-head self-attention + output projection (modules/multihead_attention.py:91-214).  ``stash``: receives
    the qkv buffer and the log-sum-exp (what the need_weights=True path recomputes the probabilities from)."""
    xd = x.data
    kw = spec.kwargs()
    kw.update(drop_p=p_attn, drop_seed=tape.next_seed())
    qkv = ops.gemm(xd, qkv_w.data, bias=None if qkv_b is None else qkv_b.data)
    ctx, lse = ops.attention_fwd(qkv, spec.nseq, spec.S, spec.H, **kw)
    if stash is not None:
        stash.update(qkv=qkv, lse=lse, kw=spec.kwargs())
    out = ops.gemm(ctx, o_w.data, bias=None if o_b is None else o_b.data)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        wgrad(tape, g, ctx, o_w, o_b)
        dctx = ops.gemm(g, o_w.data, trans_b=True)
        extra = {}
        if spec.sp_table is not None:
            extra = dict(d_sp_table=tape.pgrad(spec.sp_table),
                         d_virt=None if tape.pgrad(spec.virt) is None else tape.pgrad(spec.virt).view(-1))
        want_dense = spec.dense_bias_var is not None and spec.dense_bias_var.needs_grad
        dqkv, dbias = ops.attention_bwd(dctx, qkv, ctx, lse, spec.nseq, spec.S, spec.H, **kw,
                                        want_dense_dbias=want_dense, **extra)
        if want_dense:
            tape.add_grad(spec.dense_bias_var, dbias)
        wgrad(tape, dqkv, xd, qkv_w, qkv_b)
        if x.needs_grad:
            tape.add_grad(x, ops.gemm(dqkv, qkv_w.data, trans_b=True))

    tape.record(bwd)
    return o


def layernorm(tape: Tape, x: Var, w, b, eps: float) -> Var:
    y, mean, rstd = ops.layernorm_fwd(x.data, w.data, b.data, eps)
    o = Var(y)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        dx = _ln_bwd(tape, g, x.data, w, b, mean, rstd)
        tape.add_grad(x, dx)

    tape.record(bwd)
    return o


def dropout(tape: Tape, x: Var, p: float) -> Var:
    """Stand-alone inverted dropout (embedding / pooler dropouts); identity when p == 0."""
    if p <= 0.0:
        return x
    seed = tape.next_seed()
    o = Var(ops.dropout(x.data, p, seed))

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        tape.add_grad(x, ops.dropout(g, p, seed))

    tape.record(bwd)
    return o


def bert_embeddings(tape: Tape, ids, types, word, pos, typ) -> Var:
    """word + position + token-type sum (HF BertEmbeddings before its LayerNorm)."""
    M, Lq = ids.shape
    D = word.shape[1]
    out = torch.empty(M * Lq, D, dtype=word.dtype, device=word.device)
    ops.bert_embed_sum(ids, types, word.data, pos.data, typ.data, out, out_seq_stride=Lq, out_off=0)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        gw, gp, gt = tape.pgrad(word), tape.pgrad(pos), tape.pgrad(typ)
        if gw is not None:
            ops.row_scatter_add(gw, ids.view(-1), g, M * Lq)
        if gp is not None:
            ops.colsum(g.view(M, Lq * D), out=gp[:Lq].view(-1))
        if gt is not None:
            assert typ.shape[0] == 2, "token-type gradient is implemented for type_vocab_size == 2"
            tot = ops.colsum(g)
            one = ops.colsum(g, row_weight=types.view(-1))
            ops.row_axpby(gt, 1, d_off=1, a=one.view(1, D), accumulate=True)
            ops.row_axpby(gt, 1, d_off=0, a=tot.view(1, D), b=one.view(1, D), beta=-1.0, accumulate=True)

    tape.record(bwd)
    return o


def _embed_rows_grads(tape: Tape, g, rows, ids, types, pos_ids, word, pos, typ):
    gw, gp, gt = tape.pgrad(word), tape.pgrad(pos), tape.pgrad(typ)
    D = word.shape[1]
    if gw is not None:
        ops.row_scatter_add(gw, ids, g, rows)
    if gp is not None:
        ops.row_scatter_add(gp, pos_ids, g, rows)
    if gt is not None:
        assert typ.shape[0] == 2, "token-type gradient is implemented for type_vocab_size == 2"
        tot = ops.colsum(g)
        one = ops.colsum(g, row_weight=types)
        ops.row_axpby(gt, 1, d_off=1, a=one.view(1, D), accumulate=True)
        ops.row_axpby(gt, 1, d_off=0, a=tot.view(1, D), b=one.view(1, D), beta=-1.0, accumulate=True)


def bert_embeddings_rows(tape: Tape, ids, types, pos_ids, word, pos, typ) -> Var:
    """Ragged form of ``bert_embeddings``: one row per VALID token (ids / types / pos_ids i32[rows])."""
    rows = ids.numel()
    D = word.shape[1]
    out = torch.empty(rows, D, dtype=word.dtype, device=word.device)
    ops.bert_embed_rows(ids, types, pos_ids, word.data, pos.data, typ.data, out)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        _embed_rows_grads(tape, g, rows, ids, types, pos_ids, word, pos, typ)

    tape.record(bwd)
    return o


def bert_embeddings_ln_rows(tape: Tape, ids, types, pos_ids, word, pos, typ, w, b, eps: float) -> Var:
    """``layernorm(bert_embeddings_rows(...))`` in one pass (ops.bert_embed_ln_rows): same bits; the summed rows are
    written out only when a backward pass will read them."""
    rows = ids.numel()
    y, mean, rstd, xs = ops.bert_embed_ln_rows(ids, types, pos_ids, word.data, pos.data, typ.data, w.data, b.data, eps,
                                               keep_sum=not tape.inference)
    o = Var(y)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        dx = _ln_bwd(tape, g, xs, w, b, mean, rstd)
        _embed_rows_grads(tape, dx, rows, ids, types, pos_ids, word, pos, typ)

    tape.record(bwd)
    return o


def vit_embeddings(tape: Tape, images, proj_w, proj_b, cls, pos, patch: int) -> Var:
    """Conv2d(k = s = patch) as a patch gather + one GEMM, then [CLS] and position add."""
    I = images.shape[0]
    D = proj_w.shape[0]
    g_ = images.shape[-1] // patch
    npatch = g_ * g_
    wmat = proj_w.data.view(D, -1)
    tokens = torch.empty(I * (npatch + 1), D, dtype=proj_w.dtype, device=proj_w.device)
    # One launch (the GEMM's A loader gathers from the pixels, ops.vit_patch_embed) when nothing downstream needs the gathered
    # matrix: inference, or a frozen projection (the shipped launch freezes the ViT embeddings).  A trainable projection's
    # weight gradient contracts over that matrix, so it is made once here and kept.
    fused = (PATCH_EMBED_FUSED and proj_w.dtype == torch.bfloat16 and patch == 16 and D % 128 == 0
             and (tape.inference or not (proj_w.requires_grad or proj_b.requires_grad)))
    cols = patches = None
    if fused:
        ops.vit_patch_embed(images, patch, wmat, proj_b.data, cls.data.view(-1), pos.data.view(npatch + 1, D), tokens,
                            seq_stride=npatch + 1, off=0)
    else:
        # K = C * patch^2 is 588 for ViT-L/14: zero-padded to a multiple of 128, the projection and its weight gradient run on
        # the MFMA tile kernels instead of the any-shape fp32 one (large config: 1.3 + 1.7 ms per step -> 0.4 + 0.4)
        kin = wmat.shape[1]
        kp = kin if (kin % 128 == 0 or proj_w.dtype != torch.bfloat16 or not PATCH_K_PAD) else (kin + 127) // 128 * 128
        cols = ops.vit_patchify(images, patch, proj_w.dtype, k_pad=kp)
        patches = ops.gemm(cols, _k_padded(proj_w, wmat, kp) if kp != kin else wmat, bias=proj_b.data)
        ops.vit_assemble(patches, cls.data.view(-1), pos.data.view(npatch + 1, D), tokens, I, npatch,
                         seq_stride=npatch + 1, off=0)
    o = Var(tokens)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        gw, gb, gc, gp = tape.pgrad(proj_w), tape.pgrad(proj_b), tape.pgrad(cls), tape.pgrad(pos)
        if gw is not None or gb is not None:
            dpatch = torch.empty(I * npatch, D, dtype=g.dtype, device=g.device)
            ops.row_axpby(dpatch, I * npatch, a=g, a_inner=npatch, a_stride=npatch + 1, a_off=1)
            if gw is not None:
                kin_, kp_ = gw.view(D, -1).shape[1], cols.shape[1]
                acc = gw.view(D, -1) if kp_ == kin_ else torch.zeros(D, kp_, dtype=torch.float32, device=g.device)
                ops.gemm(dpatch, cols, trans_a=True, trans_b=True, out=acc, epilogue=ops.EPI_ATOMIC,
                         split_k=_split_k(D, cols.shape[1], dpatch.shape[0]))
                if kp_ != kin_:
                    gw.view(D, -1).add_(acc[:, :kin_])
            if gb is not None:
                ops.colsum(dpatch, out=gb)
        if gp is not None:
            ops.colsum(g.view(I, (npatch + 1) * D), out=gp.view(-1))
        if gc is not None:
            ops.colsum(g.view(I, (npatch + 1) * D)[:, :D], out=gc.view(-1))

    tape.record(bwd)
    return o


def expand_sequences(tape: Tape, x: Var, nseq: int, s_in: int, n_front: int, front: Optional[torch.nn.Parameter]) -> Var:
    """[nseq*s_in, D] → [nseq*(n_front+s_in), D] with ``n_front`` rows prepended to every sequence:
    the learned bottleneck tokens (``front`` = bottle_neck.weight, modules/multigraphormer_graph_encoder.py:339)
    or zeros (image sequences, whose bottleneck rows are refreshed before every fusion layer)."""
    D = x.data.shape[1]
    S = n_front + s_in
    out = torch.empty(nseq * S, D, dtype=x.data.dtype, device=x.data.device)
    ops.row_axpby(out, nseq * s_in, d_inner=s_in, d_stride=S, d_off=n_front, a=x.data)
    if front is not None:
        ops.row_axpby(out, nseq * n_front, d_inner=n_front, d_stride=S, d_off=0, a=front.data, a_inner=n_front,
                      a_stride=0, a_off=0)
    else:
        ops.row_axpby(out, nseq * n_front, d_inner=n_front, d_stride=S, d_off=0)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        if front is not None and tape.pgrad(front) is not None:
            ops.colsum(g.view(nseq, S * D)[:, : n_front * D], out=tape.pgrad(front).view(-1))
        if x.needs_grad:
            dx = torch.empty_like(x.data)
            ops.row_axpby(dx, nseq * s_in, a=g, a_inner=s_in, a_stride=S, a_off=n_front)
            tape.add_grad(x, dx)

    tape.record(bwd)
    return o


def expand_rows(tape: Tape, x: Var, rows_out: int, body_idx, front_idx, n_front: int, front: torch.nn.Parameter) -> Var:
    """Index-driven ``expand_sequences`` for the text side (padded or ragged layout alike): row r of ``x`` goes to
    row ``body_idx[r]`` of the [rows_out, D] result, and row ``front_idx[k]`` receives learned bottleneck token
    ``k % n_front`` (modules/multigraphormer_graph_encoder.py:339)."""
    D = x.data.shape[1]
    rows_in = x.data.shape[0]
    nfront = front_idx.numel()
    assert rows_in + nfront == rows_out, "expand_rows: body and front rows must tile the output exactly"
    out = torch.empty(rows_out, D, dtype=x.data.dtype, device=x.data.device)
    ops.row_axpby(out, rows_in, di=body_idx, a=x.data)
    ops.row_axpby(out, nfront, di=front_idx, a=front.data, a_inner=n_front, a_stride=0, a_off=0)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        gf = tape.pgrad(front)
        if gf is not None:
            fr = torch.empty(nfront, D, dtype=g.dtype, device=g.device)
            ops.row_axpby(fr, nfront, a=g, ai=front_idx)
            ops.colsum(fr.view(nfront // n_front, n_front * D), out=gf.view(-1))
        if x.needs_grad:
            dx = torch.empty_like(x.data)
            ops.row_axpby(dx, rows_in, a=g, ai=body_idx)
            tape.add_grad(x, dx)

    tape.record(bwd)
    return o


def rows_mix(tape: Tape, dst: Var, src: Var, nrows: int, *, alpha: float, beta: float, d_idx=None, d_map=(1, 1, 0),
             s_idx=None, s_map=(1, 1, 0)) -> Var:
    """In place: dst[dr] = alpha * src[sr] + beta * dst[dr] for ``nrows`` row pairs.
    alpha=1, beta=0 is the reference's ``dst[index] = src[index2]`` assignment; alpha=beta=0.5 the
    image/text bottleneck average (modules/multi_graphormer_fusion_layer.py:63-66).
    Adjoint (also in place on the gradient buffers): g_src[sr] += alpha * g_dst[dr]; g_dst[dr] *= beta."""
    di, ds_, do = d_map
    si, ss, so = s_map
    ops.row_axpby(dst.data, nrows, di=d_idx, d_inner=di, d_stride=ds_, d_off=do,
                  a=src.data, ai=s_idx, a_inner=si, a_stride=ss, a_off=so, alpha=alpha,
                  b=dst.data if beta != 0.0 else None, bi=d_idx, b_inner=di, b_stride=ds_, b_off=do, beta=beta)

    def bwd():
        g = dst.grad
        if g is None:
            return
        if src.needs_grad:
            gs = tape.grad_buf(src)
            ops.row_axpby(gs, nrows, di=s_idx, d_inner=si, d_stride=ss, d_off=so,
                          a=g, ai=d_idx, a_inner=di, a_stride=ds_, a_off=do, alpha=alpha, accumulate=True)
        ops.row_axpby(g, nrows, di=d_idx, d_inner=di, d_stride=ds_, d_off=do,
                      a=g if beta != 0.0 else None, ai=d_idx, a_inner=di, a_stride=ds_, a_off=do, alpha=beta)

    tape.record(bwd)
    return dst


def graph_node_features(tape: Tape, text: Var, text_row_of_node, in_degree, out_degree, in_emb, out_emb, graph_token,
                        B: int, T: int, src_rows_of_comment, graph_rows_of_comment, M: int, in_scatter_idx,
                        out_scatter_idx) -> Var:
    """modules/graphormer_layers.py:39-50 fused with the masked scatter that builds graph_data
    (modules/multigraphormer_graph_encoder.py:363-371): node n of tree b reads bottleneck token 0
    of its comment (row ``text_row_of_node``, -1 = padding node → zeros), both add the degree
    embeddings.  ``*_scatter_idx`` i32[B*T]: embedding row that graph row r feeds in backward,
    -1 for the graph token and for padding_idx (degree 0) rows."""
    x = ops.graph_node_feature(text.data, text_row_of_node, in_degree, out_degree, in_emb.data, out_emb.data,
                               graph_token.data.view(-1), B, T)
    o = Var(x)
    D = x.shape[1]

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        if text.needs_grad:
            gt = tape.grad_buf(text)
            ops.row_axpby(gt, M, di=src_rows_of_comment, a=g, ai=graph_rows_of_comment, accumulate=True)
        gi, go, gk = tape.pgrad(in_emb), tape.pgrad(out_emb), tape.pgrad(graph_token)
        if gi is not None:
            ops.row_scatter_add(gi, in_scatter_idx, g, B * T)
        if go is not None:
            ops.row_scatter_add(go, out_scatter_idx, g, B * T)
        if gk is not None:
            ops.colsum(g.view(B, T * D)[:, :D], out=gk.view(-1))

    tape.record(bwd)
    return o


def classifier_head(tape: Tape, text: Var, M: int, cls_rows, bn0_rows, pool_w, pool_b, cls_w, cls_b, p_drop: float = 0.0) -> Var:
    """models/multi_modal_discussion_transformer.py:265-274: the SAME pooler (dense + tanh on
    token 0) and classifier are applied to the text sequence ([CLS], row nb of each comment)
    and to the bottleneck sequence (bottleneck token 0, row 0); logits are their mean.
    ``cls_rows`` / ``bn0_rows``: i32[M] rows of [CLS] / bottleneck token 0 of every comment in ``text``."""
    D = text.data.shape[1]
    rows = torch.empty(2 * M, D, dtype=text.data.dtype, device=text.data.device)
    ops.row_axpby(rows, M, d_off=0, a=text.data, ai=cls_rows)
    ops.row_axpby(rows, M, d_off=M, a=text.data, ai=bn0_rows)
    pre = ops.gemm(rows, pool_w.data, bias=pool_b.data)
    pooled = ops.tanh_fwd(pre)
    seed = tape.next_seed()
    pooled_d = ops.dropout(pooled, p_drop, seed) if p_drop > 0 else pooled   # text_dropout, independent masks per branch
    l2 = ops.gemm(pooled_d, cls_w.data, bias=cls_b.data)                     # [2M, C]
    C = l2.shape[1]
    logits = torch.empty(M, C, dtype=l2.dtype, device=l2.device)
    ops.row_axpby(logits, M, a=l2, alpha=0.5, b=l2, b_off=M, beta=0.5)
    o = Var(logits)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None:
            return
        dl2 = torch.empty_like(l2)
        ops.row_axpby(dl2, M, d_off=0, a=g, alpha=0.5)
        ops.row_axpby(dl2, M, d_off=M, a=g, alpha=0.5)
        wgrad(tape, dl2, pooled_d, cls_w, cls_b)
        dpooled = ops.gemm(dl2, cls_w.data, trans_b=True)
        if p_drop > 0:
            dpooled = ops.dropout(dpooled, p_drop, seed)
        dpre = ops.tanh_bwd(pooled, dpooled)
        wgrad(tape, dpre, rows, pool_w, pool_b)
        if text.needs_grad:
            drows = ops.gemm(dpre, pool_w.data, trans_b=True)
            gt = tape.grad_buf(text)
            ops.row_axpby(gt, M, di=cls_rows, a=drows, a_off=0, accumulate=True)
            ops.row_axpby(gt, M, di=bn0_rows, a=drows, a_off=M, accumulate=True)
        if tape.on_params_ready:
            tape.on_params_ready([pool_w, pool_b, cls_w, cls_b])

    tape.record(bwd)
    return o


def take_rows(tape: Tape, src: Var, nrows: int, s_map=(1, 1, 0), s_idx=None) -> Var:
    """out[r] = src[row(r)] (e.g. the global embedding = graph-token row of every tree)."""
    si, ss, so = s_map
    out = torch.empty(nrows, src.data.shape[1], dtype=src.data.dtype, device=src.data.device)
    ops.row_axpby(out, nrows, a=src.data, ai=s_idx, a_inner=si, a_stride=ss, a_off=so)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None or not src.needs_grad:
            return
        gs = tape.grad_buf(src)
        ops.row_axpby(gs, nrows, di=s_idx, d_inner=si, d_stride=ss, d_off=so, a=g, accumulate=True)

    tape.record(bwd)
    return o


def scatter_rows(tape: Tape, src: Var, rows_out: int, d_idx, s_idx) -> Var:
    """out = zeros[rows_out, D]; out[d_idx[r]] = src[s_idx[r]] (ragged rows back into the reference's padded shape)."""
    n = d_idx.numel()
    out = torch.zeros(rows_out, src.data.shape[1], dtype=src.data.dtype, device=src.data.device)
    ops.row_axpby(out, n, di=d_idx, a=src.data, ai=s_idx)
    o = Var(out)

    def bwd():
        g = o.grad
        o.grad = None
        if g is None or not src.needs_grad:
            return
        gs = tape.grad_buf(src)
        ops.row_axpby(gs, n, di=s_idx, a=g, ai=d_idx, accumulate=True)

    tape.record(bwd)
    return o


# ------------------------------------------------------------------------------------------
# autograd bridge
class TapeFunction(torch.autograd.Function):
    """One autograd node around a whole tape.

    ``TapeFunction.apply(run, use_main_grad, hook, inference, n_in, *tensors)`` where ``tensors`` are the
    ``n_in`` differentiable tensor inputs followed by every parameter the tape may touch;
    ``run(tape, *input_vars) -> tuple[Var]``."""

    @staticmethod
    def forward(ctx, run, use_main_grad, hook, inference, n_in, *tensors):
        ctx.set_materialize_grads(False)
        tape = Tape(use_main_grad=use_main_grad, on_params_ready=hook, inference=inference)
        ins = [Var(t, needs_grad=t.requires_grad) for t in tensors[:n_in]]
        try:
            outs = run(tape, *ins)
        finally:
            tape._leave_side()
        tape.sync_streams()
        if inference and F8.ACTIVE is not None:
            F8.ACTIVE.end_of_step()      # no backward will close this pass: its |x| maxima become the next scales here (and are reset)
        ctx.tape, ctx.ins, ctx.outs = tape, ins, outs
        ctx.params = tensors[n_in:]
        ctx.n_in = n_in
        return tuple(o.data for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        tape = ctx.tape
        for o, g in zip(ctx.outs, gouts):
            if g is not None:
                g = g.contiguous()
                o.grad = g if o.grad is None else o.grad + g
        tape.backward()
        gin = [v.grad if v.needs_grad else None for v in ctx.ins]
        if tape.use_main_grad:
            gp = [None] * len(ctx.params)
        else:
            gp = []
            for p in ctx.params:
                g = tape.tmp_grads.get(id(p))
                gp.append(None if g is None else (g if p.dtype == torch.float32 else ops.cast(g, p.dtype)))
        ctx.tape = ctx.ins = ctx.outs = None
        return (None, None, None, None, None, *gin, *gp)


def run_tape(run, inputs: Sequence[torch.Tensor], params: Sequence[torch.nn.Parameter], *, use_main_grad=False, hook=None):
    """Execute ``run`` under the autograd bridge; returns a tuple of output tensors."""
    probe = inputs[0] if len(inputs) else params[0]
    if not probe.is_cuda:
        raise RuntimeError("the mDT HIP path has no CPU fallback: move the module and its inputs to the GPU")
    # grad mode is read HERE: inside Function.forward it is always off
    inference = not torch.is_grad_enabled()
    return TapeFunction.apply(run, use_main_grad, hook, inference, len(inputs), *inputs, *params)
