"""Data-parallel gradient exchange for mDT: discussion trees are sharded across the GPUs of one
node (one process per GPU), the only exchange step is the sum-all-reduce of parameter
gradients (RCCL over xGMI through ``torch.distributed``, backend "nccl" on ROCm) plus one
tiny all-reduce of the logging scalars.

The reference has no DDP code (FairSeq's trainer does it, ``--distributed-world-size``,
mDT/experiments/hateful_discussions/run_train.sh:52); FairSeq's convention is reproduced:
gradients are summed over ranks and divided by the global sample size (the criterion uses
``reduction="sum"`` and summable logging outputs, criterions/hatespeech_loss.py:116,175-182).

Design for xGMI (point-to-point links, ring collectives are per-link bound): few, large
buckets.  All trainable gradients live in ONE flat fp32 arena (``GraphormerModel.
prepare_main_grads``); after the first backward the arena is re-laid-out in the order the tape
finished each parameter group, so every bucket is a contiguous slice that becomes final
early — it is all-reduced in place on a side stream while backward continues (no copies, no
per-parameter hooks, statically-dead parameters never enter the arena's hot prefix).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def balance_trees(n_nodes: List[int], n_images: List[int], world: int, text_tokens: int = 104, image_tokens: int = 201) -> List[List[int]]:
    """Deal the trees of a global batch to ``world`` ranks by token cost: tree i costs
    ``N_i * (L + nb) + I_i * (P + nb)`` encoder tokens (an image comment is ~3x a text comment).  Greedy longest-
    processing-time assignment: trees in decreasing cost (ties by index) each go to the rank with the smallest load so far
    (ties by rank) — deterministic, so every rank computes the same deal from the same seeded batch without any
    communication.  Returns, per rank, the tree indices in their original order."""
    cost = [int(n) * text_tokens + int(i) * image_tokens for n, i in zip(n_nodes, n_images)]
    order = sorted(range(len(cost)), key=lambda k: (-cost[k], k))
    load = [0] * world
    share: List[List[int]] = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        share[r].append(k)
        load[r] += cost[k]
    return [sorted(sh) for sh in share]


class GradientBucketer:
    """Backend-agnostic core (works on CPU tensors with gloo — that is how it is tested)."""

    def __init__(self, params: List[torch.nn.Parameter], flat: torch.Tensor, bucket_bytes: int = 64 << 20,
                 process_group=None, comm_stream: Optional["torch.cuda.Stream"] = None, wire_dtype: Optional[torch.dtype] = None):
        self.params = [p for p in params if p.requires_grad]
        self.flat = flat
        # what travels over xGMI: the fp32 arena itself (default: the sum over ranks is then exactly what one process
        # would have accumulated) or a bf16 copy of each bucket (half the bytes; FairSeq's --fp16 trainer all-reduces
        # half-precision gradients too) that is summed by RCCL and added back into the fp32 arena
        self.wire_dtype = wire_dtype if wire_dtype not in (None, torch.float32) else None
        self.bucket_events = []
        self.backward_end = None
        self.last_launch_log = []
        self.defer = False         # verification: hold every bucket back until finish() (no overlap with backward)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.pg = process_group
        self.comm_stream = comm_stream
        self.compute_streams: List["torch.cuda.Stream"] = []    # beside the current one (two-stream tapes: engine.side_stream)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # MDT_DDP_FORCE=1: issue the collectives even at world size 1 (exercises the RCCL / side-stream path
        # on a single-GPU box; an all-reduce over one rank is the identity)
        import os
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MDT_DDP_FORCE") == "1")
        self.order_observed: List[int] = []       # ids in completion order (first backward)
        self.layout_final = False
        self.sync_this_backward = True
        self.aligned = flat.numel() == sum((p.numel() + 63) // 64 * 64 for p in self.params)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.bucket_ends = []
        self._assign_views(self.params)
        self.reset()

    # -- layout -------------------------------------------------------------------------
    def _assign_views(self, ordered):
        off = 0
        self.slots = []
        for p in ordered:
            n = p.numel()
            p.main_grad = self.flat[off:off + n].view(p.shape)
            span = (n + 63) // 64 * 64 if self.aligned else n       # 256-byte aligned slots
            self.slots.append((id(p), off, span))
            off += span
        assert off == self.flat.numel(), (off, self.flat.numel())
        self._pos = {pid: i for i, (pid, _, _) in enumerate(self.slots)}

    def finalize_layout(self):
        """Re-lay the arena out in observed completion order (parameters never reported go last).
        Gradient values already in the arena are carried over to the new positions.
        Every rank must end up with the SAME layout and the same bucket boundaries — a rank whose first batch
        had no image comment observes a different order — so rank 0's order is broadcast and adopted by all."""
        seen = set()
        order_idx = []
        for pid in self.order_observed:
            if pid not in seen and pid in self._index:
                seen.add(pid)
                order_idx.append(self._index[pid])
        order_idx += [i for i, p in enumerate(self.params) if id(p) not in seen]
        if self.world > 1:
            t = torch.tensor(order_idx, dtype=torch.int64, device=self.flat.device)
            dist.broadcast(t, 0, group=self.pg)
            order_idx = [int(i) for i in t.tolist()]
        ordered = [self.params[i] for i in order_idx]
        old = self.flat.clone()
        old_slots = {pid: off for pid, off, _ in self.slots}
        self._assign_views(ordered)
        for p in ordered:
            off = old_slots[id(p)]
            p.main_grad.view(-1).copy_(old[off:off + p.numel()])
        # static buckets: whole slots, at least bucket_elems each, the same on every rank
        self.bucket_ends = []
        start = 0
        for _, off, span in self.slots:
            if off + span - start >= self.bucket_elems:
                self.bucket_ends.append(off + span)
                start = off + span
        if not self.bucket_ends or self.bucket_ends[-1] != self.flat.numel():
            self.bucket_ends.append(self.flat.numel())
        self.layout_final = True
        self.reset()

    # -- per-step state -------------------------------------------------------------------
    def reset(self):
        self._pending_wire = []
        self.launch_log = []       # (start, end) element ranges in the order they were handed to all_reduce this step
        self.ready = [False] * len(self.slots)
        self.cursor = 0            # slots [0, cursor) are final
        self.launched = 0          # elements [0, launched) already handed to all_reduce
        self.next_bucket = 0       # index into bucket_ends of the next bucket to launch
        self.handles = []

    def on_params_ready(self, params):
        """Tape hook: the gradients of ``params`` are final for this step."""
        if not self.sync_this_backward:      # gradient accumulation: only the last micro-batch is reduced
            return
        for p in params:
            i = self._pos.get(id(p))
            if i is None:
                continue
            if not self.layout_final:
                self.order_observed.append(id(p))
            self.ready[i] = True
        while self.cursor < len(self.slots) and self.ready[self.cursor]:
            self.cursor += 1
        if self.active and self.layout_final and not self.defer:
            # launch every static bucket that is now entirely final — the SEQUENCE of collectives (sizes and
            # order) is fixed by bucket_ends, only its timing depends on this rank's batch
            end = self.slots[self.cursor - 1][1] + self.slots[self.cursor - 1][2] if self.cursor else 0
            while self.next_bucket < len(self.bucket_ends) and self.bucket_ends[self.next_bucket] <= end:
                self._launch(self.launched, self.bucket_ends[self.next_bucket])
                self.next_bucket += 1

    def _backend_is_stream_ordered(self) -> bool:
        """RCCL / NCCL work handles are stream-ordered: ``wait()`` returns at once on the host and makes the CURRENT stream
        wait for the collective.  gloo's ``wait()`` blocks the host until the result is there."""
        if getattr(self, "_stream_ordered", None) is None:
            try:
                self._stream_ordered = dist.is_initialized() and "nccl" in str(dist.get_backend(self.pg)).lower()
            except (RuntimeError, ValueError):
                self._stream_ordered = False
        return self._stream_ordered

    def _reduce(self, chunk: torch.Tensor):
        """One bucket.  Called with the comm stream current (GPU) or plainly (CPU / gloo).  On a stream-ordered backend the
        handle is waited for right here — that only orders the COMM stream behind RCCL's internal stream, the host goes
        on issuing backward — so everything enqueued on the comm stream afterwards (the bf16 copy-back below, the
        bucket's closing event, finally the compute stream's ``wait_stream``) sees the reduced bytes.  (Round 2 waited
        with the compute stream current and then copied on the comm stream, which had never been ordered behind the
        collective: a race gloo's blocking wait could not show.)"""
        import os
        ordered = self.comm_stream is not None and self._backend_is_stream_ordered() and os.environ.get("MDT_DDP_EARLY_WAIT", "1") != "0"
        if self.wire_dtype is None:
            h = dist.all_reduce(chunk, group=self.pg, async_op=True)
            if ordered:
                h.wait()
            else:
                self.handles.append(h)
            return
        wire = chunk.to(self.wire_dtype)
        h = dist.all_reduce(wire, group=self.pg, async_op=True)
        if ordered:
            h.wait()
            chunk.copy_(wire)
        else:
            self.handles.append(h)
            self._pending_wire.append((h, chunk, wire))

    def _launch(self, a: int, b: int):
        if b <= a:
            return
        chunk = self.flat[a:b]
        if self.comm_stream is not None:
            self._comm_waits_for_compute()
            with torch.cuda.stream(self.comm_stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._reduce(chunk)
                e1.record()           # stream-ordered backend: behind the collective (and the copy-back); gloo: behind its issue
                self.bucket_events.append((e0, e1, (b - a) * 4))
        else:
            self._reduce(chunk)
        self.launched = b
        self.launch_log.append((a, b))

    def finish(self, scalars: Optional[torch.Tensor] = None, sample_size_index: int = 1, fold_scale: bool = False):
        """End of backward: reduce what is left, wait, scale by 1 / global sample size.
        ``scalars`` (fp32 vector: loss, sample_size, counters...) is summed over ranks in place.
        ``fold_scale``: do not touch the arena (a read + write of every gradient) but return the device scalar
        1 / max(sample size, 1) for the optimiser to apply while it reads the gradients (``FusedAdam.step(grad_scale=...)``);
        the arena then holds the SUM over samples and ranks."""
        if self.active:
            self.bucket_events = self.bucket_events[-64:]
            if self.comm_stream is not None:
                self.backward_end = torch.cuda.Event(enable_timing=True)
                self.backward_end.record()                          # on the compute stream: where backward's kernels end
            if self.layout_final:
                while self.next_bucket < len(self.bucket_ends):     # buckets whose parameters never reported
                    self._launch(self.launched, self.bucket_ends[self.next_bucket])
                    self.next_bucket += 1
            else:
                self._launch(0, self.flat.numel())                  # first step: one all-reduce of the whole arena
            if scalars is not None:
                if self.comm_stream is None:
                    self.handles.append(dist.all_reduce(scalars, group=self.pg, async_op=True))
                else:
                    self._scalar_reduce(scalars)
            for h in self.handles:
                h.wait()
            for _, chunk, wire in self._pending_wire:       # (host-blocking backends) reduced low-precision copies back into the fp32 arena
                chunk.copy_(wire)                            # on the current stream, which the wait above has ordered
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        scale = None
        if scalars is not None:
            if fold_scale:
                scale = scalars[sample_size_index].clamp(min=1.0).reciprocal()
            else:
                self.flat.div_(scalars[sample_size_index].clamp(min=1.0))
        self.last_launch_log = list(self.launch_log)
        self.reset()
        return scale

    def _comm_waits_for_compute(self):
        """The gradients of a bucket were written by kernels on the current stream and, with a two-stream tape, on the
        other branch's stream: the collective waits for both."""
        cur = torch.cuda.current_stream()
        for s in [cur] + [s for s in self.compute_streams if s != cur]:
            self.comm_stream.wait_stream(s)

    def _scalar_reduce(self, scalars):
        self._comm_waits_for_compute()
        with torch.cuda.stream(self.comm_stream):
            h = dist.all_reduce(scalars, group=self.pg, async_op=True)
            if self._backend_is_stream_ordered():
                h.wait()
            else:
                self.handles.append(h)


class DataParallel:
    """Wrap a ``GraphormerModel`` for one-process-per-GPU data parallelism."""

    def __init__(self, model, bucket_mb: int = 64, process_group=None, wire_dtype: Optional[torch.dtype] = None):
        import os
        self.model = model
        flat = model.prepare_main_grads()
        ge = model.encoder.graph_encoder
        params = [p for p in model.parameters() if p.requires_grad and hasattr(p, "main_grad")]
        stream = torch.cuda.Stream() if flat.is_cuda else None
        if wire_dtype is None and os.environ.get("MDT_DDP_WIRE", "").lower() in ("bf16", "bfloat16"):
            wire_dtype = torch.bfloat16
        self.bucketer = GradientBucketer(params, flat, bucket_mb << 20, process_group, stream, wire_dtype=wire_dtype)
        # RCCL's kernels run beside backward and take whole CUs (a ping-pong GEMM workgroup leaves no LDS for a neighbour).
        # The persistent GEMM's STATIC tile walk assumes one resident workgroup per CU: a workgroup whose CU is busy
        # elsewhere starts late and finishes its statically assigned tiles late.  Its dynamic tile queue
        # (MDT_GEMM_DYNAMIC=1) lets such a CU simply take fewer tiles, but costs 1.5 % on an idle chip and 4 % once the
        # two branch streams overlap (155.1 against 149.1 ms, RCCL path forced on one GPU) — more than the 3-7 % of a
        # step during which the ~13 bucket all-reduces are resident at all, and with two streams the other branch's
        # kernels fill a CU that waits.  The static walk therefore stays the default at every world size; the switch
        # is for a node where the collectives turn out to be slow.
        import os as _os
        if flat.is_cuda and _os.environ.get("MDT_GEMM_DYNAMIC", "0") not in ("", "0"):
            from . import _lib
            _lib.enable_dynamic_tile_queue(flat.device)       # the queue heads live in memory this process owns
        if flat.is_cuda and getattr(ge, "two_streams", False):
            from . import engine
            self.bucketer.compute_streams = [torch.cuda.current_stream(), engine.side_stream(flat.device)]
        ge.grad_ready_hook = self.bucketer.on_params_ready
        self._steps = 0

    def zero_grad(self):
        self.model.main_grad_flat.zero_()

    def accumulate(self, last_micro_batch: bool):
        """--update-freq: call before each micro-batch's backward; overlapped all-reduce only on the last one."""
        self.bucketer.sync_this_backward = bool(last_micro_batch)

    def finish_backward(self, logging_scalars: Optional[torch.Tensor] = None, fold_scale: bool = False):
        """``fold_scale``: see ``GradBucketer.finish`` — returns the gradient scale to hand to the optimiser's step."""
        scale = self.bucketer.finish(logging_scalars, fold_scale=fold_scale)
        self._steps += 1
        if not self.bucketer.layout_final:
            self.bucketer.finalize_layout()       # after the first backward: completion-ordered buckets
        return scale

    def broadcast_parameters(self, src: int = 0, chunk_bytes: int = 256 << 20):
        """Rank ``src``'s parameters and buffers to everyone: a few large flat broadcasts (one per dtype and per
        ``chunk_bytes``) instead of one collective per tensor (~660 tensors at the launch configuration)."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        seen, groups = set(), {}
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if id(t) in seen:
                continue
            seen.add(id(t))
            groups.setdefault(t.dtype, []).append(t.data)
        self.broadcast_calls = 0
        for dtype, ts in groups.items():
            batch, size = [], 0
            for t in ts + [None]:
                if t is not None and (not batch or size + t.numel() * t.element_size() <= chunk_bytes):
                    batch.append(t)
                    size += t.numel() * t.element_size()
                    continue
                flat = torch.cat([b.reshape(-1) for b in batch])
                dist.broadcast(flat, src, group=self.bucketer.pg)
                self.broadcast_calls += 1
                off = 0
                for b in batch:
                    b.copy_(flat[off:off + b.numel()].view(b.shape))
                    off += b.numel()
                batch, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
        from . import engine
        engine.weights_changed()          # written through .data: cached transposed copies (engine.dgrad) are stale

    def replicas_checksum(self) -> dict:
        """After a finished exchange every rank must hold the SAME bytes in its gradient arena (an all-reduce hands every
        rank the identical sum; the scale is the same all-reduced scalar).  Position-weighted checksums of the arena and of
        the parameters are gathered and compared exactly: two ranks that issued their buckets in a different order — which
        sums one rank's bucket i with the other's bucket j and is invisible to each rank alone — disagree here."""
        b = self.bucketer
        f = b.flat.double()
        n = f.numel()
        w = torch.arange(n, device=f.device, dtype=torch.float64).remainder_(8191.0).add_(1.0)
        psum = torch.zeros((), dtype=torch.float64, device=f.device)
        for p in self.model.parameters():
            psum += p.detach().double().sum()
        digest = torch.stack([f.sum(), (f * w).sum(), f.abs().sum(), psum])
        world = dist.get_world_size(b.pg) if dist.is_initialized() else 1
        if world > 1:
            both = [torch.zeros_like(digest) for _ in range(world)]
            dist.all_gather(both, digest, group=b.pg)
        else:
            both = [digest]
        equal = all(torch.equal(both[0], x) for x in both[1:])
        worst = max(float((both[0] - x).abs().max()) for x in both) if world > 1 else 0.0
        return dict(replicas_equal=bool(equal), checksum_max_abs_diff=worst, arena_abs_sum=float(both[0][2]), finite=bool(torch.isfinite(both[0]).all()))

    def verify_exchange(self, run_step) -> dict:
        """Self-check of the overlapped, bucketed exchange against the simplest thing that can be right.  ``run_step()``
        must run one full step (zero_grad → forward → backward → finish_backward) on the SAME batch with the same seeds
        each time it is called.  Three runs:
          1. as configured (buckets launched from inside backward, on the comm stream);
          2. every bucket held back until backward has finished (same sequence of collectives, no overlap) — a missing
             stream dependency in 1 (a bucket reduced before its last gradient kernel finished) shows as a difference;
          3. buckets held back AND the arena exchanged as ONE flat all-reduce — wrong bucket boundaries show here.
        Dropout seeds and batch being equal, the runs differ only by the order of fp32 atomic adds inside backward."""
        b = self.bucketer
        out = {}
        run_step()
        g1 = b.flat.clone()
        out["overlapped_launches"] = len(b.last_launch_log)
        out.update(self.replicas_checksum())
        b.defer = True
        try:
            run_step()
            g2 = b.flat.clone()
            ends = b.bucket_ends
            b.bucket_ends = [b.flat.numel()]
            try:
                run_step()
            finally:
                b.bucket_ends = ends
            g3 = b.flat
            den = float(g3.norm()) + 1e-30
            out["overlapped_vs_deferred_rel_l2"] = float((g1 - g2).norm()) / den
            out["bucketed_vs_flat_rel_l2"] = float((g2 - g3).norm()) / den
        finally:
            b.defer = False
        out["ok"] = bool(out["replicas_equal"] and out["finite"] and out["overlapped_vs_deferred_rel_l2"] < 1e-4 and out["bucketed_vs_flat_rel_l2"] < 1e-4)
        if not out["ok"]:
            # name the parameters that differ most between the runs (what a failure report needs first)
            names = {id(p): n for n, p in self.model.named_parameters()}
            rows = []
            for pid, off, span in b.slots:
                a1, a2, a3 = g1[off:off + span], g2[off:off + span], g3[off:off + span]
                d = float(a3.norm()) + 1e-30
                rows.append((max(float((a1 - a2).norm()), float((a2 - a3).norm())) / d, names.get(pid, "?"), float(a3.norm())))
            rows.sort(reverse=True)
            out["worst_parameters"] = [(n, round(r, 6), float(f"{gn:.3g}")) for r, n, gn in rows[:8]]
        return out

    def diagnostics(self) -> dict:
        """What the first multi-GPU run should print about itself (bench.py puts it into its JSON line)."""
        b = self.bucketer
        ends = b.bucket_ends or [b.flat.numel()]
        ev = b.bucket_events[-len(ends):]
        times = []
        overlap = None
        try:
            for e0, e1, n in ev:
                e1.synchronize()
                times.append(round(e0.elapsed_time(e1), 3))
            if ev and b.backward_end is not None and len(ev) == len(ends):
                # the part of the exchange that backward did not hide: from the end of backward's last kernel (compute stream)
                # to the end of the last bucket (comm stream), against the time the collectives took in all
                exposed = max(0.0, b.backward_end.elapsed_time(ev[-1][1]))
                span = max(1e-6, ev[0][0].elapsed_time(ev[-1][1]))
                busy = sum(times)
                overlap = dict(collective_ms_sum=round(busy, 3), first_launch_to_last_end_ms=round(span, 3),
                               exposed_after_backward_ms=round(exposed, 3), overlap_frac=round(max(0.0, 1.0 - exposed / max(busy, 1e-6)), 4))
        except RuntimeError:
            pass
        sizes = [round((e - s) * 4 / 2 ** 20, 1) for s, e in zip([0] + ends[:-1], ends)]
        world = dist.get_world_size(b.pg) if dist.is_initialized() else 1
        backend = str(dist.get_backend(b.pg)) if dist.is_initialized() else None
        return dict(arena_mb=round(b.flat.numel() * 4 / 2 ** 20, 1), buckets=len(ends), bucket_mb=sizes,
                    wire_dtype=str(b.wire_dtype or torch.float32).replace("torch.", ""), layout_final=bool(b.layout_final),
                    world_seen_by_backend=world, backend_seen=backend, stream_ordered_waits=bool(b.comm_stream is not None and b._backend_is_stream_ordered()),
                    bucket_allreduce_ms=times, overlap=overlap, launches_last_step=len(b.last_launch_log), steps=self._steps,
                    broadcast_calls=getattr(self, "broadcast_calls", 0))
