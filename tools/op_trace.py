"""Which launch is the first to differ?  Training step 0 of the fp8 model (or the bf16 one with --bf16) from identical weights, batch
and seeds, `reps` times in one process; every call into `ops` (the C-ABI wrappers) is traced: clones of all its tensor arguments
before and after the call in repetition 0, compared on the fly in the later repetitions.  Prints the calls whose results differ
by more than what fp32 atomics explain, whether their inputs still agreed (suspicious) or an output buffer already differed before
the call (unwritten rows a later call fills), and where in the tensor the difference sits.
--poison: fresh allocations are filled with 0 in repetition 0 and with NaN / 1e30 / -1e30 / 65504 in the later ones — a launch that reads a
buffer nobody wrote differs on EVERY device then.
--steps=N: N consecutive training steps per repetition (the fp8 model's producer-quantised operands start at its second step).
GPU box only:  python tools/op_trace.py [reps] [--bf16] [--alone] [--one-stream] [--poison] [--steps=N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import fp8, ops  # noqa: E402
from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy  # noqa: E402
from multimodaldiscussiontransformer_amd.data.packer import pack_batch  # noqa: E402
from multimodaldiscussiontransformer_amd.models import GraphormerModel  # noqa: E402
from tests.test_oracle_golden import full_case  # noqa: E402
from tests.util_model import fill_hash_weights, model_args  # noqa: E402

TOL = 1e-5          # relative to the tensor's max |value|: fp32 atomics in another order stay far below


class Tracer:
    def __init__(self):
        self.ref = None          # repetition 0: list of (name, before clones, after clones)
        self.cur = []
        self.idx = 0
        self.found = []
        self.recording = False

    def tensors(self, args, kwargs):
        out = []
        for k, v in list(enumerate(args)) + sorted(kwargs.items(), key=lambda kv: str(kv[0])):
            if isinstance(v, torch.Tensor) and v.is_cuda:
                out.append((str(k), v))
            elif isinstance(v, (tuple, list)):
                for j, w in enumerate(v):
                    if isinstance(w, torch.Tensor) and w.is_cuda:
                        out.append((f"{k}[{j}]", w))
        return out

    def wrap(self, name, fn):
        def inner(*args, **kwargs):
            if not self.recording:
                return fn(*args, **kwargs)
            ts = self.tensors(args, kwargs)
            before = [(k, t.detach().clone()) for k, t in ts]
            res = fn(*args, **kwargs)
            rts = self.tensors((res,) if not isinstance(res, tuple) else res, {})
            after = [(k, t.detach().clone()) for k, t in ts] + [("ret" + k, t.detach().clone()) for k, t in rts]
            if name == "attention_fwd":
                self.blank_unwritten(args, kwargs, after)
            if self.ref is None:
                self.cur.append((name, before, after))
            elif len(self.found) < 60:
                self.compare(name, before, after)
            self.idx += 1
            return res
        return inner

    @staticmethod
    def blank_unwritten(args, kwargs, after):
        """attention_fwd leaves unwritten what nobody reads: output rows past q_limit's 16-row tile, log-sum-exp positions past that
        or past a ragged sequence's length (tests/test_kernels_gpu.py::test_attention_backward_never_reads_what_forward_did_not_write).
        Blank them in the clones so that stale bytes do not count as a difference."""
        qkv, nseq, S = args[0], int(args[1]), int(args[2])
        qlim = int(kwargs.get("q_limit", 0) or 0)
        off = kwargs.get("seq_offsets")
        lim = min(S, (qlim + 15) & ~15) if qlim else S
        pos = torch.arange(S, device=qkv.device)
        if off is not None:
            lens = (off[1:] - off[:-1]).long()
            start = off[:-1].long()
        else:
            lens = torch.full((nseq,), S, dtype=torch.long, device=qkv.device)
            start = torch.arange(nseq, device=qkv.device) * S
        valid = pos[None, :] < torch.clamp(lens, max=lim)[:, None]                  # [nseq, S]
        rows = (start[:, None] + pos[None, :])[valid]
        for k, t in after:
            if k == "ret0":
                keep = torch.zeros(t.shape[0], dtype=torch.bool, device=t.device)
                keep[rows] = True
                t[~keep] = 0
            elif k == "ret1":
                t.masked_fill_(~valid[:, None, :].expand_as(t), 0.0)

    @staticmethod
    def diff(a, b):
        if a.shape != b.shape or a.dtype != b.dtype:
            return float("inf"), "shape/dtype"
        if a.numel() == 0:
            return 0.0, ""
        af, bf_ = a.float(), b.float()
        bad_a, bad_b = ~torch.isfinite(af), ~torch.isfinite(bf_)
        if bool((bad_a != bad_b).any()):
            return float("inf"), f"non-finite pattern differs ({int(bad_a.sum())} vs {int(bad_b.sum())})"
        af, bf_ = torch.where(bad_a, torch.zeros_like(af), af), torch.where(bad_b, torch.zeros_like(bf_), bf_)
        d = (af - bf_).abs()
        m = float(d.max())
        scale = max(float(bf_.abs().max()), 1e-30)
        where = ""
        if m / scale > TOL:
            flat = d.reshape(-1)
            nbad = int((flat > TOL * scale).sum())
            i = int(flat.argmax())
            where = f"{nbad} of {flat.numel()} elements beyond tolerance, worst at flat index {i} (shape {tuple(a.shape)}): {float(af.reshape(-1)[i]):.6g} vs {float(bf_.reshape(-1)[i]):.6g}"
            if a.dim() == 2:
                rows = (d > TOL * scale).any(1).nonzero().flatten()
                cols = (d > TOL * scale).any(0).nonzero().flatten()
                where += f"; rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({rows.numel()}), cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({cols.numel()})"
        return m / scale, where

    def compare(self, name, before, after):
        if self.idx >= len(self.ref) or self.ref[self.idx][0] != name:
            self.found.append((False, f"call {self.idx}: the sequence itself differs ({name} vs {self.ref[self.idx][0] if self.idx < len(self.ref) else 'end'})"))
            return
        _, rb, ra = self.ref[self.idx]
        for (k, t), (k2, r) in zip(after, ra):
            d, where = self.diff(t, r)
            if d > TOL:
                ins, all_same, self_differs = [], True, False
                for (kb, tb), (_, rbb) in zip(before, rb):
                    db, _ = self.diff(tb, rbb)
                    ins.append(f"{kb}: {'same' if db == 0 else f'{db:.1e}'}")
                    all_same &= db == 0
                    self_differs |= (kb == k and db != 0)
                # benign: the differing argument already differed BEFORE the call (an output buffer whose unwritten part a later
                # call fills); suspicious: every argument agreed before the call and the result does not
                kind = "SUSPICIOUS (all arguments agreed before the call)" if all_same else \
                       ("output buffer differed before the call" if self_differs else "an INPUT already differed")
                self.found.append((all_same, f"call {self.idx} {name}: argument {k} differs after the call by {d:.3e} of its max [{kind}] — {where}\n"
                                             f"      arguments BEFORE the call vs repetition 0: {', '.join(ins)}"))
                return


def run(tr, bf16, fused, one_stream, trace, steps=1):
    fname, hp, trees, over = full_case("C2")
    fp8.FUSED_Q = fused
    torch.manual_seed(11)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model, overrides=over)
    model = model.cuda().bfloat16().train()
    model.prepare_main_grads()
    if one_stream:
        model.encoder.graph_encoder.two_streams = False
    st = None if bf16 else model.enable_fp8()
    try:
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
        torch.cuda.synchronize()
        tr.recording, tr.idx = trace, 0
        for step in range(steps):                      # later steps of the fp8 model run the producer-quantised (fused) operands
            torch.manual_seed(100 + step)
            model.zero_main_grads()
            loss, n, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
        torch.cuda.synchronize()
        tr.recording = False
        return float(loss), model.main_grad_flat.clone(), (st.scale[:len(st.sites)].clone() if st is not None else None)
    finally:
        fp8.ACTIVE = None


_REAL = {}


def poison_allocations(value):
    """Every torch.empty / empty_like / new_empty returns memory filled with `value` (None: leave them alone): what a kernel reads
    from a buffer nobody wrote is then the same on every device — 0 in repetition 0, NaN / 1e30 afterwards — and a launch that lets
    it reach a result shows up as SUSPICIOUS in the trace.  uint8 buffers (fp8 operands) get 0x00 / 0x7f (e4m3 NaN)."""
    if not _REAL:
        _REAL.update(empty=torch.empty, empty_like=torch.empty_like, new_empty=torch.Tensor.new_empty)
    if value is None:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = _REAL["empty"], _REAL["empty_like"], _REAL["new_empty"]
        return

    def fill(t):
        if t.is_cuda and t.numel():
            if t.dtype == torch.uint8:
                t.fill_(0 if value == 0 else 0x7F)
            elif t.is_floating_point():
                t.fill_(value)
        return t
    torch.empty = lambda *a, **k: fill(_REAL["empty"](*a, **k))
    torch.empty_like = lambda *a, **k: fill(_REAL["empty_like"](*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(_REAL["new_empty"](self, *a, **k))


def main():
    argv = sys.argv[1:]
    reps = int(argv[0]) if argv and argv[0].isdigit() else 5
    bf16, alone, one_stream = "--bf16" in argv, "--alone" in argv, "--one-stream" in argv
    tr = Tracer()
    for name in ops.__all__:
        fn = getattr(ops, name, None)
        if callable(fn) and not isinstance(fn, type):
            setattr(ops, name, tr.wrap(name, fn))
    poison = "--poison" in argv
    steps = int(next((a.split("=")[1] for a in argv if a.startswith("--steps=")), "1"))
    fills = [0.0, float("nan"), 1e30, -1e30, 65504.0]
    if poison:
        poison_allocations(fills[0])
    loss0, g0, s0 = run(tr, bf16, not alone, one_stream, True, steps)
    tr.ref, tr.cur = tr.cur, []
    print(f"repetition 0: {len(tr.ref)} traced calls, loss {loss0:.8f}, |g| {float(g0.norm()):.6f}", flush=True)
    for rep in range(1, reps):
        tr.found = []
        if poison:
            poison_allocations(fills[rep % len(fills)])
            print(f"   (fresh allocations filled with {fills[rep % len(fills)]})", flush=True)
        loss, g, s = run(tr, bf16, not alone, one_stream, True, steps)
        gd = float((g - g0).norm() / g0.norm())
        sd = int((s != s0).sum()) if s is not None else 0
        print(f"repetition {rep}: loss d {loss - loss0:+.2e}, gradient rel-L2 {gd:.2e}, first scales differing {sd}; "
              + (f"{len(tr.found)} differing call(s), {sum(1 for a, _ in tr.found if a)} suspicious" if tr.found else "no traced call differs"), flush=True)
        shown = [m for a, m in tr.found if a][:6] or ([m for _, m in tr.found][:3] if rep == 1 else [])
        for m in shown:
            print("   " + m, flush=True)


if __name__ == "__main__":
    main()
