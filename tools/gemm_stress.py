"""Deterministic stress of the 8-wave persistent GEMM (gemm_bf16_pp256p) — VERDICT r3 item 1, DESIGN.md §4 (history: docs/experiment_log.md §12).

Every case is ONE launch shape with slightly more 256 x 256 tiles than compute units (so some workgroups walk a tile boundary with the
ring still turning) at a short or long K loop.  The launch is repeated `reps` times in two builds of the same source:

  plain   the production instantiation
  jitter  MDT_GEMM_DIAG=16: the instantiation with a pseudo-random 0-900-cycle sleep per wave at every point of the barrier / vmcnt
          protocol (csrc/gemm.hip, template parameter JIT) — a hole in the RAW / WAR argument shows under SOME interleaving, and
          the jitter walks through thousands of them per launch

and every output (C, the saved derivative, the column sums) is compared BIT FOR BIT on the device against the same problem on the
128 x 128 kernel (MDT_GEMM_TILE=128: another ring, another epilogue, same MFMA and k order).  Differences are accumulated as an
element mask, so a failure names its tiles.  `python tools/gemm_stress.py [reps] [--quick]`; prints one line per case and
"STRESS_OK" / "STRESS_FAILED n".  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import _lib, ops  # noqa: E402

bf = torch.bfloat16


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def set_env(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    _lib.reload_env()


FORMS = ("gelu_saved_dropout", "gelu_saved", "mulaux_colsum_kmajor", "bias_dropout_residual", "plain")


def launch(form, t, out, aux_out, cs):
    a, b, bt, bias, res, aux_in = t
    if form == "gelu_saved_dropout":          # runtime-flag instantiation (EPK = -1): the failing test's first launch
        ops.gemm(a, b, bias=bias, aux=aux_out, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=0.3, drop_seed=4711)
    elif form == "gelu_saved":                # E_FC1: every fc1 forward of a training step
        ops.gemm(a, b, bias=bias, aux=aux_out, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)
    elif form == "mulaux_colsum_kmajor":      # E_DFC2 against a k-major B: the failing test's second launch
        cs.zero_()
        ops.gemm(a, bt, trans_b=True, aux=aux_in, out=out, epilogue=ops.EPI_MULAUX, colsum=cs)
    elif form == "bias_dropout_residual":     # E_DENSE
        ops.gemm(a, b, bias=bias, residual=res, out=out, drop_p=0.4, drop_seed=99)
    else:
        ops.gemm(a, b, out=out)


def stress(M, N, K, form, reps):
    t = (rnd(M, K, seed=1).to(bf).cuda(), rnd(N, K, seed=2, scale=0.3).to(bf).cuda(), None, rnd(N, seed=3).to(bf).cuda(),
         rnd(M, N, seed=4).to(bf).cuda(), rnd(M, N, seed=5, scale=0.5).to(bf).cuda())
    t = (t[0], t[1], t[1].t().contiguous(), t[3], t[4], t[5])
    new = lambda: (torch.empty(M, N, dtype=bf, device="cuda"), torch.empty(M, N, dtype=bf, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda"))
    # reference: the 128 x 128 kernel
    set_env(MDT_GEMM_TILE="128", MDT_GEMM_DIAG=None)
    ref = new()
    launch(form, t, *ref)
    torch.cuda.synchronize()
    res = {}
    for mode, diag in (("plain", None), ("jitter", 16)):
        set_env(MDT_GEMM_TILE=None, MDT_GEMM_DIAG=diag, MDT_GEMM_PERSIST=2)      # 2: the persistent walk also for K < 512
        cur = new()
        mask = torch.zeros(M, N, dtype=torch.bool, device="cuda")
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        cs_bad = torch.zeros((), dtype=torch.int64, device="cuda")
        uses_aux = form.startswith("gelu")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(reps):
            cur[0].fill_(float("nan"))
            if uses_aux:
                cur[1].fill_(float("nan"))
            launch(form, t, *cur)
            d = cur[0].view(torch.int16) != ref[0].view(torch.int16)          # bit compare (NaN-safe)
            if uses_aux:
                d |= cur[1].view(torch.int16) != ref[1].view(torch.int16)
            mask |= d
            bad += d.any()
            if form == "mulaux_colsum_kmajor":                                  # fp32 atomics: order-dependent, equal to rounding
                cs_bad += ((cur[2] - ref[2]).abs() > 2e-3 * ref[2].abs().clamp(min=1.0)).any()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps          # launch + fills + compare: the jitter build must be visibly slower
        nb, ncs = int(bad), int(cs_bad)
        where = ""
        if nb:
            idx = mask.nonzero()
            tiles = sorted({(int(r) // 256, int(c) // 256) for r, c in idx[:: max(1, idx.shape[0] // 4096)].tolist()})
            where = f"  {int(mask.sum())} distinct elements, tiles (row, col) {tiles[:12]}{' ...' if len(tiles) > 12 else ''}"
        res[mode] = (nb, ncs, where, us)
    set_env(MDT_GEMM_TILE=None, MDT_GEMM_DIAG=None, MDT_GEMM_PERSIST=None)
    tiles_n = ((M + 255) // 256) * (N // 256)
    line = f"M{M:6d} N{N:5d} K{K:4d} tiles {tiles_n:4d} {form:24s}"
    fails = 0
    for mode in ("plain", "jitter"):
        nb, ncs, where, us = res[mode]
        line += f" | {mode} {nb}/{reps} differ ({us:.0f} us/it)" + (f", colsum {ncs}" if ncs else "") + where
        fails += nb + ncs
    print(line, flush=True)
    if fails:
        discriminate(form, t, new, reps)
    return fails


def discriminate(form, t, new, reps):
    """A case failed: is it this kernel or this device?  The same problem repeated on kernels that share nothing with the 8-wave
    ring — the 128 x 128 kernel against its own first result, and the vendor GEMM (torch.matmul) against its own first result."""
    set_env(MDT_GEMM_TILE="128", MDT_GEMM_DIAG=None, MDT_GEMM_PERSIST=None)
    ref = new()
    launch(form, t, *ref)
    cur = new()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for it in range(reps):
        launch(form, t, *cur)
        bad += (cur[0].view(torch.int16) != ref[0].view(torch.int16)).any()
    a, b = t[0], t[1]
    v0 = a @ b.t()
    vbad = torch.zeros((), dtype=torch.int64, device="cuda")
    for it in range(reps):
        vbad += ((a @ b.t()).view(torch.int16) != v0.view(torch.int16)).any()
    torch.cuda.synchronize()
    set_env(MDT_GEMM_TILE=None)
    print(f"    discriminator: 128 x 128 kernel vs itself {int(bad)}/{reps} differ; vendor GEMM vs itself {int(vbad)}/{reps} differ "
          f"(both 0: the 8-wave kernel is at fault; either > 0: the DEVICE does not repeat its own results)", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
    quick = "--quick" in sys.argv
    uid = ""
    try:
        import glob
        uid = open(sorted(glob.glob("/sys/class/drm/card*/device/unique_id"))[0]).read().strip()
    except (OSError, IndexError):
        pass
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    print(f"device {torch.cuda.get_device_name(0)} unique_id {uid} CUs {cus} reps {reps}", flush=True)
    # tile counts just above the 256 workgroups of a launch: 264 (8 workgroups walk two tiles), 260, 404 (148 walk two), 516 (two or three)
    shapes = [(16640 + 37, 1024), (256 * 129 + 5, 512), (256 * 100 + 77, 1024)] if not quick else [(16640 + 37, 1024)]
    ks = (128, 192, 256, 320, 768) if not quick else (256, 768)
    total = 0
    for K in ks:
        for (M, N) in shapes:
            for form in FORMS:
                total += stress(M, N, K, form, reps if K <= 320 else max(reps // 4, 50))
    print("STRESS_OK" if total == 0 else f"STRESS_FAILED {total}", flush=True)
    return 0 if total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
