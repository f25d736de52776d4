"""Bitwise repeatability of the persistent GEMM launches (the same launch N times: any difference is a race).  Written after
tests/test_dropout_gpu.py::test_gemm_saved_derivative_epilogue[dtype2-shape2] was seen to fail intermittently on one box
(docs/experiment_log.md §12).  GPU box only."""
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402

bf = torch.bfloat16
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def stress(name, M, N, K, fwd=True, side_load=False):
    a, b = rnd(M, K, seed=1).to(bf).cuda(), rnd(N, K, seed=2, scale=0.3).to(bf).cuda()
    bias = rnd(N, seed=3).to(bf).cuda()
    aux = torch.empty(M, N, dtype=bf, device="cuda")
    out = torch.empty(M, N, dtype=bf, device="cuda")
    ref_out = ref_aux = ref_cs = None
    bad = 0
    other = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda", dtype=bf)
    for it in range(N_IT):
        if side_load and it % 3 == 0:       # perturb the timing: an unrelated kernel on another stream
            with torch.cuda.stream(other):
                junk2 = junk @ junk
        if fwd:
            aux.fill_(0); out.fill_(0)
            ops.gemm(a, b, bias=bias, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=0.3, drop_seed=4711)
            cur = (out.clone(), aux.clone(), None)
        else:
            cs = torch.zeros(N, dtype=torch.float32, device="cuda")
            out.fill_(0)
            bt = b.t().contiguous()
            ops.gemm(a, bt, trans_b=True, aux=aux_fixed, out=out, epilogue=ops.EPI_MULAUX, colsum=cs)
            cur = (out.clone(), None, cs.clone())
        if ref_out is None:
            ref_out, ref_aux, ref_cs = cur
            continue
        same = torch.equal(cur[0], ref_out) and (cur[1] is None or torch.equal(cur[1], ref_aux))
        if cur[2] is not None:
            same = same and bool(((cur[2] - ref_cs).abs() <= 1e-3 * ref_cs.abs().clamp(min=1.0)).all())     # atomics: order-dependent sums
        if not same:
            bad += 1
            if bad <= 3:
                d = (cur[0].float() - ref_out.float()).abs()
                idx = d.nonzero()
                print(f"   {name}: iteration {it}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}, rows {int(idx[:,0].min())}-{int(idx[:,0].max())}, cols {int(idx[:,1].min())}-{int(idx[:,1].max())}", flush=True)
    torch.cuda.synchronize()
    print(f"{name:58s} M{M} N{N} K{K}: {bad} of {N_IT - 1} repeats differ", flush=True)


for side in (False, True):
    tag = " (+ side-stream load)" if side else ""
    stress("GELU + saved derivative + dropout, K 256, 264 tiles" + tag, 16640 + 37, 1024, 256, side_load=side)
    aux_fixed = (torch.randn(16677, 1024, device="cuda") * 0.5).to(bf)
    stress("saved-derivative multiply + column sums, K 256" + tag, 16677, 1024, 256, fwd=False, side_load=side)
    stress("GELU + saved derivative + dropout, K 768, fc1 shape" + tag, 26624 + 37, 3072, 768, side_load=side)
    stress("GELU + saved derivative + dropout, K 128" + tag, 33000, 256, 128, side_load=side)
