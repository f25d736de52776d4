"""BASELINE.json configs[4]: per-tensor-scaled fp8 operands.  Kernel level: the quantiser against torch's float8 casts
(bit-equal), the 8-bit GEMM against fp32 math on the SAME quantised operands (exact up to accumulation order), and its
fused epilogues against the bf16 GEMM's."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from multimodaldiscussiontransformer_amd import ops as o
    return o


def _scale_for(x, fmax):
    return (fmax / x.abs().max().float()).reshape(1)


@pytest.mark.parametrize("fmt,tdt,fmax", [(0, torch.float8_e4m3fn, 448.0), (1, torch.float8_e5m2, 57344.0)])
@pytest.mark.parametrize("src_dtype", [torch.bfloat16, torch.float32])
def test_quantiser_equals_torch_float8_cast(ops, fmt, tdt, fmax, src_dtype):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(300, 768, device="cuda", generator=g) * 3).to(src_dtype)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-9, -1e-9, 1e4, -1e4, 0.3, -0.3], device="cuda").to(src_dtype)
    scale = _scale_for(x, fmax) * 4.0                       # deliberately too large: part of the tensor saturates
    amax = torch.zeros(1, device="cuda")
    q = ops.fp8_quantize(x, fmt, scale=scale, amax=amax)
    want = (x.float() * scale).clamp(-fmax, fmax).to(tdt).view(torch.uint8)
    assert torch.equal(q, want)
    assert float(amax) == float(x.float().abs().max())
    q1 = ops.fp8_quantize(x, fmt)                            # no scale: 1
    assert torch.equal(q1, x.float().clamp(-fmax, fmax).to(tdt).view(torch.uint8))


@pytest.mark.parametrize("a_fmt,tdt,fmax", [(0, torch.float8_e4m3fn, 448.0), (1, torch.float8_e5m2, 57344.0)])
@pytest.mark.parametrize("M,N,K", [(66000, 768, 768), (70001, 2304, 768), (66560, 768, 3072), (300, 768, 768), (2080, 256, 256)])
def test_gemm_fp8_equals_fp32_math_on_the_quantised_operands(ops, a_fmt, tdt, fmax, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    sa, sw = _scale_for(a, fmax), _scale_for(w, 448.0)
    a8, w8 = ops.fp8_quantize(a, a_fmt, scale=sa), ops.fp8_quantize(w, 0, scale=sw)
    out = ops.gemm_fp8(a8, w8, 1.0 / sa, 1.0 / sw, a_format=a_fmt)
    rows = torch.randint(0, M, (4096,), device="cuda", generator=g)      # spot rows (a 66000 x 3072 fp32 reference is 0.8 GB)
    rows[:256] = torch.arange(max(0, M - 256), max(0, M - 256) + 256, device="cuda").clamp(max=M - 1)   # the ragged last tile included
    ref = (a8[rows].view(tdt).float() @ w8.view(torch.float8_e4m3fn).float().t()) / (sa * sw)
    torch.testing.assert_close(out[rows].float(), ref, atol=2e-2 * float(ref.abs().max()), rtol=1.6e-2)      # bf16 output rounding
    # and against the unquantised product: fp8 resolution (e4m3: 2^-4 relative per element, averaged down over K)
    full = a[rows].float() @ w.float().t()
    rel = float((out[rows].float() - full).norm() / full.norm())
    assert rel < (0.05 if a_fmt == 0 else 0.09), rel


def test_gemm_fp8_epilogues_match_bf16_gemm_epilogues(ops):
    """bias + GELU + saved derivative, bias + dropout + residual, saved-derivative multiply + column sums: the same code
    path as the bf16 kernel's register epilogue, fed with the fp8 product."""
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(9)
    M, K, N = 66048, 768, 3072
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.04).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    sa, sw = _scale_for(a, 448.0), _scale_for(w, 448.0)
    a8, w8 = ops.fp8_quantize(a, 0, scale=sa), ops.fp8_quantize(w, 0, scale=sw)
    u = (a8.view(torch.float8_e4m3fn).float()[:2048] @ w8.view(torch.float8_e4m3fn).float().t()) / (sa * sw) + bias.float()
    aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = ops.gemm_fp8(a8, w8, 1 / sa, 1 / sw, bias=bias, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)
    tol = dict(atol=0.06, rtol=2e-2)
    torch.testing.assert_close(h[:2048].float(), F.gelu(u), **tol)
    ur = u.clone().requires_grad_(True)
    F.gelu(ur).sum().backward()
    torch.testing.assert_close(aux[:2048].float(), ur.grad, **tol)
    # fc2-style: [M, 3072] x [768, 3072]^T with bias, dropout and residual
    w2 = (torch.randn(K, N, device="cuda", generator=g) * 0.03).bfloat16()
    b2 = torch.randn(K, device="cuda", generator=g).bfloat16()
    res = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    sh, sw2 = _scale_for(h, 448.0), _scale_for(w2, 448.0)
    h8, w28 = ops.fp8_quantize(h, 0, scale=sh), ops.fp8_quantize(w2, 0, scale=sw2)
    y = ops.gemm_fp8(h8, w28, 1 / sh, 1 / sw2, bias=b2, residual=res, drop_p=0.4, drop_seed=77)
    m = ops.dropout_mask(M * K, 0.4, 77).view(M, K)[:2048].float() / 0.6
    ref = ((h8.view(torch.float8_e4m3fn).float()[:2048] @ w28.view(torch.float8_e4m3fn).float().t()) / (sh * sw2) + b2.float()) * m + res[:2048].float()
    torch.testing.assert_close(y[:2048].float(), ref, atol=0.08, rtol=2e-2)
    # input gradient through fc2: dY (e5m2) against the transposed weight copy, times the saved derivative, + column sums
    dy = (torch.randn(M, K, device="cuda", generator=g) * 0.1).bfloat16()
    w2t = w2.t().contiguous()                                 # [3072, 768]: the weight as the k-contiguous B operand of dX = dY W
    sd, swt = _scale_for(dy, 57344.0), _scale_for(w2t, 448.0)
    d8, wt8 = ops.fp8_quantize(dy, 1, scale=sd), ops.fp8_quantize(w2t, 0, scale=swt)
    cs = torch.zeros(N, device="cuda")
    du = ops.gemm_fp8(d8, wt8, 1 / sd, 1 / swt, a_format=1, aux=aux, epilogue=ops.EPI_MULAUX, colsum=cs)
    full = (d8.view(torch.float8_e5m2).float() @ wt8.view(torch.float8_e4m3fn).float().t()) / (sd * swt) * aux.float()
    torch.testing.assert_close(du[:2048].float(), full[:2048], atol=0.02, rtol=2e-2)
    torch.testing.assert_close(cs, full.sum(0), atol=0.5, rtol=3e-2)


@pytest.mark.parametrize("M,N,K", [(66000, 2304, 768), (70001, 3072, 768), (25700, 768, 3072), (300, 768, 1280)])
def test_gemm_fp8_block_mfma_kernel_against_the_8_wave_kernel_and_fp32_math(ops, monkeypatch, M, N, K):
    """The 4-wave kernel on v_mfma_f32_16x16x128_f8f6f4 (gemm_f8.hip: bias / GELU forward with its saved derivative / plain and
    saved-derivative input gradients) against the 8-wave kernel on the 16x16x32 form (MDT_GEMM_F8W=0) on the same quantised
    operands: the two differ only in how 128 products are summed before they join the fp32 accumulator, far below the bf16
    rounding of the outputs — and both against fp32 math on spot rows."""
    from multimodaldiscussiontransformer_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(21)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dy = (torch.randn(M, K, device="cuda", generator=g) * 0.1).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    saved = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    sa, sd, sw = _scale_for(a, 448.0), _scale_for(dy, 57344.0), _scale_for(w, 448.0)
    a8, d8, w8 = ops.fp8_quantize(a, 0, scale=sa), ops.fp8_quantize(dy, 1, scale=sd), ops.fp8_quantize(w, 0, scale=sw)

    def run():
        aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        cs = torch.zeros(N, device="cuda")
        outs = [ops.gemm_fp8(a8, w8, 1 / sa, 1 / sw, bias=bias),
                ops.gemm_fp8(a8, w8, 1 / sa, 1 / sw, bias=bias, aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD), aux,
                ops.gemm_fp8(d8, w8, 1 / sd, 1 / sw, a_format=1),
                ops.gemm_fp8(d8, w8, 1 / sd, 1 / sw, a_format=1, aux=saved, epilogue=ops.EPI_MULAUX, colsum=cs), cs]
        torch.cuda.synchronize()
        return outs

    try:
        monkeypatch.setenv("MDT_GEMM_F8W", "0"); _lib.reload_env()
        old = run()
        monkeypatch.setenv("MDT_GEMM_F8W", "1"); _lib.reload_env()
        new = run()
    finally:
        monkeypatch.delenv("MDT_GEMM_F8W", raising=False); _lib.reload_env()
    for name, o, n in zip(("bias", "gelu", "gelu derivative", "dgrad", "dgrad x saved", "column sums"), old, new):
        o, n = o.float(), n.float()
        # at most one bf16 rounding step apart where the two sums fall on either side of a rounding boundary
        torch.testing.assert_close(n, o, atol=1e-2 * float(o.abs().max()), rtol=8e-3, msg=lambda m: f"{name}: {m}")
        assert float((n - o).norm() / o.norm()) < 2e-3, name
    rows = torch.randint(0, M, (1024,), device="cuda", generator=g)
    rows[:256] = torch.arange(max(0, M - 256), max(0, M - 256) + 256, device="cuda").clamp(max=M - 1)
    ref = (a8[rows].view(torch.float8_e4m3fn).float() @ w8.view(torch.float8_e4m3fn).float().t()) / (sa * sw) + bias.float()
    torch.testing.assert_close(new[0][rows].float(), ref, atol=2e-2 * float(ref.abs().max()), rtol=1.6e-2)
    refd = (d8[rows].view(torch.float8_e5m2).float() @ w8.view(torch.float8_e4m3fn).float().t()) / (sd * sw)
    torch.testing.assert_close(new[3][rows].float(), refd, atol=2e-2 * float(refd.abs().max()), rtol=1.6e-2)


@pytest.mark.parametrize("M", [66000, 25700, 300])
def test_gemm_fp8_quantised_second_output_equals_a_quantisation_pass(ops, M):
    """mdt_gemm_fp8_q8: the GELU forward (→ e4m3) and the saved-derivative input gradient (→ e5m2) also write the fp8 copy of
    their output from the epilogue — the same bytes and the same running maximum as mdt_fp8_quantize on the bf16 output."""
    g = torch.Generator(device="cuda").manual_seed(31)
    K, N = 768, 3072
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dy = (torch.randn(M, K, device="cuda", generator=g) * 0.1).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    saved = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    sa, sd, sw = _scale_for(a, 448.0), _scale_for(dy, 57344.0), _scale_for(w, 448.0)
    a8, d8, w8 = ops.fp8_quantize(a, 0, scale=sa), ops.fp8_quantize(dy, 1, scale=sd), ops.fp8_quantize(w, 0, scale=sw)
    for fmt, fmax, run in ((0, 448.0, lambda **kw: ops.gemm_fp8(a8, w8, 1 / sa, 1 / sw, bias=bias, aux=torch.empty(M, N, device="cuda", dtype=torch.bfloat16),
                                                                epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, **kw)),
                           (1, 57344.0, lambda **kw: ops.gemm_fp8(d8, w8, 1 / sd, 1 / sw, a_format=1, aux=saved, epilogue=ops.EPI_MULAUX,
                                                                  colsum=torch.zeros(N, device="cuda"), **kw))):
        plain = run()
        # a scale that saturates part of the tensor on purpose, and a running maximum that starts above zero
        scale = (fmax / plain.float().abs().max() * 3.0).reshape(1)
        amax_a, amax_b = torch.full((1,), 1e-3, device="cuda"), torch.full((1,), 1e-3, device="cuda")
        out8 = torch.full((M, N), 0x55, device="cuda", dtype=torch.uint8)
        fused = run(q8_out=out8, q8_format=fmt, q8_scale=scale, q8_amax=amax_a)
        want8 = ops.fp8_quantize(plain, fmt, scale=scale, amax=amax_b)
        assert torch.equal(fused, plain)                       # the bf16 output is not affected
        assert torch.equal(out8, want8), int((out8 != want8).sum())
        assert float(amax_a) == float(amax_b) == float(plain.float().abs().max())


@pytest.mark.parametrize("rows,D", [(70001, 768), (5, 768), (4099, 1024)])
@pytest.mark.parametrize("fmt,fmax", [(0, 448.0), (1, 57344.0)])
def test_layernorm_quantised_second_output_equals_a_quantisation_pass(ops, rows, D, fmt, fmax):
    """mdt_layernorm_fwd_q8: the LayerNorm in front of an 8-bit GEMM writes that GEMM's operand itself — same bytes and same
    running maximum as mdt_fp8_quantize on its bf16 output, and the bf16 output / statistics unchanged."""
    g = torch.Generator(device="cuda").manual_seed(41)
    x = (torch.randn(rows, D, device="cuda", generator=g) * 3.0).bfloat16()
    gamma = (1.0 + 0.3 * torch.randn(D, device="cuda", generator=g)).bfloat16()
    beta = (0.2 * torch.randn(D, device="cuda", generator=g)).bfloat16()
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-12)
    scale = (fmax / y.float().abs().max() * 2.5).reshape(1)               # saturates part of the tensor on purpose
    amax_a, amax_b = torch.full((1,), 1e-3, device="cuda"), torch.full((1,), 1e-3, device="cuda")
    y8 = torch.full((rows, D), 0x55, device="cuda", dtype=torch.uint8)
    y2, mean2, rstd2 = ops.layernorm_fwd(x, gamma, beta, 1e-12, q8=(y8, fmt, scale, amax_a))
    want8 = ops.fp8_quantize(y, fmt, scale=scale, amax=amax_b)
    assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    assert torch.equal(y8, want8), int((y8 != want8).sum())
    assert float(amax_a) == float(amax_b) == float(y.float().abs().max())


def test_unsupported_shapes_are_refused_not_miscomputed(ops):
    from multimodaldiscussiontransformer_amd._lib import MdtError
    one = torch.ones(1, device="cuda")
    for (M, N, K) in ((70000, 640, 768), (70000, 768, 96), (70000, 768, 128)):       # N % 256; K % 64; short K
        a8 = torch.zeros(M, K, dtype=torch.uint8, device="cuda")
        w8 = torch.zeros(N, K, dtype=torch.uint8, device="cuda")
        with pytest.raises(MdtError, match="status -2|UNSUPPORTED|mdt_gemm_fp8"):
            ops.gemm_fp8(a8, w8, one, one)


def test_scale_update_delayed_scaling(ops):
    amax = torch.tensor([2.0, 0.0, float("inf"), 7.0], device="cuda")
    scale = torch.ones(4, device="cuda")
    inv = torch.ones(4, device="cuda")
    fmax = torch.tensor([448.0, 448.0, 448.0, 57344.0], device="cuda")
    ops.fp8_scale_update(amax, scale, inv, fmax, margin=2.0)
    assert scale.tolist() == [112.0, 1.0, 1.0, 4096.0] and amax.tolist() == [0.0, 0.0, 0.0, 0.0]
    torch.testing.assert_close(inv.cpu(), torch.tensor([1 / 112.0, 1.0, 1.0, 1 / 4096.0]), rtol=1e-6, atol=0)


# ----------------------------------------------------------------------------- model level (configs[4])
# The tolerance BASELINE.json configs[4] asks to be RE-STATED.  e4m3 keeps 3 mantissa bits: every 8-bit GEMM output carries
# ~5 % relative noise (rounding of both operands; it does not average out over K because the signal adds incoherently
# too), and 12 blocks x 2-3 such GEMMs compound it; e5m2 gradients keep 2 bits.  On this fixture — hash weights uniform in
# +-0.06, 1.7 x the std of the reference's N(0, 0.02) initialiser, logits spanning about +-0.6 — measured on MI355X per preset of
# fp8.PRESETS (C2 / C4): logits |err| and worst parameter-gradient relative L2
#     "all"    0.141 / 0.197    0.48 / 0.42        (round 2's three sites: 0.128 / 0.122, 0.39)
#     "fast4"  0.128 / 0.122    0.43 / 0.40
#     "grads"  0.0097 / 0.0097  0.16 / 0.13        8-bit only in the input gradients: logits at the bf16 level
# On the bench's random-init batch (weights N(0, 0.02), logits within +-0.22) the logits of "all" differ from the fp32 path by
# 0.03.  Gates = those observations with headroom:
FP8_GATES = {"all": (0.3, 0.6), "grads": (0.05, 0.25)}     # preset -> (logits |err|, gradient relative L2 per parameter tensor of >= 64 k elements)


@pytest.mark.parametrize("preset", ["all", "grads"])
@pytest.mark.parametrize("kind", ["C2", "C4"])
def test_fp8_real_geometry_vs_fp32_oracle(kind, preset):
    """mDT-base at its true geometry (and the mDT-large shapes) with fp8 operands in the blocks' big GEMMs (preset "all": QKV, fc1,
    fc2 forward, the input gradients of fc2 and fc1; "grads": the two input gradients only), delayed scaling, three training
    steps on the same batch (step 1 derives every scale from the tensor, steps 2-3 run on delayed scales and on operands
    quantised by the producing GEMM): logits within the preset's gate of the fp32 oracle, predictions identical wherever the
    fp32 margin is clear of that tolerance, parameter gradients within its relative-L2 gate on every tensor of at least 64 k
    elements."""
    FP8_LOGIT_ABS, FP8_GRAD_REL_L2 = FP8_GATES[preset]
    from multimodaldiscussiontransformer_amd import fp8
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from tests.test_oracle_golden import full_case
    from tests.test_real_shapes_gpu import oracle_run
    from tests.util_model import fill_hash_weights, model_args, named_canonical_params, split_qkv_grad
    o = oracle_run(kind, rounded=True)
    fname, hp, trees, over = full_case(kind)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model, overrides=over)
    model = model.cuda().bfloat16().train()
    model.prepare_main_grads()
    st = model.enable_fp8(sites=preset)
    try:
        pb = pack_batch(trees, 5)
        crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
        for step in range(3):
            model.zero_main_grads()
            loss, n, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
        assert st.gemms > 0 and len(st.sites) > (10 if preset == "all" else 4), (st.gemms, len(st.sites))          # the 8-bit kernel really ran
        assert st.fused_outputs > 0, "no GEMM handed its output to the next one as fp8 (fc1 -> fc2, d_fc2 -> d_fc1)"
        with torch.no_grad():
            logits, _ = model(pb.batched_data)
        torch.cuda.synchronize()
    finally:
        fp8.ACTIVE = None
    lg = logits.float().cpu()
    d = float((lg - o["logits"]).abs().max())
    margin = o["logits"][:, 1] - o["logits"][:, 0]
    clear = margin.abs() > 2 * FP8_LOGIT_ABS
    agree = bool((((lg[:, 1] - lg[:, 0]) > 0)[clear] == (margin > 0)[clear]).all())
    grads = {k: getattr(p, "main_grad", None) for k, p in named_canonical_params(model).items()}
    rows = []
    for name, ref in o["grads"].items():
        if ref is None or ref.numel() < 65536:
            continue
        gr = split_qkv_grad(name, grads)
        rn = float(ref.double().norm())
        if rn < 1e-6:
            continue
        rows.append((float((gr.float().cpu().double() - ref.double()).norm()) / rn, name))
    rows.sort(reverse=True)
    print(f"[{kind} fp8 {preset}] logits |err| {d:.3e}; {st.gemms} 8-bit GEMM launches over 3 steps, {len(st.sites)} sites; worst gradient rel-L2: "
          + "; ".join(f"{n} {r:.3e}" for r, n in rows[:4]))
    import os
    if os.environ.get("MDT_FP8_PROBE"):
        return
    assert d < FP8_LOGIT_ABS, d
    assert agree
    assert rows[0][0] < FP8_GRAD_REL_L2, rows[:5]


def test_fp8_producer_quantised_operands_equal_stand_alone_passes(monkeypatch):
    """fc1 -> fc2 and d_fc2 -> d_fc1 hand the operand over as fp8 from the producing GEMM's epilogue (fp8.py ``q8_site``).  The same
    model run with MDT_FP8_FUSED_Q=0 quantises those operands with passes of its own: same bytes, same maxima, hence the same
    scales — three steps of either route must give bit-identical logits, and gradients equal up to the summation order of atomics."""
    from multimodaldiscussiontransformer_amd import fp8
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from tests.test_oracle_golden import full_case
    from tests.util_model import fill_hash_weights, model_args
    fname, hp, trees, over = full_case("C2")

    def run(fused):
        monkeypatch.setattr(fp8, "FUSED_Q", fused)
        torch.manual_seed(11)
        model = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(model, overrides=over)
        model = model.cuda().bfloat16().train()
        model.prepare_main_grads()
        st = model.enable_fp8()
        try:
            pb = pack_batch(trees, 5)
            crit = GraphPredictionNodeCrossEntropy(None, positive_weight=hp.pos_weight, negative_weight=hp.neg_weight)
            for step in range(3):
                torch.manual_seed(100 + step)               # dropout seeds
                model.zero_main_grads()
                loss, n, log = crit(model, {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}})
                loss.backward()                             # closes the fp8 step: maxima -> next scales
            with torch.no_grad():
                logits, _ = model(pb.batched_data)
            torch.cuda.synchronize()
            return logits.float().cpu(), model.main_grad_flat.clone().cpu(), st.fused_outputs, st.gemms
        finally:
            fp8.ACTIVE = None

    lg_f, g_f, fused_n, gemms_f = run(True)
    lg_s, g_s, fused_0, gemms_s = run(False)
    assert fused_n > 0 and fused_0 == 0 and gemms_f == gemms_s, (fused_n, fused_0, gemms_f, gemms_s)
    print(f"[fp8 fused vs stand-alone] logits sums {float(lg_f.double().sum()):.10f} / {float(lg_s.double().sum()):.10f}, max |diff| {float((lg_f - lg_s).abs().max()):.3e}, "
          f"first rows {lg_f[0].tolist()} / {lg_s[0].tolist()}")
    if not torch.equal(lg_f, lg_s):
        # Seen on SOME devices of the pool only (DESIGN.md §4 "Still open"): there ONE route does not repeat its own result either.
        # Tell the two cases apart: a route that repeats itself and differs from the other one is a bug of the fused producers
        # (fail); a device on which the same route gives two answers is the open issue (xfail, with the launch that did not repeat).
        import os
        import subprocess
        import sys
        lg_f2, _, _, _ = run(True)
        lg_s2, _, _, _ = run(False)
        repeats = torch.equal(lg_f2, lg_f) and torch.equal(lg_s2, lg_s)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        tr = subprocess.run([sys.executable, os.path.join(root, "tools", "op_trace.py"), "4"], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in tr.stdout.splitlines() if ln.startswith("repetition") or "SUSPICIOUS" in ln]
        msg = ("fused and stand-alone routes differ (max |diff| %.3e); each route repeats itself: %s; tools/op_trace.py on this device:\n%s" %
               (float((lg_f - lg_s).abs().max()), repeats, "\n".join(lines)[:4000]))
        if repeats:
            pytest.fail(msg)
        pytest.xfail("this device does not repeat ONE fp8 route's own result (open issue, DESIGN.md §4): " + msg)
    # gradients: the split-K weight gradients add their slabs with fp32 atomics in whatever order they finish — equal up to that
    assert float((g_f.double() - g_s.double()).norm() / g_s.double().norm()) < 1e-5


def test_fp8_training_learns_what_bf16_learns(capsys):
    """Training-quality evidence for configs[4]: the launcher's learnable synthetic task (a comment is hateful
    iff its second token id lies in the upper half of the vocabulary) with a 64-word vocabulary, so that the rule GENERALISES from
    the 640 labelled comments of the training stream to held-out trees; a model wide enough for every fp8 site to engage (D 256,
    FFN 1024, 2 + 2 blocks); 160 updates from the SAME seed in bf16 and with fp8 operands at preset "all" (QKV / fc1 / fc2 forward,
    the input gradients of fc2 and fc1; delayed scaling, producer-side quantisation).  Both must learn the rule (training loss
    over the last 30 updates far below the first, held-out accuracy >= 0.9), and the fp8 run must learn what bf16 learns: mean
    training loss of the last 30 updates within 5 % + 0.01, held-out loss within 5 % + 0.01, held-out F1 within 0.03 and accuracy
    within 0.02 (3 of the 96 held-out comments)."""
    from multimodaldiscussiontransformer_amd import fp8, train
    base = ["--task", "node_prediction", "--arch", "multi_graphormer_base", "--criterion", "node_cross_entropy",
            "--dataset-name", "synthetic", "--batch-size", "16", "--max-update", "160", "--validate-interval-updates", "160",
            "--lr", "4e-4", "--end-learning-rate", "1e-5", "--warmup-updates", "8", "--total-num-update", "160",
            "--adam-betas", "(0.9, 0.999)", "--adam-eps", "1e-8", "--weight-decay", "0.01",
            "--encoder-embed-dim", "256", "--encoder-ffn-embed-dim", "256", "--encoder-attention-heads", "4",
            "--num_fusion_layers", "1", "--num_bottleneck_tokens", "4", "--num_graph_stack", "1", "--num_fusion_stack", "1",
            "--attention-dropout", "0.1", "--act-dropout", "0.1", "--dropout", "0.1", "--spatial-pos-max", "5",
            "--positive-weight", "1.5", "--negative-weight", "1", "--log-interval", "1",
            "--synthetic-nodes", "8", "--synthetic-seq-len", "32", "--synthetic-batches", "40", "--synthetic-valid-batches", "6",
            "--bert-config", '{"dim": 256, "layers": 4, "heads": 4, "intermediate": 1024, "vocab": 64, "max_pos": 64}',
            "--vit-config", '{"dim": 256, "layers": 4, "heads": 4, "intermediate": 1024, "image_size": 32, "patch": 16}',
            "--no-save", "--seed", "7", "--random-init-encoders"]
    res = {}
    try:
        for tag, extra in (("bf16", ["--bf16"]), ("fp8", ["--fp8", "--fp8-sites", "all"])):
            hist = train.main(base + extra)
            vh = train.main.valid_history[-1]
            st = train.main.last_run["fp8"]
            tail = [h["loss"] for h in hist[-30:]]
            res[tag] = dict(first=hist[0]["loss"], last=sum(tail) / len(tail), vloss=vh["valid_loss"], f1=vh["valid_f1"], acc=vh["valid_accuracy"],
                            gemms=0 if st is None else st.gemms, fused=0 if st is None else st.fused_outputs)
    finally:
        fp8.ACTIVE = None
    capsys.readouterr()
    b, f = res["bf16"], res["fp8"]
    print(f"[fp8 convergence] bf16: loss {b['first']:.4f} -> {b['last']:.4f} (mean of the last 30 updates), held-out loss {b['vloss']:.4f} F1 {b['f1']:.4f} acc {b['acc']:.4f} | "
          f"fp8 all: loss {f['first']:.4f} -> {f['last']:.4f}, held-out loss {f['vloss']:.4f} F1 {f['f1']:.4f} acc {f['acc']:.4f}; "
          f"{f['gemms']} 8-bit GEMM launches, {f['fused']} operands quantised by their producer")
    assert f["gemms"] > 160 * 10 and f["fused"] > 0, f           # the 8-bit kernels really carried the run
    assert b["last"] < 0.5 * b["first"] and f["last"] < 0.5 * f["first"], (b, f)
    assert b["acc"] >= 0.9 and f["acc"] >= 0.9, (b, f)           # the rule was learnt, not memorised
    assert abs(f["last"] - b["last"]) <= 0.05 * b["last"] + 0.01, (b["last"], f["last"])
    assert abs(f["vloss"] - b["vloss"]) <= 0.05 * b["vloss"] + 0.01, (b["vloss"], f["vloss"])
    assert abs(f["f1"] - b["f1"]) <= 0.03 and abs(f["acc"] - b["acc"]) <= 0.02, (b, f)
