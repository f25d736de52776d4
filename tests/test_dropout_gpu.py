"""Dropout parity on the GPU.  The kernels' masks are a pure function of (site seed, element
counter); ``ops.dropout_mask`` exposes them, so every dropout site — GEMM epilogues, attention
probabilities, the stand-alone op, a whole Graphormer layer in training mode — is checked
EXACTLY (fp32 tolerance) against a PyTorch fp32 computation that uses the same masks."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

C = 0x9E3779B97F4A7C15
M63 = 0x7FFFFFFFFFFFFFFF


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.fixture(scope="module")
def ops():
    from multimodaldiscussiontransformer_amd import ops as o
    return o


def mask(ops, shape, p, seed):
    n = 1
    for s in shape:
        n *= s
    return ops.dropout_mask(n, p, seed).view(*shape).float().cpu() / (1.0 - p)


def attn_mask(ops, nseq, H, S, p, seed):
    """Keep/scale mask of attention-probability dropout: counters run over rows of even length (csrc/attention_common.hpp)."""
    S2 = S + (S & 1)
    return mask(ops, (nseq, H, S, S2), p, seed)[..., :S].contiguous()


def test_standalone_dropout_statistics_and_determinism(ops):
    x = torch.ones(4096, 256).cuda()
    for p in (0.1, 0.3, 0.4):
        y = ops.dropout(x, p, 12345)
        m = mask(ops, (4096, 256), p, 12345)
        assert torch.equal(y.cpu(), m)
        kept = float((y != 0).float().mean())
        sigma = math.sqrt(p * (1 - p) / x.numel())
        assert abs(kept - (1 - p)) < 5 * sigma
        assert abs(float(y.mean()) - 1.0) < 5 * sigma / (1 - p)
        assert torch.equal(ops.dropout(x, p, 12345), y)
        assert not torch.equal(ops.dropout(x, p, 12346), y)
    xb = rnd(100, 64, seed=1).bfloat16().cuda()
    yb = ops.dropout(xb, 0.25, 99)
    ref = (xb.float().cpu() * mask(ops, (100, 64), 0.25, 99)).bfloat16()
    assert torch.equal(yb.cpu(), ref)
    # rows / columns are not correlated: column keep-rates are all near 1-p
    col = (ops.dropout(x, 0.3, 5) != 0).float().mean(0)
    assert float((col - 0.7).abs().max()) < 0.05


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_dropout(ops, dtype):
    M, N, K = (50, 40, 36) if dtype == torch.float32 else (33000, 256, 128)
    p, seed = 0.3, 777
    a, b = rnd(M, K, seed=1).to(dtype), rnd(N, K, seed=2, scale=0.3).to(dtype)
    bias, res = rnd(N, seed=3).to(dtype), rnd(M, N, seed=4).to(dtype)
    m = mask(ops, (M, N), p, seed)
    u = a.float() @ b.float().t() + bias.float()
    tol = dict(atol=2e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=0.04, rtol=2e-2)
    out = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda(), drop_p=p, drop_seed=seed)
    torch.testing.assert_close(out.float().cpu(), u * m + res.float(), **tol)
    aux = torch.empty(M, N, dtype=dtype).cuda()
    out = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), aux=aux, epilogue=ops.EPI_GELU, drop_p=p, drop_seed=seed)
    ur = aux.float().cpu()
    torch.testing.assert_close(ur, u, **tol)
    torch.testing.assert_close(out.float().cpu(), F.gelu(ur) * m, **tol)
    # backward of gelu+dropout: (g * mask) * gelu'(u)
    x = ur.clone().requires_grad_(True)
    (F.gelu(x) * m).backward(a.float() @ b.float().t())
    out = ops.gemm(a.cuda(), b.cuda(), aux=aux, epilogue=ops.EPI_DGELU, drop_p=p, drop_seed=seed, out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), x.grad, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (50, 40, 36)), (torch.bfloat16, (33000, 256, 128)),
                                         (torch.bfloat16, (16640 + 37, 1024, 256))])
def test_gemm_saved_derivative_epilogue(ops, dtype, shape):
    """MDT_EPI_GELU | MDT_EPI_AUX_GRAD saves d out / d u = GELU'(u) * dropout scale; backward is MDT_EPI_MULAUX.
    The third shape has more 256x256 tiles than CUs: the persistent ping-pong kernel with a ragged last row tile."""
    M, N, K = shape
    p, seed = 0.3, 4711
    a, b = rnd(M, K, seed=1).to(dtype), rnd(N, K, seed=2, scale=0.3).to(dtype)
    bias = rnd(N, seed=3).to(dtype)
    m = mask(ops, (M, N), p, seed)
    u = (a.float() @ b.float().t() + bias.float()).requires_grad_(True)
    h = F.gelu(u) * m
    h.backward(torch.ones_like(h))
    tol = dict(atol=2e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=0.04, rtol=2e-2)
    aux = torch.empty(M, N, dtype=dtype).cuda()
    out = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), aux=aux, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, drop_p=p, drop_seed=seed)
    torch.testing.assert_close(out.float().cpu(), h.detach(), **tol)
    torch.testing.assert_close(aux.float().cpu(), u.grad, **tol)
    # backward GEMM: (g @ W) * aux, with the bias gradient (column sums) fused
    g = rnd(M, K, seed=5).to(dtype)
    cs = torch.zeros(N, dtype=torch.float32).cuda()
    du = ops.gemm(g.cuda(), b.t().contiguous().cuda(), trans_b=True, aux=aux, epilogue=ops.EPI_MULAUX, colsum=cs)
    ref = (g.float() @ b.float().t()) * aux.float().cpu()
    torch.testing.assert_close(du.float().cpu(), ref, **tol)
    torch.testing.assert_close(cs.cpu(), ref.sum(0), atol=0.5 if dtype == torch.bfloat16 else 1e-3, rtol=2e-2)


@pytest.mark.parametrize("dtype,bwd", [(torch.float32, None), (torch.bfloat16, None), (torch.bfloat16, "v1"), (torch.bfloat16, "v3")])
@pytest.mark.parametrize("nseq,S,H", [(3, 20, 2), (2, 104, 3), (2, 201, 2)])
def test_attention_dropout(ops, dtype, bwd, nseq, S, H, monkeypatch):
    hd, p, seed = 64, 0.3, 4242
    if bwd:
        monkeypatch.setenv("MDT_ATTN_BWD", bwd)   # every backward kernel family regenerates the forward's mask
        from multimodaldiscussiontransformer_amd import _lib
        _lib.reload_env()
    D = H * hd
    qkv = rnd(nseq, S, 3 * D, seed=7).to(dtype)
    dout = rnd(nseq, S, D, seed=8).to(dtype)
    km = torch.ones(nseq, S, dtype=torch.uint8)
    km[1, S - 3:] = 0
    m = attn_mask(ops, nseq, H, S, p, seed)
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr.split(D, dim=-1)
    hv = lambda t: t.view(nseq, S, H, hd).transpose(1, 2)
    s = hv(q) @ hv(k).transpose(-1, -2) * hd ** -0.5
    s = s.masked_fill(~km.bool()[:, None, None, :], -math.inf)
    pr = torch.softmax(s, -1) * m
    oref = (pr @ hv(v)).transpose(1, 2).reshape(nseq, S, D)
    oref.backward(dout.float())
    q2, d2 = qkv.view(nseq * S, 3 * D).cuda(), dout.view(nseq * S, D).cuda()
    out, lse = ops.attention_fwd(q2, nseq, S, H, key_mask=km.cuda(), drop_p=p, drop_seed=seed)
    tol = dict(atol=2e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=0.04, rtol=3e-2)
    torch.testing.assert_close(out.float().cpu().view(nseq, S, D), oref.detach(), **tol)
    dqkv, _ = ops.attention_bwd(d2, q2, out, lse, nseq, S, H, key_mask=km.cuda(), drop_p=p, drop_seed=seed)
    gtol = dict(atol=5e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=0.08, rtol=6e-2)
    torch.testing.assert_close(dqkv.float().cpu().view(nseq, S, 3 * D), qr.grad, **gtol)


@pytest.mark.parametrize("pre_ln", [False, True])
def test_graphormer_layer_training_dropout_exact(ops, pre_ln):
    """A whole layer in train() mode with the launch script's dropout rates, against torch with
    the same masks (site seeds are derived from the CPU generator: base + C * counter)."""
    from multimodaldiscussiontransformer_amd.modules import GraphormerGraphEncoderLayer
    T, B, D, H, Fg = 9, 3, 128, 8, 128
    pd, pa, pact = 0.4, 0.3, 0.3
    layer = GraphormerGraphEncoderLayer(embedding_dim=D, ffn_embedding_dim=Fg, num_attention_heads=H, dropout=pd,
                                        attention_dropout=pa, activation_dropout=pact, activation_fn="gelu",
                                        pre_layernorm=pre_ln).cuda().train()
    with torch.no_grad():
        for i, prm in enumerate(layer.parameters()):
            prm.copy_(rnd(*prm.shape, seed=100 + i, scale=0.2).cuda() + (1.0 if prm.dim() == 1 and i % 2 == 0 else 0.0))
    x = rnd(T, B, D, seed=1).cuda().requires_grad_(True)
    bias = rnd(B, H, T, T, seed=2).cuda()
    kpm = torch.zeros(B, T, dtype=torch.bool)
    kpm[1, T - 2:] = True
    cot = rnd(T, B, D, seed=3).cuda()
    torch.manual_seed(31337)
    y, _ = layer(x, self_attn_bias=bias, self_attn_padding_mask=kpm.cuda())
    (y * cot).sum().backward()
    # reference
    torch.manual_seed(31337)
    base = int(torch.randint(0, 2 ** 62, (1,)).item())
    s_attn, s_o, s_act, s_f2 = [(base + C * k) & M63 for k in (1, 2, 3, 4)]
    W = {n: prm.detach().cpu().clone().requires_grad_(True) for n, prm in layer.named_parameters()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    m_attn = attn_mask(ops, B, H, T, pa, s_attn)
    m_o = mask(ops, (T * B, D), pd, s_o).view(T, B, D)
    m_act = mask(ops, (T * B, Fg), pact, s_act).view(T, B, Fg)
    m_f2 = mask(ops, (T * B, D), pd, s_f2).view(T, B, D)

    def ln(t, n):
        return F.layer_norm(t, (D,), W[n + ".weight"], W[n + ".bias"], 1e-5)

    def attn(h):
        qkv = F.linear(h, W["self_attn.qkv_weight"], W["self_attn.qkv_bias"])
        q, k, v = qkv.split(D, dim=-1)
        hv = lambda t: t.view(T, B, H, D // H).permute(1, 2, 0, 3)
        s = hv(q) @ hv(k).transpose(-1, -2) * (D // H) ** -0.5 + bias.cpu()
        s = s.masked_fill(kpm[:, None, None, :], -math.inf)
        pr = torch.softmax(s, -1) * m_attn
        o = (pr @ hv(v)).permute(2, 0, 1, 3).reshape(T, B, D)
        return F.linear(o, W["self_attn.out_proj.weight"], W["self_attn.out_proj.bias"])

    r = xr
    h = ln(xr, "self_attn_layer_norm") if pre_ln else xr
    h = r + attn(h) * m_o
    if not pre_ln:
        h = ln(h, "self_attn_layer_norm")
    r = h
    g = ln(h, "final_layer_norm") if pre_ln else h
    g = F.gelu(F.linear(g, W["fc1.weight"], W["fc1.bias"])) * m_act
    g = r + F.linear(g, W["fc2.weight"], W["fc2.bias"]) * m_f2
    if not pre_ln:
        g = ln(g, "final_layer_norm")
    (g * cot.cpu()).sum().backward()
    torch.testing.assert_close(y.detach().cpu(), g.detach(), atol=3e-4, rtol=1e-3)
    torch.testing.assert_close(x.grad.cpu(), xr.grad, atol=1e-3, rtol=2e-3)
    for n, prm in layer.named_parameters():
        torch.testing.assert_close(prm.grad.cpu(), W[n].grad, atol=2e-3, rtol=2e-3, msg=n)
    # eval mode: dropout is the identity
    layer.eval()
    y1, _ = layer(x, self_attn_bias=bias, self_attn_padding_mask=kpm.cuda())
    y2, _ = layer(x, self_attn_bias=bias, self_attn_padding_mask=kpm.cuda())
    assert torch.equal(y1, y2)


def test_full_model_training_with_launch_dropout():
    """mDT with the reference launch's dropout rates (0.4 / 0.3 / 0.3): finite, seed-reproducible,
    different across seeds, and the expected loss is in family with the no-dropout loss."""
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    from oracle import cases
    from tests.util_model import fill_hash_weights, model_args
    hp = cases.tiny_hparams("A")
    trees = cases.tiny_trees("A", hp)
    model = GraphormerModel.build_model(model_args(hp, dropout=0.4, attention_dropout=0.3, act_dropout=0.3), task=None)
    fill_hash_weights(model)
    model = model.cuda().train()
    pb = pack_batch(trees, 5)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
    sample = {"nsamples": len(trees), "net_input": {"batched_data": pb.batched_data}}

    def run(seed):
        torch.manual_seed(seed)
        for prm in model.parameters():
            prm.grad = None
        loss, _, _ = crit(model, sample)
        loss.backward()
        g = model.encoder.graph_encoder.fusion_layers[1].fusion_layers[0].bert_encoder.intermediate.dense.weight.grad
        return float(loss), g.clone()

    l1, g1 = run(1)
    l1b, g1b = run(1)
    l2, g2 = run(2)
    assert l1 == l1b and torch.equal(g1, g1b)
    assert l1 != l2 and not torch.equal(g1, g2)
    assert all(math.isfinite(v) for v in (l1, l2)) and bool(torch.isfinite(g1).all())
    model.eval()
    with torch.no_grad():
        la, _ = model(pb.batched_data)
        lb, _ = model(pb.batched_data)
    assert torch.equal(la, lb)
