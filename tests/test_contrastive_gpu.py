"""criterion ``contrastive_loss`` on the GPU: the HIP kernels behind ``mdt_contrastive_loss`` and
the end-to-end contrastive step against the REAL reference's outputs (tests/golden/contrastive.npz, made by
oracle/gen_golden.py from mDT/src/criterions/contrastive_loss.py) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import mdt_ref_cpu as R
from oracle import structure as S
from tests.util_model import fill_hash_weights, model_args, named_canonical_params, split_qkv_grad

pytestmark = pytest.mark.gpu

MODES = (("adaptive", dict(scale=20.0, soft_negative_weight=0.0, adaptive=True)),
         ("fixed", dict(scale=20.0, soft_negative_weight=0.25, adaptive=False)),
         ("strict", dict(scale=1.0, soft_negative_weight=0.0, adaptive=False)))


def _half_ulp(v):
    return 2.0 ** (int(np.floor(np.log2(max(abs(v), 2.0 ** -14)))) - 10)


@pytest.mark.parametrize("tag,kw", MODES)
def test_contrastive_kernel_vs_reference_golden(golden_dir, tag, kw):
    from multimodaldiscussiontransformer_amd import ops
    g = np.load(os.path.join(golden_dir, "contrastive.npz"))
    emb = torch.from_numpy(g["emb"]).cuda()
    y, hard = torch.from_numpy(g["y"]).cuda(), torch.from_numpy(g["hard_y"]).cuda()
    loss, counters, d_emb = ops.contrastive_loss(emb, y, hard, kw["scale"], kw["soft_negative_weight"], kw["adaptive"])
    ref = float(g[f"{tag}/loss"])
    assert abs(float(loss.cpu()) - ref) <= _half_ulp(ref), (float(loss.cpu()), ref)       # the reference's loss is a half value
    assert counters.cpu().tolist() == [int(g[f"{tag}/{k}"]) for k in ("ncorrect", "positive_correct", "total_positive", "pred_positive")]
    np.testing.assert_allclose(d_emb.cpu().numpy(), g[f"{tag}/d_emb"], atol=2e-6, rtol=2e-5)


def test_contrastive_kernel_vs_oracle_random_bf16_and_edge_cases():
    """Random labels / sizes against the oracle; one batch where a row has no soft negative (the reference divides by
    zero there: inf weights, inf loss — reproduced, not repaired); bf16 embeddings."""
    from multimodaldiscussiontransformer_amd import ops
    rng = np.random.default_rng(5)
    for B, D, ncomm, dtype in ((7, 768, 3, torch.float32), (33, 1024, 5, torch.float32), (64, 768, 4, torch.bfloat16)):
        y = torch.from_numpy(rng.integers(0, ncomm, B).astype(np.float32))
        hard = torch.from_numpy(((y.numpy() + 1 + rng.integers(0, ncomm - 1, B)) % ncomm).astype(np.float32))
        e0 = torch.from_numpy(rng.normal(0, 1, (B, D)).astype(np.float32)).to(dtype)
        er = e0.float().clone().requires_grad_(True)
        lo, c = R.contrastive_loss(er, y, hard)
        lo.backward()
        loss, counters, d_emb = ops.contrastive_loss(e0.cuda(), y.cuda(), hard.cuda(), 20.0, 0.0, True)
        assert abs(float(loss.cpu()) - float(lo)) <= _half_ulp(float(lo)) + 1e-4 * abs(float(lo)) * (dtype == torch.bfloat16)
        assert counters.cpu().tolist() == [c["ncorrect"], c["positive_correct"], c["total_positive"], c["pred_positive"]]
        tol = dict(atol=2e-6, rtol=2e-5) if dtype == torch.float32 else dict(atol=2e-3, rtol=2e-2)
        torch.testing.assert_close(d_emb.float().cpu(), er.grad, **tol)
    y = torch.tensor([0.0, 0.0, 1.0, 1.0])          # every pair is either a positive or a hard negative for rows 0, 1
    hard = torch.tensor([1.0, 1.0, 2.0, 2.0])
    e0 = torch.from_numpy(rng.normal(0, 1, (4, 64)).astype(np.float32))
    lo, _ = R.contrastive_loss(e0, y, hard)
    loss, _, _ = ops.contrastive_loss(e0.cuda(), y.cuda(), hard.cuda(), 20.0, 0.0, True)
    assert (not np.isfinite(float(lo))) == (not np.isfinite(float(loss.cpu())))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_contrastive_end_to_end_vs_reference_golden_and_oracle(golden_dir, dtype):
    """One contrastive step of the full (tiny "A"-shaped) model: global embeddings, loss, counters and every parameter
    gradient against the reference run — the final graph stack trains under this objective, the classifier head does not."""
    from multimodaldiscussiontransformer_amd.criterions import GraphContrastiveLoss
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    g = np.load(os.path.join(golden_dir, "contrastive.npz"))
    hp = cases.tiny_hparams("A")
    trees = cases.contrastive_trees(hp)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    fill_hash_weights(model)
    model = model.cuda().to(dtype).train()
    pb = pack_batch(trees, 5)
    ref_b = S.collate(trees, 5)
    assert np.array_equal(pb.batched_data["hard_y"].cpu().numpy(), ref_b["hard_y"]) and "y_mask" not in pb.batched_data
    crit = GraphContrastiveLoss(None)
    loss, n, log = crit(model, {"net_input": {"batched_data": pb.batched_data}})
    loss.backward()
    with torch.no_grad():
        _, glob = model(pb.batched_data)
    assert n == int(g["full/sample_size"]) == len(trees) ** 2
    fp32 = dtype == torch.float32
    np.testing.assert_allclose(glob.float().cpu().numpy(), g["full/global"], atol=1e-3 if fp32 else 6e-2)
    ref_loss = float(g["full/loss"])
    assert abs(float(loss) - ref_loss) <= (_half_ulp(ref_loss) if fp32 else 0.02 * abs(ref_loss))
    if fp32:
        for i, k in enumerate(("ncorrect", "positive_correct", "total_positive", "pred_positive")):
            assert int(log[k]) == int(g[f"full/{k}"]), k
    grads = {k: p.grad for k, p in named_canonical_params(model).items()}
    n_checked = 0
    for key in [k for k in g.files if k.startswith("full/gnorm/")]:
        name = key[len("full/gnorm/"):]
        gn = float(g[key])
        gr = split_qkv_grad(name, grads)
        if gn < 0:
            assert gr is None or float(gr.abs().max()) == 0.0, f"{name}: the reference gives no gradient"
            continue
        assert gr is not None, name
        n_checked += 1
        if fp32:
            assert abs(float(gr.double().norm()) - gn) <= 1e-3 * max(1.0, gn), (name, float(gr.norm()), gn)
            d = float(np.abs(gr.flatten()[:64].float().cpu().numpy() - g["full/gslice/" + name]).max())
            assert d <= 1e-3 * max(1.0, gn), (name, d)
        elif gn > 1e-3 and gr.numel() >= 4096:
            assert abs(float(gr.double().norm()) - gn) <= 0.06 * gn, (name, float(gr.norm()), gn)
    assert n_checked == int(g["full/n_trainable_with_grad"])
    assert float(split_qkv_grad("layers.2.layers.0.fc1.weight", grads).abs().max()) > 0      # final graph stack: live
