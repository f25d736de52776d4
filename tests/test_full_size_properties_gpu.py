"""BASELINE.json configs[1] at its FULL size (mDT-base, 32 bushy 64-comment trees, 25 % image comments, L = 100, 224-px images) —
where the CPU oracle would take an hour — through properties that do not depend on the size:

  * a discussion tree's logits do not depend on the company it keeps: every tree of the 32-tree batch gets, BIT FOR BIT, the logits it
    gets in a batch of its own and in the batch reversed (attention is per tree / per comment; a GEMM row's k-sum does not depend
    on where the row sits in its tile — the property that lets data-parallel ranks deal trees freely);
  * the loss is the sum of the trees' losses (criterions/hatespeech_loss.py sums), its counters the sums of theirs;
  * gradients are additive over trees: the gradient arena of the batch equals the sum of the arenas of its two halves up to the
    order of fp32 additions (split-K atomics, accumulation into the arena);
  * the step is finite everywhere and, with dropout off, repeats itself bit for bit in the logits.

bf16, the production kernels (persistent MFMA GEMMs, ragged text, length-binned attention, pruned last fusion layer, two streams)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    import bench
    from multimodaldiscussiontransformer_amd import synthetic
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    args = argparse.Namespace(config="base", num_fusion_layers=5, freeze_initial_encoders=False, dropout=0.0, attention_dropout=0.0, act_dropout=0.0)
    torch.manual_seed(4321)
    model = GraphormerModel.build_model(bench.base_args(args), task=None).cuda().bfloat16().eval()
    model.prepare_main_grads()
    trees = synthetic.make_trees(32, 64, seed=1234, seq_len=100, image_frac=0.25, image_size=224, shape="bushy")
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)

    def run(ts, backward=False):
        pb = pack_batch(ts, spatial_pos_max=5)
        if not backward:
            with torch.no_grad():
                logits, _ = model(pb.batched_data)
            return logits.float().cpu()
        model.zero_main_grads()
        loss, n, log = crit(model, {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), n, {k: int(log[k]) for k in ("ncorrect", "num_positive_correct", "total_positive", "num_pred_positive")}, model.main_grad_flat.clone()
    return trees, run


def per_tree(logits, trees):
    """logits [M, 2], one row per comment in batch order → one [N_i, 2] block per tree"""
    assert logits.shape == (sum(len(t["parent"]) for t in trees), 2)
    return list(torch.split(logits, [len(t["parent"]) for t in trees]))


def test_a_trees_logits_do_not_depend_on_its_batch(full):
    trees, run = full
    whole = per_tree(run(trees), trees)
    assert all(bool(torch.isfinite(w).all()) for w in whole)
    again = per_tree(run(trees), trees)
    assert all(torch.equal(a, b) for a, b in zip(whole, again))                       # the step repeats itself
    rev = per_tree(run(trees[::-1]), trees[::-1])[::-1]
    assert all(torch.equal(a, b) for a, b in zip(whole, rev))                         # order in the batch
    for i in (0, 13, 31):
        alone = per_tree(run([trees[i]]), [trees[i]])[0]
        assert torch.equal(alone, whole[i]), i                                        # a batch of its own
    half = per_tree(run(trees[8:24]), trees[8:24])
    assert all(torch.equal(a, b) for a, b in zip(half, whole[8:24]))


def test_loss_counters_and_gradients_are_additive_over_trees(full):
    trees, run = full
    loss, n, counters, arena = run(trees, backward=True)
    assert bool(torch.isfinite(arena).all()) and n == 32
    l0, n0, c0, a0 = run(trees[:16], backward=True)
    l1, n1, c1, a1 = run(trees[16:], backward=True)
    assert n0 + n1 == n and all(c0[k] + c1[k] == counters[k] for k in counters)
    assert abs((l0 + l1) - loss) <= 2e-3 * max(1.0, abs(loss))                        # half-precision loss values summed in fp32
    s = a0.double() + a1.double()
    rel = float((arena.double() - s).norm() / s.norm())
    assert rel <= 1e-5, rel                                                           # the order of fp32 additions only
