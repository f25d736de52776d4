"""Worker of tests/test_ddp_gpu.py: two ranks (gloo, both on cuda:0) run three data-parallel steps through
ddp.DataParallel with the two-stream tape and compare the exchanged gradient arena with a single-process run over the
union of the two shards.  Launched by ``python -m torch.distributed.run --nproc-per-node 2 tests/ddp_gpu_worker.py``."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")     # two processes share the card: stay within its hardware queue slots

from multimodaldiscussiontransformer_amd import synthetic  # noqa: E402
from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy  # noqa: E402
from multimodaldiscussiontransformer_amd.data.packer import pack_batch  # noqa: E402
from multimodaldiscussiontransformer_amd.ddp import DataParallel  # noqa: E402
from multimodaldiscussiontransformer_amd.models import GraphormerModel  # noqa: E402
from oracle import cases  # noqa: E402
from tests.util_model import fill_hash_weights, model_args  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    hp = cases.tiny_hparams("A")
    trees = synthetic.make_trees(8, 9, seed=314, variable=True, seq_len=16, vocab_size=hp.vocab_size, image_frac=0.5,
                                 image_size=hp.image_size, min_len=3)
    crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)

    def build():
        m = GraphormerModel.build_model(model_args(hp), task=None)
        fill_hash_weights(m)
        return m.cuda().eval()          # eval: no dropout, so the union batch is the sum of the shards

    # single-process reference over all trees (ordinary autograd gradients of the summed loss)
    ref = build()
    pb_all = pack_batch(trees, 5)
    loss, n_all, _ = crit(ref, {"nsamples": len(trees), "net_input": {"batched_data": pb_all.batched_data}})
    loss.backward()
    ref_grads = {n: p.grad.detach() / float(n_all) for n, p in ref.named_parameters() if p.grad is not None}

    model = build()
    assert model.encoder.graph_encoder.two_streams
    dp = DataParallel(model, bucket_mb=16)
    assert dp.bucketer.world == world and dp.bucketer.active
    mine = trees[rank::world]
    pb = pack_batch(mine, 5)
    worst = 0.0
    for step in range(3):       # step 0: whole-arena reduce + re-layout; steps 1, 2: static buckets launched during backward
        dp.zero_grad()
        loss, n_mine, _ = crit(model, {"nsamples": len(mine), "net_input": {"batched_data": pb.batched_data}})
        loss.backward()
        scal = torch.zeros(6, device="cuda")
        scal[0] = loss.detach().float()
        scal[1] = float(n_mine)
        dp.finish_backward(scal)
        torch.cuda.synchronize()
        assert abs(float(scal[1]) - float(n_all)) < 0.5, (float(scal[1]), float(n_all))
        if step > 0:
            assert len(dp.bucketer.bucket_ends) > 2, dp.bucketer.bucket_ends
        for n, p in model.named_parameters():
            if n not in ref_grads:
                continue
            g, r = p.main_grad, ref_grads[n]
            err = float((g - r).abs().max()) / max(1e-6, float(r.abs().max()))
            worst = max(worst, err)
            assert err < 2e-4, (step, n, err)
    t = torch.tensor([worst])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"DDP_GPU_OK worst_rel={float(t):.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
