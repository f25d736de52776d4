"""Host enqueue time of one step with the GPU idle at the start (is the step CPU- or GPU-bound?). GPU box only."""
import sys, time
sys.argv = ["bench.py"]
sys.path.insert(0, ".")
import torch
import bench
from multimodaldiscussiontransformer_amd import synthetic
from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy
from multimodaldiscussiontransformer_amd.data.packer import pack_batch
from multimodaldiscussiontransformer_amd.models import GraphormerModel
import argparse
a = argparse.Namespace(num_fusion_layers=5, freeze_initial_encoders=False, dropout=0.4, attention_dropout=0.3, act_dropout=0.3)
torch.manual_seed(1234)
model = GraphormerModel.build_model(bench.base_args(a), task=None).cuda().bfloat16().train()
model.prepare_main_grads()
crit = GraphPredictionNodeCrossEntropy(None, positive_weight=1.5, negative_weight=1.0)
pb = pack_batch(synthetic.make_trees(32, 64, seed=1234, seq_len=100, image_frac=0.25, image_size=224), spatial_pos_max=5)
sample = {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}}
def step():
    model.main_grad_flat.zero_()
    loss, _, _ = crit(model, sample)
    loss.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, then wait {1e3*(t2-t1):.1f} ms (total {1e3*(t2-t0):.1f} ms)")
if "--profile" in sys.orig_argv if hasattr(sys, "orig_argv") else False:
    pass
import cProfile, pstats, os
if os.environ.get("MDT_ISSUE_PROFILE") == "1":
    pr = cProfile.Profile()
    pr.enable()
    step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
