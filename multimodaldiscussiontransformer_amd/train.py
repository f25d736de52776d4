"""Minimal ``fairseq-train``-compatible launcher for the HIP-backed mDT (fairseq itself is optional).

Accepts the flags of mDT/experiments/hateful_discussions/run_train.sh:28-65 with the same
spellings (unknown FairSeq flags such as --save-dir / --wandb-project are accepted and
ignored with a note), resolves --task / --arch / --criterion / --dataset-name through
``registry.py``, and runs the reference recipe: Adam(0.9, 0.999, eps 1e-8, wd 0.01),
polynomial-decay LR with warm-up, --update-freq gradient accumulation, gradient scaling by
1 / (global) sample size, bf16 (``--fp16`` in the reference) with fp32 master weights, RCCL
data parallelism when launched under ``torch.distributed.run``.

Without a registered dataset (the HatefulDiscussions graphs are private) ``--dataset-name
synthetic`` trains on generated discussion trees whose label depends on the labelled comment's
text, which is what the smoke test uses.

  python -m multimodaldiscussiontransformer_amd.train --task node_prediction --arch multi_graphormer_base \\
      --criterion node_cross_entropy --dataset-name synthetic --batch-size 12 --max-update 50 ...
"""
from __future__ import annotations

import argparse
import ast
import json
import os
import sys
import time

import numpy as np
import torch

from . import synthetic
from .registry import ARCH_CONFIG_REGISTRY, ARCH_MODEL_REGISTRY, CRITERION_REGISTRY, MODEL_REGISTRY, TASK_REGISTRY


def build_parser():
    from . import criterions, models, tasks  # noqa: F401  (registration side effects)
    # argument_default=SUPPRESS: model flags that are not given stay absent, so the architecture functions
    # fill their defaults exactly as under fairseq-train
    p = argparse.ArgumentParser(allow_abbrev=False, argument_default=argparse.SUPPRESS)
    p.add_argument("--user-dir", default=None)
    p.add_argument("--num-workers", type=int, default=0)
    p.add_argument("--task", default="node_prediction")
    p.add_argument("--criterion", default="node_cross_entropy")
    p.add_argument("--arch", default="multi_graphormer_base")
    p.add_argument("--optimizer", default="adam")
    p.add_argument("--adam-betas", default="(0.9, 0.999)")
    p.add_argument("--adam-eps", type=float, default=1e-8)
    p.add_argument("--weight-decay", type=float, default=0.01)
    p.add_argument("--lr-scheduler", default="polynomial_decay")
    p.add_argument("--power", type=float, default=1.0)
    p.add_argument("--warmup-updates", type=int, default=0)
    p.add_argument("--total-num-update", type=int, default=10820)
    p.add_argument("--lr", type=float, default=3e-5)
    p.add_argument("--end-learning-rate", type=float, default=3e-7)
    p.add_argument("--batch-size", type=int, default=12)
    p.add_argument("--update-freq", type=int, default=1)
    p.add_argument("--max-epoch", type=int, default=0)
    p.add_argument("--max-update", type=int, default=0)
    p.add_argument("--fp16", action="store_true", default=False, help="the reference's half precision; runs as bf16 here")
    p.add_argument("--bf16", action="store_true", default=False)
    p.add_argument("--log-interval", type=int, default=10)
    p.add_argument("--positive-weight", type=float, default=1.0)
    p.add_argument("--negative-weight", type=float, default=1.0)
    p.add_argument("--distributed-world-size", type=int, default=1)
    p.add_argument("--save-checkpoint", default="", help="write model.state_dict() (reference key names) here at the end")
    # synthetic data controls
    p.add_argument("--synthetic-nodes", type=int, default=16)
    p.add_argument("--synthetic-seq-len", type=int, default=32)
    p.add_argument("--synthetic-image-frac", type=float, default=0.0)
    p.add_argument("--synthetic-batches", type=int, default=8)
    p.add_argument("--bert-config", type=json.loads, default=None, help="JSON overrides of the BERT shape (tests)")
    p.add_argument("--vit-config", type=json.loads, default=None)
    # task flags (mDT/src/tasks/task.py:29-113)
    from .tasks import TaskConfig
    for f in TaskConfig.__dataclass_fields__.values():
        flag = "--" + f.name.replace("_", "-")
        if f.type in (bool, "bool"):
            p.add_argument(flag, action="store_true", default=False)
        else:
            p.add_argument(flag, type=type(f.default), default=f.default)
    MODEL_REGISTRY["multi_graphormer"].add_args(p)
    return p


def synthetic_batches(args, task, rank):
    """Discussion trees whose labelled comment is hateful iff its second token id is in the upper half of
    the vocabulary — learnable from the text encoder alone, so a few dozen updates move the loss."""
    vocab = (args.bert_config or {}).get("vocab", 30522)
    img = (args.vit_config or {}).get("image_size", 224)
    out = []
    for i in range(args.synthetic_batches):
        trees = synthetic.make_trees(args.batch_size, args.synthetic_nodes, seed=args.seed * 7919 + rank * 1000 + i,
                                     seq_len=args.synthetic_seq_len, vocab_size=vocab, image_frac=args.synthetic_image_frac,
                                     image_size=img, min_len=4)
        for t in trees:
            m = int(np.nonzero(t["y_mask"])[0][0])
            t["y"] = np.asarray([1.0 if t["input_ids"][m, 1] >= vocab // 2 else 0.0], dtype=np.float32)
        out.append(task.collate(trees))
    return out


def main(argv=None):
    parser = build_parser()
    args, unknown = parser.parse_known_args(argv)
    if unknown:
        print(f"[mdt-train] ignoring FairSeq flags outside the accelerated path: {' '.join(unknown)}", file=sys.stderr)
    import torch.distributed as dist
    from .ddp import DataParallel
    from .optim import FusedAdam, PolynomialDecayLR
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    torch.manual_seed(args.seed)

    task_cls, task_cfg_cls = TASK_REGISTRY[args.task]
    cfg = task_cfg_cls(**{k: getattr(args, k) for k in task_cfg_cls.__dataclass_fields__})
    task = task_cls.setup_task(cfg)
    ARCH_CONFIG_REGISTRY[args.arch](args)
    assert ARCH_MODEL_REGISTRY[args.arch] in MODEL_REGISTRY
    model = task.build_model(args).cuda()
    half = args.fp16 or args.bf16
    if half:
        model = model.bfloat16()
    model.train()
    dp = DataParallel(model)
    dp.broadcast_parameters()
    crit_cls, _ = CRITERION_REGISTRY[args.criterion]
    crit = crit_cls(task, positive_weight=args.positive_weight, negative_weight=args.negative_weight)
    betas = ast.literal_eval(args.adam_betas) if isinstance(args.adam_betas, str) else args.adam_betas
    opt = FusedAdam([p for p in model.parameters() if hasattr(p, "main_grad")], lr=args.lr, betas=betas, eps=args.adam_eps,
                    weight_decay=args.weight_decay)
    sched = PolynomialDecayLR(args.lr, args.end_learning_rate, args.warmup_updates, args.total_num_update, args.power)

    if args.dataset_name == "synthetic" or task.dm is None:
        batches = synthetic_batches(args, task, rank)
    else:
        raise SystemExit("registered datasets are iterated by FairSeq's data pipeline, which is outside this launcher")
    max_update = args.max_update or (args.max_epoch * len(batches) // max(1, args.update_freq)) or 50
    scal = torch.zeros(6, dtype=torch.float32, device="cuda")
    history = []
    it = 0
    t0 = time.time()
    for upd in range(1, max_update + 1):
        dp.zero_grad()
        acc = torch.zeros(6, dtype=torch.float32, device="cuda")
        for micro in range(args.update_freq):
            dp.accumulate(micro == args.update_freq - 1)
            pb = batches[it % len(batches)]
            it += 1
            loss, sample_size, log = crit(model, {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
            acc[0] += loss.detach().float()
            acc[1] += float(sample_size)
            acc[2:6] += torch.stack([log["ncorrect"], log["num_positive_correct"], log["total_positive"],
                                     log["num_pred_positive"]]).float()
        scal.copy_(acc)
        dp.finish_backward(scal)                 # all-reduce (world > 1) and divide by the global sample size
        opt.step(lr=sched(upd))
        if upd % args.log_interval == 0 or upd == max_update:
            s = scal.tolist()                    # the only host sync, once per log interval
            m = crit_cls.compute_metrics([dict(loss=s[0], sample_size=s[1], ncorrect=s[2], num_positive_correct=s[3],
                                               total_positive=s[4], num_pred_positive=s[5])])
            m.update(num_updates=upd, lr=sched(upd), wall=round(time.time() - t0, 2))
            history.append(m)
            if rank == 0:
                print(json.dumps(m), flush=True)
    if args.save_checkpoint and rank == 0:
        torch.save({"model": model.state_dict(), "args": vars(args)}, args.save_checkpoint)
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
