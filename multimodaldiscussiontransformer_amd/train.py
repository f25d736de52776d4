"""Minimal ``fairseq-train``-compatible launcher for the HIP-backed mDT (fairseq itself is optional).

Accepts the flags of mDT/experiments/hateful_discussions/run_train.sh:28-65 with the same
spellings (flags it does not implement are refused; the few it accepts without effect —
--wandb-project, --num-workers, --required-batch-size-multiple — say so on stderr), resolves --task / --arch / --criterion / --dataset-name through
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
    p.add_argument("--fp8", action="store_true", default=False,
                   help="BASELINE.json configs[4] (not a reference flag): bf16 run with per-tensor-scaled fp8 operands in the blocks' big GEMMs (fp8.py)")
    p.add_argument("--fp8-sites", default=None, help="preset (all | fast4 | grads) or comma list of fp8 sites (fp8.PRESETS); default all")
    p.add_argument("--log-interval", type=int, default=10)
    # criterion flags: every field of every registered criterion's config dataclass (FairSeq derives them the same way)
    from .registry import CRITERION_REGISTRY as _CR
    for _cls, _dc in _CR.values():
        for f in getattr(_dc, "__dataclass_fields__", {}).values():
            if f.name.startswith("_"):
                continue
            flag = "--" + f.name.replace("_", "-")
            if f.type in (bool, "bool"):
                p.add_argument(flag, type=lambda v: str(v).lower() in ("1", "true", "yes"), nargs="?", const=True, default=f.default)
            else:
                p.add_argument(flag, type=type(f.default), default=f.default)
    p.add_argument("--distributed-world-size", type=int, default=1)
    p.add_argument("--save-checkpoint", default="", help="write a FairSeq-layout checkpoint here at the end")
    p.add_argument("--save-dir", default="", help="FairSeq: checkpoint_last.pt (and checkpoint<epoch>.pt) are written here")
    p.add_argument("--save-interval-updates", type=int, default=0)
    p.add_argument("--no-save", action="store_true", default=False)
    p.add_argument("--restore-file", default="", help="FairSeq checkpoint to start from (run_train.sh:58)")
    p.add_argument("--reset-optimizer", action="store_true", default=False)
    p.add_argument("--reset-lr-scheduler", action="store_true", default=False)
    p.add_argument("--reset-meters", action="store_true", default=False)
    p.add_argument("--reset-dataloader", action="store_true", default=False)
    p.add_argument("--clip-norm", type=float, default=0.0)
    p.add_argument("--required-batch-size-multiple", type=int, default=1)
    p.add_argument("--validate-interval-updates", type=int, default=0,
                   help="run_train.sh:42 — every N updates (and after the last one) an eval-mode pass over --valid-subset")
    p.add_argument("--valid-subset", default="valid")
    p.add_argument("--disable-validation", action="store_true", default=False)
    p.add_argument("--synthetic-valid-batches", type=int, default=2)
    p.add_argument("--wandb-project", default=None)
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
        # a flag this launcher does not know may change what is trained: refuse instead of training something else
        raise SystemExit(f"[mdt-train] unsupported FairSeq flag(s): {' '.join(unknown)} — this launcher implements the flags of "
                         "mDT/experiments/hateful_discussions/run_train.sh; run fairseq-train --user-dir <repo>/src for the rest")
    if args.clip_norm and args.clip_norm > 0:
        raise SystemExit("[mdt-train] --clip-norm is not implemented (the reference launch does not use it)")
    if args.wandb_project:
        print("[mdt-train] --wandb-project: metrics are printed as JSON lines, nothing is sent to wandb", file=sys.stderr)
    # accepted for launch-script compatibility, without effect here — said, not silent
    if args.num_workers:
        print("[mdt-train] --num-workers: ignored — batches are packed by the native packer in ONE prefetch thread (data/prefetch.py)", file=sys.stderr)
    if args.required_batch_size_multiple != 1:
        print("[mdt-train] --required-batch-size-multiple: ignored — batches hold exactly --batch-size trees (the last one of an epoch may be smaller)",
              file=sys.stderr)
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
    # the reference's encoders come from from_pretrained() (multigraphormer_graph_encoder.py:236-245): build_model loads the same
    # HuggingFace weights from LOCAL files (--pretrained-bert / --pretrained-vit) unless the run starts from a checkpoint,
    # opts out (--random-init-encoders) or uses custom encoder shapes — a recipe that silently trained from random weights
    # would not be the reference's recipe
    args._skip_pretrained = bool(args.restore_file)
    try:
        model = task.build_model(args)
    except FileNotFoundError as e:
        raise SystemExit(f"[mdt-train] {e}\n[mdt-train] give --pretrained-bert / --pretrained-vit a local path, start from a "
                         f"checkpoint (--restore-file), or pass --random-init-encoders to train from random weights")
    model = model.cuda()
    half = args.fp16 or args.bf16 or args.fp8
    if half:
        model = model.bfloat16()
    model.train()
    fp8_state = model.enable_fp8(sites=args.fp8_sites) if args.fp8 else None      # the optimiser tells it when the weight copies go stale
    dp = DataParallel(model)
    dp.broadcast_parameters()
    crit_cls, _ = CRITERION_REGISTRY[args.criterion]
    crit = crit_cls.build_criterion(args, task)      # constructor arguments by name from the flags, as FairSeq does
    betas = ast.literal_eval(args.adam_betas) if isinstance(args.adam_betas, str) else args.adam_betas
    opt = FusedAdam([p for p in model.parameters() if hasattr(p, "main_grad")], lr=args.lr, betas=betas, eps=args.adam_eps,
                    weight_decay=args.weight_decay)
    sched = PolynomialDecayLR(args.lr, args.end_learning_rate, args.warmup_updates, args.total_num_update, args.power)
    from . import checkpoint as ckpt
    start_update, epoch0 = 0, 1
    if args.restore_file:
        if not os.path.exists(args.restore_file):
            raise SystemExit(f"[mdt-train] --restore-file {args.restore_file} does not exist")
        info = ckpt.load_checkpoint(args.restore_file, model, optimizer=opt, reset_optimizer=args.reset_optimizer,
                                    reset_lr_scheduler=args.reset_lr_scheduler, reset_meters=args.reset_meters,
                                    reset_dataloader=args.reset_dataloader, allow_missing_prefixes=("node_encoder_stack.",))
        start_update, epoch0 = info["num_updates"], info["epoch"]
        if rank == 0:
            print(json.dumps(dict(restored=args.restore_file, num_updates=start_update, optimizer=info["loaded_optimizer"],
                                  missing=len(info["missing"]), unexpected=len(info["unexpected"]))), flush=True)
        dp.broadcast_parameters()

    skip_batches = 0          # batches of the running epoch that the restored checkpoint had already consumed
    if args.restore_file and not args.reset_dataloader:
        skip_batches = int(info.get("iterations_in_epoch", 0))      # (not read from extra_state: --reset-meters empties that, and must not rewind the data)
    valid_batches = None      # callable → iterable of packed batches of the held-out split (this rank's share)
    if args.dataset_name == "synthetic":
        batches = synthetic_batches(args, task, rank)
        n_batches = len(batches)

        def batch_stream():
            while True:
                yield from batches

        if not args.disable_validation and args.synthetic_valid_batches > 0:
            import copy
            va = copy.copy(args)
            va.seed, va.synthetic_batches = args.seed + 104729, args.synthetic_valid_batches     # trees the training stream never shows
            held_out = synthetic_batches(va, task, rank)
            valid_batches = lambda: held_out      # noqa: E731
    elif task.dm is None:
        raise SystemExit(f"[mdt-train] dataset {args.dataset_name!r} is not registered: pass --user-data-dir <package whose "
                         "modules call register_dataset> (mDT/src/tasks/task.py:123-137) or --dataset-name synthetic")
    else:
        # a registered dataset (mDT/src/tasks/task.py:168-204): rank r takes every world-th batch of the epoch order;
        # batches are packed, uploaded and indexed one step ahead by the prefetch thread.  FairSeq keeps the last,
        # partial batch of an epoch (no drop-last): the number of batches per epoch is the CEILING, which is also what
        # --max-epoch turns into a number of updates.
        from .data.prefetch import Prefetcher
        task.collate_device = "cuda"
        ds = task.load_dataset("train")
        per_step = args.batch_size * world
        n_batches = max(1, (len(ds) + per_step - 1) // per_step)

        def index_stream():
            epoch, first = epoch0, skip_batches % n_batches
            while True:
                if hasattr(ds, "set_epoch"):
                    ds.set_epoch(epoch)
                order = list(ds.ordered_indices())
                for b in range(first, n_batches):
                    lo = (b * world + rank) * args.batch_size
                    idx = [int(i) for i in order[lo:lo + args.batch_size]]
                    # the tail batch of an epoch may leave the last ranks without trees; collectives must stay matched, so such
                    # a rank takes the tail batch's first tree again (at most world - 1 trees per epoch are seen twice)
                    yield idx if idx else [int(order[min(b * per_step, len(order) - 1)])]
                epoch, first = epoch + 1, 0

        ge = model.encoder.graph_encoder

        def make(idx):
            sample = ds.collater([ds[i] for i in idx])
            return sample["net_input"]["batched_data"]["_packed"]

        def warm(pb):
            ix = ge._indices(pb)
            if ge.prune_last_layer:
                ge._prune_indices(pb, ix)

        batch_stream = lambda: Prefetcher(index_stream(), make, depth=2, warm=warm)   # noqa: E731
        if not args.disable_validation:
            try:
                vds = task.load_dataset(args.valid_subset)
            except (KeyError, ValueError, AttributeError, FileNotFoundError):
                vds = None
            if vds is not None and len(vds) > 0:
                def valid_batches():
                    order = list(range(len(vds)))
                    nb_ = (len(order) + per_step - 1) // per_step
                    for b in range(nb_):
                        lo = (b * world + rank) * args.batch_size
                        idx = order[lo:lo + args.batch_size]
                        if idx:                         # held-out trees are counted once: a rank without a share sits the batch out
                            yield vds.collater([vds[i] for i in idx])["net_input"]["batched_data"]["_packed"]
    stream = iter(batch_stream())
    if args.dataset_name == "synthetic" and start_update and not args.reset_dataloader:
        for _ in range(start_update * args.update_freq):        # resume the batch order where the checkpoint left it
            next(stream)
    max_update = args.max_update or (args.max_epoch * n_batches // max(1, args.update_freq)) or 50
    scal = torch.zeros(6, dtype=torch.float32, device="cuda")
    history = []
    it = 0
    t0 = time.time()
    lr_for = sched.for_update        # FairSeq's timing: update k runs with the rate of num_updates = k - 1

    best = {"value": None}

    def save(path, upd):
        done = skip_batches + it                     # batches consumed since the restored epoch began
        ckpt.save_checkpoint(path, model, args, optimizer=opt, num_updates=upd, criterion_name=crit_cls.__name__,
                             lr_scheduler_state={"best": best["value"]}, epoch=epoch0 + done // max(1, n_batches),
                             training_time=time.time() - t0,
                             extra_state={"train_iterator": {"version": 2, "epoch": epoch0 + done // max(1, n_batches),
                                                             "iterations_in_epoch": done % max(1, n_batches), "shuffle": True},
                                          "val_loss": best["value"]})

    valid_history = []
    main.valid_history = valid_history

    def validate(upd):
        """FairSeq's ``validate``: eval mode, no gradients, the summed logging outputs of every held-out batch (all ranks) go
        through the criterion's own ``reduce_metrics`` arithmetic (criterions/hatespeech_loss.py:133-173: accuracy,
        precision, recall, F1 with its zero guards)."""
        if valid_batches is None:
            return None
        was = model.training
        model.eval()
        tot = torch.zeros(6, dtype=torch.float32, device="cuda")
        keys = None
        with torch.no_grad():
            for pb in valid_batches():
                loss, sample_size, log = crit(model, {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}})
                keys = [k for k in log if k not in ("loss", "sample_size", "nsentences", "ntokens")]
                tot[0] += loss.detach().float()
                tot[1] += float(sample_size)
                tot[2:6] += torch.stack([log[k] for k in keys]).float()
        model.train(was)
        if world > 1:
            dist.all_reduce(tot)
        if keys is None:
            keys = ["ncorrect", "num_positive_correct", "total_positive", "num_pred_positive"]
        s_ = tot.tolist()
        summed = dict(loss=s_[0], sample_size=s_[1], **dict(zip(keys, s_[2:6])))
        crit_cls.reduce_metrics([summed])            # logs through fairseq.metrics when fairseq is there
        m = {("valid_" + k): v for k, v in crit_cls.compute_metrics([summed]).items()}
        m.update(valid_counters=summed, num_updates=upd, subset=args.valid_subset)
        if best["value"] is None or m["valid_loss"] < best["value"]:
            best["value"] = m["valid_loss"]
        m["valid_best_loss"] = best["value"]
        valid_history.append(m)
        if rank == 0:
            print(json.dumps(m), flush=True)
        return m

    for upd in range(start_update + 1, max_update + 1):
        dp.zero_grad()
        acc = torch.zeros(6, dtype=torch.float32, device="cuda")
        for micro in range(args.update_freq):
            dp.accumulate(micro == args.update_freq - 1)
            pb = next(stream)
            it += 1
            loss, sample_size, log = crit(model, {"nsamples": pb.B, "net_input": {"batched_data": pb.batched_data}})
            loss.backward()
            counter_keys = [k for k in log if k not in ("loss", "sample_size", "nsentences", "ntokens")]   # 4 per criterion
            acc[0] += loss.detach().float()
            acc[1] += float(sample_size)
            acc[2:6] += torch.stack([log[k] for k in counter_keys]).float()
        scal.copy_(acc)
        gscale = dp.finish_backward(scal, fold_scale=True)     # all-reduce (world > 1); 1 / global sample size is applied by the optimiser
        opt.step(lr=lr_for(upd), grad_scale=gscale)
        if upd % args.log_interval == 0 or upd == max_update:
            s = scal.tolist()                    # the only host sync, once per log interval
            m = crit_cls.compute_metrics([dict(loss=s[0], sample_size=s[1], **dict(zip(counter_keys, s[2:6])))])
            m.update(num_updates=upd, lr=lr_for(upd), wall=round(time.time() - t0, 2))
            history.append(m)
            if rank == 0:
                print(json.dumps(m), flush=True)
        if args.validate_interval_updates > 0 and upd % args.validate_interval_updates == 0 and upd != max_update:
            validate(upd)
        if rank == 0 and args.save_dir and not args.no_save and args.save_interval_updates > 0 and upd % args.save_interval_updates == 0:
            for path in ckpt.checkpoint_paths(args.save_dir, epoch0, upd):
                save(path, upd)
    if max_update > start_update:
        validate(max_update)                         # FairSeq validates at the end of training as well
    if rank == 0 and args.save_checkpoint:
        save(args.save_checkpoint, max_update)
    if rank == 0 and args.save_dir and not args.no_save:
        save(os.path.join(args.save_dir, "checkpoint_last.pt"), max_update)
    main.last_run = dict(model=model, criterion=crit, valid_batches=valid_batches, optimizer=opt, fp8=fp8_state)     # for tests / notebooks
    if fp8_state is not None:
        from . import fp8 as _fp8
        _fp8.ACTIVE = None                           # process-wide switch: a later model in this process starts in bf16
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
