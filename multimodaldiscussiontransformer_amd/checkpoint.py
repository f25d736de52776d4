"""FairSeq-compatible checkpoints: what ``--save-dir`` writes and ``--restore-file`` /
``--reset-optimizer`` read in the reference launch (mDT/experiments/hateful_discussions/run_train.sh:57-58,63).

The reference leaves the format to FairSeq's trainer (``Trainer.state_dict`` / ``checkpoint_utils``): ONE ``torch.save``d
dict —

    args                 None (legacy slot)
    cfg                  {"common", "task", "model", "criterion", "optimizer", "lr_scheduler", ...}   nested config
    model                model.state_dict()     the reference's key names: separate q/k/v projections, HF transformers-4.x
                         inner names, alias entries (text_pooler.* = text_model.pooler.* = node_encoder_stack.0.*), the
                         dead parameters; legacy fused ``in_proj_weight`` entries are split on load
                         (modules/multihead_attention.py:219-248), ``embed_out`` / ``lm_output_learned_bias`` dropped
                         when the head was removed (models/multi_modal_discussion_transformer.py:282-287)
    criterion            None (the criteria have no parameters)
    optimizer_history    [{"criterion_name", "optimizer_name", "lr_scheduler_state", "num_updates"}]
    task_state           {}
    extra_state          {"metrics", "previous_training_time", "train_iterator": {"epoch", ...}}
    last_optimizer_state torch-optimizer state dict of Adam: {"state": {i: {"step", "exp_avg", "exp_avg_sq"}},
                         "param_groups": [{"lr", "betas", "eps", "weight_decay", "params": [0 .. T-1]}]}  — numbered over
                         the TRAINABLE parameters in ``model.parameters()`` order, as torch numbers them (frozen ones are
                         not counted; a parameter that never got a gradient has no ``state`` entry), FairSeq's layout for
                         fp32 training and for
                         ``--fp16-no-flatten-grads``; its flattened-fp32-copy layout holds a single entry instead and is
                         accepted on load when the element counts add up)

``load_checkpoint`` follows ``Trainer.load_checkpoint``: the model is loaded strictly (missing / unexpected keys raise),
optimizer state, update count and LR schedule come back unless ``reset_optimizer`` — the combination the launch script
uses to start fine-tuning from a contrastive pre-training checkpoint.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional

import torch


def _model_state_fp32(model, optimizer) -> "OrderedDict[str, torch.Tensor]":
    """state_dict with the fp32 master copy substituted where the optimizer keeps one (bf16 training)."""
    sd = model.state_dict()
    if optimizer is None:
        return OrderedDict((k, v.detach().cpu()) for k, v in sd.items())
    master = {}
    for p in optimizer.params:
        st = optimizer.state.get(id(p), {})
        if "master" in st:
            master[p.data.data_ptr()] = (p, st["master"])
    out = OrderedDict()
    for k, v in sd.items():
        hit = None
        for ptr0, (p, m) in master.items():
            # a state-dict entry is the parameter itself or a row slice of a fused q/k/v parameter
            beg, end = p.data.data_ptr(), p.data.data_ptr() + p.numel() * p.element_size()
            if beg <= v.data_ptr() < end and v.dtype == p.dtype:
                off = (v.data_ptr() - beg) // p.element_size()
                hit = m.view(-1)[off:off + v.numel()].view(v.shape)
                break
        out[k] = (hit if hit is not None else v).detach().float().cpu() if v.is_floating_point() else v.detach().cpu()
    return out


def _trainable(model, criterion=None):
    """The list FairSeq builds its optimizer over (fairseq/trainer.py ``_build_optimizer``): the parameters of model then
    criterion that require gradients, in ``parameters()`` order.  A torch / FairSeq Adam ``state_dict`` numbers exactly
    these 0 .. T-1 — frozen parameters (``--freeze_initial_encoders``: the BERT / ViT prefix, which comes FIRST in
    ``model.parameters()``) are not counted — and holds a ``state`` entry only for parameters that ever received a
    gradient (the reference's dead parameters never do)."""
    ps = [p for p in model.parameters() if p.requires_grad]
    if criterion is not None:
        ps += [p for p in criterion.parameters() if p.requires_grad]
    return ps


def optimizer_state_dict(optimizer, model, criterion=None) -> dict:
    params = _trainable(model, criterion)
    index = {id(p): i for i, p in enumerate(params)}
    state = {}
    for p in optimizer.params:
        i = index[id(p)]
        st = optimizer.state[id(p)]
        state[i] = {"step": optimizer.step_count, "exp_avg": st["m"].detach().cpu(), "exp_avg_sq": st["v"].detach().cpu()}
    group = {"lr": optimizer.lr, "betas": tuple(optimizer.betas), "eps": optimizer.eps, "weight_decay": optimizer.weight_decay,
             "amsgrad": False, "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(optimizer, model, osd: dict, criterion=None):
    """Positional, as ``torch.optim.Optimizer.load_state_dict`` maps them: the ids listed in ``param_groups[*]["params"]``
    (in order) belong to the trainable parameters (in order).  A parameter without a ``state`` entry (never stepped in
    the run that wrote the checkpoint) starts from zero moments; shapes must agree exactly — a shifted numbering must
    fail loudly, not broadcast another parameter's moments in."""
    params = _trainable(model, criterion)
    st_in = osd["state"]
    ids = [i for g in osd["param_groups"] for i in g["params"]]
    if len(st_in) == 1 and len(ids) <= 1 and len(optimizer.params) > 1:
        # FairSeq FP16Optimizer with flattened fp32 copies: one flat tensor over the trainable parameters in order
        (flat,) = st_in.values()
        total = sum(p.numel() for p in params)
        if flat["exp_avg"].numel() != total:
            raise ValueError(f"flattened optimizer state holds {flat['exp_avg'].numel()} elements, the model has {total} trainable")
        off = 0
        for p in params:
            n = p.numel()
            st = optimizer.state.get(id(p))
            if st is not None:
                st["m"].copy_(flat["exp_avg"].view(-1)[off:off + n].view(p.shape))
                st["v"].copy_(flat["exp_avg_sq"].view(-1)[off:off + n].view(p.shape))
            off += n
        step = flat.get("step", 0)
        optimizer.step_count = int(step.item() if torch.is_tensor(step) else step)
    else:
        if len(ids) != len(params):
            raise ValueError(f"optimizer state numbers {len(ids)} parameters, the model has {len(params)} trainable ones "
                             f"(frozen parameters are not numbered: check --freeze_initial_encoders against the checkpoint)")
        key_of = {id(p): k for p, k in zip(params, ids)}
        steps = set()
        for p in optimizer.params:
            st = optimizer.state[id(p)]
            ent = st_in.get(key_of[id(p)])
            if ent is None:                          # never stepped in the run that wrote the checkpoint: zero moments
                st["m"].zero_()
                st["v"].zero_()
                continue
            for name, dst in (("exp_avg", st["m"]), ("exp_avg_sq", st["v"])):
                if tuple(ent[name].shape) != tuple(dst.shape):
                    raise ValueError(f"optimizer state #{key_of[id(p)]}: {name} has shape {tuple(ent[name].shape)}, the parameter "
                                     f"at that position has {tuple(dst.shape)}")
                dst.copy_(ent[name])
            step = ent["step"]
            steps.add(int(step.item() if torch.is_tensor(step) else step))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ: {sorted(steps)}")
        optimizer.step_count = steps.pop() if steps else 0
    g = osd["param_groups"][0]
    optimizer.lr, optimizer.betas, optimizer.eps, optimizer.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


def save_checkpoint(path: str, model, args, *, optimizer=None, lr_scheduler_state: Optional[dict] = None, num_updates: int = 0,
                    criterion_name: str = "GraphPredictionNodeCrossEntropy", epoch: int = 1, extra_state: Optional[dict] = None,
                    training_time: float = 0.0) -> dict:
    cfg_model = {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, type(None), list, tuple, dict))}
    state = {
        "args": None,
        "cfg": {"_name": None, "common": {"seed": cfg_model.get("seed", 1), "fp16": bool(cfg_model.get("fp16", False)),
                                          "bf16": bool(cfg_model.get("bf16", False)), "user_dir": cfg_model.get("user_dir")},
                "task": {"_name": cfg_model.get("task")}, "model": dict(cfg_model, _name=cfg_model.get("arch")),
                "criterion": {"_name": cfg_model.get("criterion")},
                "optimizer": {"_name": cfg_model.get("optimizer", "adam")},
                "lr_scheduler": {"_name": cfg_model.get("lr_scheduler", "polynomial_decay")}},
        "model": _model_state_fp32(model, optimizer),
        "criterion": None,
        "optimizer_history": [{"criterion_name": criterion_name, "optimizer_name": "FairseqAdam",
                               "lr_scheduler_state": dict(lr_scheduler_state or {}), "num_updates": int(num_updates)}],
        "task_state": {},
        "extra_state": dict({"metrics": {}, "previous_training_time": float(training_time),
                             "train_iterator": {"version": 2, "epoch": int(epoch), "iterations_in_epoch": 0, "shuffle": True}},
                            **(extra_state or {})),
    }
    if optimizer is not None:
        state["last_optimizer_state"] = optimizer_state_dict(optimizer, model)
    tmp = path + ".tmp"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, tmp)
    os.replace(tmp, path)               # a killed run never leaves a half-written checkpoint behind
    return state


def upgrade_state_dict(model, sd: dict) -> dict:
    """The in-repo ``upgrade_state_dict_named`` hooks FairSeq calls before loading (fairseq BaseFairseqModel
    .upgrade_state_dict): every sub-module that defines one sees the dict with its own prefix."""
    sd = OrderedDict(sd)
    for name, m in model.named_modules():
        fn = getattr(m, "upgrade_state_dict_named", None)
        if fn is not None and m is not model:
            fn(sd, name)
    return sd


def load_checkpoint(path: str, model, *, optimizer=None, reset_optimizer: bool = False, reset_lr_scheduler: bool = False,
                    reset_meters: bool = False, reset_dataloader: bool = False, strict: bool = True,
                    allow_missing_prefixes=()) -> dict:
    """→ dict(num_updates, lr_scheduler_state, epoch, extra_state, missing, unexpected).  ``allow_missing_prefixes``:
    model-level keys a checkpoint of another task legitimately lacks (``node_encoder_stack.`` — the fresh classifier
    list tasks/node_prediction.py:44-53 attaches AFTER the checkpointed model was built)."""
    state = torch.load(path, map_location="cpu", weights_only=False)
    if "model" not in state:
        raise KeyError(f"{path}: not a FairSeq checkpoint (no 'model' entry; keys: {sorted(state)[:8]})")
    sd = upgrade_state_dict(model, state["model"])
    target = model.state_dict()
    for k, v in sd.items():                      # load_state_dict casts, but silently accepts only equal shapes: say which key
        if k in target and tuple(target[k].shape) != tuple(v.shape):
            raise ValueError(f"{path}: shape of {k} is {tuple(v.shape)}, the model expects {tuple(target[k].shape)}")
    # bf16 training keeps fp32 master weights: load INTO the masters (exact), then round them into the working copies
    swapped = []
    if optimizer is not None:
        for p_ in optimizer.params:
            st = optimizer.state[id(p_)]
            if "master" in st:
                swapped.append((p_, p_.data))
                p_.data = st["master"]
    try:
        res = model.load_state_dict(sd, strict=False)
    finally:
        for p_, low in swapped:
            low.copy_(p_.data)
            p_.data = low
    missing = [k for k in res.missing_keys if not k.startswith(tuple(allow_missing_prefixes))]
    if strict and (missing or res.unexpected_keys):
        raise RuntimeError(f"{path}: missing keys {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected keys {list(res.unexpected_keys)[:8]}")
    out = dict(num_updates=0, lr_scheduler_state={}, epoch=1, iterations_in_epoch=0, extra_state={}, missing=list(res.missing_keys),
               unexpected=list(res.unexpected_keys), loaded_optimizer=False)
    hist = state.get("optimizer_history") or []
    if optimizer is not None and not reset_optimizer and state.get("last_optimizer_state") is not None and hist:
        load_optimizer_state_dict(optimizer, model, state["last_optimizer_state"])
        out["num_updates"] = int(hist[-1].get("num_updates", 0))
        out["loaded_optimizer"] = True
        if not reset_lr_scheduler:
            out["lr_scheduler_state"] = dict(hist[-1].get("lr_scheduler_state") or {})
    extra = state.get("extra_state") or {}
    if not reset_meters:
        out["extra_state"] = extra
    if not reset_dataloader:      # FairSeq ties the iterator state (epoch AND position inside it) to --reset-dataloader alone
        it = extra.get("train_iterator") or {}
        out["epoch"] = int(it.get("epoch", 1))
        out["iterations_in_epoch"] = int(it.get("iterations_in_epoch", 0) or 0)
    from . import engine
    engine.weights_changed()          # cached transposed weight copies (engine.dgrad) are stale
    return out


def checkpoint_paths(save_dir: str, epoch: Optional[int] = None, num_updates: Optional[int] = None):
    """File names FairSeq uses under --save-dir."""
    names = ["checkpoint_last.pt"]
    if epoch is not None:
        names.insert(0, f"checkpoint{epoch}.pt")
    if num_updates is not None:
        names.insert(0, f"checkpoint_{epoch or 1}_{num_updates}.pt")
    return [os.path.join(save_dir, n) for n in names]
