"""``--user-dir`` alias: the reference's launch scripts say ``fairseq-train --user-dir ../../src``
(mDT/experiments/hateful_discussions/run_train.sh:29), and the reference's own files import each other through the
package name ``src`` (``from src.models import GraphormerModel`` — mDT/src/criterions/hatespeech_loss.py:18,
``from src.data import register_dataset`` — mDT/experiments/hateful_discussions/datasets/dataset.py:1).

Put this directory where ``mDT/src`` was (or symlink it) and the scripts run unchanged: importing ``src`` imports the
MI355X package once — which registers model ``multi_graphormer`` (+ archs), tasks ``node_prediction`` /
``contrastive_learning`` and criterions ``node_cross_entropy`` / ``contrastive_loss`` with FairSeq — and then maps every
one of its modules under ``src.*`` as the SAME module object (no second execution, hence no duplicate registration),
so ``src.models``, ``src.tasks.node_prediction``, ``src.data.collator`` … resolve exactly as they did in the reference.
Like the reference's ``src/__init__.py`` (:5-16) it forces the ``fork`` start method for DataLoader workers.
"""
import importlib
import os
import sys

_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _repo not in sys.path:
    sys.path.insert(0, _repo)

_pkg = importlib.import_module("multimodaldiscussiontransformer_amd")
for _sub in ("criterions", "data", "models", "modules", "tasks"):
    importlib.import_module(f"{_pkg.__name__}.{_sub}")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_pkg.__name__ + ".") and _mod is not None:
        sys.modules[__name__ + _name[len(_pkg.__name__):]] = _mod
        _head = _name[len(_pkg.__name__) + 1:]
        if "." not in _head:
            globals()[_head] = _mod

try:
    import torch

    torch.multiprocessing.set_start_method("fork", force=True)
except Exception:  # noqa: BLE001  (same guard as the reference)
    print("Your OS does not support multiprocessing based on fork, please use num_workers=0", file=sys.stderr, flush=True)
