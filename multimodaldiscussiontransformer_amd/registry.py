"""Registration surface of the reference (FairSeq ``--user-dir`` plugins): the same names resolve
whether or not fairseq is installed.

  models      multi_graphormer  (archs: multi_graphormer, multi_graphormer_base)
  tasks       node_prediction, contrastive_learning
  criterions  node_cross_entropy, contrastive_loss
  datasets    via ``register_dataset`` (mDT/src/data/__init__.py:1-8)

The reference's classes extend FairSeq's base classes — ``GraphormerModel(FairseqEncoderModel)`` /
``GraphormerEncoder(FairseqEncoder)`` (mDT/src/models/multi_modal_discussion_transformer.py:22-23,181),
``Task(ABC, FairseqTask)`` (mDT/src/tasks/task.py:116), ``GraphPredictionNodeCrossEntropy(FairseqCriterion)``
(mDT/src/criterions/hatespeech_loss.py:40-43), configs ``FairseqDataclass`` — and FairSeq's registries REJECT
classes that do not (``register_model``: "must extend BaseFairseqModel", ``register_task``: "must extend
FairseqTask", the criterion registry: "must extend FairseqCriterion", dataclasses: "must extend FairseqDataclass").
So the product classes extend the names exported HERE:

  * fairseq importable  → these ARE fairseq's classes and the decorators forward to fairseq's registries, so
    ``fairseq-train --user-dir <repo>/src`` builds the HIP-backed model / task / criterion;
  * fairseq absent      → minimal local base classes with the same constructor signatures, and
    ``multimodaldiscussiontransformer_amd.train`` reads the tables below.

tests/test_fairseq_boundary_cpu.py imports the package under an in-repo stand-in ``fairseq`` that enforces the same
``issubclass`` checks as the real registries.
"""
from __future__ import annotations

import torch.nn as nn

MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}
TASK_REGISTRY = {}
CRITERION_REGISTRY = {}
DATASET_REGISTRY = {}

try:
    import fairseq  # noqa: F401
    HAVE_FAIRSEQ = True
except ImportError:
    HAVE_FAIRSEQ = False

if HAVE_FAIRSEQ:
    # a fairseq that imports but lacks these names is a broken install: let the ImportError surface
    import fairseq.criterions as _fs_crit
    import fairseq.models as _fs_models
    import fairseq.tasks as _fs_tasks
    from fairseq.criterions import FairseqCriterion
    from fairseq.dataclass.configs import FairseqDataclass
    from fairseq.models import FairseqEncoder, FairseqEncoderModel
    from fairseq.tasks import FairseqTask
else:
    from dataclasses import dataclass

    @dataclass
    class FairseqDataclass:
        """Stand-in for fairseq.dataclass.configs.FairseqDataclass (a plain dataclass base)."""

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary=None):
            super().__init__()
            self.dictionary = dictionary

    class FairseqEncoderModel(nn.Module):
        def __init__(self, encoder):
            super().__init__()
            self.encoder = encoder

    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task

        @classmethod
        def build_criterion(cls, cfg, task):
            """fairseq.criterions.FairseqCriterion.build_criterion: constructor arguments by name from ``cfg``."""
            import inspect
            kw = {}
            for p in inspect.signature(cls).parameters.values():
                if p.name == "task":
                    kw["task"] = task
                elif hasattr(cfg, p.name):
                    kw[p.name] = getattr(cfg, p.name)
                elif p.default is p.empty:
                    raise NotImplementedError(f"unable to infer criterion argument {p.name!r} from the config")
            return cls(**kw)

    class FairseqTask:
        def __init__(self, cfg, **kwargs):
            self.cfg = cfg
            self.datasets = {}
            self.dataset_to_epoch_iter = {}

        @classmethod
        def setup_task(cls, cfg, **kwargs):
            return cls(cfg, **kwargs)

        def dataset(self, split):
            if split not in self.datasets:
                raise KeyError("Dataset not loaded: " + split)
            return self.datasets[split]

        def build_criterion(self, cfg):
            name = getattr(cfg, "_name", None) or getattr(cfg, "criterion", None)
            cls, _ = CRITERION_REGISTRY[name]
            return cls.build_criterion(cfg, self)


def register_model(name):
    def deco(cls):
        if not issubclass(cls, FairseqEncoderModel):
            raise ValueError(f"Model ({name}: {cls.__name__}) must extend FairseqEncoderModel")
        MODEL_REGISTRY[name] = cls
        if HAVE_FAIRSEQ:
            _fs_models.register_model(name)(cls)
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError(f"Cannot register model architecture for unknown model type ({model_name})")
        ARCH_MODEL_REGISTRY[arch_name] = model_name
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        if HAVE_FAIRSEQ:
            _fs_models.register_model_architecture(model_name, arch_name)(fn)
        return fn
    return deco


def _check_dataclass(dataclass, what):
    if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
        raise ValueError(f"Dataclass {dataclass} of {what} must extend FairseqDataclass")


def register_task(name, dataclass=None):
    def deco(cls):
        if not issubclass(cls, FairseqTask):
            raise ValueError(f"Task ({name}: {cls.__name__}) must extend FairseqTask")
        _check_dataclass(dataclass, f"task {name}")
        TASK_REGISTRY[name] = (cls, dataclass)
        if HAVE_FAIRSEQ:
            _fs_tasks.register_task(name, dataclass=dataclass)(cls)
        return cls
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        if not issubclass(cls, FairseqCriterion):
            raise ValueError(f"{cls.__name__} must extend FairseqCriterion")
        _check_dataclass(dataclass, f"criterion {name}")
        CRITERION_REGISTRY[name] = (cls, dataclass)
        if HAVE_FAIRSEQ:
            _fs_crit.register_criterion(name, dataclass=dataclass)(cls)
        return cls
    return deco


def register_dataset(name: str):
    """mDT/src/data/__init__.py:4-8 (the reference's decorator returns None; returning the function is harmless)."""
    def deco(fn):
        DATASET_REGISTRY[name] = fn
        return fn
    return deco
