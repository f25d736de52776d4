"""Registration surface of the reference (FairSeq ``--user-dir`` plugins): the same names resolve
whether or not fairseq is installed.

  models      multi_graphormer  (archs: multi_graphormer, multi_graphormer_base)
  tasks       node_prediction, contrastive_learning
  criterions  node_cross_entropy
  datasets    via ``register_dataset`` (mDT/src/data/__init__.py:1-8)

When fairseq is importable the decorators forward to fairseq's registries as well, so
``fairseq-train --user-dir <this package>`` picks the HIP-backed implementations up;
otherwise ``multimodaldiscussiontransformer_amd.train`` reads these tables.
"""
from __future__ import annotations

MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}
TASK_REGISTRY = {}
CRITERION_REGISTRY = {}
DATASET_REGISTRY = {}

try:  # pragma: no cover - fairseq is absent from the build image
    import fairseq.models as _fs_models
    import fairseq.tasks as _fs_tasks
    import fairseq.criterions as _fs_crit
    HAVE_FAIRSEQ = True
except Exception:  # noqa: BLE001
    HAVE_FAIRSEQ = False


def register_model(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        if HAVE_FAIRSEQ:
            _fs_models.register_model(name)(cls)
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCH_MODEL_REGISTRY[arch_name] = model_name
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        if HAVE_FAIRSEQ:
            _fs_models.register_model_architecture(model_name, arch_name)(fn)
        return fn
    return deco


def register_task(name, dataclass=None):
    def deco(cls):
        TASK_REGISTRY[name] = (cls, dataclass)
        if HAVE_FAIRSEQ:
            _fs_tasks.register_task(name, dataclass=dataclass)(cls)
        return cls
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        CRITERION_REGISTRY[name] = (cls, dataclass)
        if HAVE_FAIRSEQ:
            _fs_crit.register_criterion(name, dataclass=dataclass)(cls)
        return cls
    return deco


def register_dataset(name: str):
    def deco(fn):
        DATASET_REGISTRY[name] = fn
        return fn
    return deco
