import torch.nn as nn

from .dataclass import FairseqDataclass

MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_MODEL_NAME_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}


class BaseFairseqModel(nn.Module):
    def __init__(self):
        super().__init__()
        self._is_generation_fast = False

    @classmethod
    def add_args(cls, parser):
        pass

    @classmethod
    def build_model(cls, args, task):
        raise NotImplementedError("Model must implement the build_model method")


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class FairseqEncoderModel(BaseFairseqModel):
    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        assert isinstance(self.encoder, FairseqEncoder), "FairseqEncoderModel needs a FairseqEncoder"


def register_model(name, dataclass=None):
    def register_model_cls(cls):
        if name in MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model ({})".format(name))
        if not issubclass(cls, BaseFairseqModel):
            raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
        if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
            raise ValueError("Dataclass {} must extend FairseqDataclass".format(dataclass))
        MODEL_REGISTRY[name] = cls
        return cls
    return register_model_cls


def register_model_architecture(model_name, arch_name):
    def register_model_arch_fn(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model architecture ({})".format(arch_name))
        if not callable(fn):
            raise ValueError("Model architecture must be callable ({})".format(arch_name))
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_MODEL_NAME_REGISTRY[arch_name] = model_name
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn
    return register_model_arch_fn


def build_model(cfg, task, from_checkpoint=False):
    arch = getattr(cfg, "arch", None)
    if arch not in ARCH_MODEL_REGISTRY:
        raise ValueError("Could not infer model type from " + repr(arch))
    ARCH_CONFIG_REGISTRY[arch](cfg)
    return ARCH_MODEL_REGISTRY[arch].build_model(cfg, task)
