from .dataclass import FairseqDataclass

TASK_REGISTRY = {}
TASK_DATACLASS_REGISTRY = {}
TASK_CLASS_NAMES = set()


class FairseqTask:
    def __init__(self, cfg, **kwargs):
        self.cfg = cfg
        self.datasets = dict()
        self.dataset_to_epoch_iter = dict()

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        return cls(cfg, **kwargs)

    def dataset(self, split):
        if split not in self.datasets:
            raise KeyError("Dataset not loaded: " + split)
        return self.datasets[split]

    def build_model(self, cfg, from_checkpoint=False):
        from . import models
        return models.build_model(cfg, self, from_checkpoint)

    def build_criterion(self, cfg):
        from . import criterions
        return criterions.build_criterion(cfg, self)


def register_task(name, dataclass=None):
    def register_task_cls(cls):
        if name in TASK_REGISTRY:
            raise ValueError("Cannot register duplicate task ({})".format(name))
        if not issubclass(cls, FairseqTask):
            raise ValueError("Task ({}: {}) must extend FairseqTask".format(name, cls.__name__))
        if cls.__name__ in TASK_CLASS_NAMES:
            raise ValueError("Cannot register task with duplicate class name ({})".format(cls.__name__))
        if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
            raise ValueError("Dataclass {} must extend FairseqDataclass".format(dataclass))
        TASK_REGISTRY[name] = cls
        TASK_CLASS_NAMES.add(cls.__name__)
        cls.__dataclass = dataclass
        if dataclass is not None:
            TASK_DATACLASS_REGISTRY[name] = dataclass
        return cls
    return register_task_cls


def setup_task(cfg, **kwargs):
    return TASK_REGISTRY[getattr(cfg, "task", None) or getattr(cfg, "_name")].setup_task(cfg, **kwargs)
