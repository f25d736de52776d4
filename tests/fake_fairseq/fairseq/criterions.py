import inspect

from torch.nn.modules.loss import _Loss

from .dataclass import FairseqDataclass

CRITERION_REGISTRY = {}
CRITERION_DATACLASS_REGISTRY = {}
CRITERION_CLASS_NAMES = set()


class FairseqCriterion(_Loss):
    def __init__(self, task):
        super().__init__()
        self.task = task
        if hasattr(task, "target_dictionary"):
            tgt_dict = task.target_dictionary
            self.padding_idx = tgt_dict.pad() if tgt_dict is not None else -100

    @classmethod
    def build_criterion(cls, cfg, task):
        """Construct a criterion from command-line args: constructor arguments by name from ``cfg``."""
        init_args = {}
        for p in inspect.signature(cls).parameters.values():
            if p.kind in (p.POSITIONAL_ONLY, p.VAR_POSITIONAL, p.VAR_KEYWORD):
                raise NotImplementedError("{} not supported".format(p.kind))
            if p.name == "task":
                init_args["task"] = task
            elif p.name == "cfg":
                init_args["cfg"] = cfg
            elif hasattr(cfg, p.name):
                init_args[p.name] = getattr(cfg, p.name)
            elif p.default != p.empty:
                pass
            else:
                raise NotImplementedError("Unable to infer Criterion arguments, please implement "
                                          "{}.build_criterion".format(cls.__name__))
        return cls(**init_args)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return False


def register_criterion(name, dataclass=None):
    def register_x_cls(cls):
        if name in CRITERION_REGISTRY:
            raise ValueError("Cannot register duplicate criterion ({})".format(name))
        if cls.__name__ in CRITERION_CLASS_NAMES:
            raise ValueError("Cannot register criterion with duplicate class name ({})".format(cls.__name__))
        if not issubclass(cls, FairseqCriterion):
            raise ValueError("{} must extend {}".format(cls.__name__, FairseqCriterion.__name__))
        if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
            raise ValueError("Dataclass {} must extend FairseqDataclass".format(dataclass))
        CRITERION_CLASS_NAMES.add(cls.__name__)
        cls.__dataclass = dataclass
        CRITERION_REGISTRY[name] = cls
        if dataclass is not None:
            CRITERION_DATACLASS_REGISTRY[name] = dataclass
        return cls
    return register_x_cls


def build_criterion(cfg, task):
    name = getattr(cfg, "criterion", None) or getattr(cfg, "_name")
    return CRITERION_REGISTRY[name].build_criterion(cfg, task)
