"""Registration surface of mDT/src/tasks/task.py: ``TaskConfig`` (the task-level CLI flags,
same names and defaults, :29-113) and a ``Task`` base that resolves ``--dataset-name`` through
the dataset registry (:116-204) and batches with the native packer.  FairSeq's dataset
wrappers / iterators are out of scope; ``train.py`` iterates directly."""
from __future__ import annotations

import importlib
import os
import sys
from dataclasses import dataclass, field

from ..registry import DATASET_REGISTRY


@dataclass
class TaskConfig:
    dataset_name: str = field(default="hateful_discussions", metadata={"help": "name of the dataset"})
    num_classes: int = field(default=1, metadata={"help": "number of classes or regression targets"})
    max_nodes: int = field(default=128, metadata={"help": "max nodes per graph"})
    dataset_source: str = field(default="pyg", metadata={"help": "source of graph dataset"})
    num_atoms: int = field(default=512 * 9, metadata={"help": "number of atom types in the graph"})
    num_edges: int = field(default=512 * 3, metadata={"help": "number of edge types in the graph"})
    num_in_degree: int = field(default=512, metadata={"help": "number of in degree types in the graph"})
    num_out_degree: int = field(default=512, metadata={"help": "number of out degree types in the graph"})
    num_spatial: int = field(default=512, metadata={"help": "number of spatial types in the graph"})
    num_edge_dis: int = field(default=128, metadata={"help": "number of edge dis types in the graph"})
    multi_hop_max_dist: int = field(default=5, metadata={"help": "max distance of multi-hop edges"})
    spatial_pos_max: int = field(default=1024, metadata={"help": "max distance of multi-hop edges"})
    edge_type: str = field(default="multi_hop", metadata={"help": "edge type in the graph"})
    seed: int = field(default=1, metadata={"help": "common.seed"})
    pretrained_model_name: str = field(default="none", metadata={"help": "name of used pretrained model"})
    load_pretrained_model_output_layer: bool = field(default=False, metadata={"help": "whether to load the output layer of pretrained model"})
    train_epoch_shuffle: bool = field(default=False, metadata={"help": "whether to shuffle the dataset at each epoch"})
    user_data_dir: str = field(default="", metadata={"help": "path to the module of user-defined dataset"})


class Task:
    def __init__(self, cfg: TaskConfig):
        self.cfg = cfg
        if cfg.user_data_dir:
            self._import_user_datasets(cfg.user_data_dir)
        self.dm = None
        if cfg.dataset_name in DATASET_REGISTRY:
            self.dm = DATASET_REGISTRY[cfg.dataset_name]()     # {dataset, train_idx, valid_idx, test_idx, source}

    @staticmethod
    def _import_user_datasets(path):
        path = os.path.abspath(path)
        parent, name = os.path.split(path)
        if parent not in sys.path:
            sys.path.insert(0, parent)
        for f in sorted(os.listdir(path)):
            if f.endswith(".py") and not f.startswith("_"):
                importlib.import_module(f"{name}.{f[:-3]}")

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        assert cfg.num_classes > 0, "Must set task.num_classes"
        return cls(cfg)

    def build_model(self, args):
        from ..models import GraphormerModel
        for k, v in vars(self.cfg).items():
            if not hasattr(args, k):
                setattr(args, k, v)
        args.max_nodes = self.cfg.max_nodes
        return GraphormerModel.build_model(args, self)

    def collate(self, trees, device="cuda"):
        from ..data.packer import pack_batch
        return pack_batch(trees, self.cfg.spatial_pos_max, device=device)
