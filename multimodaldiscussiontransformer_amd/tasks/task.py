"""Registration surface of mDT/src/tasks/task.py: ``TaskConfig`` (the task-level CLI flags, same names and defaults,
:29-113) and the ``Task`` base (:116-228) — a ``FairseqTask`` that resolves ``--dataset-name`` through the dataset
registry (``--user-data-dir`` modules register with ``register_dataset``), wraps the three splits in the batching
datasets of ``data/dataset.py`` and hands FairSeq the ``{"nsamples", "net_input": {"batched_data"}}`` envelope
(:188-194).  Batches are collated by the native packer (``data.packer.pack_batch``), which also emits the CSR index
vectors of the fused path."""
from __future__ import annotations

import importlib
import logging
import os
import sys
from dataclasses import dataclass, field

import numpy as np

from ..registry import DATASET_REGISTRY, FairseqDataclass, FairseqTask

logger = logging.getLogger(__name__)


@dataclass
class TaskConfig(FairseqDataclass):
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


class Task(FairseqTask):
    """Generic task for discussion trees: dataset loading and model construction (task.py:116-228)."""

    def __init__(self, cfg: TaskConfig):
        super().__init__(cfg)
        from ..data.dataset import GraphormerDataset
        self.dm = None
        self.collate_device = "cpu"        # DataLoader workers collate on the host; train.py collates straight to the GPU
        if cfg.user_data_dir != "":
            self._import_user_defined_datasets(cfg.user_data_dir)
            if cfg.dataset_name not in DATASET_REGISTRY:
                raise ValueError(f"dataset {cfg.dataset_name} is not found in customized dataset module {cfg.user_data_dir}")
        if cfg.dataset_name in DATASET_REGISTRY:
            d = DATASET_REGISTRY[cfg.dataset_name]()      # {dataset, train_idx, valid_idx, test_idx, source}
            self.dm = GraphormerDataset(dataset=d["dataset"], dataset_source=d["source"], train_idx=d["train_idx"],
                                        valid_idx=d["valid_idx"], test_idx=d["test_idx"], seed=cfg.seed)

    @staticmethod
    def _import_user_defined_datasets(dataset_dir: str):
        """task.py:146-161: the directory is a package; every module in it is imported for its ``register_dataset``s."""
        dataset_dir = os.path.abspath(dataset_dir.rstrip("/"))
        parent, name = os.path.split(dataset_dir)
        if parent not in sys.path:
            sys.path.insert(0, parent)
        importlib.import_module(name)
        for f in sorted(os.listdir(dataset_dir)):
            path = os.path.join(dataset_dir, f)
            if f.startswith("_") or f.startswith("."):
                continue
            if f.endswith(".py") or os.path.isdir(path):
                importlib.import_module(name + "." + (f[:-3] if f.endswith(".py") else f))

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        assert cfg.num_classes > 0, "Must set task.num_classes"
        return cls(cfg)

    def load_dataset(self, split, **kwargs):
        """Load a dataset split (task.py:168-204): the envelope FairSeq's trainer iterates."""
        from ..data.dataset import EpochShuffleDataset, SampleEnvelopeDataset
        assert split in ("train", "valid", "test"), f"split {split} is not supported. Must be one of train, valid, test"
        if self.dm is None:
            raise ValueError(f"dataset {self.cfg.dataset_name} is not registered (use --user-data-dir / register_dataset)")
        data = {"train": self.dm.dataset_train, "valid": self.dm.dataset_val, "test": self.dm.dataset_test}[split]
        batched = self.get_batched_dataset(data)
        sizes = np.array([self.max_nodes()] * len(batched))
        dataset = SampleEnvelopeDataset(batched, sizes)
        if split == "train" and self.cfg.train_epoch_shuffle:
            dataset = EpochShuffleDataset(dataset, num_samples=len(dataset), seed=self.cfg.seed)
        logger.info("Loaded %s with #samples: %d", split, len(dataset))
        self.datasets[split] = dataset
        return dataset

    def get_batched_dataset(self, dataset):
        raise NotImplementedError

    def build_model(self, args):
        """NodePredictionTask / ContrastiveLearningTask.build_model (:34-55 / :37-48): the task's max_nodes goes into
        the model config, then the registered architecture builds the model."""
        from ..models import GraphormerModel
        for k, v in vars(self.cfg).items():
            if not hasattr(args, k):
                setattr(args, k, v)
        args.max_nodes = self.cfg.max_nodes
        return GraphormerModel.build_model(args, self)

    def max_nodes(self):
        return self.cfg.max_nodes

    @property
    def source_dictionary(self):
        return None

    @property
    def target_dictionary(self):
        return None

    @property
    def label_dictionary(self):
        return None

    def collate(self, trees, device="cuda"):
        from ..data.packer import pack_batch
        return pack_batch(trees, self.cfg.spatial_pos_max, device=device)
