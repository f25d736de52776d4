"""Dataset wrappers of mDT/src/data/dataset.py and mDT/src/data/pyg_datasets/pyg_dataset.py, without torch_geometric:

  DiscussionDataset        list-like container of discussion TREES (dicts of numpy arrays, see synthetic.py / discussions.py)
                           — stands where the reference holds a PyG ``Dataset`` of per-graph ``Data`` objects
  GraphormerPYGDataset     split handling of pyg_dataset.py:13-95 (given index lists are shuffled with RandomState(seed),
                           absent ones are drawn with sklearn's train_test_split exactly as the reference does)
  GraphormerDataset        dataset.py:34-70 (the three split pointers the task reads)
  BatchedDataDataset / NodeBatchedDataDataset / ContrastiveBatchedDataDataset
                           dataset.py:73-214: ``collater(samples)`` → the reference's batch dict (+ ``y_mask`` / ``hard_y``)
  SampleEnvelopeDataset    what tasks/task.py:188-194 builds from NestedDictionaryDataset + NumSamplesDataset
  EpochShuffleDataset      dataset.py:217-233

What changed: the reference runs ``preprocess_item`` (pure-Python O(N^2) loops) per item in ``__getitem__`` behind a
32-entry lru_cache and pads per-tree tensors in ``collator``; here items stay raw parent arrays and ``collater`` calls the
native packer once per batch (``mdt_pack_structure``, C++), which emits the same integer tensors bit for bit plus the
CSR index vectors of the fused path (``batched_data["_csr"]``: plain tensors, so FairSeq's ``move_to_cuda`` carries them
to the device with the rest of the sample).
"""
from __future__ import annotations

import copy
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..registry import HAVE_FAIRSEQ
from .packer import pack_batch

if HAVE_FAIRSEQ:
    from fairseq.data import BaseWrapperDataset, FairseqDataset
else:
    class FairseqDataset(torch.utils.data.Dataset):
        def set_epoch(self, epoch):
            pass

        @property
        def supports_prefetch(self):
            return False

        @property
        def can_reuse_epoch_itr_across_epochs(self):
            return True

    class BaseWrapperDataset(FairseqDataset):
        def __init__(self, dataset):
            super().__init__()
            self.dataset = dataset

        def __getitem__(self, index):
            return self.dataset[index]

        def __len__(self):
            return len(self.dataset)

        def collater(self, samples):
            return self.dataset.collater(samples)

        @property
        def sizes(self):
            return self.dataset.sizes

        def num_tokens(self, index):
            return self.dataset.num_tokens(index)

        def size(self, index):
            return self.dataset.size(index)

        def ordered_indices(self):
            return self.dataset.ordered_indices()

        def set_epoch(self, epoch):
            if hasattr(self.dataset, "set_epoch"):
                self.dataset.set_epoch(epoch)


class DiscussionDataset:
    """Random-access container of discussion trees.  ``trees`` is a sequence of tree dicts or a callable
    ``get(i) -> tree`` with ``length`` (lazy loading: data/discussions.py reads one JSON line per tree)."""

    def __init__(self, trees, length: Optional[int] = None, indices: Optional[Sequence[int]] = None):
        self._get = trees if callable(trees) else trees.__getitem__
        n = length if length is not None else len(trees)
        self._indices = np.arange(n) if indices is None else np.asarray(indices, dtype=np.int64)

    def __len__(self):
        return int(self._indices.shape[0])

    def __getitem__(self, i: int):
        return self._get(int(self._indices[int(i)]))

    def index_select(self, idx):
        idx = idx.numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
        out = copy.copy(self)
        out._indices = self._indices[idx.astype(np.int64)]
        return out


class GraphormerPYGDataset:
    def __init__(self, dataset, seed: int = 0, train_idx=None, valid_idx=None, test_idx=None, train_set=None,
                 valid_set=None, test_set=None):
        self.dataset = dataset
        self.num_data = len(dataset) if dataset is not None else 0
        self.seed = seed
        self.train_data = self.valid_data = self.test_data = None
        self.train_idx = self.valid_idx = self.test_idx = None
        if train_idx is None and train_set is None:
            from sklearn.model_selection import train_test_split
            train_idx, test_valid_idx = train_test_split(np.arange(self.num_data), test_size=self.num_data // 5,
                                                         random_state=seed)
            test_idx, valid_idx = train_test_split(test_valid_idx, test_size=self.num_data // 10, random_state=seed)
            self.train_idx, self.valid_idx, self.test_idx = (torch.from_numpy(a) for a in (train_idx, valid_idx, test_idx))
        elif train_set is not None:
            self.num_data = len(train_set) + len(valid_set) + len(test_set)
            self.train_data, self.valid_data, self.test_data = (self._subset(s) for s in (train_set, valid_set, test_set))
            return
        else:
            self.num_data = len(train_idx) + len(valid_idx) + len(test_idx)
            rng = np.random.RandomState(seed)
            train_idx, valid_idx, test_idx = (np.array(a, copy=True) for a in (train_idx, valid_idx, test_idx))
            rng.shuffle(train_idx)
            rng.shuffle(valid_idx)
            rng.shuffle(test_idx)
            self.train_idx, self.valid_idx, self.test_idx = train_idx, valid_idx, test_idx
        self.train_data = self._subset(self.dataset.index_select(self.train_idx))
        self.valid_data = self._subset(self.dataset.index_select(self.valid_idx))
        self.test_data = self._subset(self.dataset.index_select(self.test_idx))

    def _subset(self, subset):
        out = copy.copy(self)
        out.dataset = subset
        out.num_data = len(subset)
        out.train_data = out.valid_data = out.test_data = None
        out.train_idx = out.valid_idx = out.test_idx = None
        return out

    def __getitem__(self, idx):
        if not isinstance(idx, (int, np.integer)):
            raise TypeError("index to a GraphormerPYGDataset can only be an integer.")
        item = dict(self.dataset[int(idx)])
        item["idx"] = int(idx)
        item["y"] = np.asarray(item["y"]).reshape(-1)
        return item

    get = __getitem__

    def __len__(self):
        return self.num_data

    len = __len__


class GraphormerDataset:
    def __init__(self, dataset=None, dataset_source: Optional[str] = None, seed: int = 0, train_idx=None, valid_idx=None,
                 test_idx=None, dataset_spec=None):
        if dataset is None:
            raise ValueError(f"built-in dataset specs ({dataset_spec!r}) do not exist for mDT: register one with register_dataset")
        if dataset_source != "pyg":
            raise ValueError("Customized dataset can only have source pyg")
        if not hasattr(dataset, "index_select"):
            dataset = DiscussionDataset(dataset)
        self.dataset = GraphormerPYGDataset(dataset, seed=seed, train_idx=train_idx, valid_idx=valid_idx, test_idx=test_idx)
        self.train_idx, self.valid_idx, self.test_idx = self.dataset.train_idx, self.dataset.valid_idx, self.dataset.test_idx
        self.dataset_train, self.dataset_val, self.dataset_test = (self.dataset.train_data, self.dataset.valid_data,
                                                                   self.dataset.test_data)


class BatchedDataDataset(FairseqDataset):
    """Batches trees with the native packer.  ``device``: where ``collater`` leaves the batch — "cpu" inside DataLoader
    workers (FairSeq moves the sample to the GPU), "cuda" when the trainer collates on the main process."""

    def __init__(self, dataset, spatial_pos_max: int = 1024, device: str = "cpu", sample_filter=None):
        super().__init__()
        self.dataset = dataset
        self.spatial_pos_max = spatial_pos_max
        self.device = device
        self.sample_filter = sample_filter       # e.g. Task.filter_oversized

    def __getitem__(self, index):
        return self.dataset[int(index)]

    def __len__(self):
        return len(self.dataset)

    def _pack(self, samples: List[dict]):
        samples = [s for s in samples if s is not None]
        if self.sample_filter is not None:
            samples = self.sample_filter(samples)
        return samples, pack_batch(samples, self.spatial_pos_max, device=self.device).batched_data

    def collater(self, samples):
        raise NotImplementedError


class NodeBatchedDataDataset(BatchedDataDataset):
    def collater(self, samples):
        """dataset.py:183-214: the collated dict + ``y_mask`` (bool[M], which comments carry a label)."""
        _, bd = self._pack(samples)
        return bd


class ContrastiveBatchedDataDataset(BatchedDataDataset):
    def collater(self, samples):
        """dataset.py:148-179: ``y`` = one community label per tree, ``hard_y`` = its polar-opposite community."""
        samples, bd = self._pack(samples)
        bd["hard_y"] = torch.cat([torch.as_tensor(np.asarray(s["hard_y"]).reshape(-1)) for s in samples]).to(bd["y"].device)
        return bd


class SampleEnvelopeDataset(FairseqDataset):
    """{"nsamples": B, "net_input": {"batched_data": ...}} (tasks/task.py:188-194)."""

    def __init__(self, batched: BatchedDataDataset, sizes):
        super().__init__()
        self.batched = batched
        self.sizes = np.asarray(sizes)

    def __getitem__(self, index):
        return self.batched[index]

    def __len__(self):
        return len(self.batched)

    def collater(self, samples):
        if len(samples) == 0:
            return {}
        return {"nsamples": len(samples), "net_input": {"batched_data": self.batched.collater(samples)}}

    def num_tokens(self, index):
        return int(self.sizes[index])

    def size(self, index):
        return int(self.sizes[index])

    def ordered_indices(self):
        return np.arange(len(self))


class EpochShuffleDataset(BaseWrapperDataset):
    def __init__(self, dataset, num_samples, seed):
        super().__init__(dataset)
        self.num_samples = num_samples
        self.seed = seed
        self.set_epoch(1)

    def set_epoch(self, epoch):
        # fairseq.data.data_utils.numpy_seed(seed + epoch - 1): a scoped numpy seed
        state = np.random.get_state()
        np.random.seed(self.seed + epoch - 1)
        try:
            self.sort_order = np.random.permutation(self.num_samples)
        finally:
            np.random.set_state(state)

    def ordered_indices(self):
        return self.sort_order

    @property
    def can_reuse_epoch_itr_across_epochs(self):
        return False
