import numpy as np
import torch.utils.data


class FairseqDataset(torch.utils.data.Dataset):
    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def collater(self, samples):
        raise NotImplementedError

    def num_tokens(self, index):
        raise NotImplementedError

    def size(self, index):
        raise NotImplementedError

    def ordered_indices(self):
        return np.arange(len(self), dtype=np.int64)

    @property
    def supports_prefetch(self):
        return False

    @property
    def can_reuse_epoch_itr_across_epochs(self):
        return True

    def set_epoch(self, epoch):
        pass


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
        super().set_epoch(epoch)
        if hasattr(self.dataset, "set_epoch"):
            self.dataset.set_epoch(epoch)
