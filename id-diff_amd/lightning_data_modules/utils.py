"""Data-module registry (drop-in for /root/reference/lightning_data_modules/utils.py:4-30), without Lightning."""
from ..registry import Registry

_DATA_MODULES = Registry("data module")
register_lightning_datamodule = _DATA_MODULES.register
get_lightning_datamodule_by_name = _DATA_MODULES.get


def create_lightning_datamodule(config):
    return get_lightning_datamodule_by_name(config.data.datamodule)(config)


class SplitDataModule:
    """80/10/10-style split + shuffled loaders, as every data module of the reference does
    (e.g. KSphereDataset.py:120-135).  Loaders are plain generators of CPU tensors (host-side plumbing)."""

    def __init__(self, config):
        self.config = config
        self.split = config.data.get('split', [0.8, 0.1, 0.1])
        self.train_batch = config.training.batch_size
        self.val_batch = config.get('validation.batch_size', self.train_batch)

    def make_dataset(self):
        raise NotImplementedError

    def setup(self, stage=None):
        from torch.utils.data import random_split
        self.dataset = self.make_dataset()
        n = len(self.dataset)
        sizes = [int(self.split[0] * n), int(self.split[1] * n), int(self.split[2] * n)]
        rest = n - sum(sizes)  # the reference passes the three ints as they are; leftovers stay unused here
        parts = random_split(self.dataset, sizes + ([rest] if rest else []))
        self.train_data, self.valid_data, self.test_data = parts[0], parts[1], parts[2]

    def _loader(self, data, batch, shuffle=True):
        from torch.utils.data import DataLoader
        return DataLoader(data, batch_size=batch, num_workers=0, shuffle=shuffle)

    def train_dataloader(self):
        return self._loader(self.train_data, self.train_batch)

    def val_dataloader(self):
        return self._loader(self.valid_data, self.val_batch)

    def test_dataloader(self):
        return self._loader(self.test_data, self.val_batch, shuffle=False)
