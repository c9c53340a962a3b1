"""``Gan`` data module (drop-in for /root/reference/lightning_data_modules/GanDataset.py:9-68).

Same config keys, same files, same split arithmetic as the reference:
* ``data.data_path`` + ``data.latent_dim`` + ``data.style_gan`` (GanDataset.py:14-22; the authors' config sets them at
  configs/dimension_estimation/extra_experiments/styleGAN/style_gan_base.py:81-84):
  ``style_gan`` true  -> ``<data_path>/style_gan_horvat/gan_<latent_dim>d_train.npy`` (``np.load`` -> float32 tensor, as stored),
  ``style_gan`` false -> ``<data_path>/latent_dim_<latent_dim>/data.pt`` (``torch.load``);
* a file that is not there raises ``FileNotFoundError`` -- never a silent stand-in;
* split: ``train = int(split[0] * l)``, ``val = int(split[1] * l)``, ``test = the rest`` (GanDataset.py:50-54).

The reference repository does not ship the StyleGAN ``.npy``.  This repo's own benchmark config asks for generated images with
the EXPLICIT key ``data.synthetic = True`` (not a reference key; an authors' config never carries it, so an authors' config
never gets made-up data).
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset, random_split

from . import utils
from .SyntheticImages import SyntheticImageDataset


class GanDataset(Dataset):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.data_path = config.data.data_path
        self.latent_dim = config.data.latent_dim
        if self.data_path is None:
            raise FileNotFoundError("GanDataset: config.data.data_path is not set (the directory that holds style_gan_horvat/ or "
                                    "latent_dim_<d>/, GanDataset.py:14-22)")
        if not hasattr(config.data, 'style_gan'):
            # the reference leaves ``self.data`` unset here and fails later with an AttributeError in __len__ (GanDataset.py:17-22)
            raise AttributeError("GanDataset: config.data.style_gan is not set (True: style_gan_horvat/gan_<d>d_train.npy, "
                                 "False: latent_dim_<d>/data.pt)")
        if config.data.style_gan:
            path = os.path.join(self.data_path, f'style_gan_horvat/gan_{self.latent_dim}d_train.npy')
            self._require(path)
            self.data = torch.from_numpy(np.load(path)).float()
        else:
            path = os.path.join(self.data_path, f'latent_dim_{self.latent_dim}/data.pt')
            self._require(path)
            self.data = torch.load(path, map_location='cpu')

    @staticmethod
    def _require(path):
        if not os.path.isfile(path):
            raise FileNotFoundError(f"GanDataset: {path} does not exist (config.data.data_path / latent_dim / style_gan name this "
                                    "file, GanDataset.py:17-22); set data.synthetic = True only if generated images are what you want")

    def __getitem__(self, index):
        return self.data[index]

    def __len__(self):
        return len(self.data)


@utils.register_lightning_datamodule(name='Gan')
class SyntheticDataModule(utils.SplitDataModule):
    """The reference's class name for this module (GanDataset.py:31-32)."""

    def make_dataset(self):
        if bool(self.config.data.get('synthetic', False)):
            return SyntheticImageDataset(self.config)
        return GanDataset(self.config)

    def setup(self, stage=None):
        self.dataset = self.make_dataset()
        n = len(self.dataset)
        train_len = int(self.split[0] * n)
        val_len = int(self.split[1] * n)
        self.train_data, self.valid_data, self.test_data = random_split(self.dataset, [train_len, val_len, n - train_len - val_len])
