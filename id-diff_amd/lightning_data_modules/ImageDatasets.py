"""Folder-of-image-files data module ``'image'`` (reference: lightning_data_modules/ImageDatasets.py:25-105), without
torchvision or Lightning: PIL decodes, numpy/torch do what ``transforms.ToTensor / Lambda(crop) / Resize / Normalize`` did.

* ``config.data.crop`` (the CelebA recipe, :33-45): 108x108 centre crop of the 218x178 frame, bicubic resize on the PIL
  image to ``config.data.shape[1:]``, then scale to [-1, 1].
* otherwise (:46-49): to [0, 1] tensor, then the tensor resize torchvision applies (bilinear, antialiased).
* ``config.data.dataset == 'mnist'`` needs torchvision's downloader in the reference (:10-23); there is no network
  here, so that branch reads the four raw idx files if they are already under ``base_dir/MNIST/raw`` and raises
  otherwise.  Images are padded 2+2 to 32x32 like the reference.

Host-side data plumbing only: batches are CPU tensors, the driver moves the selected points to the GPU.
"""
import gzip
import os
import struct

import numpy as np
import torch
from torch.utils.data import Dataset

from . import utils


def load_file_paths(dataset_base_dir):
    return [os.path.join(dataset_base_dir, f) for f in os.listdir(dataset_base_dir)]


def _to_tensor(img):
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).float().div_(255.0)


class ImageDataset(Dataset):
    def __init__(self, config):
        from PIL import Image
        self._Image = Image
        path = os.path.join(config.data.base_dir, config.data.dataset)
        self.res = (int(config.data.shape[1]), int(config.data.shape[2]))
        self.crop = bool(config.data.get('crop', False))
        self.image_paths = sorted(load_file_paths(path))

    def __getitem__(self, index):
        Image = self._Image
        image = Image.open(self.image_paths[index]).convert('RGB')
        if self.crop:
            crop_size = 108
            top, left = (218 - crop_size) // 2, (178 - crop_size) // 2
            image = image.crop((left, top, left + crop_size, top + crop_size))
            image = image.resize((self.res[1], self.res[0]), Image.BICUBIC)
            return (_to_tensor(image) - 0.5) / 0.5
        x = _to_tensor(image)
        if tuple(x.shape[1:]) != self.res:
            x = torch.nn.functional.interpolate(x[None], size=self.res, mode='bilinear', antialias=True, align_corners=False)[0]
        return x

    def __len__(self):
        return len(self.image_paths)


def _read_idx(path):
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        magic, = struct.unpack('>I', f.read(4))
        dims = [struct.unpack('>I', f.read(4))[0] for _ in range(magic & 0xFF)]
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(dims)


class MNISTDataset(Dataset):
    def __init__(self, config):
        raw = os.path.join(config.data.base_dir, 'MNIST', 'raw')
        def find(stem):
            for name in (stem, stem + '.gz'):
                if os.path.exists(os.path.join(raw, name)):
                    return os.path.join(raw, name)
            raise FileNotFoundError(f"{stem}[.gz] not found under {raw}: the reference downloads MNIST through torchvision; "
                                    "this build has no network -- place the raw idx files there")
        self.images = _read_idx(find('train-images-idx3-ubyte'))
        self.labels = _read_idx(find('train-labels-idx1-ubyte'))
        self.return_labels = bool(config.data.get('return_labels', False))

    def __getitem__(self, index):
        x = torch.from_numpy(self.images[index].copy()).float().div_(255.0)[None]
        x = torch.nn.functional.pad(x, (2, 2, 2, 2))
        return (x, int(self.labels[index])) if self.return_labels else x

    def __len__(self):
        return len(self.images)


@utils.register_lightning_datamodule(name='image')
class ImageDataModule(utils.SplitDataModule):
    def __init__(self, config):
        super().__init__(config)
        self.val_batch = config.get('eval.batch_size', self.val_batch)

    def make_dataset(self):
        return MNISTDataset(self.config) if self.config.data.dataset == 'mnist' else ImageDataset(self.config)

    def setup(self, stage=None):
        from torch.utils.data import random_split
        self.dataset = data = self.make_dataset()
        n = len(data)
        a, b = int(self.split[0] * n), int(self.split[1] * n)          # the reference gives the remainder to test (:88)
        self.train_data, self.valid_data, self.test_data = random_split(data, [a, b, n - a - b])
