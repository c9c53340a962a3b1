"""Synthetic image data for the image-shaped workloads (no dataset ships with the reference and there is no
network): the reference's ``image`` / ``Gan`` data modules (ImageDatasets.py:10-61, GanDataset.py:9-29) read
MNIST folders / the authors' StyleGAN ``.npy``; here images are a fixed random smooth decoder applied to
latents z in R^latent_dim, float32 in [0, 1], shape ``config.data.shape`` -- an image manifold of known
intrinsic dimension <= latent_dim.  (The ``Gan`` data module of the reference lives in GanDataset.py.)
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import utils


def smooth_decoder_images(n, shape, latent_dim, seed):
    c, h, w = shape
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(n, latent_dim, generator=g)
    # low-frequency cosine basis with random phases/weights -> smooth images
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    freq = torch.randint(0, 4, (latent_dim, c, 2), generator=g).float()
    phase = torch.rand(latent_dim, c, generator=g) * 2 * np.pi
    basis = torch.cos(2 * np.pi * (freq[..., 0, None, None] * yy + freq[..., 1, None, None] * xx)
                      + phase[..., None, None])                      # [latent, c, h, w]
    imgs = torch.einsum("nl,lchw->nchw", z, basis) / np.sqrt(latent_dim)
    return torch.sigmoid(imgs).float().contiguous()


class SyntheticImageDataset(Dataset):
    """With ``data.return_labels`` items are (image, label) pairs, label = index % 2, which is what the
    conditional estimator iterates over (dim_reduction.py:50-57 keeps label == 1)."""

    def __init__(self, config):
        d = config.data
        self.data = smooth_decoder_images(d.get('data_samples', 256), list(d.shape), d.get('latent_dim', 64),
                                          d.get('data_seed', 0))
        self.return_labels = bool(d.get('return_labels', False))

    def __getitem__(self, index):
        if self.return_labels:
            return self.data[index], index % 2
        return self.data[index]

    def __len__(self):
        return len(self.data)


@utils.register_lightning_datamodule(name='image_synthetic')
class SyntheticImageDataModule(utils.SplitDataModule):
    def make_dataset(self):
        return SyntheticImageDataset(self.config)
