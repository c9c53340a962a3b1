"""k-sphere data (drop-in for /root/reference/lightning_data_modules/KSphereDataset.py:7-135).

Uniform points on S^k in R^{k+1} (normalised Gaussians, :87-91), embedded in R^ambient by the Q factor of a
seed-0 Gaussian matrix (:38-44; restated with ``.numpy()`` because the reference's ``np.linalg.qr(Tensor)``
breaks under numpy 2), plus optional isotropic noise.  Host-side generation: it is a few MB, done once.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import utils


class KSphereDataset(Dataset):
    def __init__(self, config):
        super().__init__()
        d = config.data
        self.data = self.generate_data(d.get('data_samples'), d.get('n_spheres'), d.get('ambient_dim'),
                                       d.get('manifold_dim'), d.get('noise_std'), d.get('embedding_type'),
                                       d.get('radii', []), d.get('angle_std', -1))

    @staticmethod
    def isometry(ambient_dim, manifold_dim):
        g = torch.Generator().manual_seed(0)
        a = torch.randn(size=(ambient_dim, manifold_dim + 1), generator=g)
        q, _ = np.linalg.qr(a.numpy())
        return torch.from_numpy(q)

    def generate_data(self, n_samples, n_spheres, ambient_dim, manifold_dim, noise_std, embedding_type, radii,
                      angle_std):
        if radii == []:
            radii = [1] * n_spheres
        dims = [manifold_dim] * n_spheres if isinstance(manifold_dim, int) else list(manifold_dim)
        if angle_std != -1:
            raise NotImplementedError("angle_std sampling is not used by the dimension-estimation configs")
        chunks = []
        for i in range(n_spheres):
            k = dims[i]
            pts = torch.randn((n_samples, k + 1))
            pts = pts / torch.linalg.norm(pts, dim=1)[:, None]
            pts = pts * radii[i]
            if embedding_type == 'random_isometry':
                pts = (self.isometry(ambient_dim, k) @ pts.T).T
            elif embedding_type == 'first':
                pts = torch.cat([pts, torch.zeros([n_samples, ambient_dim - pts.shape[1]])], dim=1)
            else:
                raise NotImplementedError(f"embedding_type {embedding_type!r} is not used by the "
                                          "dimension-estimation configs")
            pts = pts + noise_std * torch.randn_like(pts)
            chunks.append(pts)
        return torch.cat(chunks, dim=0)

    def __getitem__(self, index):
        return self.data[index]

    def __len__(self):
        return len(self.data)


@utils.register_lightning_datamodule(name='KSphere')
class KSphereDataModule(utils.SplitDataModule):
    def make_dataset(self):
        return KSphereDataset(self.config)
