"""Oracle restatement of the k-sphere data on the hot path, plus the exact score of that data.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import numpy as np
import torch
from scipy.special import ive


def isometry(ambient_dim, manifold_dim):
    """Q of the reference's fixed random isometric embedding.

    /root/reference/lightning_data_modules/KSphereDataset.py:38-43: QR of
    randn(ambient, k+1) drawn from ``torch.Generator().manual_seed(0)``.
    """
    g = torch.Generator().manual_seed(0)
    a = torch.randn(size=(ambient_dim, manifold_dim + 1), generator=g)
    q, _ = np.linalg.qr(a.numpy())
    return torch.from_numpy(q)


def ksphere_data(n_samples, ambient_dim, manifold_dim, noise_std=0.0, radius=1.0):
    """Uniform points on S^k embedded in R^ambient by ``isometry``.

    KSphereDataset.py:21-69,87-91 for n_spheres=1, embedding_type='random_isometry',
    angle_std=-1; consumes the global torch RNG exactly like the reference
    (randn(n, k+1), then randn_like for the noise term even when noise_std=0).
    """
    pts = torch.randn((n_samples, manifold_dim + 1))
    pts = pts / torch.linalg.norm(pts, dim=1)[:, None]
    pts = pts * radius
    q = isometry(ambient_dim, manifold_dim)
    data = (q @ pts.T).T
    return data + noise_std * torch.randn_like(data)


class KSphereExact(torch.nn.Module):
    """Exact score of (uniform k-sphere in the span of Q) * N(0, sigma^2 I)  [SURVEY 8-c, builder's oracle].

    With a = Q^T x, r = |a|, x_perp = x - Q a, p = k+1 and A_p(kappa) = I_{p/2}(kappa)/I_{p/2-1}(kappa):
        score(x) = -x_perp/sigma^2 + Q (a/r) (A_p(r/sigma^2) - r)/sigma^2.
    ``forward`` returns ``-sigma * score`` so that ``get_score_fn`` (-out/std) yields the score.
    Not part of the reference (its ``ksphere_gt`` is radial-only and gives ID ~ 96).
    """

    def __init__(self, ambient_dim, manifold_dim, sigma_min, sigma_max, N=1000):
        super().__init__()
        self.register_buffer("Q", isometry(ambient_dim, manifold_dim).double())
        self.k = manifold_dim
        self.sigma_min, self.sigma_max, self.N = sigma_min, sigma_max, N

    def forward(self, x, labels):
        t = labels / (self.N - 1)
        lo = torch.tensor(self.sigma_min).type_as(t)
        hi = torch.tensor(self.sigma_max).type_as(t)
        sigma = (lo * (hi / lo) ** t).double()[:, None]
        xd = x.double()
        a = xd @ self.Q
        r = torch.linalg.norm(a, dim=1, keepdim=True)
        perp = xd - a @ self.Q.T
        p = self.k + 1
        kappa = (r / sigma ** 2).numpy()
        ratio = torch.from_numpy(ive(p / 2, kappa) / ive(p / 2 - 1, kappa))
        score = -perp / sigma ** 2 + (a / r) @ self.Q.T * (ratio - r) / sigma ** 2
        return (-sigma * score).float()
