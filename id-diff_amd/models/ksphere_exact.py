"""Analytic score of the noised k-sphere (acceptance model for "correct ID on the 10-sphere in R^100").

Not part of the reference (its ``ksphere_gt`` is radial-only, models/ksphere_gt.py:14-27, and yields ID ~ 96;
SURVEY.md 8-c).  With a = Q^T x, r = |a|, x_perp = x - Q a, p = k+1:

    score(x) = -x_perp / sigma^2 + Q (a / r) (A_p(r / sigma^2) - r) / sigma^2,   A_p = I_{p/2} / I_{p/2-1}.

At the sigma the ID estimator uses (1e-2) kappa = r/sigma^2 ~ 1e4, where the uniform asymptotic expansion
A_p(kappa) = 1 - (p-1)/(2 kappa) + (p-1)(p-3)/(8 kappa^2) + (p-1)(p-3)/(8 kappa^3) + O(kappa^-4)  is exact to
fp64 rounding for the p <= 101 used here; the model refuses smaller kappa instead of guessing.  The two
projections are GEMMs on the MFMA path; the radial factor is a [B]-sized torch expression (plumbing).
"""
import numpy as np
import torch

from .. import _lib
from . import utils
from .base import HipScoreModel


def isometry(ambient_dim, manifold_dim):
    """Q of lightning_data_modules/KSphereDataset.py:38-43 (QR of a seed-0 Gaussian matrix)."""
    g = torch.Generator().manual_seed(0)
    a = torch.randn(size=(ambient_dim, manifold_dim + 1), generator=g)
    q, _ = np.linalg.qr(a.numpy())
    return torch.from_numpy(np.ascontiguousarray(q))


@utils.register_model(name='ksphere_exact')
class KSphereExact(HipScoreModel):
    def __init__(self, config):
        super().__init__()
        d = config.data
        self.k, self.n = d.manifold_dim, d.ambient_dim
        self.sigma_min, self.sigma_max, self.N = config.model.sigma_min, config.model.sigma_max, config.model.num_scales
        self.Q = torch.nn.Parameter(isometry(self.n, self.k), requires_grad=False)  # [n, k+1]

    def _pack(self):
        q = self.Q.detach().float()
        kp = (self.k + 1 + 3) // 4 * 4
        qpad = torch.zeros(self.n, kp, device=q.device)
        qpad[:, :self.k + 1] = q
        return {"Q": qpad.contiguous(), "Qt": qpad.t().contiguous(), "kp": kp}

    def forward(self, x, labels, out_rowscale=None):
        x, labels = self._check_inputs(x, labels)
        pk = self.packed()
        t = labels / (self.N - 1)
        lo = torch.tensor(self.sigma_min).type_as(t)
        hi = torch.tensor(self.sigma_max).type_as(t)
        sigma = lo * (hi / lo) ** t                                  # [B]
        a = _lib.gemm(x, pk["Qt"])                                   # a = x Q        [B, kp]
        r = torch.linalg.vector_norm(a.double(), dim=1)              # [B] (plumbing-sized)
        kappa = r / sigma.double() ** 2
        if float(kappa.min()) < 5e3:
            raise NotImplementedError("ksphere_exact: asymptotic Bessel ratio needs r/sigma^2 >= 5e3")
        p = self.k + 1
        ratio = 1 - (p - 1) / (2 * kappa) + (p - 1) * (p - 3) / (8 * kappa ** 2) + (p - 1) * (p - 3) / (8 * kappa ** 3)
        # score * sigma^2 = -(x - Q a) + Q a * (ratio - r)/r = -x + Q a * (1 + (ratio - r)/r) = -x + (Q a) * ratio / r
        coef = (ratio / r).float().contiguous()                      # [B]
        # model output convention: score = -out/std  ->  out = -sigma * score = (x - (Q a) coef) / sigma
        inv_sigma = (1.0 / sigma).contiguous()
        scale = inv_sigma if out_rowscale is None else (inv_sigma * out_rowscale).contiguous()
        proj = _lib.gemm(a, pk["Q"], epilogue=_lib.make_epilogue(rowscale=(-coef).contiguous()))   # -(Q a) coef
        out = torch.empty_like(x)
        _lib.add_scale(x, proj, out, x.numel(), 1.0)
        res = torch.empty_like(x)
        _lib.affine_act(out, res, out.numel(), 1.0, 0.0, None, scale, x.shape[1])
        return res
