"""Golden spectrum at BASELINE config 5 size (S 16768 x 12288, dim_reduction.py:190-199 for 64x64x3 images, B = 128).

Run once in the build container (8 cores, ~64 GB): fp64 ``torch.linalg.svdvals`` of the fp32-centred matrix, i.e. the
accuracy yardstick of oracle.dim.spectrum_f64 at the full size.  The matrix is NOT stored (824 MB): it is rebuilt from
the seed by ``cfg5_matrix`` (torch's CPU generator is deterministic for a given build), the fixture keeps the singular
values the -m gpu test compares (top, around the planted cliff, bottom), their sum of squares and ||S_c||_F^2.

    python tests/golden/make_spectrum_cfg5.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
M, D, K_PLANTED, CLIFF, SEED = 16768, 12288, 64, 70.0, 2026


def cfg5_matrix(m=M, d=D, k=K_PLANTED, cliff=CLIFF, seed=SEED):
    """Gaussian [m, d] fp32 whose last k columns are scaled by 1/cliff (a (d-k)-dimensional normal space and a
    k-dimensional tangent space, as the score matrix of a k-manifold has), plus a constant so that centring matters."""
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(m, d, generator=g)
    s[:, d - k:] *= 1.0 / cliff
    s += 0.37
    return s


if __name__ == "__main__":
    torch.set_num_threads(8)
    t0 = time.time()
    s = cfg5_matrix()
    centred = (s - s.mean(dim=0, keepdim=True)).double()     # fp32 centring as dim_reduction.py:193-194, then fp64
    fro2 = float((centred * centred).sum())
    del s
    sv = torch.linalg.svdvals(centred).numpy()
    print("svdvals took", time.time() - t0, "s", sv[:3], sv[D - K_PLANTED - 2:D - K_PLANTED + 2], sv[-3:])
    np.savez_compressed(os.path.join(HERE, "spectrum_cfg5.npz"), sv_f64=sv, fro2=np.array(fro2),
                        params=np.array([M, D, K_PLANTED, CLIFF, SEED], dtype=np.float64))
