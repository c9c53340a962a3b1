"""Golden point at BASELINE config 3, FULL size: the CPU oracle's score matrix S [4480, 3072] of ONE data point of the
bench's workload (nf = 128 NCSN++, init_scale = 1, torch.manual_seed(0) weights, bench image 1, the bench's point seed),
on the draws of the device path's Philox stream restated on the CPU (oracle/philox.py).

Run once in the build container (8 cores; ~10 min of oracle forwards + the SVDs):

    python tests/golden/make_cfg3_point.py [point]      (default 1 -> cfg3_point.npz; 3 -> cfg3_point3.npz)

What the fixture keeps (S itself is 55 MB and is not stored):
* ``rows`` / ``S_rows``: 192 rows of the oracle's S, spread over both 2240-row launch sets of the HIP driver and
  straddling their boundary (rows the device computes in different launches);
* ``sv_f32``: ``torch.linalg.svd`` of the fp32-centred oracle S exactly as dim_reduction.py:193-198 (LAPACK gesdd);
  ``sv_f64``: float64 singular values of the same centred matrix (the accuracy yardstick);
* ``id_f32`` / ``id_f64``: the ID rule (plot_utils.py:173-183) on each; ``gaps``: the five largest gaps s[i]-s[i+1], i >= 1,
  of sv_f64 with their indices, so that a tie of the rule's argmax is visible;
* ``colmean``: the oracle S's column means (fp64) and ``fro2``: ||S_c||_F^2 -- size-independent checks of the whole matrix;
* ``weight_abs_sums``: abs-sum of every parameter (the test rebuilds the weights from the seed and checks them against this);
* ``x0``: the data point.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import id_diff_amd  # noqa: E402,F401  (package alias)
from helpers import weight_abs_sums  # noqa: E402
from id_diff_amd.configs.utils import read_config  # noqa: E402
from id_diff_amd.lightning_data_modules.SyntheticImages import smooth_decoder_images  # noqa: E402
from oracle import dim as odim, models as omodels, philox, sde as osde  # noqa: E402

MODEL_SEED, IMAGE_SEED, N_IMAGES, IMAGE_INDEX, POINT = 0, 100, 9, 1, 1   # bench.py defaults: Workload on rank 0, 8 steps + 1 warm-up, first timed point
POINT_SEED = 1234 + 1000003 * (POINT + 1)


def point_seed(point):
    """bench.py: Workload.point(i) on rank 0."""
    return 1234 + 1000003 * (point + 1)


def fixture_name(point):
    return "cfg3_point.npz" if point == POINT else f"cfg3_point{point}.npz"
ROWS = np.r_[0:32, 1100:1132, 2208:2272, 3400:3432, 4448:4480]      # 192 rows; 2208..2271 straddle the launch-set boundary at 2240
T = 1e-5                                                             # sampling_eps of the VE SDE


def cfg3():
    cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
    cfg.model.init_scale = 1.0
    return cfg


def oracle_model(cfg):
    torch.manual_seed(MODEL_SEED)
    return omodels.create_model(cfg).eval()


def data_point(point=POINT):
    return smooth_decoder_images(N_IMAGES, [3, 32, 32], 64, seed=IMAGE_SEED)[point]


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
    point = int(sys.argv[1]) if len(sys.argv) > 1 else POINT       # the bench's points 1 .. 8: different images, different IDs
    POINT_SEED = point_seed(point)
    cfg = cfg3()
    model = oracle_model(cfg)
    sde = osde.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    score_fn = osde.get_score_fn(sde, model)
    x0 = data_point(point)
    B = int(cfg.training.batch_size)
    nb, extra, rows = odim.batching(tuple(x0.shape), B)
    assert rows == 4480 and extra == 0
    D = x0.numel()
    S = torch.empty(rows, D)
    t0 = time.time()
    with torch.no_grad():
        for lo in range(0, rows, B):                                 # the reference's batch shape (dim_reduction.py:167-183)
            z = torch.from_numpy(philox.normal_rows(POINT_SEED, D, lo, B)).view(B, *x0.shape)
            vec_t = torch.ones(B) * T
            mean, std = sde.marginal_prob(x0.unsqueeze(0).repeat(B, 1, 1, 1), vec_t)
            S[lo:lo + B] = score_fn(mean + std[:, None, None, None] * z, vec_t).reshape(B, D)
            if (lo // B) % 5 == 0:
                print(f"rows {lo + B}/{rows}  {time.time() - t0:.0f}s", flush=True)
    centred = S - S.mean(dim=0, keepdim=True)                        # fp32, dim_reduction.py:193-194
    sv32 = torch.linalg.svd(centred)[1].numpy()
    sv64 = torch.linalg.svdvals(centred.double()).numpy()
    gaps = sv64[1:-1] - sv64[2:]
    order = np.argsort(-gaps)[:5]
    print("top gaps (index i of s[i]-s[i+1], value):", [(int(i) + 1, float(gaps[i])) for i in order])
    print("ID fp32", odim.estimate_dim(sv32.tolist()), "ID fp64", odim.estimate_dim(sv64.tolist()), "sv[:4]", sv64[:4], "sv[-3:]", sv64[-3:])
    np.savez_compressed(
        os.path.join(HERE, fixture_name(point)), x0=x0.numpy(), rows=ROWS, S_rows=S[ROWS].numpy(), sv_f32=sv32, sv_f64=sv64,
        id_f32=np.array(odim.estimate_dim(sv32.tolist())), id_f64=np.array(odim.estimate_dim(sv64.tolist())),
        gap_index=(order + 1).astype(np.int64), gap_value=gaps[order], colmean=S.double().mean(dim=0).numpy(),
        fro2=np.array(float((centred.double() ** 2).sum())), weight_abs_sums=weight_abs_sums(model),
        point_seed=np.array(POINT_SEED), point=np.array(point), torch_version=np.array(torch.__version__))
    torch.save(S, os.path.join(ROOT, "gpurun_out", f"cfg3_oracle_S_{point}.pt"))   # scratch copy (not tracked) for offline experiments
    print("wrote", fixture_name(point), "in", time.time() - t0, "s")
