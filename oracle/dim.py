"""Oracle restatement of the ID driver: score matrix -> centred SVD -> spectrum -> integer ID.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch


def batching(sample_shape, batchsize):
    """(num_batches, rows_kept_from_last_batch, rows_total) of dim_reduction.py:166-171.

    ``ambient_dim`` is prod(x.shape[1:]) of ONE un-batched sample, i.e. the
    leading axis is dropped: [100] -> 1, [3,32,32] -> 1024.
    """
    ambient = math.prod(sample_shape[1:])
    num_batches = (ambient // batchsize + 1) * 4
    extra = ambient - (ambient // batchsize) * batchsize
    return num_batches, extra, (num_batches - 1) * batchsize + extra


def score_matrix(score_fn, sde, x, batchsize, t, noise=None, generator=None):
    """Rows of S for one data point, dim_reduction.py:166-191.

    ``noise`` ([num_batches, B, *x.shape]) replaces the reference's unseeded
    ``randn_like`` so that two implementations can be fed the same draws.
    """
    num_batches, extra, _ = batching(x.shape, batchsize)
    rep = x.unsqueeze(0).repeat([batchsize] + [1] * x.ndim)
    vec_t = torch.ones(batchsize) * t
    rows = []
    for i in range(1, num_batches + 1):
        mean, std = sde.marginal_prob(rep.clone(), vec_t)
        z = noise[i - 1] if noise is not None else torch.randn(rep.shape, generator=generator)
        batch = mean + std.reshape((-1,) + (1,) * x.ndim) * z
        with torch.no_grad():
            score = score_fn(batch, vec_t)
        rows.append(score if i < num_batches else score[:extra])
    return torch.flatten(torch.cat(rows, dim=0), start_dim=1)


def spectrum(scores):
    """Centre the columns, full SVD, singular values descending: dim_reduction.py:193-198."""
    centred = scores - scores.mean(dim=0, keepdim=True)
    _, s, _ = torch.linalg.svd(centred)
    return s


def spectrum_f64(scores):
    """Float64 singular values of the fp32-centred matrix (accuracy yardstick for fp32 gesdd and the HIP solver)."""
    centred = scores - scores.mean(dim=0, keepdim=True)
    return torch.linalg.svdvals(centred.double())


def estimate_dim(s):
    """The ID rule of plot_utils.py:173-183 / :224-230, in float64 numpy.

    diff[j] = (s[j+1]-s[j+2])/(s[1]-s[2]); softmax; dim = len(diff) - argmax.
    """
    s = [float(v) for v in s]
    nf = s[1] - s[2]
    diff = np.array([(s[i] - s[i + 1]) / nf for i in range(1, len(s) - 1)])
    e = np.exp(diff - np.max(diff))
    soft = e / e.sum(axis=0)
    return int(len(soft) - soft.argmax())


def estimate_dims(svd, mode="all"):
    """extract_sing_vals + rule, plot_utils.py:197-205,173-183."""
    sv = svd["singular_values"]
    if mode == "first":
        sv = [sv[0]]
    elif mode == "mean":
        sv = [np.mean(sv, axis=0)]
    return [estimate_dim(s) for s in sv]


def get_manifold_dimension(score_fn, sde, sampling_eps, loader, batchsize, num_datapoints, generator=None):
    """The loop of dim_reduction.py:150-211 over an iterable of data batches.

    Processes ``num_datapoints - 1`` points (the ``idx+1 >= num_datapoints``
    test, :159-164) and returns ``{'singular_values': [[...], ...]}``.
    """
    out, idx = [], 0
    for orig_batch in loader:
        if idx + 1 >= num_datapoints:
            break
        for x in orig_batch:
            if idx + 1 >= num_datapoints:
                break
            s = spectrum(score_matrix(score_fn, sde, x, batchsize, sampling_eps, generator=generator))
            out.append(s.tolist())
            idx += 1
    return {"singular_values": out}
