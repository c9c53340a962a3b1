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


def conditional_times(sampling_eps):
    """The 12 noise levels of dim_reduction.py:39."""
    return torch.linspace(sampling_eps, 0.3, 12)


def get_conditional_manifold_dimension(score_fn, sde, sampling_eps, loader, num_datapoints=26, noise=None):
    """The loop of dim_reduction.py:39-114 over an iterable of (images, labels) validation batches.

    For every level of ``linspace(sampling_eps, 0.3, 12)`` (:39) the loader is walked again (:49), only items with
    label == 1 are used (:56-57), ``num_datapoints - 1`` of them (:52-53, :59-60); rows per score batch = the
    LOADER batch size (:51, :65), batching arithmetic as in the unconditional driver (:64-69), noise from the global
    torch RNG (``randn_like``, :79) unless ``noise(level, point, batch_index, shape)`` supplies the draws.
    Returns one dict per level with what the reference pickles (:103-114) plus the level's time:
    ``{'t', 'images', 'singular_values', 'labels'}``; the directory name of a level is ``'%.3f' % t`` (:41).
    """
    out = []
    for level, t_slice in enumerate(conditional_times(sampling_eps)):
        singular_values, labels, imgs, idx = [], [], [], 0
        for orig_batch, orig_labels in loader:
            batchsize = orig_batch.size(0)
            if idx + 1 >= num_datapoints:
                break
            for x, y in zip(orig_batch, orig_labels):
                if y.item() != 1:
                    continue
                if idx + 1 >= num_datapoints:
                    break
                imgs.append(x.permute(1, 2, 0))
                num_batches, extra, _ = batching(x.shape, batchsize)
                rep = x.repeat([batchsize] + [1] * x.ndim)
                vec_t = torch.ones(batchsize) * t_slice
                rows = []
                for i in range(1, num_batches + 1):
                    batch = rep.clone()
                    mean, std = sde.marginal_prob(batch, vec_t)
                    z = torch.randn_like(batch) if noise is None else noise(level, idx, i - 1, batch.shape)
                    batch = mean + std[(...,) + (None,) * len(batch.shape[1:])] * z
                    with torch.no_grad():
                        score = score_fn(batch, vec_t)
                    rows.append(score if i < num_batches else score[:extra])
                singular_values.append(spectrum(torch.flatten(torch.cat(rows, dim=0), start_dim=1)).tolist())
                labels.append(y.item())
                idx += 1
        out.append({"t": float(t_slice), "images": torch.stack(imgs).numpy() if imgs else [],
                    "singular_values": singular_values, "labels": labels})
    return out


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
