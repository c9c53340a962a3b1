"""Spectrum -> integer intrinsic dimension (the part of /root/reference/plot_utils.py that is on the hot path).

``plot_distribution`` (:158-195) and ``plot_dims`` (:207-238) both apply the same 8-line rule to every spectrum
and return the list of estimated dimensions; the matplotlib rendering around it is out of scope and only runs
when ``render=True`` and matplotlib is importable.  The rule is evaluated in float64 numpy exactly like the
reference (it is a few hundred flops per point; an argmax, not a kernel).
"""
import numpy as np


def softmax(x):
    e_x = np.exp(x - np.max(x))
    return e_x / e_x.sum(axis=0)


def gap_profile(s):
    """Normalised consecutive gaps diff[j] = (s[j+1]-s[j+2])/(s[1]-s[2]), j = 0..n-3 (plot_utils.py:175-176)."""
    s = [float(v) for v in s]
    norm_factor = s[1] - s[2]
    return np.array([(s[i] - s[i + 1]) / norm_factor for i in range(1, len(s) - 1)])


def estimate_dim(s, tail=None):
    """dim = len(soft) - argmax(soft) (plot_utils.py:177-183); ``tail`` keeps only the last entries first."""
    soft = softmax(gap_profile(s))
    if tail:
        soft = soft[-tail:]
    return int(len(soft) - soft.argmax())


def extract_sing_vals(svd, mode='first'):
    """plot_utils.py:197-205."""
    singular_vals = svd['singular_values']
    if mode == 'first':
        return [singular_vals[0]]
    elif mode == 'all':
        return singular_vals
    elif mode == 'mean':
        return [np.mean(singular_vals, axis=0)]
    raise ValueError(f"unknown aggregation mode {mode}")


def plot_distribution(svd, mode='first', return_tensor=False, tail=None, render=False):
    dims = [estimate_dim(s, tail) for s in extract_sing_vals(svd, mode)]
    if render:
        _render_distribution(svd, mode, tail)
    return (None, dims) if return_tensor else dims


def plot_dims(svd, title='Histogram of dimensions', tick_step=2, tick_start=1, render=False):
    dims = [estimate_dim(s) for s in extract_sing_vals(svd, 'all')]
    fig = _render_hist(dims, title) if render else None
    return fig, dims


def _render_distribution(svd, mode, tail):
    from matplotlib import pyplot as plt
    plt.figure(figsize=(15, 10))
    plt.grid(alpha=0.5)
    plt.title('Dimension distribution')
    for s in extract_sing_vals(svd, mode):
        soft = softmax(gap_profile(s))
        if tail:
            soft = soft[-tail:]
        plt.plot(list(range(1, 1 + len(soft)))[::-1], soft)
    return plt.gcf()


def _render_hist(dims, title):
    from matplotlib import pyplot as plt
    plt.figure(figsize=(15, 10))
    plt.grid(alpha=0.5)
    plt.xlabel('dimension')
    plt.ylabel('count')
    plt.title(title)
    plt.hist(dims, bins=np.arange(1, max(dims) + 1, 0.5))
    return plt.gcf()
