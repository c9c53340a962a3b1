"""Spectrum -> integer intrinsic dimension, and the figures either side of it (/root/reference/plot_utils.py).

``plot_distribution`` (:158-195) and ``plot_dims`` (:207-238) both apply the same 8-line rule to every spectrum
and return the list of estimated dimensions; ``plot_spectrum`` (:111-139) draws the spectra.  The rule is evaluated in
float64 numpy exactly like the reference (it is a few hundred flops per point; an argmax, not a kernel).  Figures are
drawn on matplotlib's off-screen Agg canvas only when asked for (``render=True`` / ``return_tensor=True``):
``return_tensor=True`` gives the [3, H, W] float image in [0, 1] the reference hands to TensorBoard (there through a
JPEG round trip and torchvision's ToTensor; here straight from the canvas).
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


def _pyplot():
    import matplotlib
    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt
    return plt


def figure_to_tensor(fig):
    """RGB image of a drawn figure as a [3, H, W] float32 torch tensor in [0, 1] (what ToTensor gives, plot_utils.py:131-137)."""
    import torch
    fig.canvas.draw()
    rgba = np.asarray(fig.canvas.buffer_rgba())
    img = torch.from_numpy(np.ascontiguousarray(rgba[..., :3])).permute(2, 0, 1).float() / 255.0
    _pyplot().close(fig)
    return img


def plot_spectrum(svd, return_tensor=False, mode='first', title='Score Spectrum', ground_truth=None):
    """plot_utils.py:111-139: one curve per spectrum, red dashed lines at ``len - ground_truth``."""
    plt = _pyplot()
    singular_values = extract_sing_vals(svd, mode)
    n = len(singular_values[0])
    plt.rcParams.update({'font.size': 24})
    fig = plt.figure(figsize=(15, 10))
    plt.grid(alpha=0.5)
    plt.title(title)
    plt.xticks(np.arange(0, n + 1, 10))
    if ground_truth:
        for gt in (ground_truth if isinstance(ground_truth, list) else [ground_truth]):
            plt.axvline(x=n - gt, color='red', ls='--')
    for sing_vals in singular_values:
        plt.plot(list(range(1, len(sing_vals) + 1)), [float(v) for v in sing_vals])
    return figure_to_tensor(fig) if return_tensor else fig


def plot_distribution(svd, mode='first', return_tensor=False, tail=None, render=False):
    """plot_utils.py:158-195.  ``return_tensor=True`` -> (image, dims) like the reference; otherwise the dims (and the
    figure is drawn only when ``render=True``)."""
    dims = [estimate_dim(s, tail) for s in extract_sing_vals(svd, mode)]
    if return_tensor:
        return figure_to_tensor(_render_distribution(svd, mode, tail)), dims
    if render:
        _render_distribution(svd, mode, tail)
    return dims


def plot_dims(svd, title='Histogram of dimensions', tick_step=2, tick_start=1, render=False):
    dims = [estimate_dim(s) for s in extract_sing_vals(svd, 'all')]
    fig = _render_hist(dims, title) if render else None
    return fig, dims


def _render_distribution(svd, mode, tail):
    plt = _pyplot()
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
    plt = _pyplot()
    plt.figure(figsize=(15, 10))
    plt.grid(alpha=0.5)
    plt.xlabel('dimension')
    plt.ylabel('count')
    plt.title(title)
    plt.hist(dims, bins=np.arange(1, max(dims) + 1, 0.5))
    return plt.gcf()
