"""Oracle restatement of the perturbation kernels the hot path uses.

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math
import torch


class VESDE:
    """Variance-exploding SDE, /root/reference/sde_lib.py:316-347.

    Only ``marginal_prob`` is on the hot path: mean = x,
    std = sigma_min * (sigma_max / sigma_min) ** t, computed in t's dtype.
    """

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min, self.sigma_max, self.N = sigma_min, sigma_max, N

    def marginal_prob(self, x, t):
        lo = torch.tensor(self.sigma_min).type_as(t)
        hi = torch.tensor(self.sigma_max).type_as(t)
        return x, lo * (hi / lo) ** t


class VPSDE:
    """Variance-preserving SDE, /root/reference/sde_lib.py:222-252 (marginal_prob only)."""

    def __init__(self, beta_min=0.1, beta_max=20., N=1000):
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N

    def marginal_prob(self, x, t):
        log_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_coeff).reshape((-1,) + (1,) * (x.ndim - 1)) * x
        return mean, torch.sqrt(1. - torch.exp(2. * log_coeff))


class subVPSDE(VPSDE):
    """sub-VP SDE, /root/reference/sde_lib.py:276-304: the VP mean, std = 1 - exp(2 log_mean_coeff) (no square root)."""

    def marginal_prob(self, x, t):
        log_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_coeff).reshape((-1,) + (1,) * (x.ndim - 1)) * x
        return mean, 1 - torch.exp(2. * log_coeff)


class SNRSDE:
    """/root/reference/sde_lib.py:153-187 with the default gamma(t) = a t + b t^c (a, b, c = 2, 3, 6), normalised so that
    -log SNR runs from -10 at t = 0 to 5 at t = 1: mean = sqrt(SNR / (1 + SNR)) x, std = sqrt(1 / (1 + SNR)).
    The unconditional get_score_fn has a branch for it (models/utils.py:270-277: labels = t (N - 1), score = -model / std),
    pinned by tests/golden/ncsnpp_snr.npz (the reference's own score_fn output)."""

    def __init__(self, N, a=2, b=3, c=6, minus_log_SNR_0=-10, minus_log_SNR_1=5):
        self.N = N
        gamma = lambda t: a * t + b * t ** c
        k = (minus_log_SNR_1 - minus_log_SNR_0) / (gamma(1) - gamma(0))
        self.log_SNR = lambda t: -(minus_log_SNR_0 + k * (gamma(t) - gamma(0)))

    def marginal_prob(self, x, t):
        snr = torch.exp(self.log_SNR(t))
        alpha = torch.sqrt(snr / (1 + snr)).reshape((-1,) + (1,) * (x.ndim - 1))
        return alpha * x, torch.sqrt(1 / (1 + snr))


def make_sde(config):
    """(sde, sampling_eps) as /root/reference/lightning_modules/BaseSdeGenerativeModel.py:27-47."""
    kind = config.training.sde.lower()
    if kind == "vesde":
        return VESDE(config.model.sigma_min, config.model.sigma_max, config.model.num_scales), 1e-5
    if kind == "vpsde":
        return VPSDE(config.model.beta_min, config.model.beta_max, config.model.num_scales), 1e-3
    if kind == "subvpsde":
        return subVPSDE(config.model.beta_min, config.model.beta_max, config.model.num_scales), 1e-3
    if kind == "snrsde":
        return SNRSDE(config.model.num_scales), 1e-3
    raise NotImplementedError(f"SDE {config.training.sde} is not on the manifold_dimension path")


def get_score_fn(sde, model, conditional=False, train=False, continuous=True):
    """Unconditional branch of /root/reference/models/utils.py:236-280.

    labels = t*(N-1); out = model.eval()(x, labels); std = marginal_prob(0,t)[1];
    score = -out/std -- the same three lines in the VP / subVP (:238-255, continuous), VE (:257-268) and SNR (:270-277)
    branches; any other SDE class is refused (:279-280).
    """
    if conditional or not continuous:
        raise NotImplementedError("only the unconditional continuous branch is on the hot path")
    if not isinstance(sde, (VESDE, VPSDE, SNRSDE)):      # subVPSDE is a VPSDE here, as in the reference's isinstance test
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def score_fn(x, t):
        model.eval()
        out = model(x, t * (sde.N - 1))
        std = sde.marginal_prob(torch.zeros_like(x), t)[1]
        return -out / std.reshape((-1,) + (1,) * (x.ndim - 1))

    return score_fn
