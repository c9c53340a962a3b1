"""SDE perturbation kernels used by the manifold_dimension path (host-side scalars + tiny device vectors).

Mirrors /root/reference/sde_lib.py: ``VESDE`` (:316-347) and ``VPSDE`` (:222-252), ``marginal_prob`` only --
the reverse SDE, discretisation and priors belong to sampling/training and are out of scope.
The per-sample std vector has B entries; it is produced with torch elementwise ops on the device the time
vector lives on (plumbing, a few hundred bytes) and consumed by the fused HIP kernels.
"""
import torch


class SDE:
    def __init__(self, N):
        self.N = N

    @property
    def T(self):
        return 1


class VESDE(SDE):
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, data_mean=None):
        super().__init__(N)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.diffused_mean = data_mean

    def marginal_prob(self, x, t):
        # sde_lib.py:342-347 of the reference: `torch.tensor(sigma).type_as(t)`.  The two scalars are kept per (device,
        # dtype): made afresh they are a pageable host-to-device copy per call, and that copy is stream-ordered -- the host
        # sat in it until the previous forward had drained (2 x 240 ms of blocking per point, a ~1 ms bubble on the GPU
        # at every forward boundary).  Same tensors, same arithmetic, same bits.
        key = (t.device, t.dtype, float(self.sigma_min), float(self.sigma_max))
        cached = getattr(self, "_sigma_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, torch.tensor(self.sigma_min).type_as(t), torch.tensor(self.sigma_max).type_as(t))
            self._sigma_cache = cached
        lo, hi = cached[1], cached[2]
        return x, lo * (hi / lo) ** t


class VPSDE(SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max

    def marginal_prob(self, x, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff).reshape((-1,) + (1,) * (x.ndim - 1)) * x
        return mean, torch.sqrt(1. - torch.exp(2. * log_mean_coeff))


class subVPSDE(VPSDE):
    """sde_lib.py:276-304 of the reference: the VP mean with std = 1 - exp(2 log_mean_coeff); get_score_fn treats it as VP
    (models/utils.py:238)."""

    def marginal_prob(self, x, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff).reshape((-1,) + (1,) * (x.ndim - 1)) * x
        return mean, 1 - torch.exp(2. * log_mean_coeff)


class SNRSDE(SDE):
    """sde_lib.py:153-187 of the reference (default gamma); get_score_fn evaluates it through its own branch
    (models/utils.py:270-277), the driver perturbs with mean = alpha(t) x (the mean-coefficient path of idiff_perturb_f32)."""

    def __init__(self, N, a=2, b=3, c=6, minus_log_SNR_0=-10, minus_log_SNR_1=5):
        super().__init__(N)
        gamma = lambda t: a * t + b * t ** c
        k = (minus_log_SNR_1 - minus_log_SNR_0) / (gamma(1) - gamma(0))
        self.log_SNR = lambda t: -(minus_log_SNR_0 + k * (gamma(t) - gamma(0)))

    def marginal_prob(self, x, t):
        snr = torch.exp(self.log_SNR(t))
        alpha = torch.sqrt(snr / (1 + snr)).reshape((-1,) + (1,) * (x.ndim - 1))
        return alpha * x, torch.sqrt(1 / (1 + snr))


def configure_sde(config):
    """(sde, sampling_eps) exactly as BaseSdeGenerativeModel.configure_sde (lightning_modules/BaseSdeGenerativeModel.py:27-47)."""
    kind = config.training.sde.lower()
    if kind == "vesde":
        if config.data.get("use_data_mean", False):
            raise NotImplementedError("data.use_data_mean needs the authors' datasets_mean/*.pt files")
        return VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max, N=config.model.num_scales), 1e-5
    if kind == "vpsde":
        return VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales), 1e-3
    if kind == "subvpsde":
        return subVPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales), 1e-3
    if kind == "snrsde":
        return SNRSDE(N=config.model.num_scales), 1e-3
    raise NotImplementedError(f"SDE {config.training.sde} unknown.")
