"""Common base of the HIP-executed score networks."""
import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

# ---- admissibility of the fp16-pair kernels for a GroupNorm-fed contraction (csrc/winograd43h.hip, igemm.hip SPLIT == 2) --------
# The pair kernels cut the ACTIVATION operand as it is: the convolution's transformed patch V = B^T d B (the GEMM's row) must stay
# below the largest fp16 number or the outputs are NaN, and a tensor whose scale is far below one keeps only an absolute 2^-25.
# A GroupNorm's output is bounded by construction: a normalised group of n elements has |v| <= sqrt(n - 1), so
#     |gamma v + beta| <= sqrt(n) max|gamma| + max|beta|
# (the activations of models/layers.py:29-41 and the FIR / box resamplers never increase the bound beyond the FIR's absolute tap sum),
# and B^T's largest absolute row sum, for the points 0, +-2/3, +-3/2, inf of winograd43_shared.h, is 1 + b^2 + a (1 + b^2) = 5.4167
# per axis.  Both are known at pack time from the checkpoint's gamma / beta: a layer that COULD leave the range is sent to the
# fp32-contraction kernel, so no checkpoint can make this route produce a NaN the reference's fp32 arithmetic would not.
F43_INPUT_GAIN = (1.0 + 1.5 ** 2 + (2.0 / 3.0) * (1.0 + 1.5 ** 2)) ** 2          # 29.34
PAIRS_MAX = 60000.0              # fp16: 65504
PAIRS_MIN_RMS = 2.0 ** -6        # below this scale of the whole tensor the 2^-25 absolute floor is worse than fp32's 2^-24 relative
PAIRS_BOUND_CHECK = True         # tests switch it off to drive a NaN into the drivers' re-run (dim_reduction: safe rebuild)


class HipScoreModel(nn.Module):
    """Holds the reference's parameters (same ``state_dict`` keys) plus kernel-ready packed copies.

    Packed weights are rebuilt lazily after anything that can change parameters
    (``load_state_dict``, ``.to()``, ``.cuda()``).  ``forward`` refuses CPU tensors: the CPU restatement of
    these networks lives in ``oracle/`` and is test infrastructure, not a fallback.
    """

    def __init__(self):
        super().__init__()
        self._packed = None

    def _invalidate(self):
        self._packed = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._invalidate()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    @property
    def device(self):
        return next(self.parameters()).device

    def packed(self):
        if self._packed is None:
            dev = self.device
            if dev.type != "cuda":
                raise RuntimeError(f"{type(self).__name__}: parameters are on {dev}; move the model to the MI355X "
                                   "(`.to('cuda')`) -- id-diff_amd has no CPU path")
            _lib.lib()
            with torch.no_grad():
                self._packed = self._pack()
        return self._packed

    def _pack(self):
        raise NotImplementedError

    def pairs_admissible(self, gn, group_elems, gain=1.0, transform=True, modulated=False):
        """May the contraction fed by GroupNorm ``gn`` (normalising groups of ``group_elems`` elements; ``gain``: absolute tap sum of
        a resampling FIR in between; ``transform``: a Winograd input transform follows) run on fp16 pairs?  Decided ONCE per layer
        from max|gamma|, max|beta| (see the constants above) and remembered in the pack; a refusal is reported once.
        ``modulated``: the norm's output is further multiplied and shifted by a per-sample projection of the time embedding
        (BeatGANsblocks.py:258-332) whose range is not a property of the weights alone -- such layers keep the pair route
        and rely on the drivers' re-run of a non-finite point on the fp32 route (dim_reduction.ScoreMatrixBuilder.build(safe=True))."""
        if not PAIRS_BOUND_CHECK:
            return True
        cache = self.packed().setdefault("pairs_ok", {})
        key = (id(gn), int(group_elems), float(gain), bool(transform))
        if key not in cache:
            w, b = gn.weight.detach().double(), gn.bias.detach().double()
            stats = torch.stack([w.abs().max(), b.abs().max(), (w * w).mean() + (b * b).mean()]).cpu().numpy()   # one copy per layer
            bound = (np.sqrt(float(group_elems)) * stats[0] + stats[1]) * float(gain) * (F43_INPUT_GAIN if transform else 1.0)
            rms = float(np.sqrt(stats[2])) * float(gain)
            ok = bool(np.isfinite(bound)) and (modulated or bound < PAIRS_MAX) and rms >= PAIRS_MIN_RMS
            if not ok:
                warnings.warn(f"id-diff_amd: a GroupNorm({gn.num_channels}) with max|gamma| = {stats[0]:.3g}, max|beta| = {stats[1]:.3g} "
                              f"(rms scale {rms:.3g}) can leave the range of the fp16-pair kernels (bound {bound:.3g}): the "
                              "contraction it feeds runs on the fp32 route")
            cache[key] = ok
        return cache[key]

    @staticmethod
    def _out_buffer(out, B, C, H, W, device):
        """The NCHW output tensor of an image model: the caller's ``out`` (e.g. the rows of the device-resident score matrix, so
        that the last kernel of the network writes S directly) after checking it is what that kernel will write, else a new one."""
        if out is None:
            return torch.empty(B, C, H, W, device=device, dtype=torch.float32)
        _lib._dev(out, "out")
        if out.numel() != B * C * H * W or out.device != device:
            raise RuntimeError(f"out: expected {B * C * H * W} contiguous fp32 values on {device}, got {tuple(out.shape)} on {out.device}")
        return out.view(B, C, H, W)

    @staticmethod
    def _check_inputs(x, t):
        """Device / dtype checks; returns contiguous views (the reference's ops call .contiguous() themselves)."""
        _lib._dev(x, "x", contiguous=False)
        _lib._dev(t, "time/labels", contiguous=False)
        if t.ndim != 1 or t.shape[0] != x.shape[0]:
            raise RuntimeError(f"time vector must be [batch]; got {tuple(t.shape)} for x {tuple(x.shape)}")
        return x.contiguous(), t.contiguous()
