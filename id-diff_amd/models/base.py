"""Common base of the HIP-executed score networks."""
import torch
import torch.nn as nn

from .. import _lib


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

    @staticmethod
    def _check_inputs(x, t):
        """Device / dtype checks; returns contiguous views (the reference's ops call .contiguous() themselves)."""
        _lib._dev(x, "x", contiguous=False)
        _lib._dev(t, "time/labels", contiguous=False)
        if t.ndim != 1 or t.shape[0] != x.shape[0]:
            raise RuntimeError(f"time vector must be [batch]; got {tuple(t.shape)} for x {tuple(x.shape)}")
        return x.contiguous(), t.contiguous()
