"""The ``op`` package of the drop-in: the three public names of the reference's native-op package
(``FusedLeakyReLU``, ``fused_leaky_relu``, ``upfirdn2d``), here backed by libidiff_hip.so instead of a JIT-built
torch extension."""
import importlib as _importlib

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d"]

_act = _importlib.import_module(__name__ + ".fused_act")
_fir = _importlib.import_module(__name__ + ".upfirdn2d")
FusedLeakyReLU = _act.FusedLeakyReLU
fused_leaky_relu = _act.fused_leaky_relu
upfirdn2d = _fir.upfirdn2d      # the function shadows the submodule, exactly as in the reference
