"""Drop-in for the reference's ``op`` package (op/__init__.py:1-2): same three names."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
