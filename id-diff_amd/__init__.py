"""id_diff_amd -- MI355X-native manifold_dimension hot path of GBATZOLIS/ID-diff.

Host side mirrors the reference's module layout (``op``, ``sde_lib``,
``models``, ``dim_reduction``, ``plot_utils``, ``configs``, ``main``,
``get_dim``) and runs every array operation through hand-written gfx950 HIP
kernels behind the C-ABI library ``libidiff_hip.so`` (include/idiff_hip.h).
There is no CPU fallback: without the library, or with CPU tensors, the ops
raise.
"""
import sys as _sys

__version__ = "0.1.0"

_DROPIN = ("op", "sde_lib", "models", "dim_reduction", "plot_utils", "configs", "lightning_callbacks")


def install_dropin():
    """Alias the reference's top-level module names to this package.

    After this call ``from op import upfirdn2d``, ``import sde_lib``,
    ``from models import utils as mutils``, ``import dim_reduction`` -- the
    imports the reference's own scripts use (/root/reference/get_dim.py:1-3,
    models/up_or_down_sampling.py:10) -- resolve to the MI355X implementation.
    """
    import importlib
    for name in _DROPIN:
        mod = importlib.import_module(f"{__name__}.{name}")
        _sys.modules.setdefault(name, mod)
    return [n for n in _DROPIN]
