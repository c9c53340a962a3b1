"""Score networks of the manifold_dimension path, executed with the gfx950 kernels of libidiff_hip.so.

Importing the package registers ``fcn``, ``ncsnpp``, ``BeatGANsUNetModel`` (and ``ksphere_exact``) under the names
``config.model.name`` selects (reference: models/utils.py:24-47, models/fcn.py:6, models/ncsnpp.py:39).
"""
from . import utils  # noqa: F401
from . import fcn, ncsnpp, ddpm, beatgans, ksphere_exact  # noqa: F401
