"""Import shim: makes the package directory ``id-diff_amd/`` importable as ``id_diff_amd``.

The directory name carries a hyphen (it is the name the project layout
prescribes), which Python cannot import directly; this module loads
``id-diff_amd/__init__.py`` as the package ``id_diff_amd`` and replaces itself
in ``sys.modules``, so ``import id_diff_amd.op`` etc. work from the repo root.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "id-diff_amd")
_spec = importlib.util.spec_from_file_location(
    "id_diff_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["id_diff_amd"] = _pkg
_spec.loader.exec_module(_pkg)
