"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every declared symbol."""
import ctypes
import os
import re

import pytest

import id_diff_amd
from id_diff_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def library():
    if not os.path.exists(_lib.library_path()):
        _lib.build()
    return ctypes.CDLL(_lib.library_path())


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "idiff_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(idiff_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(library):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(library, name), f"{name} declared in include/idiff_hip.h but not exported"


def test_python_binding_covers_the_header():
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_abi_version_and_loader(library):
    library.idiff_abi_version.restype = ctypes.c_int
    assert library.idiff_abi_version() == 1
    assert _lib.lib() is not None  # binds argtypes for every symbol


def test_epilogue_struct_layout_matches_header():
    # 8 + 8 + 8 + 4 + 4 + 8 + 8 + 4 (+4 pad) + 8 + 8
    assert ctypes.sizeof(_lib.Epilogue) == 72
    assert _lib.Epilogue.rowscale.offset == 56 and _lib.Epilogue.residual.offset == 32


def test_product_refuses_cpu_tensors():
    import torch
    from id_diff_amd import op
    with pytest.raises(RuntimeError, match="no CPU path"):
        op.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
    with pytest.raises(RuntimeError, match="no CPU path"):
        op.fused_leaky_relu(torch.zeros(1, 2, 3), torch.zeros(2))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "id-diff_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
