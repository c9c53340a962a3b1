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


def test_no_kernel_spills_to_scratch_beyond_the_known_ones(library, tmp_path):
    """both GPU faults of the build were inside multi-workgroup kernels whose register allocation had spilled
    hundreds of bytes per lane to scratch; the rule since then is that no kernel of the library carries a scratch segment
    beyond a few dwords.  Read from the code objects."""
    import shutil, subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("no LLVM binutils")
    so = shutil.copy(_lib.library_path(), tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objs = [p for p in os.listdir(tmp_path) if "gfx950" in p]
    assert objs, "no gfx950 code object in the library"
    allowed = {}                            # no exceptions left since the factorisations keep two lanes per row
    seen = 0
    for o in objs:
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", str(tmp_path / o)], check=True, capture_output=True, text=True).stdout
        for name, size in re.findall(r"\.name:\s+(\S+)\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            seen += 1
            limit = next((v for k, v in allowed.items() if k in name), 64)
            assert int(size) <= limit, f"{name}: {size} bytes of scratch per lane"
    assert seen >= 40


def test_thread_option_acts_on_the_calling_thread_only():
    """idiff_set_thread_option: the fail-soft re-solve selects a slower eigensolver form for ITS launch; another host thread's
    launches keep reading the process-wide value (checked through a host-only query that reads a switch)."""
    import threading
    ask = lambda: _lib.conv2d_winograd_split_ok(4, 16, 16, 64, 64)       # reads IDIFF_WINO_SPLIT, no device call
    assert ask() is False
    seen = {}
    with _lib.thread_option("IDIFF_WINO_SPLIT", 1):
        assert ask() is True
        t = threading.Thread(target=lambda: seen.setdefault("other", ask()))
        t.start(); t.join()
    assert seen["other"] is False and ask() is False
    with pytest.raises(KeyError):
        with _lib.thread_option("IDIFF_NO_SUCH_SWITCH", 1):
            pass


def test_diagnostic_builds_cannot_reach_the_production_path(library):
    """probe scripts used to rebuild libidiff_hip.so in place with timing-only kernels.  Now a build with extra flags must
    name a variant (its own file), the product library reports no variant flags, and no probe script patches or rewrites build.sh."""
    import subprocess
    library.idiff_variant_flags.restype = ctypes.c_char_p
    assert library.idiff_variant_flags() == b""
    build_sh = os.path.join(ROOT, "id-diff_amd", "csrc", "build.sh")
    out = subprocess.run(["bash", build_sh], env=dict(os.environ, IDIFF_VARIANT_FLAGS="-DIDIFF_W43H_DIAG_NO_U", IDIFF_VARIANT=""),
                         capture_output=True, text=True)
    assert out.returncode == 4 and "must name its own output file" in out.stderr          # refused before anything is compiled
    out = subprocess.run(["bash", build_sh], env=dict(os.environ, IDIFF_VARIANT="../x"), capture_output=True, text=True)
    assert out.returncode == 4
    scripts = os.path.join(ROOT, "scripts")
    for name in os.listdir(scripts):
        if name.endswith(".py") and name != "_variant.py":
            text = open(os.path.join(scripts, name)).read()
            assert "build.sh" not in text or "_variant" in text, f"scripts/{name} drives build.sh itself"
            assert "IDIFF_SCRATCH_LIMIT" not in text, f"scripts/{name} lifts the scratch limit of a build it does not name"
