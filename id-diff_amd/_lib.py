"""ctypes binding of libidiff_hip.so (C ABI declared in include/idiff_hip.h).

PyTorch is used for device memory and streams only: every wrapper below takes
CUDA(=HIP) fp32/fp64 tensors, checks device / dtype / contiguity / shape on
the host, passes ``tensor.data_ptr()`` and the current stream handle to the
library and raises ``RuntimeError`` if the call reports an error.  There is no
CPU path: a missing library or a CPU tensor is an error, never a fallback.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDIFF_LIB_VARIANT=<name>: a diagnostic / A-B build made by csrc/build.sh with IDIFF_VARIANT=<name> (scripts/_variant.py); the
# production library is never overwritten by such a build and never reports variant flags (checked in lib())
_VARIANT = os.environ.get("IDIFF_LIB_VARIANT", "")
_LIB_PATH = os.path.join(_HERE, "csrc", f"libidiff_hip.{_VARIANT}.so" if _VARIANT else "libidiff_hip.so")
_lib = None

ACT = {None: 0, "none": 0, "linear": 0, "silu": 1, "swish": 1, "elu": 2, "relu": 3, "lrelu": 4}

c_i, c_i64, c_f, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class Epilogue(ctypes.Structure):
    """Mirror of ``idiff_epilogue`` (include/idiff_hip.h)."""
    _fields_ = [("bias", c_p), ("rowbias", c_p), ("ld_rowbias", c_i64), ("rows_per_group", c_i), ("act", c_i),
                ("residual", c_p), ("ld_residual", c_i64), ("out_scale", c_f), ("rowscale", c_p), ("colstats", c_p)]


_SIGNATURES = {
    "idiff_abi_version": (c_i, []),
    "idiff_last_error": (ctypes.c_char_p, []),
    "idiff_source_stamp": (ctypes.c_char_p, []),
    "idiff_variant_flags": (ctypes.c_char_p, []),
    "idiff_set_option": (c_i, [ctypes.c_char_p, c_i]),
    "idiff_gemm_pairs_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "idiff_gemm_pairs_scale_f32": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p]),
    "idiff_gemm_pairs_f32": (c_i, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_p, c_p]),
    "idiff_pairs_act_scale_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "idiff_gemm_pairs_2src_f32": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_i64, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
    "idiff_set_thread_option": (c_i, [ctypes.c_char_p, c_i, c_i]),
    "idiff_upfirdn2d_f32": (c_i, [c_p, c_p, c_p] + [c_i] * 14 + [c_p]),
    "idiff_fused_bias_act_f32": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "idiff_upfirdn2d_f16": (c_i, [c_p, c_p, c_p] + [c_i] * 14 + [c_p]),
    "idiff_upfirdn2d_f64": (c_i, [c_p, c_p, c_p] + [c_i] * 14 + [c_p]),
    "idiff_fused_bias_act_f16": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "idiff_fused_bias_act_f64": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "idiff_gemm_f32": (c_i, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i,
                             ctypes.POINTER(Epilogue), c_p]),
    "idiff_gemm_2src_f32": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_i64, c_p, c_i64, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_p]),
    "idiff_conv2d_nhwc_f32": (c_i, [c_p, c_p, c_p] + [c_i] * 10 + [ctypes.POINTER(Epilogue), c_p]),
    "idiff_gemm_colstats_split": (c_i, [c_i, c_i, c_i, c_i64, c_i64, c_i]),
    "idiff_conv2d_colstats_split": (c_i, [c_i] * 10),
    "idiff_conv2d_winograd_ok": (c_i, [c_i] * 5),
    "idiff_winograd_weight_floats": (c_i64, [c_i, c_i]),
    "idiff_winograd_pack_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "idiff_conv2d_winograd_f32": (c_i, [c_p, c_p, c_p] + [c_i] * 5 + [ctypes.POINTER(Epilogue), c_p]),
    "idiff_conv2d_winograd_colstats_split": (c_i, [c_i] * 5),
    "idiff_conv2d_winograd43_ok": (c_i, [c_i] * 5),
    "idiff_conv2d_winograd43_colstats_split": (c_i, [c_i] * 5),
    "idiff_winograd43_weight_floats": (c_i64, [c_i, c_i]),
    "idiff_winograd43_pack_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "idiff_conv2d_winograd43_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_p]),
    "idiff_conv2d_winograd43h_ok": (c_i, [c_i] * 5),
    "idiff_winograd43h_weight_floats": (c_i64, [c_i, c_i]),
    "idiff_winograd43h_pack_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "idiff_conv2d_winograd43h_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_p]),
    "idiff_conv2d_wino1d_ok": (c_i, [c_i] * 5),
    "idiff_conv2d_wino1d_colstats_split": (c_i, [c_i] * 5),
    "idiff_wino1d_weight_floats": (c_i64, [c_i, c_i]),
    "idiff_wino1d_pack_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "idiff_conv2d_wino1d_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_p]),
    "idiff_conv2d_winograd_split_ok": (c_i, [c_i] * 5),
    "idiff_winograd_split_weight_floats": (c_i64, [c_i, c_i]),
    "idiff_winograd_pack_split_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "idiff_conv2d_winograd_split_f32": (c_i, [c_p, c_p, c_p] + [c_i] * 5 + [ctypes.POINTER(Epilogue), c_p]),
    "idiff_groupnorm_finalize_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "idiff_groupnorm_nsplit": (c_i, [c_i, c_i, c_i]),
    "idiff_groupnorm_stats_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    "idiff_groupnorm_apply_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i64, c_i, c_p, c_p]),
    "idiff_groupnorm_apply_colstats_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_p, c_p, c_p, c_i64,
                                                 c_i, c_p, c_p]),
    "idiff_softmax_rows_f32": (c_i, [c_p, c_p, c_i64, c_i, c_f, c_p]),
    "idiff_attention256_ok": (c_i, [c_i, c_i, c_i]),
    "idiff_attention256_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "idiff_affine_act_f32": (c_i, [c_p, c_p, c_i64, c_f, c_f, c_i, c_p, c_i64, c_p]),
    "idiff_add_scale_f32": (c_i, [c_p, c_p, c_p, c_i64, c_f, c_p]),
    "idiff_fourier_embed_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "idiff_positional_embed_f32": (c_i, [c_p, c_p, c_i, c_i, c_f, c_i, c_p]),
    "idiff_concat_cols_f32": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i64, c_p]),
    "idiff_nchw_to_nhwc_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "idiff_nhwc_to_nchw_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "idiff_resample2x_nhwc_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "idiff_perturb_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "idiff_perturb_randn_f32": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, ctypes.c_uint64, c_p, c_p]),
    "idiff_spectrum_workspace_bytes": (c_i64, [c_i, c_i, c_i]),
    "idiff_spectrum_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i64, c_p, c_p, c_p]),
    "idiff_colmean_f64": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "idiff_centered_gram_f64": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "idiff_centered_gram_rows_f64": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "idiff_symmetrize_upper_f64": (c_i, [c_p, c_i, c_p]),
    "idiff_symtridiag_scratch_doubles": (c_i64, [c_i]),
    "idiff_symband_ld": (c_i, []),
    "idiff_symband_f64": (c_i, [c_p, c_i, c_p, c_p]),
    "idiff_symtridiag_f64": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "idiff_symtridiag_plan": (c_i, [c_i]),
    "idiff_tridiag_eigvals_f64": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path():
    return _LIB_PATH


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (works without a GPU)."""
    out = subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libidiff_hip.so failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout.strip())
    return _LIB_PATH


def source_stamp():
    """The stamp csrc/build.sh computes: sha256 over csrc/*.hip, csrc/*.h (sorted) and include/idiff_hip.h."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") or f.endswith(".h"))
    h = hashlib.sha256()
    for f in [os.path.join(csrc, n) for n in names] + [os.path.join(os.path.dirname(_HERE), "include", "idiff_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib():
    """The loaded library; raises (never falls back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(id-diff_amd has no CPU or PyTorch fallback).")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header/library mismatch
            fn.restype, fn.argtypes = res, args
        if handle.idiff_abi_version() != 1:
            raise RuntimeError("libidiff_hip.so ABI version mismatch; rebuild it")
        built, tree = handle.idiff_source_stamp().decode(), source_stamp()
        if built != tree:
            raise RuntimeError(f"{_LIB_PATH} was built from other sources (stamp {built}, tree {tree}): rebuild it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
        flags = handle.idiff_variant_flags().decode()
        if flags and not _VARIANT:
            raise RuntimeError(f"{_LIB_PATH} is a diagnostic build (compiled with {flags!r}): its kernels may be wrong by construction. "
                               "Rebuild the product library with `python -c 'import __graft_entry__ as g; g.build()'`")
        if _VARIANT:
            import sys
            print(f"[id-diff_amd] DIAGNOSTIC library {os.path.basename(_LIB_PATH)} (flags: {flags or 'none'}) -- not the product",
                  file=sys.stderr, flush=True)
        _lib = handle
    return _lib


def set_option(name, value):
    """Flip a library debug switch (``IDIFF_NO_WINOGRAD`` ...); returns the previous value."""
    prev = lib().idiff_set_option(name.encode(), int(value))
    if prev < 0:
        raise KeyError(f"unknown libidiff_hip option {name!r}")
    return prev if prev > 1 else bool(prev)


class thread_option:
    """``with thread_option("IDIFF_CHASE_WAVEFRONT", 1): ...`` -- the switch for the launches THIS host thread makes inside the
    block, and for nobody else's (idiff_set_thread_option: launchers read their switches on the calling thread).  This is
    what the fail-soft re-solve uses: flipping the process-wide switch around a launch would also redirect an eigensolve
    that another host thread launches in that window."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        if lib().idiff_set_thread_option(self.name, self.value, 1) != 0:
            raise KeyError(f"unknown libidiff_hip option {self.name.decode()!r}")
        return self

    def __exit__(self, *exc):
        lib().idiff_set_thread_option(self.name, 0, 0)
        return False


def _check(rc, what):
    if rc != 0:
        msg = lib().idiff_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=torch.float32, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda device), got {t.device}; "
                           "id-diff_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def make_epilogue(bias=None, rowbias=None, rows_per_group=1, act=None, residual=None, out_scale=1.0,
                  ld_rowbias=None, ld_residual=None, rowscale=None, colstats=None):
    ep = Epilogue()
    ep.bias = _ptr(bias)
    ep.rowbias = _ptr(rowbias)
    ep.ld_rowbias = (rowbias.stride(0) if ld_rowbias is None else ld_rowbias) if rowbias is not None else 0
    ep.rows_per_group = int(rows_per_group)
    ep.act = ACT[act]
    ep.residual = _ptr(residual)
    ep.ld_residual = (residual.shape[-1] if ld_residual is None else ld_residual) if residual is not None else 0
    ep.out_scale = float(out_scale)
    ep.rowscale = _ptr(rowscale)
    ep.colstats = _ptr(colstats)
    # the struct only carries raw addresses: keep the tensors alive until the launch that consumes `ep` is enqueued
    ep._keepalive = (bias, rowbias, residual, rowscale, colstats)
    return ep


# ------------------------------------------------------------------------------------------- native ops
# the dtypes of the reference's native-op dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF) -> entry-point suffix
OP_DTYPES = {torch.float32: "f32", torch.float16: "f16", torch.float64: "f64"}


def op_suffix(t, name):
    """Entry-point suffix for a native-op tensor; any other dtype is refused as the reference's dispatch refuses it."""
    try:
        return OP_DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"{name}: dtype {t.dtype} is not one of float32 / float16 / float64 (the reference's native ops "
                           "dispatch on floating types and half, op/upfirdn2d_kernel.cu:311)") from None


def upfirdn2d_raw(x, k, out, major, in_h, in_w, minor, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    kh, kw = k.shape
    sfx = op_suffix(x, "upfirdn2d")
    if k.dtype != x.dtype or out.dtype != x.dtype:
        raise RuntimeError(f"upfirdn2d: input {x.dtype}, kernel {k.dtype} and output {out.dtype} must share one dtype")
    fn = getattr(lib(), "idiff_upfirdn2d_" + sfx)
    _check(fn(x.data_ptr(), k.data_ptr(), out.data_ptr(), major, in_h, in_w, minor, kh, kw,
              up_x, up_y, down_x, down_y, px0, px1, py0, py1, _stream()), "idiff_upfirdn2d_" + sfx)


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, k):
    return (in_size * up + pad0 + pad1 - k) // down + 1


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    sfx = op_suffix(x, "fused_bias_act")
    _dev(x, "input", dtype=x.dtype)
    out = torch.empty_like(x)
    has_b = bias is not None and bias.numel() > 0
    has_r = ref is not None and ref.numel() > 0
    if has_b:
        _dev(bias, "bias", dtype=x.dtype)
        if x.ndim < 2 or bias.numel() != x.shape[1]:
            raise RuntimeError(f"bias has {bias.numel()} entries but input dim 1 is {tuple(x.shape)}")
    if has_r:
        _dev(ref, "refer", dtype=x.dtype)
        if ref.shape != x.shape:
            raise RuntimeError("refer must have the shape of input")
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    _check(getattr(lib(), "idiff_fused_bias_act_" + sfx)(x.data_ptr(), _ptr(bias if has_b else None), _ptr(ref if has_r else None),
                                                         out.data_ptr(), x.numel(), step_b, bias.numel() if has_b else 0, act, grad,
                                                         alpha, scale, _stream()), "idiff_fused_bias_act_" + sfx)
    return out


# ------------------------------------------------------------------------------------------- contractions
def gemm(a, bt, out=None, epilogue=None, M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
         batch=1, stride_a=0, stride_b=0, stride_c=0):
    """out[b] = epilogue(a[b] @ bt[b].T); 2-D tensors by default, explicit geometry for batched views."""
    explicit = M is not None  # strided views of larger buffers: the caller supplies the geometry
    _dev(a, "a", contiguous=not explicit); _dev(bt, "bt", contiguous=not explicit)
    if M is None:
        M, K = a.shape
        N = bt.shape[0]
        if bt.shape[1] != K:
            raise RuntimeError(f"gemm: inner dimensions differ: {tuple(a.shape)} x {tuple(bt.shape)}^T")
        lda, ldb = a.stride(0), bt.stride(0)
    if out is None:
        out = torch.empty((M, N) if batch == 1 else (batch, M, N), device=a.device, dtype=torch.float32)
        ldc = N
        stride_c = M * N
    elif ldc is None:
        ldc = out.stride(-2)
    _dev(out, "out", contiguous=not explicit)
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    _check(lib().idiff_gemm_f32(a.data_ptr(), lda, stride_a, bt.data_ptr(), ldb, stride_b, out.data_ptr(), ldc, stride_c,
                                M, N, K, batch, ep, _stream()), "idiff_gemm_f32")
    return out


def gemm_pairs_ok(M, N, K, batch=1):
    """True when gemm_pairs serves this shape (IDIFF_NO_PAIRS / IDIFF_NO_SPLIT / IDIFF_NO_PIPE turn it off)."""
    return bool(lib().idiff_gemm_pairs_ok(M, N, K, batch))


def gemm_pairs_scale(w):
    """Device tensor {s, 1 / s}: the power of two a weight [rows, K] is multiplied by before its cut into fp16 pairs (once per weight)."""
    _dev(w, "w")
    out = torch.empty(2, device=w.device, dtype=torch.float32)
    _check(lib().idiff_gemm_pairs_scale_f32(w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], out.data_ptr(), _stream()),
           "idiff_gemm_pairs_scale_f32")
    return out


def gemm_pairs(a, bt, w_scale, out, epilogue=None, weight_is_a=False, M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
               batch=1, stride_a=0, stride_b=0, stride_c=0, act_scale=None):
    """out[b] = epilogue(a[b] @ bt[b].T) on fp16 pairs (three matrix instructions per block instead of six).  One operand is a
    weight -- ``bt``, or ``a`` with ``weight_is_a`` -- whose ``w_scale`` comes from gemm_pairs_scale; the other an activation of
    order one (a GroupNorm's output): see idiff_gemm_pairs_f32.  2-D tensors by default, explicit geometry for batched views."""
    explicit = M is not None
    _dev(a, "a", contiguous=not explicit); _dev(bt, "bt", contiguous=not explicit); _dev(out, "out", contiguous=not explicit)
    _dev(w_scale, "w_scale")
    if M is None:
        M, K = a.shape
        N = bt.shape[0]
        if bt.shape[1] != K or out.shape[0] != M or out.shape[1] != N:
            raise RuntimeError(f"gemm_pairs: shapes {tuple(a.shape)} x {tuple(bt.shape)}^T -> {tuple(out.shape)}")
        lda, ldb, ldc = a.stride(0), bt.stride(0), out.stride(0)
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    _check(lib().idiff_gemm_pairs_f32(a.data_ptr(), lda, stride_a, bt.data_ptr(), ldb, stride_b, w_scale.data_ptr(), int(bool(weight_is_a)),
                                      _ptr(act_scale), out.data_ptr(), ldc, stride_c, M, N, K, batch, ep, _stream()), "idiff_gemm_pairs_f32")
    return out


def pairs_act_scale(stats1, C1, stats2, C2, B, HW):
    """Device tensor whose first two floats are {s, 1 / s}, s the power of two that brings the root mean square of a tensor (or of
    cat[x1, x2]) into [0.71, 1.41), from the column sums ``stats = (ws, nsplit)`` its producing contraction(s) wrote."""
    out = torch.empty(8, device=stats1[0].device, dtype=torch.float32)
    _check(lib().idiff_pairs_act_scale_f32(stats1[0].data_ptr(), stats1[1], C1, _ptr(stats2[0]) if stats2 is not None else None,
                                           stats2[1] if stats2 is not None else 0, C2, B, HW, out.data_ptr(), _stream()),
           "idiff_pairs_act_scale_f32")
    return out


def gemm_pairs_2src(a1, a2, act_scale, bt, w_scale, out, epilogue=None):
    """out = epilogue([a1 | a2] @ bt.T) on fp16 pairs; ``act_scale`` from pairs_act_scale over both sources, ``w_scale`` from
    gemm_pairs_scale(bt)."""
    _dev(a1, "a1"); _dev(a2, "a2"); _dev(bt, "bt"); _dev(out, "out"); _dev(act_scale, "act_scale"); _dev(w_scale, "w_scale")
    M, K1 = a1.shape
    K = K1 + a2.shape[1]
    if a2.shape[0] != M or a1.stride(0) != a2.stride(0) or bt.shape[1] != K:
        raise RuntimeError(f"gemm_pairs_2src: shapes {tuple(a1.shape)} | {tuple(a2.shape)} x {tuple(bt.shape)}^T")
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    _check(lib().idiff_gemm_pairs_2src_f32(a1.data_ptr(), a2.data_ptr(), a1.stride(0), K1, act_scale.data_ptr(), bt.data_ptr(), bt.stride(0),
                                           w_scale.data_ptr(), out.data_ptr(), out.stride(0), M, bt.shape[0], K, ep, _stream()),
           "idiff_gemm_pairs_2src_f32")
    return out


def gemm_normed(cache, a, w, out, epilogue=None, pairs=True):
    """out = epilogue(a @ w.T) for ``a`` [M, K] = the output of a GroupNorm (order one by construction) and a weight ``w`` [N, K]:
    on fp16 pairs where that form serves the shape, else on gemm's six bf16 products.  ``cache``: a dict that lives as long as
    the weights (the executor's pack), holding each weight's power-of-two scale."""
    M, K = a.shape
    if not pairs or not gemm_pairs_ok(M, w.shape[0], K):     # `pairs`: the caller's range verdict (models/base.py: pairs_admissible)
        return gemm(a, w, out=out, epilogue=epilogue)
    return gemm_pairs(a, w, _pairs_scale_of(cache, w), out, epilogue=epilogue)


def _pairs_scale_of(cache, w):
    sc = cache.setdefault("pairs_scale", {})
    if id(w) not in sc:
        sc[id(w)] = (gemm_pairs_scale(w), w)                  # the weight itself keeps the id unique while the entry lives
    return sc[id(w)][0]


def gemm_weight_times_normed_t(cache, w, x, out, B, HW, C, pairs=True):
    """out[b] = w [C, C] @ x[b]^T for x [B, HW, C] = the output of a GroupNorm: V^T of an attention block, K-contiguous for P.V."""
    kw = dict(M=C, N=HW, K=C, lda=C, ldb=C, ldc=HW, batch=B, stride_a=0, stride_b=HW * C, stride_c=C * HW)
    if not pairs or not gemm_pairs_ok(C, HW, C, B):
        return gemm(w, x, out=out, **kw)
    return gemm_pairs(w, x, _pairs_scale_of(cache, w), out, weight_is_a=True, **kw)


def gemm_2src(a1, a2, bt, out, epilogue=None):
    """out = epilogue([a1 | a2] @ bt.T) for two [M, K/2]-shaped sources of equal row pitch (concatenation never formed)."""
    _dev(a1, "a1"); _dev(a2, "a2"); _dev(bt, "bt"); _dev(out, "out")
    M, K1 = a1.shape
    K = K1 + a2.shape[1]
    if a2.shape[0] != M or a1.stride(0) != a2.stride(0) or bt.shape[1] != K:
        raise RuntimeError(f"gemm_2src: shapes {tuple(a1.shape)} | {tuple(a2.shape)} x {tuple(bt.shape)}^T")
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    _check(lib().idiff_gemm_2src_f32(a1.data_ptr(), a2.data_ptr(), a1.stride(0), K1, bt.data_ptr(), bt.stride(0), out.data_ptr(),
                                     out.stride(0), M, bt.shape[0], K, ep, _stream()), "idiff_gemm_2src_f32")
    return out


def conv2d_nhwc(x, wt, out, B, H, W, Cin, Cout, KH, KW, stride, pad, epilogue=None, pad_hi=None):
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    _check(lib().idiff_conv2d_nhwc_f32(x.data_ptr(), wt.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, KH, KW, stride,
                                       pad, pad if pad_hi is None else pad_hi, ep, _stream()), "idiff_conv2d_nhwc_f32")
    return out


def conv2d_winograd_ok(B, H, W, Cin, Cout):
    return bool(lib().idiff_conv2d_winograd_ok(B, H, W, Cin, Cout))


def conv2d_winograd_split_ok(B, H, W, Cin, Cout):
    """True when the opt-in split-precision Winograd kernel (IDIFF_WINO_SPLIT) serves this geometry NOW: asked per call, so a
    switch flipped after a bank was packed, or another (B, H, W) through the same layer, is seen."""
    return bool(lib().idiff_conv2d_winograd_split_ok(B, H, W, Cin, Cout))


def winograd_pack(wt, Cin, Cout, split=False):
    """wt [Cout, 3, 3, Cin] (the direct kernel's panel) -> the transformed filter bank of idiff_conv2d_winograd_f32, or --
    ``split=True`` -- of idiff_conv2d_winograd_split_f32 (three bf16 per weight)."""
    _dev(wt, "wt")
    if wt.numel() != Cout * 9 * Cin:
        raise RuntimeError(f"winograd_pack: expected {Cout}x3x3x{Cin} weights, got {tuple(wt.shape)}")
    if split:
        u = torch.empty(lib().idiff_winograd_split_weight_floats(Cin, Cout), device=wt.device, dtype=torch.float32)
        _check(lib().idiff_winograd_pack_split_f32(wt.data_ptr(), u.data_ptr(), Cin, Cout, _stream()), "idiff_winograd_pack_split_f32")
        return u
    u = torch.empty(lib().idiff_winograd_weight_floats(Cin, Cout), device=wt.device, dtype=torch.float32)
    _check(lib().idiff_winograd_pack_f32(wt.data_ptr(), u.data_ptr(), Cin, Cout, _stream()), "idiff_winograd_pack_f32")
    return u


def conv2d_winograd(x, u, out, B, H, W, Cin, Cout, epilogue=None, split=False):
    """``split`` names the kernel; the bank must be the one ``winograd_pack(..., split=split)`` made (sizes differ: checked)."""
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    want = (lib().idiff_winograd_split_weight_floats if split else lib().idiff_winograd_weight_floats)(Cin, Cout)
    if u.numel() != want:
        raise RuntimeError(f"conv2d_winograd: a filter bank of {u.numel()} floats for the {'split' if split else 'fp32'} kernel "
                           f"({want} expected): pack it with winograd_pack(..., split={split})")
    if split:
        _check(lib().idiff_conv2d_winograd_split_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ep, _stream()),
               "idiff_conv2d_winograd_split_f32")
        return out
    _check(lib().idiff_conv2d_winograd_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ep, _stream()),
           "idiff_conv2d_winograd_f32")
    return out


def conv2d_winograd43_ok(B, H, W, Cin, Cout):
    """True when the F(4x4, 3x3) kernel serves this geometry (asked per call: IDIFF_NO_WINO43 / IDIFF_NO_WINOGRAD switch it off)."""
    return bool(lib().idiff_conv2d_winograd43_ok(B, H, W, Cin, Cout))


def conv2d_winograd43_colstats_split(B, H, W, Cin, Cout):
    return lib().idiff_conv2d_winograd43_colstats_split(B, H, W, Cin, Cout)


def conv2d_winograd43h_ok(B, H, W, Cin, Cout):
    """True when the fp16-pair F(4x4, 3x3) kernel serves this geometry (IDIFF_NO_WINO43H and the fp32 form's switches turn it off)."""
    return bool(lib().idiff_conv2d_winograd43h_ok(B, H, W, Cin, Cout))


def winograd43_pack(wt, Cin, Cout, pairs=False):
    """wt [Cout, 3, 3, Cin] -> the transformed filter bank of idiff_conv2d_winograd43_f32 (36 * Cin * Cout floats), or with
    pairs=True that of idiff_conv2d_winograd43h_f32 (scaled fp16 pairs, 36 * Cin * Cout + 4 floats)."""
    _dev(wt, "wt")
    if wt.numel() != Cout * 9 * Cin:
        raise RuntimeError(f"winograd43_pack: expected {Cout}x3x3x{Cin} weights, got {tuple(wt.shape)}")
    if pairs:
        u = torch.empty(lib().idiff_winograd43h_weight_floats(Cin, Cout), device=wt.device, dtype=torch.float32)
        _check(lib().idiff_winograd43h_pack_f32(wt.data_ptr(), u.data_ptr(), Cin, Cout, _stream()), "idiff_winograd43h_pack_f32")
        return u
    u = torch.empty(lib().idiff_winograd43_weight_floats(Cin, Cout), device=wt.device, dtype=torch.float32)
    _check(lib().idiff_winograd43_pack_f32(wt.data_ptr(), u.data_ptr(), Cin, Cout, _stream()), "idiff_winograd43_pack_f32")
    return u


def conv2d_winograd43(x, u, out, B, H, W, Cin, Cout, epilogue=None, pairs=False):
    """pairs: `u` is a bank of fp16 pairs (winograd43_pack(..., pairs=True)) and the contraction runs on the fp16 matrix cores."""
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    if pairs:
        if u.numel() != 36 * Cin * Cout + 4:
            raise RuntimeError(f"conv2d_winograd43: a bank of {u.numel()} floats ({36 * Cin * Cout + 4} expected): pack it with "
                               "winograd43_pack(..., pairs=True)")
        _check(lib().idiff_conv2d_winograd43h_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ep, _stream()),
               "idiff_conv2d_winograd43h_f32")
        return out
    if u.numel() != 36 * Cin * Cout:
        raise RuntimeError(f"conv2d_winograd43: a filter bank of {u.numel()} floats ({36 * Cin * Cout} expected): pack it with winograd43_pack")
    _check(lib().idiff_conv2d_winograd43_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ep, _stream()),
           "idiff_conv2d_winograd43_f32")
    return out


def conv2d_wino1d_ok(B, H, W, Cin, Cout):
    """True when the row-wise F(4, 3) kernel on fp16 pairs (csrc/wino1d.hip) serves this geometry (W in {4, 8, 16, 32, 64}; off under
    IDIFF_NO_WINOGRAD / IDIFF_NO_WINO43H / IDIFF_NO_WINO1D)."""
    return bool(lib().idiff_conv2d_wino1d_ok(B, H, W, Cin, Cout))


def conv2d_wino1d_colstats_split(B, H, W, Cin, Cout):
    return lib().idiff_conv2d_wino1d_colstats_split(B, H, W, Cin, Cout)


def wino1d_pack(wt, Cin, Cout):
    """wt [Cout, 3, 3, Cin] -> the filter bank of idiff_conv2d_wino1d_f32 (scaled fp16 pairs of (G g[ky])[i], 18 * Cin * Cout + 4 floats)."""
    if wt.dtype != torch.float32 or not wt.is_contiguous() or tuple(wt.shape) != (Cout, 3, 3, Cin):
        raise RuntimeError(f"wino1d_pack: expected contiguous float32 {Cout}x3x3x{Cin} weights, got {wt.dtype} {tuple(wt.shape)}")
    u = torch.empty(lib().idiff_wino1d_weight_floats(Cin, Cout), device=wt.device, dtype=torch.float32)
    _check(lib().idiff_wino1d_pack_f32(wt.data_ptr(), u.data_ptr(), Cin, Cout, _stream()), "idiff_wino1d_pack_f32")
    return u


def conv2d_wino1d(x, u, out, B, H, W, Cin, Cout, epilogue=None):
    ep = ctypes.byref(epilogue) if epilogue is not None else None
    if u.numel() != 18 * Cin * Cout + 4:
        raise RuntimeError(f"conv2d_wino1d: a bank of {u.numel()} floats ({18 * Cin * Cout + 4} expected): pack it with wino1d_pack")
    _check(lib().idiff_conv2d_wino1d_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ep, _stream()), "idiff_conv2d_wino1d_f32")
    return out


def conv2d_winograd_colstats_split(B, H, W, Cin, Cout):
    return lib().idiff_conv2d_winograd_colstats_split(B, H, W, Cin, Cout)


# ------------------------------------------------------------------------------------------- norm / pointwise
def gemm_colstats_split(M, N, K, lda, ldb, rows_per_sample):
    return lib().idiff_gemm_colstats_split(M, N, K, lda, ldb, rows_per_sample)


def conv2d_colstats_split(B, H, W, Cin, Cout, KH, KW, stride, pad, pad_hi=None):
    return lib().idiff_conv2d_colstats_split(B, H, W, Cin, Cout, KH, KW, stride, pad, pad if pad_hi is None else pad_hi)


def groupnorm_finalize(ws1, nsplit1, C1, ws2, nsplit2, C2, B, HW, G, eps, stats):
    _check(lib().idiff_groupnorm_finalize_f32(ws1.data_ptr(), nsplit1, C1, _ptr(ws2), nsplit2, C2, B, HW, G, eps,
                                              stats.data_ptr(), _stream()), "idiff_groupnorm_finalize_f32")


def groupnorm_nsplit(B, HW, C):
    return lib().idiff_groupnorm_nsplit(B, HW, C)


def groupnorm_stats(x, C, x2, C2, B, HW, G, eps, workspace, stats):
    _check(lib().idiff_groupnorm_stats_f32(x.data_ptr(), C, _ptr(x2), C2, B, HW, G, eps, workspace.data_ptr(),
                                           stats.data_ptr(), _stream()), "idiff_groupnorm_stats_f32")


def groupnorm_apply(x, C, x2, C2, B, HW, G, stats, gamma, beta, act, y, mod=None):
    _check(lib().idiff_groupnorm_apply_f32(x.data_ptr(), C, _ptr(x2), C2, B, HW, G, stats.data_ptr(), gamma.data_ptr(),
                                           beta.data_ptr(), _ptr(mod), mod.stride(0) if mod is not None else 0,
                                           ACT[act], y.data_ptr(), _stream()),
           "idiff_groupnorm_apply_f32")


def groupnorm_apply_colstats(x, C, x2, C2, B, HW, G, ws1, ns1, ws2, ns2, eps, gamma, beta, act, y, mod=None):
    """GroupNorm apply whose statistics come from the producers' epilogue column sums (finalize + apply in one launch)."""
    _check(lib().idiff_groupnorm_apply_colstats_f32(x.data_ptr(), C, _ptr(x2), C2, B, HW, G, ws1.data_ptr(), ns1, _ptr(ws2), ns2,
                                                    eps, gamma.data_ptr(), beta.data_ptr(), _ptr(mod),
                                                    mod.stride(0) if mod is not None else 0, ACT[act], y.data_ptr(), _stream()),
           "idiff_groupnorm_apply_colstats_f32")


def softmax_rows(x, y, rows, cols, scale):
    _check(lib().idiff_softmax_rows_f32(x.data_ptr(), y.data_ptr(), rows, cols, scale, _stream()),
           "idiff_softmax_rows_f32")


def attention256_ok(B, tokens, C):
    """True when the one-launch attention serves this shape (256 tokens, 128 / 256 channels; off under IDIFF_NO_FUSED_ATTN / IDIFF_NO_PAIRS)."""
    return bool(lib().idiff_attention256_ok(B, tokens, C))


def pairs_scale_from_rows(w, bias=None, extra=1.0):
    """Device tensor {s, 1 / s}: the power of two that brings the output of ``w @ n + bias`` -- n a GroupNorm's output, unit variance per
    element by construction -- to a root mean square near one: rms^2 = mean_i |w_i|^2 (+ mean b^2).  Computed on the device, once per
    weight (no host synchronisation).  ``extra``: a known factor of the input's scale (GroupNorm gamma's rms)."""
    ms = (w.double() ** 2).sum(dim=1).mean() * float(extra) ** 2
    if bias is not None:
        ms = ms + (bias.double() ** 2).mean()
    e = torch.round(-0.5 * torch.log2(ms.clamp_min(1e-300))).clamp(-24, 24)
    s = torch.exp2(e)
    return torch.stack([s, 1.0 / s]).to(torch.float32).contiguous()


def attention256(qk, vt, out, B, C, s_qk, s_v, scale, bias_v=None):
    """out [B * 256, C] = softmax(q k^T * scale) v (+ bias_v) per sample; qk [B * 256, 2 C] (q | k), vt [B, C, 256]."""
    _dev(qk, "qk"); _dev(vt, "vt"); _dev(out, "out"); _dev(s_qk, "s_qk"); _dev(s_v, "s_v")
    if qk.shape != (B * 256, 2 * C) or vt.shape != (B, C, 256) or out.numel() != B * 256 * C:
        raise RuntimeError(f"attention256: shapes qk {tuple(qk.shape)}, vt {tuple(vt.shape)}, out {tuple(out.shape)} for B = {B}, C = {C}")
    _check(lib().idiff_attention256_f32(qk.data_ptr(), qk.stride(0), vt.data_ptr(), _ptr(bias_v), s_qk.data_ptr(), s_v.data_ptr(),
                                        out.data_ptr(), B, 256, C, float(scale), _stream()), "idiff_attention256_f32")
    return out


def affine_act(a, y, n, alpha=1.0, beta=0.0, act=None, rowscale=None, inner=0):
    _check(lib().idiff_affine_act_f32(a.data_ptr(), y.data_ptr(), n, alpha, beta, ACT[act], _ptr(rowscale), inner,
                                      _stream()), "idiff_affine_act_f32")


def add_scale(a, b, y, n, scale):
    _check(lib().idiff_add_scale_f32(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, scale, _stream()),
           "idiff_add_scale_f32")


def fourier_embed(t, W, out, B, half):
    _check(lib().idiff_fourier_embed_f32(t.data_ptr(), W.data_ptr(), out.data_ptr(), B, half, _stream()),
           "idiff_fourier_embed_f32")


def positional_embed(t, out, B, dim, max_positions=10000.0, mode=0):
    _check(lib().idiff_positional_embed_f32(t.data_ptr(), out.data_ptr(), B, dim, max_positions, mode, _stream()),
           "idiff_positional_embed_f32")


def concat_cols(a, Ca, b, Cb, out, rows):
    _check(lib().idiff_concat_cols_f32(a.data_ptr(), Ca, b.data_ptr(), Cb, out.data_ptr(), rows, _stream()),
           "idiff_concat_cols_f32")


def nchw_to_nhwc(x, y, B, C, HW, Cpad, alpha=1.0, beta=0.0):
    _check(lib().idiff_nchw_to_nhwc_f32(x.data_ptr(), y.data_ptr(), B, C, HW, Cpad, alpha, beta, _stream()),
           "idiff_nchw_to_nhwc_f32")


def nhwc_to_nchw(x, y, B, C, HW, Cpad, rowscale=None):
    _check(lib().idiff_nhwc_to_nchw_f32(x.data_ptr(), y.data_ptr(), B, C, HW, Cpad, _ptr(rowscale), _stream()),
           "idiff_nhwc_to_nchw_f32")


def perturb(x, z, std, mean_coeff, out, rows, D):
    _check(lib().idiff_perturb_f32(x.data_ptr(), z.data_ptr(), std.data_ptr(), _ptr(mean_coeff), out.data_ptr(), rows, D,
                                   _stream()), "idiff_perturb_f32")


def perturb_randn(x, std, mean_coeff, out, rows, D, row0, seed, z_out=None):
    _check(lib().idiff_perturb_randn_f32(x.data_ptr(), std.data_ptr(), _ptr(mean_coeff), out.data_ptr(), rows, D, row0,
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(z_out), _stream()), "idiff_perturb_randn_f32")


def resample2x_nhwc(x, y, B, H, W, C, up):
    _check(lib().idiff_resample2x_nhwc_f32(x.data_ptr(), y.data_ptr(), B, H, W, C, int(up), _stream()),
           "idiff_resample2x_nhwc_f32")


# ------------------------------------------------------------------------------------------- spectrum
def spectrum_workspace_bytes(P, M, D):
    return lib().idiff_spectrum_workspace_bytes(P, M, D)


def spectrum(S, workspace=None, return_eig=False, full=False):
    """Singular values (descending, fp32, min(M, D) of them as torch.linalg.svd gives) of the column-centred matrices
    S [P, M, D] or [M, D].  ``full=True`` keeps all D values of the Gram route (the drivers gather fixed-width rows and cut
    each point's list to its own min(M, D) on the host).  ``return_eig=True`` adds the Gram eigenvalues behind them (fp64,
    ASCENDING, the same count as the singular values: ``sv[i] == sqrt(max(eig[-1 - i], 0))``)."""
    _dev(S, "scores")
    squeeze = S.ndim == 2
    if squeeze:
        S = S.unsqueeze(0)
    if S.ndim != 3:
        raise RuntimeError("scores must be [M, D] or [P, M, D]")
    P, M, D = S.shape
    need = spectrum_workspace_bytes(P, M, D)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 7) // 8, dtype=torch.float64, device=S.device)
    sv = torch.empty(P, D, dtype=torch.float32, device=S.device)
    eig = torch.empty(P, D, dtype=torch.float64, device=S.device) if return_eig else None
    _check(lib().idiff_spectrum_f32(S.data_ptr(), P, M, D, workspace.data_ptr(),
                                    workspace.numel() * workspace.element_size(), sv.data_ptr(), _ptr(eig), _stream()),
           "idiff_spectrum_f32")
    if M < D and not full:
        sv = sv[:, :M].contiguous()           # the Gram route yields D values, the last D - M of them zeros up to rounding
        if eig is not None:
            eig = eig[:, D - M:].contiguous()  # ascending: the SAME M values as sv (sv[i]^2 = eig[M - 1 - i])
    if squeeze:
        sv = sv[0]
        eig = eig[0] if eig is not None else None
    return (sv, eig) if return_eig else sv


def symtridiag_plan(D):
    """0 LDS-resident, 1 two-stage + systolic chase, 2 two-stage + wavefront chase, 3 one-stage (include/idiff_hip.h)."""
    return lib().idiff_symtridiag_plan(int(D))


# The eigensolver never returns a silently wrong spectrum: a band-reduction residual above tolerance or a stalled systolic
# chase (its workgroups wait on each other; a device that cannot keep them all resident stalls it) poisons the outputs
# with NaN.  These are the slower forms that do not share the failure: tried in turn, in the same process.
_FALLBACKS = (("IDIFF_CHASE_WAVEFRONT", "bulge chasing one launch per wavefront"),
              ("IDIFF_TRIDIAG_ONESTAGE", "one-stage Householder sweep"))


def resolve_failed_spectrum(S, full=False, log=None):
    """``spectrum(S)`` came back with NaN: solve the same matrices again with the fallback forms of the eigensolver, on
    the current stream.  Synchronises (the rare path).  Raises if S itself is non-finite or every form fails."""
    import warnings
    if not bool(torch.isfinite(S).all()):
        raise RuntimeError("the score matrix holds non-finite values (NaN / inf score vectors): no spectrum exists")
    for name, what in _FALLBACKS:
        with thread_option(name, 1):                       # this thread's launches only
            sv = spectrum(S, full=full)
        if not bool(torch.isnan(sv).any()):
            msg = f"id-diff_amd: the two-stage eigensolver reported a failure; spectrum re-solved with {what} ({name})"
            (log or warnings.warn)(msg)
            return sv
    raise RuntimeError("the spectrum kernels reported a failure (NaN singular values) and so did the wavefront chase and the "
                       "one-stage sweep")


# ---- the stages of the spectrum, for the row-sharded single-point pipeline (dim_reduction.row_sharded_spectrum)
def column_sums(S):
    """fp64 column sums [D] of S [M, D] (two-stage, deterministic: idiff_colmean_f64 times M)."""
    _dev(S, "scores")
    M, D = S.shape
    if M == 0:
        return torch.zeros(D, dtype=torch.float64, device=S.device)
    mean = torch.empty(D, dtype=torch.float64, device=S.device)
    scratch = torch.empty(32 * D, dtype=torch.float64, device=S.device)
    _check(lib().idiff_colmean_f64(S.data_ptr(), 1, M, D, mean.data_ptr(), scratch.data_ptr(), _stream()), "idiff_colmean_f64")
    return mean * M


def centered_gram(S, mean):
    """fp64 Gram [D, D] of the rows of S [M, D] after subtracting ``mean`` [D] (fp64): sum_i (s_i - mean)(s_i - mean)^T."""
    _dev(S, "scores"); _dev(mean, "mean", dtype=torch.float64)
    M, D = S.shape
    if M == 0:
        return torch.zeros(D, D, dtype=torch.float64, device=S.device)
    G = torch.empty(D, D, dtype=torch.float64, device=S.device)
    _check(lib().idiff_centered_gram_f64(S.data_ptr(), mean.data_ptr(), 1, M, D, G.data_ptr(), _stream()), "idiff_centered_gram_f64")
    return G


def centered_gram_rows(S, mean, G, row0, row1):
    """Upper-triangle rows [row0, row1) of the centred Gram into the (pre-zeroed) [D, D] fp64 matrix G."""
    _dev(S, "scores"); _dev(mean, "mean", dtype=torch.float64); _dev(G, "G", dtype=torch.float64)
    M, D = S.shape
    if M == 0:
        return G
    _check(lib().idiff_centered_gram_rows_f64(S.data_ptr(), mean.data_ptr(), M, D, row0, row1, G.data_ptr(), _stream()),
           "idiff_centered_gram_rows_f64")
    return G


def symmetrize_upper(G):
    _dev(G, "G", dtype=torch.float64)
    _check(lib().idiff_symmetrize_upper_f64(G.data_ptr(), G.shape[0], _stream()), "idiff_symmetrize_upper_f64")
    return G


def sym_band(G, dense=True):
    """Stage 1 of the two-stage eigensolver alone (G is overwritten).  ``dense=True``: the dense symmetric band matrix
    [D, D] (half-width 32) similar to G, for the parity tests; ``dense=False``: the compact band [D, ld] with
    band[j, k] = B[j + k, j], for timing."""
    _dev(G, "G", dtype=torch.float64)
    D = G.shape[0]
    scratch = torch.zeros(lib().idiff_symtridiag_scratch_doubles(D), dtype=torch.float64, device=G.device)
    _check(lib().idiff_symband_f64(G.data_ptr(), D, scratch.data_ptr(), _stream()), "idiff_symband_f64")
    ld = lib().idiff_symband_ld()
    band = scratch[:D * ld].view(D, ld)                       # band[j, k] = B[j + k, j]
    if not dense:
        return band
    B = torch.zeros(D, D, dtype=torch.float64, device=G.device)
    j = torch.arange(D, device=G.device)
    for k in range(ld):
        n = D - k
        if n <= 0:
            break
        B[j[:n] + k, j[:n]] = band[:n, k]
        B[j[:n], j[:n] + k] = band[:n, k]
    return B


def sym_eigvals(G):
    """Eigenvalues (ascending, fp64) of a symmetric fp64 matrix [D, D]; G is overwritten (Householder + Sturm bisection)."""
    _dev(G, "G", dtype=torch.float64)
    D = G.shape[0]
    diag = torch.empty(D, dtype=torch.float64, device=G.device)
    offd = torch.empty(D, dtype=torch.float64, device=G.device)
    scratch = torch.empty(max(1, lib().idiff_symtridiag_scratch_doubles(D)), dtype=torch.float64, device=G.device)
    eig = torch.empty(D, dtype=torch.float64, device=G.device)
    _check(lib().idiff_symtridiag_f64(G.data_ptr(), 1, D, diag.data_ptr(), offd.data_ptr(), scratch.data_ptr(), _stream()),
           "idiff_symtridiag_f64")
    _check(lib().idiff_tridiag_eigvals_f64(diag.data_ptr(), offd.data_ptr(), 1, D, eig.data_ptr(), _stream()), "idiff_tridiag_eigvals_f64")
    return eig
