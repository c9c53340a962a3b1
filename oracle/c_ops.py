"""ctypes front-end of oracle/native_ops.c (plain-C restatement of the two native ops).  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ops.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def upfirdn2d(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """x: [major, H, W, minor] float32 numpy; returns [major, out_h, out_w, minor]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    out = np.empty((major, out_h, out_w, minor), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().oracle_upfirdn2d_f32(x.ctypes.data_as(fp), k.ctypes.data_as(fp), out.ctypes.data_as(fp), major, in_h, in_w,
                                    minor, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    assert rc == 0
    return out


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    fp = ctypes.POINTER(ctypes.c_float)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    bptr = np.ascontiguousarray(b, dtype=np.float32).ctypes.data_as(fp) if b is not None else None
    rptr = np.ascontiguousarray(ref, dtype=np.float32).ctypes.data_as(fp) if ref is not None else None
    f = lib().oracle_fused_bias_act_f32
    f.argtypes = [fp, fp, fp, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                  ctypes.c_float]
    rc = f(x.ctypes.data_as(fp), bptr, rptr, out.ctypes.data_as(fp), x.size, step_b, len(b) if b is not None else 1, act,
           grad, alpha, scale)
    assert rc == 0
    return out
