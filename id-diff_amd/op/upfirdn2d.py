"""``upfirdn2d`` with the reference's Python signature, executed by the gfx950 kernel.

Mirrors /root/reference/op/upfirdn2d.py:145-156 (public function) and :88-124
(the CUDA branch: view as [N*C, H, W, 1], one native call, view back).  Forward
only -- the manifold_dimension path never differentiates through the op.
"""
import torch

from .. import _lib


def upfirdn2d_xy(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Separate x/y factors, as the native entry point takes them (op/upfirdn2d.cpp:12-19)."""
    _lib._dev(input, "input")
    if input.ndim != 4:
        raise RuntimeError(f"upfirdn2d: expected [N, C, H, W], got {tuple(input.shape)}")
    if kernel.ndim != 2:
        raise RuntimeError(f"upfirdn2d: expected a 2-D FIR kernel, got {tuple(kernel.shape)}")
    kernel = _lib._dev(kernel.to(device=input.device, dtype=torch.float32).contiguous(), "kernel")
    n, c, in_h, in_w = input.shape
    kh, kw = kernel.shape
    out_h = _lib.upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    out_w = _lib.upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError("upfirdn2d: empty output")
    out = torch.empty((n, c, out_h, out_w), device=input.device, dtype=torch.float32)
    _lib.upfirdn2d_raw(input, kernel, out, n * c, in_h, in_w, 1, up_x, up_y, down_x, down_y,
                       pad_x0, pad_x1, pad_y0, pad_y1)
    return out


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same arguments as op/upfirdn2d.py:145: one factor and one pad pair for both axes."""
    return upfirdn2d_xy(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
