"""``upfirdn2d`` with the reference's Python signature, executed by the gfx950 kernel.

Mirrors /root/reference/op/upfirdn2d.py:145-156 (public function), :88-124 (the CUDA branch: view as [N*C, H, W, 1],
one native call, view back) and :19-85,126-142 (first and second derivative): every derivative of upfirdn2d is again
an upfirdn2d -- the input gradient runs the op with up and down swapped, the flipped kernel and the complementary
pads -- so forward, backward and double-backward are all the same HIP kernel.
"""
import torch
from torch.autograd import Function

from .. import _lib


def _launch(x4, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x4: [N, C, H, W] contiguous CUDA fp32 / fp16 / fp64 -> [N, C, out_h, out_w] of the same dtype."""
    n, c, in_h, in_w = x4.shape
    kh, kw = kernel.shape
    out_h = _lib.upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    out_w = _lib.upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError("upfirdn2d: empty output")
    out = torch.empty((n, c, out_h, out_w), device=x4.device, dtype=x4.dtype)
    _lib.upfirdn2d_raw(x4, kernel, out, n * c, in_h, in_w, 1, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    return out


class _UpFirDn2dGrad(Function):
    """d/d(input): upfirdn2d(grad, flip(k), up=down, down=up, complementary pads); its own derivative is the
    forward op again (reference: UpFirDn2dBackward, op/upfirdn2d.py:19-85)."""

    @staticmethod
    def forward(ctx, grad_output, kernel, flipped, up, down, pad, g_pad, in_size):
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad)
        gx0, gx1, gy0, gy1 = g_pad
        grad_input = _launch(grad_output.contiguous(), flipped, down[0], down[1], up[0], up[1], gx0, gx1, gy0, gy1)
        # the derivative op may produce a border row/col beyond the input when the forward discarded samples
        return grad_input[:, :, :in_size[2], :in_size[3]].contiguous() if grad_input.shape[2:] != in_size[2:] else grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        up, down, pad = ctx.cfg
        out = _launch(gradgrad_input.contiguous(), kernel, up[0], up[1], down[0], down[1], *pad)
        return out, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kh, kw = kernel.shape
        n, c, in_h, in_w = input.shape
        out = _launch(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
        out_h, out_w = out.shape[2:]
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        # pads of the derivative op, op/upfirdn2d.py:111-116
        ctx.g_pad = (kw - pad_x0 - 1, in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kh - pad_y0 - 1, in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        ctx.cfg = (up, down, pad, tuple(input.shape))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, flipped = ctx.saved_tensors
        up, down, pad, in_size = ctx.cfg
        grad_input = _UpFirDn2dGrad.apply(grad_output, kernel, flipped, up, down, pad, ctx.g_pad, in_size)
        return grad_input, None, None, None, None


def upfirdn2d_xy(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Separate x/y factors, as the native entry point takes them (op/upfirdn2d.cpp:12-19)."""
    _lib.op_suffix(input, "upfirdn2d")
    _lib._dev(input, "input", dtype=input.dtype, contiguous=False)
    if input.ndim != 4:
        raise RuntimeError(f"upfirdn2d: expected [N, C, H, W], got {tuple(input.shape)}")
    if kernel.ndim != 2:
        raise RuntimeError(f"upfirdn2d: expected a 2-D FIR kernel, got {tuple(kernel.shape)}")
    # the reference reads the FIR kernel through data_ptr<scalar_t>() of the INPUT's dtype (op/upfirdn2d_kernel.cu:315-317)
    kernel = _lib._dev(kernel.detach().to(device=input.device, dtype=input.dtype).contiguous(), "kernel", dtype=input.dtype)
    if not (torch.is_grad_enabled() and input.requires_grad):
        # nothing to differentiate: one native call, no autograd.Function around it (host time matters on small tensors)
        return _launch(input.contiguous(), kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    return UpFirDn2d.apply(input.contiguous(), kernel, (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1))


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same arguments as op/upfirdn2d.py:145: one factor and one pad pair for both axes."""
    return upfirdn2d_xy(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
