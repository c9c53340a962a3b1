"""Oracle restatement of the reference's two native ops (CPU branches).

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F


def upfirdn2d_ref(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Upsample (zero insertion) -> pad/crop -> 2-D FIR (true convolution) -> decimate.

    Follows /root/reference/op/upfirdn2d.py:159-200 (``upfirdn2d_native``) and
    the index algebra of op/upfirdn2d_kernel.cu:49-105; x is ``[N, C, H, W]``.
    out_h = (H*up_y + pad_y0 + pad_y1 - kh) // down_y + 1 (upfirdn2d.py:103-104).
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    planes = x.reshape(n * c, 1, h, w)
    # zero insertion: sample (i, j) lands at (i*up_y, j*up_x); trailing zeros kept
    stuffed = planes.new_zeros(n * c, 1, h * up_y, w * up_x)
    stuffed[:, :, ::up_y, ::up_x] = planes
    # positive pads add zeros, negative pads crop (upfirdn2d.py:172-180)
    stuffed = F.pad(stuffed, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    hh, ww = stuffed.shape[-2:]
    stuffed = stuffed[:, :, max(-pad_y0, 0):hh - max(-pad_y1, 0), max(-pad_x0, 0):ww - max(-pad_x1, 0)]
    # F.conv2d is a correlation; the op is a convolution -> flip the taps (upfirdn2d.py:186)
    taps = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(x.dtype)
    full = F.conv2d(stuffed, taps)
    out = full[:, :, ::down_y, ::down_x]
    out_h = (h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    return out.reshape(n, c, out_h, out_w)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Public signature of /root/reference/op/upfirdn2d.py:145-156 (CPU branch)."""
    return upfirdn2d_ref(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def fused_bias_act_ref(x, bias=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """Elementwise ``act(x + b[channel]) * scale`` with the CUDA kernel's modes.

    Follows /root/reference/op/fused_bias_act_kernel.cu:18-49: bias is indexed by
    ``(i / step_b) % size_b`` with step_b = prod(dims[2:]) (:69-71), i.e. dim 1
    is the channel; ``act*10+grad``: 10/11 -> x, 12 -> 0, 30 -> lrelu(x),
    31 -> ref>0 ? x : x*alpha, 32 -> 0.
    """
    y = x
    if bias is not None and bias.numel() > 0:
        shape = [1, bias.numel()] + [1] * (x.ndim - 2)
        y = y + bias.reshape(shape)
    if act == 1:
        y = y if grad < 2 else torch.zeros_like(y)
    elif act == 3:
        if grad == 0:
            y = torch.where(y > 0, y, y * alpha)
        elif grad == 1:
            y = torch.where(ref > 0, y, y * alpha)
        else:
            y = torch.zeros_like(y)
    else:
        raise ValueError(f"unsupported act {act}")
    return y * scale


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """CPU branch of /root/reference/op/fused_act.py:86-94.

    Quirk kept on purpose: that branch hard-codes slope 0.2 and ignores
    ``negative_slope`` (fused_act.py:91); the CUDA branch honours it.
    """
    return fused_bias_act_ref(x, bias, None, act=3, grad=0, alpha=0.2, scale=scale)


def fused_bias_act_native(x, bias=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """The CUDA kernel's arithmetic for ANY dtype of its dispatch (float16 / float32 / float64), in numpy.

    /root/reference/op/fused_bias_act_kernel.cu:18-49 computes in ``scalar_t``; the op's ``float alpha, float scale``
    arguments (fused_bias_act.cpp:11-12) are converted to scalar_t at the launch (:80-93), so alpha and scale are
    first rounded to fp32 and then to the tensor's dtype.  Every operation rounds to the dtype (c10::Half's operators
    widen to fp32, operate, round back -- which is what numpy's float16 arithmetic does too):
    x = r(x + b);  y = x > 0 ? x : r(x * alpha)  (``ref > 0`` for grad = 1);  out = r(y * scale).
    Takes and returns numpy arrays (or torch CPU tensors, converted)."""
    import numpy as np
    as_np = lambda t: None if t is None else (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t))
    x, bias, ref = as_np(x), as_np(bias), as_np(ref)
    dt = x.dtype
    a, s = np.float32(alpha).astype(dt), np.float32(scale).astype(dt)
    y = x
    if bias is not None and bias.size > 0:
        y = (y + bias.astype(dt).reshape([1, bias.size] + [1] * (x.ndim - 2))).astype(dt)
    mode = act * 10 + grad
    if mode in (10, 11):
        pass
    elif mode == 30:
        y = np.where(y > 0, y, (y * a).astype(dt))
    elif mode == 31:
        y = np.where(ref > 0, y, (y * a).astype(dt))
    elif mode in (12, 32):
        y = np.zeros_like(y)
    else:
        raise ValueError(f"unsupported act/grad {act}/{grad}")
    return (y * s).astype(dt)
