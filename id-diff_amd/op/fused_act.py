"""``fused_leaky_relu`` / ``FusedLeakyReLU`` with the reference's signatures on the gfx950 kernel.

Mirrors /root/reference/op/fused_act.py:20-97.  This is the reference's GPU branch, so ``negative_slope`` is honoured
(its CPU branch hard-codes 0.2, fused_act.py:91 -- restated in oracle/ops.py).  Like the reference, the backward pass
re-uses the native op: ``fused_bias_act(grad, empty, out, act=3, grad=1)`` gates the incoming gradient by the sign of
the saved OUTPUT, and the bias gradient is its sum over every axis but the channel axis.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Native entry point, op/fused_bias_act.cpp:11-17."""
    return _lib.fused_bias_act(input.contiguous(), bias, refer, int(act), int(grad), float(alpha), float(scale))


class _FusedLeakyReLUGrad(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        grad_input = fused_bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
        dims = [0] + list(range(2, grad_input.ndim))
        return grad_input, grad_input.sum(dims).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        return fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, negative_slope, scale), None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = _FusedLeakyReLUGrad.apply(grad_output, out, *ctx.cfg)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    _lib.op_suffix(input, "fused_leaky_relu")
    _lib._dev(input, "input", dtype=input.dtype, contiguous=False)
    if not (torch.is_grad_enabled() and (input.requires_grad or (bias is not None and bias.requires_grad))):
        # nothing to differentiate (inference under no_grad, or plain tensors): the autograd.Function wrapper was most of
        # the 13 us host floor of this op on small tensors
        return fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
