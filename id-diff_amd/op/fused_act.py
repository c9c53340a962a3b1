"""``fused_leaky_relu`` / ``FusedLeakyReLU`` with the reference's signatures on the gfx950 kernel.

Mirrors /root/reference/op/fused_act.py:74-97.  This is the reference's GPU
branch, so ``negative_slope`` is honoured (its CPU branch hard-codes 0.2,
fused_act.py:91 -- restated in oracle/ops.py).  Forward only.
"""
import torch
from torch import nn

from .. import _lib


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Native entry point, op/fused_bias_act.cpp:11-17."""
    return _lib.fused_bias_act(input, bias, refer, int(act), int(grad), float(alpha), float(scale))


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
