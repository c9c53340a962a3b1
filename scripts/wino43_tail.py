"""Tail cost of winograd43_kernel by epilogue content: Cin = 8 (one K step), 32x32 -> 128, B = 2240."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B, H, Cin, Cout = 2240, 32, 8, 128
x = torch.randn(B, H * H, Cin, device=dev)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
o = torch.empty(B, H * H, Cout, device=dev)
u4 = _lib.winograd43_pack(w, Cin, Cout)
bias = torch.randn(Cout, device=dev)
res = torch.randn(B, H * H, Cout, device=dev)
temb = torch.randn(B, Cout, device=dev)
ns = _lib.conv2d_winograd43_colstats_split(B, H, H, Cin, Cout)
cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
eps = {"none": None, "bias": _lib.make_epilogue(bias=bias), "bias+silu": _lib.make_epilogue(bias=bias, act="silu", rows_per_group=H * H),
       "bias+temb+silu+stats": _lib.make_epilogue(bias=bias, rowbias=temb, act="silu", rows_per_group=H * H, colstats=cs),
       "bias+residual+scale": _lib.make_epilogue(bias=bias, residual=res, out_scale=0.7071, rows_per_group=H * H)}
for name, ep in eps.items():
    fn = lambda: _lib.conv2d_winograd43(x, u4, o, B, H, H, Cin, Cout, epilogue=ep)
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:24s} {ms*1e3:8.1f} us = {ms*1e3/35:.1f} us per workgroup slot", flush=True)
