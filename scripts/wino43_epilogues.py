"""winograd43_kernel at the network's layer shape 32x32 128 -> 128 (B = 2240) with the epilogues the network actually uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B, H, Cin, Cout = 2240, 32, 128, 128
x = torch.randn(B, H * H, Cin, device=dev)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
o = torch.empty(B, H * H, Cout, device=dev)
u4 = _lib.winograd43_pack(w, Cin, Cout)
bias = torch.randn(Cout, device=dev)
res = torch.randn(B, H * H, Cout, device=dev)
temb = torch.randn(B, Cout, device=dev)
ns = _lib.conv2d_winograd43_colstats_split(B, H, H, Cin, Cout)
cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
eps = {"none": None, "bias": _lib.make_epilogue(bias=bias, rows_per_group=H * H),
       "bias+colstats": _lib.make_epilogue(bias=bias, rows_per_group=H * H, colstats=cs),
       "bias+temb+colstats (Conv_0 of a block)": _lib.make_epilogue(bias=bias, rowbias=temb, rows_per_group=H * H, colstats=cs),
       "bias+residual+scale+colstats (Conv_1)": _lib.make_epilogue(bias=bias, residual=res, out_scale=0.7071, rows_per_group=H * H, colstats=cs),
       "bias+residual+scale": _lib.make_epilogue(bias=bias, residual=res, out_scale=0.7071, rows_per_group=H * H)}
for rnd in range(2):
    for name, ep in eps.items():
        fn = lambda: _lib.conv2d_winograd43(x, u4, o, B, H, H, Cin, Cout, epilogue=ep)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:42s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
