"""Cost of each fused epilogue term of the Winograd conv (isolated launches, [2240,32,32,128]->128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B, H, Cin, Cout = 2240, 32, 128, 128
x = torch.randn(B, H * H, Cin, device=dev)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
u = _lib.winograd_pack(w, Cin, Cout)
o = torch.empty(B, H * H, Cout, device=dev)
bias = torch.randn(Cout, device=dev); temb = torch.randn(B, Cout, device=dev); res = torch.randn(B, H * H, Cout, device=dev)
ns = _lib.conv2d_winograd_colstats_split(B, H, H, Cin, Cout)
cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
def timeit(ep, reps=5):
    f = lambda: _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=ep)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, kw in [("bias", dict(bias=bias)), ("+temb", dict(bias=bias, rowbias=temb, rows_per_group=H * H)),
                 ("+residual,scale", dict(bias=bias, residual=res, out_scale=0.7071)),
                 ("+colstats", dict(bias=bias, rows_per_group=H * H, colstats=cs)),
                 ("temb+colstats (Conv_0)", dict(bias=bias, rowbias=temb, rows_per_group=H * H, colstats=cs)),
                 ("residual+colstats (Conv_1)", dict(bias=bias, residual=res, out_scale=0.7071, rows_per_group=H * H, colstats=cs))]:
    print(f"{name:28s} {timeit(_lib.make_epilogue(**kw)):.3f} ms", flush=True)
