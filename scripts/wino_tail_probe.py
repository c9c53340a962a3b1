"""Winograd conv with the epilogues the residual blocks actually use (time + agreement with the implicit GEMM):
   plain bias | bias + per-sample bias + column statistics (conv 1) | bias + residual + 1/sqrt(2) + statistics (conv 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
shapes = [(32, 128, 128), (16, 256, 256), (32, 256, 128), (16, 512, 256), (8, 256, 256), (4, 256, 256)]
def timeit(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
torch.manual_seed(0)
for H, Cin, Cout in shapes:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    bias = torch.randn(Cout, device=dev); temb = torch.randn(B, Cout, device=dev); res = torch.randn(B, H * H, Cout, device=dev)
    u = _lib.winograd_pack(w, Cin, Cout)
    o1 = torch.empty(B, H * H, Cout, device=dev); o2 = torch.empty_like(o1)
    split = _lib.conv2d_winograd_colstats_split(B, H, H, Cin, Cout)
    row = f"{H:2d}x{H:<2d} {Cin:3d}->{Cout:3d}:"
    fl = 2.0 * B * H * H * Cin * Cout * 9 / 2.25
    for name, kw in [("bias", dict(bias=bias)),
                     ("conv1", dict(bias=bias, rowbias=temb, rows_per_group=H * H, stats=True)),
                     ("conv2", dict(bias=bias, residual=res, out_scale=0.70710678, stats=True)),
                     ("silu", dict(bias=bias, act="silu"))]:
        stats = kw.pop("stats", False) and split > 0
        cs = torch.zeros(B * split, Cout, 2, device=dev, dtype=torch.float64) if stats else None
        ep_w = _lib.make_epilogue(colstats=cs, **kw)
        ep_d = _lib.make_epilogue(**kw)
        _lib.conv2d_nhwc(x, w, o1, B, H, H, Cin, Cout, 3, 3, 1, 1, epilogue=ep_d)
        tw = timeit(lambda: _lib.conv2d_winograd(x, u, o2, B, H, H, Cin, Cout, epilogue=ep_w))
        err = float((o1.double() - o2.double()).norm() / o1.double().norm())
        serr = 0.0
        if stats:
            tot = cs.view(B, split, Cout, 2).sum(1)
            ref1 = o2.double().sum(1); ref2 = (o2.double() ** 2).sum(1)
            serr = max(float((tot[..., 0] - ref1).abs().max() / ref1.abs().max()), float((tot[..., 1] - ref2).abs().max() / ref2.abs().max()))
        row += f" | {name} {tw:6.3f} ms {fl/tw/1e9:5.1f} TF d={err:.1e}" + (f" s={serr:.0e}" if stats else "")
    say(row)
