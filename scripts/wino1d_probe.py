"""The row-wise F(4, 3) pair kernel (csrc/wino1d.hip) beside the 2-D pair kernel on the 3x3 convolutions of one nf = 128 NCSN++ forward at B = 2240:
per shape the device time of both and their relative error against each other; the sum over the forward's calls.
    python scripts/wino1d_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
          (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (16, 128, 128, 2), (16, 128, 256, 1), (4, 256, 256, 17), (4, 512, 256, 7)]
def timed(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot2, tot1 = 0.0, 0.0
for H, Cin, Cout, calls in shapes:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    o2 = torch.empty(B, H * H, Cout, device=dev); o1 = torch.empty_like(o2)
    ns = _lib.conv2d_winograd43_colstats_split(B, H, H, Cin, Cout)
    cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
    ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rowbias=torch.randn(B, Cout, device=dev), rows_per_group=H * H, colstats=cs)
    u2 = _lib.winograd43_pack(w, Cin, Cout, pairs=True)
    u1 = _lib.wino1d_pack(w, Cin, Cout)
    t2 = timed(lambda: _lib.conv2d_winograd43(x, u2, o2, B, H, H, Cin, Cout, epilogue=ep, pairs=True))
    t1 = timed(lambda: _lib.conv2d_wino1d(x, u1, o1, B, H, H, Cin, Cout, epilogue=ep))
    err = float((o1.double() - o2.double()).norm() / o2.double().norm())
    tot2 += t2 * calls; tot1 += t1 * calls
    print(f"{H:2d}x{H:<2d} {Cin:3d}->{Cout:3d} x{calls:2d}: F(4x4,3x3) {t2*1e3:7.0f} us   F(4,3) rows {t1*1e3:7.0f} us   ratio {t2/t1:5.2f}   rel diff {err:.2e}", flush=True)
print(f"per forward: F(4x4,3x3) {tot2:.1f} ms, F(4,3) rows {tot1:.1f} ms")
