"""Winograd F(2x2,3x3) vs implicit-GEMM conv on the 3x3 shapes of the benchmark forward (2240-row launch set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
shapes = [(32, 128, 128), (16, 256, 256), (32, 256, 128), (32, 256, 256), (16, 512, 256), (8, 256, 256), (32, 384, 128),
          (8, 512, 256), (4, 256, 256), (4, 512, 256)]
def timeit(fn, reps=4):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for H, Cin, Cout in shapes:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    bias = torch.randn(Cout, device=dev)
    u = _lib.winograd_pack(w, Cin, Cout)
    o1 = torch.empty(B, H * H, Cout, device=dev); o2 = torch.empty_like(o1)
    ep = _lib.make_epilogue(bias=bias)
    td = timeit(lambda: _lib.conv2d_nhwc(x, w, o1, B, H, H, Cin, Cout, 3, 3, 1, 1, epilogue=ep))
    tw = timeit(lambda: _lib.conv2d_winograd(x, u, o2, B, H, H, Cin, Cout, epilogue=ep))
    err = float((o1.double() - o2.double()).norm() / o1.double().norm())
    fl = 2.0 * B * H * H * Cin * Cout * 9
    extra = ""
    for ng in (1, 2):
        if (Cout // 64) % ng == 0 and Cout // 64 > ng:
            _lib.set_option("IDIFF_WINO_NGROUP", ng)
            tg = timeit(lambda: _lib.conv2d_winograd(x, u, o2, B, H, H, Cin, Cout, epilogue=ep))
            _lib.set_option("IDIFF_WINO_NGROUP", 0)
            extra += f" | ngroup{ng} {tg:7.3f} ms"
    say(f"{H:2d}x{H:<2d} {Cin:3d}->{Cout:3d}: direct {td:7.3f} ms {fl/td/1e9:6.1f} TF | winograd {tw:7.3f} ms {fl/tw/1e9:6.1f} TF-equiv "
        f"({fl/2.25/tw/1e9:5.1f} TF executed) | x{td/tw:4.2f} | rel diff {err:.2e}" + extra)
