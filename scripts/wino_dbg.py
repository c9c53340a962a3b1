"""Timing experiments on the Winograd kernel (IDIFF_WINO_DBG bit 0: no global loads in the K loop, bit 1: no transform/LDS writes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B = 2240
def timeit(fn, reps=4):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for H, Cin, Cout in [(32, 128, 128), (16, 512, 256), (16, 256, 256)]:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    u = _lib.winograd_pack(w, Cin, Cout)
    o = torch.empty(B, H * H, Cout, device=dev)
    ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev))
    fl = 2.0 * B * H * H * Cin * Cout * 9 / 2.25
    line = f"{H}x{H} {Cin}->{Cout}:"
    for dbg in (0,):
        os.environ["IDIFF_WINO_DBG"] = str(dbg)
        t = timeit(lambda: _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=ep))
        line += f"  dbg{dbg} {t:6.3f} ms ({fl/t/1e9:5.1f} TF)"
    print(line, flush=True)
