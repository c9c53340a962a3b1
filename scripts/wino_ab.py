"""fp32 vs split-precision Winograd kernel on the NCSN++ layer shapes, both in ONE process (boxes differ by ~10 %)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
          (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (4, 256, 256, 19), (4, 512, 256, 5), (16, 128, 128, 2), (16, 128, 256, 1)]
tot = [0.0, 0.0]
for H, Cin, Cout, calls in shapes:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    o = torch.empty(B, H * H, Cout, device=dev)
    ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), act="silu", rows_per_group=H * H)
    ms = []
    for split in (False, True):
        prev = _lib.set_option("IDIFF_WINO_SPLIT", 1)
        u = _lib.winograd_pack(w, Cin, Cout, split=bool(split))
        _lib.set_option("IDIFF_WINO_SPLIT", int(prev))
        for _ in range(2):
            _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=ep, split=bool(split))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=ep, split=bool(split))
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 5)
    tot[0] += ms[0] * calls; tot[1] += ms[1] * calls
    print(f"{H:3d}x{H:<3d} {Cin:4d}->{Cout:<4d} x{calls:<3d} fp32 {ms[0]*1e3:8.1f} us  split {ms[1]*1e3:8.1f} us  ratio {ms[0]/ms[1]:.3f}", flush=True)
print(f"per forward: fp32 {tot[0]:.1f} ms, split {tot[1]:.1f} ms, ratio {tot[0]/tot[1]:.3f}; best of both per layer would need the table above")
