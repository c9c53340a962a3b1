"""winograd43_kernel: output-channel tiles scheduled together (IDIFF_WINO_NGROUP) -- time of one forward's F(4x4) convs per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B = 2240
shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
          (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (16, 128, 128, 2), (16, 128, 256, 1)]
data = []
for H, Cin, Cout, calls in shapes:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    o = torch.empty(B, H * H, Cout, device=dev)
    ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rows_per_group=H * H)
    data.append((H, Cin, Cout, calls, x, _lib.winograd43_pack(w, Cin, Cout), o, ep))
for rnd in range(2):
    for ng in (0, 1, 2, 4):
        tot = 0.0
        per = []
        with _lib.thread_option("IDIFF_WINO_NGROUP", ng):
            for H, Cin, Cout, calls, x, u, o, ep in data:
                fn = lambda: _lib.conv2d_winograd43(x, u, o, B, H, H, Cin, Cout, epilogue=ep)
                fn(); fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                tot += ms * calls; per.append(f"{ms*1e3:.0f}")
        print(f"ngroup {ng} (0 = default rule): {tot:7.1f} ms per forward   per shape us: {' '.join(per)}", flush=True)
