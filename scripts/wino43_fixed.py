"""Per-workgroup fixed cost and per-step cost of winograd43_kernel: the same map with Cin = 8 .. 512 (1 .. 64 K steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B = 2240
for H, Cout in ((32, 128), (16, 256)):
    rows = []
    for Cin in (8, 16, 32, 64, 128, 256, 512):
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        o = torch.empty(B, H * H, Cout, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), act="silu", rows_per_group=H * H)
        u4 = _lib.winograd43_pack(w, Cin, Cout)
        fn = lambda: _lib.conv2d_winograd43(x, u4, o, B, H, H, Cin, Cout, epilogue=ep)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        wgs = ((B * (H // 4) ** 2 + 31) // 32) * (Cout // 64)
        per_wg_us = ms * 1e3 / (wgs / 256.0)
        rows.append((Cin // 8, per_wg_us))
        print(f"{H}x{H} Cin {Cin:4d} -> {Cout}: {ms*1e3:8.1f} us, {wgs} workgroups = {wgs/256:.1f} per CU -> {per_wg_us:6.1f} us per workgroup slot", flush=True)
    (s0, t0), (s1, t1) = rows[0], rows[-1]
    slope = (t1 - t0) / (s1 - s0)
    print(f"  per step {slope:.2f} us (MFMA time 1.92 us at 2.4 GHz), fixed {t0 - slope * s0:.1f} us per workgroup", flush=True)
