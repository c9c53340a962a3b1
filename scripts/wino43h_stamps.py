"""Where a workgroup of winograd43h_kernel spends its life (diagnostic build -DIDIFF_W43H_STAMP: s_memrealtime, 100 MHz, of lane 0 at
kernel start, loop start, loop end, and at the tail's phases), with the epilogue of a ResnetBlock's first convolution (bias, per-sample
bias, column sums) and of its second (residual, scale, column sums).  Run on the GPU box: python scripts/wino43h_stamps.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "wino43h_stamps"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    dev = "cuda"
    B = 2240
    holder = torch.zeros(16 * 20000, device=dev, dtype=torch.int64)
    os.environ["IDIFF_W43H_STAMP_PTR"] = hex(holder.data_ptr())
    import id_diff_amd
    from id_diff_amd import _lib
    for (H, Cin, Cout) in ((32, 128, 128), (16, 256, 256), (16, 512, 256), (8, 256, 256)):
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        u = _lib.winograd43_pack(w, Cin, Cout, pairs=True)
        out = torch.empty(B, H * H, Cout, device=dev)
        nwg = ((B * (H // 4) ** 2 + 31) // 32) * (Cout // 64)
        ns = _lib.conv2d_winograd43_colstats_split(B, H, H, Cin, Cout)
        cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
        res = torch.randn(B, H * H, Cout, device=dev)
        eps = {"plain (bias)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev)),
               "conv 0 (bias, time-embedding bias, column sums)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rowbias=torch.randn(B, Cout, device=dev), rows_per_group=H * H, colstats=cs),
               "conv 1 (bias, residual, scale, column sums)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev), residual=res, out_scale=0.7071, rows_per_group=H * H, colstats=cs)}
        for name, ep in eps.items():
            for _ in range(3):
                holder.zero_()
                _lib.conv2d_winograd43(x, u, out, B, H, H, Cin, Cout, epilogue=ep, pairs=True)
            torch.cuda.synchronize()
            t = holder[:8 * nwg].view(nwg, 8).cpu().double() * 0.01     # microseconds
            ph = [t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 6] - t[:, 5], t[:, 7] - t[:, 6]]
            m = [float(v.median()) for v in ph]
            steps = Cin // 16
            print(f"{H}x{H} {Cin}->{Cout} [{name}]: {nwg} workgroups; median us per workgroup: prologue {m[0]:.1f}, K loop {m[1]:.1f} ({steps} steps: {m[1] / steps:.2f} each), "
                  f"tail {sum(m[2:]):.1f} = exchange 0 {m[2]:.1f} + outputs 0 {m[3]:.1f} + exchange 1 {m[4]:.1f} + outputs 1 {m[5]:.1f} + column sums {m[6]:.1f};  "
                  f"kernel span {float(t[:, 7].max() - t[:, 0].min()):.0f} us", flush=True)
            ph = holder[8 * nwg:16 * nwg].view(nwg, 2, 4).cpu().double()       # [workgroup][wave 0 / wave 4][stage ticks, barrier ticks, -, -]
            loop_us = float((t[:, 2] - t[:, 1]).median())
            clk = float(ph[:, 0, 0].median() + 1) * 0                          # (ticks are shader clocks: shown as a share of the loop below)
            pro = holder[8 * nwg:16 * nwg].view(nwg, 2, 4).cpu()[:, 0]              # wave 0's prologue split, 100 MHz ticks
            a, b, c, d = (pro[:, 2] >> 32).double() * 0.01, ((pro[:, 2] >> 16) & 0xFFFF).double() * 0.01, (pro[:, 2] & 0xFFFF).double() * 0.01, pro[:, 3].double() * 0.01
            print(f"      prologue (wave 0, median us): set-up {float(a.median()):.1f}, first operands requested -> arrived {float(b.median()):.1f}, first stage {float(c.median()):.1f}, barrier {float(d.median()):.1f}", flush=True)
            for wv, nm in ((0, "wave 0"), (1, "wave 4 (same SIMD)")):
                st_t, wt_t = float(ph[:, wv, 0].median()), float(ph[:, wv, 1].median())
                print(f"      {nm}: per step {st_t / (steps - 1):.0f} clocks in its seven staging parts, {wt_t / steps:.0f} clocks at the barrier  (a step is {loop_us / steps * 2400:.0f} clocks at 2.4 GHz)", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_W43H_STAMP")
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
