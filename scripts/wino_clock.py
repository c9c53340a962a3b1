"""The clock the chip holds INSIDE winograd_kernel (MI355X_MICROARCH.md, 'DVFS give-back' item 6): a diagnostic build
(-DIDIFF_WINO_STAMP: every workgroup stamps s_memtime and s_memrealtime around its lifetime into a buffer of its own) runs
back-to-back launches on random data for ~2 s, then clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the
workgroups of the last launch.  Run on the GPU box:  python scripts/wino_clock.py   (builds a separate diagnostic library, libidiff_hip.<variant>.so)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "wino_clock"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = "cuda"
    for (H, Cin, Cout) in ((32, 128, 128), (16, 256, 256), (16, 512, 256)):
        B = 2240
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        u = _lib.winograd_pack(w, Cin, Cout)
        out = torch.empty(B, H * H, Cout, device=dev)
        nwg = (B * H * H // 4 // 32) * (Cout // 64)
        stamps = torch.zeros(2 * nwg, device=dev, dtype=torch.int64)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), colstats=stamps.view(torch.float64))
        t0 = time.time(); n = 0
        while time.time() - t0 < 2.0:
            for _ in range(20):
                _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=ep)
            torch.cuda.synchronize(); n += 20
        st = stamps.view(nwg, 2).double()
        clk = (st[:, 0] / st[:, 1] * 0.1).cpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=ep)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{H}x{H} {Cin}->{Cout}: in-kernel clock median {clk.median():.3f} GHz (5 % {clk.quantile(0.05):.3f}, 95 % {clk.quantile(0.95):.3f}) "
              f"after {n} launches; {ms:.3f} ms per launch in the stamped build = {2.0 * 16 * (B * H * H // 4) * Cin * Cout / ms / 1e9:.1f} TFLOP/s; "
              f"workgroup lifetime median {st[:, 1].median() * 10:.0f} ns", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_WINO_STAMP")
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
