"""Where a step of winograd_split_kernel spends its cycles: a diagnostic build (-DIDIFF_SPLIT_PHASES) stamps s_memtime
between the phases of every step (each stamp waits for the wave's LDS traffic, so it perturbs the schedule by ~10 %).
Run on the GPU box:  python scripts/wino_split_phases.py   (builds a separate diagnostic library, libidiff_hip.<variant>.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "wino_split_phases"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = "cuda"
    _lib.set_option("IDIFF_WINO_SPLIT", 1)
    names = ["reads+early stage", "position 0", "position 1", "late stage", "barrier"]
    for (H, Cin, Cout) in ((16, 256, 256), (32, 128, 128), (16, 512, 256)):
        B = 2240
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        u = _lib.winograd_pack(w, Cin, Cout, split=True)
        out = torch.empty(B, H * H, Cout, device=dev)
        nwg = (B * H * H // 4 // 32) * (Cout // 64)
        st = torch.zeros(nwg * 8 * 8, device=dev, dtype=torch.int32)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), colstats=st.view(torch.float64))
        for _ in range(3):
            _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=ep, split=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=ep, split=True)
        e1.record(); torch.cuda.synchronize()
        s = st.view(nwg, 8, 8).double()
        steps = s[0, 0, 7].item()
        print(f"{H}x{H} {Cin}->{Cout}: {e0.elapsed_time(e1) / 5:.3f} ms per launch (stamped build), {int(steps)} steps", flush=True)
        for half, nm in ((slice(0, 4), "early waves"), (slice(4, 8), "late waves ")):
            m = s[:, half, :].mean(dim=(0, 1))
            per = "  ".join(f"{names[k]} {m[k].item() / steps:7.0f}" for k in range(5))
            print(f"   {nm}: cycles per step: {per} | loop total {m[5].item():8.0f}  tail {m[6].item():7.0f}", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_SPLIT_PHASES")
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
