"""Alternating-build A/B of winograd43_kernel on ONE box: python scripts/wino43_ab.py "<flags A>" "<flags B>" [rounds]
e.g.  "" "-DIDIFF_W43_DYADIC_POINTS"   or   "-DIDIFF_W43_SCR_UNIT=168" "".  Each round builds a variant library (scripts/_variant.py) with the flags and times
the F(4x4) convolutions of one nf = 128 NCSN++ forward at B = 2240 (scripts/wino43_probe.py's shape table), F(2x2) beside it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "wino43_ab"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = torch.device("cuda:0")
    B = 2240
    shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
              (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (16, 128, 128, 2), (16, 128, 256, 1)]
    tot = [0.0, 0.0]
    for H, Cin, Cout, calls in shapes:
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        o = torch.empty(B, H * H, Cout, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rows_per_group=H * H)
        u2, u4 = _lib.winograd_pack(w, Cin, Cout), _lib.winograd43_pack(w, Cin, Cout)
        for k, fn in enumerate((lambda: _lib.conv2d_winograd(x, u2, o, B, H, H, Cin, Cout, epilogue=ep),
                                lambda: _lib.conv2d_winograd43(x, u4, o, B, H, H, Cin, Cout, epilogue=ep))):
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            tot[k] += e0.elapsed_time(e1) / 5 * calls
    print(f"{sys.argv[2]!r:40s} F(2x2) {tot[0]:7.1f} ms  F(4x4) {tot[1]:7.1f} ms  ratio {tot[0] / tot[1]:.3f}", flush=True)
    sys.exit(0)

A, Bf = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
try:
    for r in range(rounds):
        for flags in (A, Bf):
            build_variant(VARIANT, flags)
            run_child(__file__, VARIANT, flags or "(default)")
finally:
    remove_variant(VARIANT)
