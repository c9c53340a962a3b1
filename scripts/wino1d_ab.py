"""Builds of wino1d_kernel compared on ONE box: python scripts/wino1d_ab.py "<flags 1>" "<flags 2>" ... [--rounds N]
Each entry builds a variant library (scripts/_variant.py) with the flags ("" = as committed; the IDIFF_W1D_DIAG_* flags make timing-only kernels whose
results are wrong by construction) and times the row-wise F(4, 3) pair convolutions of one nf = 128 NCSN++ forward at B = 2240."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child
VARIANT = "wino1d_ab"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = torch.device("cuda:0")
    B = 2240
    shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
              (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (16, 128, 128, 2), (16, 128, 256, 1)]
    tot, detail = 0.0, []
    for H, Cin, Cout, calls in shapes:
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        o = torch.empty(B, H * H, Cout, device=dev)
        ns = _lib.conv2d_wino1d_colstats_split(B, H, H, Cin, Cout)
        cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rowbias=torch.randn(B, Cout, device=dev), rows_per_group=H * H, colstats=cs)
        u = _lib.wino1d_pack(w, Cin, Cout)
        fn = lambda: _lib.conv2d_wino1d(x, u, o, B, H, H, Cin, Cout, epilogue=ep)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        tot += t * calls
        if (H, Cin, Cout) in ((16, 256, 256), (32, 128, 128), (8, 256, 256)): detail.append(f"{H}x{H} {Cin}->{Cout} {t*1e3:.0f} us")
    print(f"{sys.argv[2]!r:60s} rows {tot:7.1f} ms per forward   [{', '.join(detail)}]", flush=True)
    sys.exit(0)

args = sys.argv[1:]
rounds = 1
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
try:
    for r in range(rounds):
        for flags in args:
            build_variant(VARIANT, flags, scratch_limit=100000)
            run_child(__file__, VARIANT, flags or "(as committed)")
finally:
    remove_variant(VARIANT)
