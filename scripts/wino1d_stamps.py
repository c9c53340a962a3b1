"""Where a workgroup of wino1d_kernel spends its life (diagnostic build -DIDIFF_W1D_STAMP: s_memrealtime, 100 MHz, of lane 0 at kernel start, first
operands arrived, loop start, loop end, exchange done, outputs stored, end).  Run on the GPU box: python scripts/wino1d_stamps.py ["extra flags"]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child
VARIANT = "wino1d_stamps"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    dev = "cuda"
    B = 2240
    holder = torch.zeros(8 * 40000, device=dev, dtype=torch.int64)
    os.environ["IDIFF_W1D_STAMP_PTR"] = hex(holder.data_ptr())
    import id_diff_amd
    from id_diff_amd import _lib
    for (H, Cin, Cout) in ((32, 128, 128), (16, 256, 256), (16, 512, 256), (8, 256, 256)):
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        u = _lib.wino1d_pack(w, Cin, Cout)
        out = torch.empty(B, H * H, Cout, device=dev)
        nwg = ((B * H * H + 511) // 512) * (Cout // 64)
        ns = _lib.conv2d_wino1d_colstats_split(B, H, H, Cin, Cout)
        cs = torch.empty(B * ns * Cout * 2, device=dev, dtype=torch.float64)
        res = torch.randn(B, H * H, Cout, device=dev)
        eps = {"plain (bias)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev)),
               "conv 0 (bias, time-embedding bias, column sums)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev), rowbias=torch.randn(B, Cout, device=dev), rows_per_group=H * H, colstats=cs),
               "conv 1 (bias, residual, scale, column sums)": _lib.make_epilogue(bias=torch.randn(Cout, device=dev), residual=res, out_scale=0.7071, rows_per_group=H * H, colstats=cs)}
        for name, ep in eps.items():
            for _ in range(3):
                holder.zero_()
                _lib.conv2d_wino1d(x, u, out, B, H, H, Cin, Cout, epilogue=ep)
            torch.cuda.synchronize()
            t = holder[:8 * nwg].view(nwg, 8).cpu().double()[:, :7] * 0.01     # microseconds
            m = [float((t[:, k + 1] - t[:, k]).median()) for k in range(6)]
            steps = Cin // 16
            life = float((t[:, 6] - t[:, 0]).median())
            span = float(t[:, 6].max() - t[:, 0].min())
            print(f"{H}x{H} {Cin}->{Cout} [{name}]: {nwg} workgroups ({nwg / 256:.1f} per CU); median us per workgroup: first operands {m[0]:.1f}, first stage {m[1]:.1f}, "
                  f"K loop {m[2]:.1f} ({steps} steps: {m[2] / steps:.2f} each), first park {m[3]:.1f}, outputs + second park {m[4]:.1f}, column sums {m[5]:.1f}; life {life:.1f}; "
                  f"kernel span {span:.0f} us = {span / (nwg / 256):.1f} per workgroup round", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_W1D_STAMP " + (sys.argv[1] if len(sys.argv) > 1 else ""))
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
