"""Where a workgroup of winograd43_kernel spends its life (diagnostic build -DIDIFF_W43_STAMP: s_memrealtime at start, loop start,
loop end, end + the CU it ran on): prologue / K loop / tail per workgroup, and the gap between consecutive workgroups on one CU.
Run on the GPU box: python scripts/wino43_stamps.py   (builds a separate diagnostic library, libidiff_hip.<variant>.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "wino43_stamps"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, collections
    import id_diff_amd
    from id_diff_amd import _lib
    dev = "cuda"
    B = 2240
    for (H, Cin, Cout) in ((32, 128, 128), (16, 256, 256), (16, 512, 256), (8, 256, 256)):
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        u = _lib.winograd43_pack(w, Cin, Cout)
        out = torch.empty(B, H * H, Cout, device=dev)
        nwg = ((B * (H // 4) ** 2 + 31) // 32) * (Cout // 64)
        stamps = torch.zeros(10 * nwg, device=dev, dtype=torch.int64)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), act="silu", rows_per_group=H * H, colstats=stamps.view(torch.float64))
        for _ in range(3):
            _lib.conv2d_winograd43(x, u, out, B, H, H, Cin, Cout, epilogue=ep)
        torch.cuda.synchronize()
        st = stamps.view(nwg, 10).cpu()
        t = st[:, :4].double() * 0.01                      # microseconds
        pro, loop, tail = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
        cu = st[:, 4]
        by = collections.defaultdict(list)
        for i in range(nwg):
            by[int(cu[i]) & ~0xFF].append((float(t[i, 0]), float(t[i, 3])))     # drop wave / simd / pipe bits of HW_ID
        gaps = []
        for lst in by.values():
            lst.sort()
            gaps += [b[0] - a[1] for a, b in zip(lst, lst[1:])]
        g = torch.tensor(gaps)
        span = float(t[:, 3].max() - t[:, 0].min())
        tt = st[:, 6:10].double() * 0.01
        ph = [tt[:, 0] - t[:, 2], tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2], t[:, 3] - tt[:, 3]]
        print("   tail phases (median us): z exchange 0 %.1f, output 0 %.1f, z exchange 1 %.1f, output 1 %.1f, rest %.1f" % tuple(float(v.median()) for v in ph))
        print(f"{H}x{H} {Cin}->{Cout}: {nwg} workgroups on {len(by)} CUs, kernel span {span:.0f} us; per workgroup median us: prologue {pro.median():.1f}, "
              f"K loop {loop.median():.1f} ({int(st[0, 5])} steps: {loop.median() / int(st[0, 5]):.2f} per step), tail {tail.median():.1f}, "
              f"gap to the next workgroup on the CU {g.median():.1f} (mean {g.mean():.1f}, 95 % {g.quantile(0.95):.1f})", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_W43_STAMP", scratch_limit=256)
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
