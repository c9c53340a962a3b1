"""fp16-pair F(4x4,3x3) kernel (winograd43h_kernel): correctness against an fp64 convolution (every epilogue term, column
statistics) beside the fp32-contraction kernel, and timing of both on the 3x3 convolutions of the nf = 128 NCSN++ at B = 2240
(shape, calls per forward).   python scripts/wino43h_probe.py [check|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import id_diff_amd
from id_diff_amd import _lib

def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "all"

def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())

if mode in ("check", "all"):
    for (B, H, W, Cin, Cout) in [(2, 8, 8, 32, 64), (3, 4, 4, 48, 64), (2, 16, 16, 64, 128), (5, 8, 12, 80, 64), (2, 16, 16, 512, 256),
                                 (4, 32, 32, 128, 128), (130, 4, 4, 48, 64), (1, 64, 64, 32, 64), (33, 8, 8, 256, 256), (7, 16, 16, 384, 256)]:
        g = torch.Generator().manual_seed(B * H + Cin)
        x = torch.randn(B, Cin, H, W, generator=g)
        x = x * (torch.rand(B, Cin, H, W, generator=g) < 0.9) * torch.exp(2 * torch.randn(B, Cin, H, W, generator=g))   # wide range, exact zeros
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        b = torch.randn(Cout, generator=g)
        temb = torch.randn(B, Cout, generator=g)
        res = torch.randn(B, Cout, H, W, generator=g)
        rsc = torch.rand(B, generator=g) + 0.5
        assert _lib.conv2d_winograd43h_ok(B, H, W, Cin, Cout), (B, H, W, Cin, Cout)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wt = w.permute(0, 2, 3, 1).contiguous().to(dev)
        u = _lib.winograd43_pack(wt, Cin, Cout, pairs=True)
        uf = _lib.winograd43_pack(wt, Cin, Cout)
        ref0 = F.conv2d(x.double(), w.double(), None, padding=1)
        out0 = torch.empty(B, H, W, Cout, device=dev)
        _lib.conv2d_winograd43(xd, u, out0, B, H, W, Cin, Cout, pairs=True)
        e0 = rel(out0.permute(0, 3, 1, 2).cpu(), ref0)
        outf = torch.empty(B, H, W, Cout, device=dev)
        _lib.conv2d_winograd43(xd, uf, outf, B, H, W, Cin, Cout)
        ef = rel(outf.permute(0, 3, 1, 2).cpu(), ref0)
        resd = res.permute(0, 2, 3, 1).contiguous().to(dev)
        out = torch.empty(B, H, W, Cout, device=dev)
        ns = _lib.conv2d_winograd43_colstats_split(B, H, W, Cin, Cout)
        cs = torch.zeros(B * max(ns, 1) * Cout * 2, device=dev, dtype=torch.float64) if ns > 0 else None
        _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout, pairs=True,
                               epilogue=_lib.make_epilogue(bias=b.to(dev), rowbias=temb.to(dev), rows_per_group=H * W, act="silu",
                                                           residual=resd, out_scale=0.7071, rowscale=rsc.to(dev), colstats=cs))
        ref2 = (F.silu(ref0 + b.double()[None, :, None, None] + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
        e2 = rel(out.permute(0, 3, 1, 2).cpu(), ref2)
        es = float("nan")
        if ns > 0:
            c = cs.reshape(B, ns, Cout, 2).sum(1).cpu()
            o64 = out.double().cpu()
            es = max(rel(c[..., 0], o64.sum((1, 2))), rel(c[..., 1], (o64 * o64).sum((1, 2))))
        say(f"{(B, H, W, Cin, Cout)}: pairs no-epilogue {e0:.2e} (fp32 contraction {ef:.2e}), full epilogue {e2:.2e}, colstats vs own output {es:.1e}")

if mode in ("time", "all"):
    B = int(os.environ.get("ROWS", 2240))
    shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
              (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (16, 128, 128, 2), (16, 128, 256, 1)]
    tot = [0.0, 0.0]
    for H, Cin, Cout, calls in shapes:
        if not _lib.conv2d_winograd43h_ok(B, H, H, Cin, Cout):
            say(f"{H}x{H} {Cin}->{Cout}: not served"); continue
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        o = torch.empty(B, H * H, Cout, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), act="silu", rows_per_group=H * H)
        uf, uh = _lib.winograd43_pack(w, Cin, Cout), _lib.winograd43_pack(w, Cin, Cout, pairs=True)
        ts = []
        for (u, pairs) in ((uf, False), (uh, True)):
            f = lambda: _lib.conv2d_winograd43(x, u, o, B, H, H, Cin, Cout, epilogue=ep, pairs=pairs)
            f(); f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 6)
        fl = 2.0 * B * H * H * Cin * Cout * 2.25 / 1e9      # executed Winograd-domain GFLOP
        tot[0] += ts[0] * calls; tot[1] += ts[1] * calls
        say(f"{H}x{H} {Cin}->{Cout} x{calls}: fp32 {ts[0]*1e3:7.0f} us ({fl/ts[0]:5.1f} TF/s)   pairs {ts[1]*1e3:7.0f} us ({fl/ts[1]:5.1f} TF/s executed-equivalent)   "
            f"{ts[0]/ts[1]:.2f}x")
    say(f"per forward (these shapes x calls): fp32 {tot[0]:.1f} ms, pairs {tot[1]:.1f} ms, {tot[0]/tot[1]:.2f}x")
