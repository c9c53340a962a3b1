"""F(4x4,3x3) kernel: correctness against an fp64 convolution (every epilogue term) and timing against the F(2x2,3x3) kernel on the
3x3 convolutions of the nf = 128 NCSN++ at B = 2240 (shape, calls per forward).  python scripts/wino43_probe.py [check|time|all]"""
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
    for (B, H, W, Cin, Cout) in [(2, 8, 8, 8, 64), (3, 4, 4, 16, 64), (2, 16, 16, 64, 128), (5, 8, 12, 24, 64), (2, 16, 16, 512, 256),
                                 (4, 32, 32, 128, 128), (130, 4, 4, 48, 64), (1, 64, 64, 16, 64), (33, 8, 8, 256, 256)]:
        g = torch.Generator().manual_seed(B * H + Cin)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        b = torch.randn(Cout, generator=g)
        temb = torch.randn(B, Cout, generator=g)
        res = torch.randn(B, Cout, H, W, generator=g)
        rsc = torch.rand(B, generator=g) + 0.5
        assert _lib.conv2d_winograd43_ok(B, H, W, Cin, Cout)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wt = w.permute(0, 2, 3, 1).contiguous().to(dev)
        u = _lib.winograd43_pack(wt, Cin, Cout)
        out = torch.empty(B, H, W, Cout, device=dev)
        _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(dev)))
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        e1 = rel(out.permute(0, 3, 1, 2).cpu(), ref)
        out0 = torch.empty(B, H, W, Cout, device=dev)
        _lib.conv2d_winograd43(xd, u, out0, B, H, W, Cin, Cout)
        e0 = rel(out0.permute(0, 3, 1, 2).cpu(), F.conv2d(x.double(), w.double(), None, padding=1))
        resd = res.permute(0, 2, 3, 1).contiguous().to(dev)
        _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout,
                               epilogue=_lib.make_epilogue(bias=b.to(dev), rowbias=temb.to(dev), rows_per_group=H * W, act="silu",
                                                           residual=resd, out_scale=0.7071, rowscale=rsc.to(dev)))
        ref2 = (F.silu(ref + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
        e2 = rel(out.permute(0, 3, 1, 2).cpu(), ref2)
        u2 = _lib.winograd_pack(wt, Cin, Cout)
        o2 = torch.empty_like(out)
        _lib.conv2d_winograd(xd, u2, o2, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(dev)))
        e3 = rel(o2.permute(0, 3, 1, 2).cpu(), ref)
        say(f"{(B, H, W, Cin, Cout)}: F(4x4) no-epilogue {e0:.2e}, bias {e1:.2e}, full epilogue {e2:.2e};  F(2x2) bias {e3:.2e}")

if mode in ("time", "all"):
    B = 2240
    shapes = [(32, 128, 128, 13), (16, 256, 256, 14), (32, 256, 128, 4), (32, 256, 256, 2), (16, 512, 256, 4), (8, 256, 256, 17),
              (32, 384, 128, 1), (8, 512, 256, 5), (16, 384, 256, 1), (4, 256, 256, 19), (4, 512, 256, 5), (16, 128, 128, 2), (16, 128, 256, 1)]
    tot = [0.0, 0.0]
    for H, Cin, Cout, calls in shapes:
        x = torch.randn(B, H * H, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        o = torch.empty(B, H * H, Cout, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev), act="silu", rows_per_group=H * H)
        u2, u4 = _lib.winograd_pack(w, Cin, Cout), _lib.winograd43_pack(w, Cin, Cout)
        ms = []
        for fn in (lambda: _lib.conv2d_winograd(x, u2, o, B, H, H, Cin, Cout, epilogue=ep),
                   lambda: _lib.conv2d_winograd43(x, u4, o, B, H, H, Cin, Cout, epilogue=ep)):
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 5)
        tot[0] += ms[0] * calls; tot[1] += ms[1] * calls
        fl2, fl4 = 2.0 * 16 * (B * H * H / 4) * Cin * Cout, 2.0 * 36 * (B * H * H / 16) * Cin * Cout
        say(f"{H:3d}x{H:<3d} {Cin:4d}->{Cout:<4d} x{calls:<3d} F(2x2) {ms[0]*1e3:8.1f} us ({fl2/ms[0]/1e9:6.1f} TF/s)  F(4x4) {ms[1]*1e3:8.1f} us "
            f"({fl4/ms[1]/1e9:6.1f} TF/s executed)  speed-up {ms[0]/ms[1]:.3f}")
    say(f"per forward: F(2x2) {tot[0]:.1f} ms, F(4x4) {tot[1]:.1f} ms, ratio {tot[0]/tot[1]:.3f}")
