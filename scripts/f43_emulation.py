"""What would F(4x4, 3x3) Winograd do to parity?  CPU emulation, zero GPU minutes.

The nf = 128 NCSN++ of BASELINE config 3 (oracle network, CPU) is run three ways on the same rows of one point's score matrix:
  (a) as is (ATen fp32 direct convolutions)                                   -> S_ref
  (b) the 88 Winograd-eligible 3x3 convs (stride 1, pad 1, Cin % 8 == 0, Cout % 64 == 0) as F(2x2,3x3) in fp32 (what the HIP
      kernel computes: U = G g G^T packed in fp64 and rounded once, V = B^T d B and Y = A^T M A in fp32, fp32 contraction)
  (c) the same convs as F(4x4,3x3) in fp32, for several point sets (Lavin's 0, +-1, +-2 and better-conditioned ones)
reporting per-layer error against an fp64 convolution of the same fp32 operands, rel_err(S) against an fp64 run of the whole
network, and the spectrum / ID bars (every sigma above 2e-5 sigma_max within 1e-4 of the fp64 network's, same ID).

    python scripts/f43_emulation.py [rows] [threads]        (rows: score rows evaluated, default 256)
"""
import os, sys, time
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import id_diff_amd  # noqa: F401
from id_diff_amd.configs.utils import read_config
from oracle import dim as odim, models as omodels, sde as osde


def say(*a):
    print(*a, flush=True)


def toom_cook(points, m, r):
    """A^T [m, n], G [n, r], B^T [n, n] of F(m, r) for n - 1 finite points + infinity, exact rationals -> float64.
    Transposed polynomial multiplication: y = V_A^T [(V_G g) * (V^-T d)]; the scaling N_j = prod_{l != j}(p_j - p_l) is moved from
    B^T into G (Lavin's convention: integer-ish B^T)."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1
    V = [[p ** t for t in range(n)] for p in pts] + [[Fraction(0)] * (n - 1) + [Fraction(1)]]
    # inverse of V by Gauss-Jordan in exact arithmetic
    a = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if a[i][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        pv = a[c][c]
        a[c] = [v / pv for v in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [vi - f * vc for vi, vc in zip(a[i], a[c])]
    Vinv = [row[n:] for row in a]
    N = [Fraction(1)] * n
    for j in range(n - 1):
        for l in range(n - 1):
            if l != j:
                N[j] *= pts[j] - pts[l]
    AT = [[(pts[j] ** i if j < n - 1 else Fraction(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = [[(pts[j] ** k / N[j] if j < n - 1 else Fraction(int(k == r - 1))) for k in range(r)] for j in range(n)]
    BT = [[Vinv[t][j] * N[j] for t in range(n)] for j in range(n)]
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=np.float64)
    AT, G, BT = f(AT), f(G), f(BT)
    # check the bilinear identity on random data (fp64)
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(n), rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.abs(y - ref).max() < 1e-12, (points, y, ref)
    return AT, G, BT


class WinogradConv:
    """conv3x3 / stride 1 / pad 1 as F(m x m, 3 x 3) with fp32 transforms and an fp32 contraction (torch CPU)."""

    def __init__(self, m, points, split=None):
        # split = "f16x2": both operands of the contraction as a pair of fp16 values (hi = fp16(v), lo = fp16(v - hi), V scaled by
        # 1/4 and U by a power of two that brings max|U| to 2^12), three products hi*hi + hi*lo + lo*hi accumulated in fp32 --
        # what winograd43h_kernel's v_mfma_f32_32x32x16_f16 contraction computes
        self.split = split
        self.m, self.n = m, m + 2
        AT, G, BT = toom_cook(points, m, 3)
        self.G64 = torch.from_numpy(G)
        self.AT, self.BT = torch.from_numpy(AT).float(), torch.from_numpy(BT).float()
        self.cache = {}

    def __call__(self, conv, x):
        m, n = self.m, self.n
        key = id(conv)
        if key not in self.cache:                                  # U = G g G^T in fp64, rounded once (as winograd_pack does)
            w = conv.weight.detach().double()                      # [O, C, 3, 3]
            self.cache[key] = torch.einsum("ik,ockl,jl->ijoc", self.G64, w, self.G64).float().reshape(n * n, *w.shape[:2])
        U = self.cache[key]                                        # [n*n, O, C]
        B, C, H, W = x.shape
        assert H % m == 0 and W % m == 0
        xp = F.pad(x, (1, 1, 1, 1))
        d = xp.unfold(2, n, m).unfold(3, n, m)                     # [B, C, th, tw, n, n]
        th, tw = d.shape[2], d.shape[3]
        V = torch.einsum("ik,bcxykl,jl->ijbxyc", self.BT, d, self.BT).reshape(n * n, B * th * tw, C)     # fp32
        if self.split == "f16x2":
            ku = 12 - int(torch.ceil(torch.log2(U.abs().max())))
            Vs, Us = V * 0.25, U * (2.0 ** ku)
            assert float(Vs.abs().max()) < 65504, "fp16 range"
            Vh = Vs.half().float(); Vl = (Vs - Vh).half().float()
            Uh = Us.half().float(); Ul = (Us - Uh).half().float()
            Ut = lambda t: t.transpose(1, 2)
            M = (torch.bmm(Vh, Ut(Uh)) + torch.bmm(Vh, Ut(Ul)) + torch.bmm(Vl, Ut(Uh))) * (4.0 * 2.0 ** -ku)
        else:
            M = torch.bmm(V, U.transpose(1, 2))                    # [n*n, tiles, O] fp32 contraction over C
        M = M.reshape(n, n, B, th, tw, -1)
        Y = torch.einsum("ik,klbxyo,jl->boxiyj", self.AT, M, self.AT)                                   # [B, O, th, m, tw, m]
        y = Y.reshape(B, -1, H, W)
        return y + conv.bias.reshape(1, -1, 1, 1) if conv.bias is not None else y


def eligible(mod):
    return (isinstance(mod, nn.Conv2d) and mod.kernel_size == (3, 3) and mod.stride == (1, 1) and mod.padding == (1, 1)
            and mod.in_channels % 8 == 0 and mod.out_channels % 64 == 0)


class Patched:
    """Context: eligible convs of `model` run through `wino` (None = untouched); records per-layer error vs fp64 if asked."""

    def __init__(self, model, wino, layer_err=None):
        self.model, self.wino, self.layer_err = model, wino, layer_err
        self.saved = []

    def __enter__(self):
        for name, mod in self.model.named_modules():
            if eligible(mod):
                orig = mod.forward
                self.saved.append((mod, orig))

                def fwd(x, mod=mod, name=name, orig=orig):
                    y = self.wino(mod, x) if self.wino is not None else orig(x)
                    if self.layer_err is not None and x.shape[0] <= 8:
                        ref = F.conv2d(x.double(), mod.weight.double(), mod.bias.double(), padding=1)
                        self.layer_err.setdefault(name, []).append(float((y.double() - ref).norm() / ref.norm()))
                    return y
                mod.forward = fwd
        return self

    def __exit__(self, *exc):
        for mod, orig in self.saved:
            mod.forward = orig


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
    cfg.model.init_scale = 1.0
    torch.manual_seed(0)
    model = omodels.create_model(cfg).eval()
    n_el = sum(eligible(m) for m in model.modules())
    say(f"nf=128 NCSN++: {n_el} Winograd-eligible 3x3 convs of {sum(isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) for m in model.modules())}")
    sde = osde.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    score_fn = osde.get_score_fn(sde, model)
    g = torch.Generator().manual_seed(1)
    x0 = torch.rand(3, 32, 32, generator=g)
    t = torch.full((rows,), 1e-5)
    std = sde.marginal_prob(torch.zeros(1), t[:1])[1]
    xin = x0[None] + std[:, None, None, None] * torch.randn(rows, 3, 32, 32, generator=g)

    def run(wino, dtype=torch.float32, layer_err=None):
        out = []
        with torch.no_grad(), Patched(model, wino, layer_err):
            for i in range(0, rows, 64):
                xb = xin[i:i + 64].to(dtype)
                out.append(score_fn(xb, t[i:i + 64].to(dtype)).reshape(xb.shape[0], -1))
        return torch.cat(out)

    t0 = time.time()
    model.double()
    S64 = run(None, torch.float64)
    model.float()
    say(f"fp64 network on {rows} rows: {time.time() - t0:.0f} s")
    forms = [("direct fp32 (ATen)", None),
             ("F(2x2,3x3) fp32, points 0,+-1", WinogradConv(2, [0, 1, -1])),
             ("F(4x4,3x3) fp32, points 0,+-1/2,+-2 (the kernel's)", WinogradConv(4, [0, Fraction(1, 2), Fraction(-1, 2), 2, -2])),
             ("F(4x4,3x3) fp32, points 0,+-1,+-2 (Lavin)", WinogradConv(4, [0, 1, -1, 2, -2])),
             ("F(4x4,3x3) fp32, points 0,+-1,1/2,-2", WinogradConv(4, [0, 1, -1, Fraction(1, 2), -2])),
             ("F(4x4,3x3) fp32, points 0,+-2/3,+-3/2", WinogradConv(4, [0, Fraction(2, 3), Fraction(-2, 3), Fraction(3, 2), Fraction(-3, 2)])),
             ("F(4x4,3x3) 0,+-2/3,+-3/2, contraction on fp16 pairs", WinogradConv(4, [0, Fraction(2, 3), Fraction(-2, 3), Fraction(3, 2), Fraction(-3, 2)], split="f16x2"))]
    if len(sys.argv) > 3:
        forms = [forms[int(i)] for i in sys.argv[3].split(",")]
    sv64 = odim.spectrum_f64(S64.float()) if rows >= 8 else None
    c64 = S64 - S64.mean(0, keepdim=True)
    sv64 = torch.linalg.svdvals(c64)
    id64 = odim.estimate_dim(sv64.tolist())
    for name, wino in forms:
        t0 = time.time()
        le = {}
        with torch.no_grad(), Patched(model, wino, le):                       # per-layer error on 4 rows
            score_fn(xin[:4], t[:4])
        S = run(wino)
        errS = float((S.double() - S64).norm() / S64.norm())
        c = S.double() - S.double().mean(0, keepdim=True)
        sv = torch.linalg.svdvals(c)
        keep = sv64 > 2e-5 * sv64[0]
        sv_err = float(((sv[keep] - sv64[keep]).abs() / sv64[keep]).max())
        worst = max(le.items(), key=lambda kv: max(kv[1])) if le else ("-", [0.0])
        med = float(np.median([max(v) for v in le.values()])) if le else 0.0
        say(f"{name:48s} rel_err(S) {errS:.2e}  worst sigma rel err (sigma > 2e-5 sigma_max: {int(keep.sum())} values) {sv_err:.2e}  "
            f"ID {odim.estimate_dim(sv.tolist())} (fp64 network: {id64})  per-layer err vs fp64 conv: median {med:.1e}, worst {max(worst[1]):.1e} "
            f"({worst[0]})  [{time.time() - t0:.0f} s]")


if __name__ == "__main__":
    main()
