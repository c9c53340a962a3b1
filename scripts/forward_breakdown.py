"""Per-call timing of one ncsnpp forward (BASELINE config 3) at the benchmark's launch-set size: every _lib entry point is
wrapped with HIP events, calls are grouped by (op, geometry) and printed with TFLOP/s or GB/s."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils

def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
which = sys.argv[2] if len(sys.argv) > 2 else "config3"
cfg = read_config({"config3": 'configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py',
                   "config5": 'configs/dimension_estimation/extra_experiments/styleGAN/style_gan_64d_BeatGAN.py'}[which])
cfg.model.allow_random_init = True
torch.manual_seed(0)
model = mutils.create_model(cfg)
if which == "config5":     # the U-Net zero-initialises its output layers: give every all-zero parameter small values
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for prm in model.parameters():
            if float(prm.abs().sum()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
model = model.to(dev).eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
side = int(cfg.data.image_size)
x = torch.rand(rows, 3, side, side, device=dev); t = torch.full((rows,), 1e-5, device=dev)
with torch.no_grad():
    score_fn(x, t); score_fn(x, t); torch.cuda.synchronize()

records = []
def wrap(name, describe):
    orig = getattr(_lib, name)
    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        records.append((name, describe(*a, **k), e0, e1))
        return r
    setattr(_lib, name, timed)

def d_gemm(a, bt, out=None, epilogue=None, M=None, N=None, K=None, batch=1, **k):
    if M is None:
        M, K = a.shape; N = bt.shape[0]
    return (f"b{batch} M{M} N{N} K{K}", 2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N))
def d_gemm2(a1, a2, bt, out, epilogue=None):
    M, K, N = a1.shape[0], a1.shape[1] + a2.shape[1], bt.shape[0]
    return (f"two-source M{M} N{N} K{K}", 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N))
def d_conv(x, wt, out, B, H, W, Cin, Cout, KH, KW, stride, pad, epilogue=None, pad_hi=None):
    return (f"B{B} {H}x{W} {Cin}->{Cout} k{KH} s{stride}", 2.0 * out.numel() * Cin * KH * KW,
            4.0 * (B * H * W * Cin + out.numel()))
def d_wino(x, u, out, B, H, W, Cin, Cout, epilogue=None, split=False):
    return (f"B{B} {H}x{W} {Cin}->{Cout} (F(2x2,3x3); TF = direct-equivalent)", 2.0 * out.numel() * Cin * 9,
            4.0 * (B * H * W * Cin + out.numel()))
def d_wino43(x, u, out, B, H, W, Cin, Cout, epilogue=None, pairs=False):
    return (f"B{B} {H}x{W} {Cin}->{Cout} (F(4x4,3x3){' on fp16 pairs' if pairs else ''}; TF = direct-equivalent)", 2.0 * out.numel() * Cin * 9,
            4.0 * (B * H * W * Cin + out.numel()))
def d_gnapply(x, C, x2, C2, B, HW, G, stats, gamma, beta, act, y, mod=None):
    return (f"B{B} HW{HW} C{C}+{C2} act={act}", 0.0, 8.0 * B * HW * (C + C2))
def d_gnapply_cs(x, C, x2, C2, B, HW, G, ws1, ns1, ws2, ns2, eps_, gamma, beta, act, y, mod=None):
    return (f"B{B} HW{HW} C{C}+{C2} act={act} (stats from column sums)", 0.0, 8.0 * B * HW * (C + C2))
def d_gnstats(x, C, x2, C2, B, HW, G, eps_, workspace, stats):
    return (f"B{B} HW{HW} C{C}+{C2}", 0.0, 4.0 * B * HW * (C + C2))
def d_ufd(x, k, out, major, in_h, in_w, minor, *rest):
    return (f"major{major} {in_h}x{in_w} minor{minor} up{rest[0]} down{rest[2]}", 0.0, 4.0 * (x.numel() + out.numel()))
def d_soft(x, y, rows_, cols, scale):
    return (f"rows{rows_} cols{cols}", 0.0, 8.0 * rows_ * cols)
def d_generic(*a, **k):
    n = sum(4.0 * v.numel() for v in list(a) + list(k.values()) if torch.is_tensor(v))
    return ("", 0.0, n)

wrap("gemm", d_gemm); wrap("gemm_2src", d_gemm2); wrap("conv2d_nhwc", d_conv); wrap("conv2d_winograd", d_wino); wrap("conv2d_winograd43", d_wino43); wrap("groupnorm_apply", d_gnapply); wrap("groupnorm_apply_colstats", d_gnapply_cs); wrap("groupnorm_stats", d_gnstats)
wrap("upfirdn2d_raw", d_ufd); wrap("softmax_rows", d_soft)
for nm in ("groupnorm_finalize", "affine_act", "add_scale", "fourier_embed", "positional_embed", "concat_cols", "nchw_to_nhwc",
           "nhwc_to_nchw", "resample2x_nhwc"):
    wrap(nm, d_generic)

with torch.no_grad():
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(); score_fn(x, t); w1.record(); torch.cuda.synchronize()
whole = w0.elapsed_time(w1)
groups = collections.OrderedDict()
for name, (geom, flops, nbytes), e0, e1 in records:
    g = groups.setdefault((name, geom), [0, 0.0, flops, nbytes])
    g[0] += 1; g[1] += e0.elapsed_time(e1)
tot = sum(g[1] for g in groups.values())
say(f"forward rows={rows}: {whole:.1f} ms wall (event-instrumented), sum of calls {tot:.1f} ms, {len(records)} calls")
by_op = collections.Counter()
for (name, geom), (cnt, ms, flops, nbytes) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    by_op[name] += ms
    per = ms / cnt
    rate = f"{flops / per / 1e9:7.1f} TF" if flops else " " * 10
    say(f"{ms:8.2f} ms {100 * ms / tot:5.1f}%  x{cnt:<3} {per * 1e3:9.1f} us  {rate} {nbytes / per / 1e6:8.0f} GB/s  {name} {geom}")
say("by op:", ", ".join(f"{k} {v:.1f} ms ({100 * v / tot:.1f}%)" for k, v in by_op.most_common()))
