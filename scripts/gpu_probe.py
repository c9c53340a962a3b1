"""Phase timings of the hot path on the GPU box (development aid; prints progressively)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import id_diff_amd
from id_diff_amd import _lib, dim_reduction, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils


def say(*a):
    print(*a, flush=True)


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


say("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
dev = torch.device("cuda:0")
cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
cfg.model.init_scale = 1.0
torch.manual_seed(0)
model = mutils.create_model(cfg).to(dev).eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
for n in (32, 128, 512):
    x = torch.rand(n, 3, 32, 32, device=dev)
    t = torch.full((n,), 1e-5, device=dev)
    with torch.no_grad():
        dt = timeit(lambda: score_fn(x, t), reps=2)
    say(f"score_fn rows={n}: {dt*1e3:.1f} ms  -> {n/dt:.1f} evals/s, {n*21.79e9/dt/1e12:.2f} TFLOP/s model")

# individual conv shapes
for (B, H, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 8, 256, 256), (128, 4, 256, 256), (128, 32, 256, 128), (512, 32, 128, 128)]:
    xx = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02
    out = torch.empty(B, H * H, Cout, device=dev)
    dt = timeit(lambda: _lib.conv2d_nhwc(xx, w, out, B, H, H, Cin, Cout, 3, 3, 1, 1), reps=5)
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    say(f"conv3x3 B={B} {H}x{H} {Cin}->{Cout}: {dt*1e6:.0f} us  {fl/dt/1e12:.1f} TFLOP/s")
for (M, N, K) in [(4096, 4096, 4096), (16000, 2048, 2048), (131072, 128, 1152)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    dt = timeit(lambda: _lib.gemm(a, b), reps=3)
    say(f"gemm {M}x{N}x{K}: {dt*1e6:.0f} us  {2.0*M*N*K/dt/1e12:.1f} TFLOP/s")

S = torch.randn(4480, 3072, device=dev)
dt = timeit(lambda: _lib.spectrum(S), reps=1)
say(f"spectrum 4480x3072: {dt*1e3:.1f} ms")
S = torch.randn(64, 1501, 100, device=dev)
dt = timeit(lambda: _lib.spectrum(S), reps=2)
say(f"spectrum 64 x 1501x100: {dt*1e3:.2f} ms")
