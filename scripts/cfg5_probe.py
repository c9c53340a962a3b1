"""BASELINE config 5 (64x64x3 BeatGANs U-Net): phase timings for one data point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, dim_reduction, sde_lib, plot_utils
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils

def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
cfg = read_config('configs/dimension_estimation/extra_experiments/styleGAN/style_gan_64d_BeatGAN.py')
cfg.model.allow_random_init = True
torch.manual_seed(0)
model = mutils.create_model(cfg)
g = torch.Generator().manual_seed(3)
with torch.no_grad():
    for prm in model.parameters():
        if float(prm.abs().sum()) == 0.0:
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
model = model.to(dev).eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
rows_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for n in (rows_arg,):
    x = torch.rand(n, 3, 64, 64, device=dev); t = torch.full((n,), 1e-5, device=dev)
    with torch.no_grad():
        score_fn(x, t); torch.cuda.synchronize()
        t0 = time.perf_counter(); score_fn(x, t); score_fn(x, t); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
    say(f"score_fn rows={n}: {dt*1e3:.1f} ms -> {n/dt:.1f} evals/s, {n*37.43e9/dt/1e12:.1f} TFLOP/s model")
if len(sys.argv) > 2 and sys.argv[2] == "svd":
    M, D = 16768, 12288
    S = torch.randn(M, D, device=dev)
    S[:, D - 64:] *= 0.02
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sv = _lib.spectrum(S); torch.cuda.synchronize()
    say(f"spectrum {M}x{D}: {(time.perf_counter()-t0):.2f} s; ID rule -> {plot_utils.estimate_dim(sv.tolist())} (expect 64)")
if len(sys.argv) > 2 and sys.argv[2] == "pipeline":
    from id_diff_amd.lightning_data_modules.SyntheticImages import smooth_decoder_images
    imgs = smooth_decoder_images(3, [3, 64, 64], 64, seed=0).to(dev)
    builder = dim_reduction.ScoreMatrixBuilder(score_fn, sde, eps, dev)
    pipe = dim_reduction.SpectrumPipeline(dev)
    with torch.no_grad():
        pipe.submit(builder.build(imgs[0], 128, seed=1)); pipe.results(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in (1, 2):
            pipe.submit(builder.build(imgs[i], 128, seed=2 + i))
        svs = pipe.results(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rows = dim_reduction.batching((3, 64, 64), 128)[2]
    say(f"config 5 pipeline: 2 points x {rows} rows in {dt:.2f} s -> {2*rows/dt:.0f} evals/s incl. spectra of {rows}x12288; "
        f"IDs {[plot_utils.estimate_dim(s.tolist()) for s in svs]}")
