"""BASELINE config 2 (50-sphere in R^100, fcn 2048x5, B=500): batched score evals + batched spectrum vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, dim_reduction, plot_utils
from id_diff_amd.configs.utils import read_config

def say(*a): print(*a, flush=True)
cfg = read_config('configs/dimension_estimation/paper/euclidean_data/ksphere/50dim.py')
cfg.model.allow_random_init = True
cfg.device = "cuda:0"
cfg.data.data_samples = 8000
for P, name in ((65, 'fcn'), (513, 'fcn'), (65, 'ksphere_exact')):
    cfg.model.name = name
    cfg.dim_estimation.num_datapoints = P
    t0 = time.perf_counter()
    svd = dim_reduction.get_manifold_dimension(cfg, return_svd=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    svd = dim_reduction.get_manifold_dimension(cfg, return_svd=True)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    dims = plot_utils.plot_dims(svd)[1]
    n = len(svd['singular_values'])
    say(f"{name}: {n} points: {dt2:.3f} s (first call {dt:.2f} s incl. data/model setup) -> {n*1501/dt2:.0f} evals/s end to end; "
        f"IDs min/max {min(dims)}/{max(dims)}")
for P in (64, 512, 4096):
    S = torch.randn(P, 1501, 100, device="cuda")
    _lib.spectrum(S); torch.cuda.synchronize()
    t0 = time.perf_counter(); _lib.spectrum(S); torch.cuda.synchronize()
    say(f"batched spectrum P={P} (1501x100 each): {(time.perf_counter()-t0)*1e3:.2f} ms")
torch.set_num_threads(16)
S1 = torch.randn(1501, 100)
t0 = time.perf_counter()
for _ in range(20):
    torch.linalg.svd(S1 - S1.mean(0, keepdim=True))
say(f"CPU full torch.linalg.svd 1501x100 (16 threads): {(time.perf_counter()-t0)/20*1e3:.1f} ms per matrix")
t0 = time.perf_counter()
for _ in range(20):
    torch.linalg.svdvals(S1 - S1.mean(0, keepdim=True))
say(f"CPU torch.linalg.svdvals 1501x100: {(time.perf_counter()-t0)/20*1e3:.2f} ms per matrix")
