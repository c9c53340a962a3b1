"""Is the host ahead of the GPU?  Host time to ENQUEUE one 2240-row forward (no synchronisation) vs its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, sde_lib, dim_reduction
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils
dev = torch.device("cuda:0")
cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
cfg.model.allow_random_init = True
torch.manual_seed(0)
model = mutils.create_model(cfg).to(dev).eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
x = torch.rand(2240, 3, 32, 32, device=dev); t = torch.full((2240,), 1e-5, device=dev)
with torch.no_grad():
    for _ in range(2): score_fn(x, t)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        score_fn(x, t)
        t1 = time.perf_counter()
        score_fn(x, t)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"enqueue forward 1: {(t1-t0)*1e3:7.1f} ms   enqueue forward 2: {(t2-t1)*1e3:7.1f} ms   drain: {(t3-t2)*1e3:7.1f} ms", flush=True)
    builder = dim_reduction.ScoreMatrixBuilder(score_fn, sde, eps, dev)
    img = torch.rand(3, 32, 32, device=dev)
    builder.build(img, 128, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); S = builder.build(img, 128, seed=2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"builder.build (2 forwards): enqueue {(t1-t0)*1e3:.1f} ms, drain {(t2-t1)*1e3:.1f} ms", flush=True)
