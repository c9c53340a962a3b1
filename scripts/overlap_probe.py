"""How long a spectrum takes on the side stream while the next point's score evaluations run (config 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, dim_reduction, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils
dev = torch.device("cuda:0")
cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
torch.manual_seed(0)
model = mutils.create_model(cfg).to(dev).eval()
sde, eps = sde_lib.configure_sde(cfg)
builder = dim_reduction.ScoreMatrixBuilder(mutils.get_score_fn(sde, model), sde, eps, dev, 2240)
side = torch.cuda.Stream(device=dev)
x = torch.rand(3, 32, 32, device=dev)
with torch.no_grad():
    S = builder.build(x, 128, seed=1); _lib.spectrum(S); torch.cuda.synchronize()
    t0 = time.perf_counter()
    S = builder.build(x, 128, seed=2); torch.cuda.synchronize()
    print(f"one point, score evaluations only: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    evs = []
    t0 = time.perf_counter()
    for i in range(4):
        S = builder.build(x, 128, seed=3 + i)
        ready = torch.cuda.Event(); ready.record(); side.wait_event(ready)
        with torch.cuda.stream(side):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); sv = _lib.spectrum(S); b.record()
        S.record_stream(side)
        evs.append((a, b))
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3
    print(f"4 points overlapped: {tot:.1f} ms total ({tot/4:.1f} per point); spectrum wall on the side stream: "
          + ", ".join(f"{a.elapsed_time(b):.0f}" for a, b in evs) + " ms (the last has nothing beside it)", flush=True)
