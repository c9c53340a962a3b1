"""Runs every distinct Winograd-conv call of the benchmark NCSN++ forward twice in isolation; used under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to tabulate HBM traffic per launch
(profiles/r01_wino_traffic.json, built by scripts/parse_wino_traffic.py).  Prints the shape keys in launch order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils

cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
torch.manual_seed(0)
model = mutils.create_model(cfg).to("cuda").eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
# second argument: which kernel's calls to tabulate -- "43" = winograd43_kernel (F(4x4,3x3), fp32 contraction), "43h" = winograd43h_kernel
# (the same on fp16 pairs), default winograd_kernel (F(2x2,3x3))
# "1d" = wino1d_kernel (F(4,3) along the rows on fp16 pairs)
MODE = sys.argv[2] if len(sys.argv) > 2 else ""
F43, PAIRS, ROWWISE = MODE in ("43", "43h"), MODE == "43h", MODE == "1d"
OP, PACK = ("conv2d_wino1d", _lib.wino1d_pack) if ROWWISE else (("conv2d_winograd43", _lib.winograd43_pack) if F43 else ("conv2d_winograd", _lib.winograd_pack))
calls = []
orig = getattr(_lib, OP)

def rec(x, u, out, B, H, W, Cin, Cout, epilogue=None, **kw):
    if not F43 or bool(kw.get("pairs")) == PAIRS:
        calls.append((B, H, W, Cin, Cout))
    return orig(x, u, out, B, H, W, Cin, Cout, epilogue, **kw)

setattr(_lib, OP, rec)
with torch.no_grad():
    ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
    score_fn(torch.rand(ROWS, 3, 32, 32, device="cuda"), torch.full((ROWS,), 1e-5, device="cuda"))
setattr(_lib, OP, orig)
torch.cuda.synchronize()
for (B, H, W, Cin, Cout) in sorted(set(calls)):
    x = torch.randn(B, H * W, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.02
    kw = {"pairs": True} if PAIRS else {}
    u = PACK(w, Cin, Cout, **kw)
    out = torch.empty(B, H * W, Cout, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        orig(x, u, out, B, H, W, Cin, Cout, **kw)
    torch.cuda.synchronize()
    print("KEY", f"{B}x{H}x{W}x{Cin}->{Cout}", calls.count((B, H, W, Cin, Cout)), flush=True)
