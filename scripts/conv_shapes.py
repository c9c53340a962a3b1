"""Runs every distinct 3x3-conv call of the benchmark NCSN++ forward (512 rows) twice in isolation; used under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to tabulate HBM traffic per launch (profiles/r01_conv_traffic.json).
Prints the shape keys in launch order (the parser takes the last 2*N igemm dispatches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils

cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
cfg.model.init_scale = 1.0
torch.manual_seed(0)
model = mutils.create_model(cfg).to("cuda").eval()
sde, eps = sde_lib.configure_sde(cfg)
score_fn = mutils.get_score_fn(sde, model)
calls = []
orig = _lib.conv2d_nhwc

def rec(x, wt, out, B, H, W, Cin, Cout, KH, KW, stride, pad, epilogue=None, pad_hi=None):
    if KH == 3:
        calls.append((B, H, W, Cin, Cout, stride, pad, pad if pad_hi is None else pad_hi))
    return orig(x, wt, out, B, H, W, Cin, Cout, KH, KW, stride, pad, epilogue, pad_hi)

_lib.conv2d_nhwc = rec
with torch.no_grad():
    ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
    score_fn(torch.rand(ROWS, 3, 32, 32, device="cuda"), torch.full((ROWS,), 1e-5, device="cuda"))
_lib.conv2d_nhwc = orig
torch.cuda.synchronize()
uniq = sorted(set(calls))
for (B, H, W, Cin, Cout, stride, pad, pad_hi) in uniq:
    x = torch.randn(B, H * W, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.02
    OH = (H + pad + pad_hi - 3) // stride + 1
    out = torch.empty(B, OH * OH, Cout, device="cuda")
    for _ in range(2):
        _lib.conv2d_nhwc(x, w, out, B, H, W, Cin, Cout, 3, 3, stride, pad, pad_hi=pad_hi)
    torch.cuda.synchronize()
    print("KEY", f"{B}x{H}x{W}x{Cin}->{Cout}", stride, OH, calls.count((B, H, W, Cin, Cout, stride, pad, pad_hi)), flush=True)
