"""The network's 1x1 / NIN GEMM shapes: plain, with bias + residual, with colstats, split vs fp32 MFMA, in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

dev = torch.device("cuda:0")

def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for (M, N, K, hw) in ((573440, 256, 256, 256), (573440, 512, 256, 256), (2293760, 128, 256, 1024), (2293760, 256, 256, 1024), (2293760, 128, 128, 1024)):
    a = torch.randn(M, K, device=dev); bt = torch.randn(N, K, device=dev) / K ** 0.5
    out = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
    ns = _lib.gemm_colstats_split(M, N, K, K, K, hw)
    ws = torch.zeros(max(1, (M // hw) * max(ns, 1) * N * 2), device=dev, dtype=torch.float64)
    row = []
    for nosplit in (0, 1):
        _lib.set_option("IDIFF_NO_SPLIT", nosplit)
        cases = [("plain", None), ("bias+res", _lib.make_epilogue(bias=bias, residual=res, out_scale=0.7071))]
        if ns > 0:
            cases.append(("bias+res+colstats", _lib.make_epilogue(bias=bias, residual=res, out_scale=0.7071, colstats=ws, rows_per_group=hw)))
        for name, ep in cases:
            us = t(lambda: _lib.gemm(a, bt, out=out, epilogue=ep))
            row.append(f"{'fp32 ' if nosplit else 'split'} {name:18s} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
    _lib.set_option("IDIFF_NO_SPLIT", 0)
    print(f"M={M} N={N} K={K} (colstats split {ns}):")
    for r in row: print("   ", r)
