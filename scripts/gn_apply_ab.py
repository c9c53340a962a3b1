"""gn_apply_rows_kernel at the benchmark's GroupNorm shapes beside a plain copy (torch) of the same tensor: the achievable read + write
rate of the box.  (Round 4 used it for two A/Bs that changed nothing: rotated workgroup start offsets, non-temporal accesses.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

dev = "cuda"
def timeit(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2240
for (HW, C, G) in [(1024, 128, 32), (256, 256, 32), (1024, 256, 32), (256, 512, 32), (64, 256, 32), (16, 256, 32)]:
    x = torch.randn(B, HW, C, device=dev); y = torch.empty_like(x)
    st = torch.empty(B * G * 2, device=dev); ns = _lib.groupnorm_nsplit(B, HW, C)
    ws = torch.empty(B * ns * C * 2, device=dev, dtype=torch.float64)
    ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
    _lib.groupnorm_stats(x, C, None, 0, B, HW, G, 1e-6, ws, st)
    f = lambda: _lib.groupnorm_apply(x, C, None, 0, B, HW, G, st, ga, be, "silu", y)
    res = [timeit(f) for rep in range(3)]
    tc = timeit(lambda: y.copy_(x))
    nb = 8 * x.numel()
    print(f"B={B} HW={HW} C={C}: gn_apply+silu {min(res)*1e6:7.1f} us ({nb/min(res)/1e9:5.0f} GB/s)   torch copy of the same tensor {tc*1e6:7.1f} us ({nb/tc/1e9:5.0f} GB/s)", flush=True)
