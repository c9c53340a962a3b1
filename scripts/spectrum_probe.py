import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4480
S = torch.randn(M, D, device="cuda")
_lib.spectrum(S); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    _lib.spectrum(S)
torch.cuda.synchronize()
print(f"spectrum {M}x{D}: {(time.perf_counter()-t0)/3*1e3:.1f} ms", flush=True)
