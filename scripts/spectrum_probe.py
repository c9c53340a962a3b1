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
# What the stock PyTorch-ROCm stack (rocSOLVER / hipSOLVER behind torch.linalg) takes for the same step on the same card:
# the reference's own call (torch.linalg.svd, dim_reduction.py:197), its values-only form, and the symmetric
# eigensolver on the fp64 Gram this library factors.  Library calls only; a comparison, not part of the product path.
if len(sys.argv) > 3 and sys.argv[3] == "libs":
    def timed(name, fn, reps=2):
        try:
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            print(f"  {name}: {(time.perf_counter()-t0)/reps*1e3:.1f} ms", flush=True)
        except Exception as e:       # a solver the build does not ship
            print(f"  {name}: unavailable ({type(e).__name__}: {str(e)[:80]})", flush=True)
    Sc = S - S.mean(0, keepdim=True)
    G = (Sc.double().T @ Sc.double())
    print(f"torch.linalg on cuda:0, {M}x{D}:", flush=True)
    if D <= 4096:      # (minutes at D = 12288)
        timed("torch.linalg.svd(S, full_matrices=True) fp32 (the reference's call)", lambda: torch.linalg.svd(Sc))
        timed("torch.linalg.svdvals(S) fp32", lambda: torch.linalg.svdvals(Sc))
    timed(f"torch.linalg.eigvalsh(G) fp64, G {D}x{D}", lambda: torch.linalg.eigvalsh(G))
