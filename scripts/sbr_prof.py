"""One spectrum's worth of the two-stage eigensolver for rocprofv3 (argument: D)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
G = torch.randn(D + 64, D, device="cuda", dtype=torch.float64)
G = G.T @ G
for _ in range(2):
    _lib.sym_eigvals(G.clone())
torch.cuda.synchronize()
