"""One split-precision GEMM shape, a few launches (for rocprofv3 --pmc runs): gemm_one.py M N K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev); bt = torch.randn(N, K, device=dev) / K ** 0.5; o = torch.empty(M, N, device=dev)
ep = _lib.make_epilogue(bias=torch.randn(N, device=dev))
for _ in range(4):
    _lib.gemm(a, bt, o, epilogue=ep)
torch.cuda.synchronize()
print("done", flush=True)
