"""Centred fp64 Gram: large-tile kernel vs the 64 x 64 kernel (time, TFLOP/s of the upper-triangular tile flops)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
for M, D in ((4480, 3072), (16768, 12288)):
    S = torch.randn(M, D, device="cuda"); mean = _lib.column_sums(S) / M
    for small in (0, 1):
        _lib.set_option("IDIFF_GRAM_SMALL_TILES", small)
        _lib.centered_gram(S, mean); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): _lib.centered_gram(S, mean)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{M}x{D} {'64x64 tiles ' if small else '128x128 tiles'}: {ms:8.2f} ms  {M * D * D / ms / 1e9:6.1f} TFLOP/s (2*M*D^2/2)", flush=True)
    _lib.set_option("IDIFF_GRAM_SMALL_TILES", 0)
    del S
