"""Wall clock of the two-stage eigensolver (idiff_sym_eigvals on a resident fp64 Gram matrix) with and without the band
reduction's look-ahead, both in ONE process:  python scripts/sbr_ab.py [D ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

for D in [int(a) for a in sys.argv[1:]] or [3072, 12288]:
    G = torch.randn(D + 64, D, device="cuda", dtype=torch.float64)
    G = G.T @ G
    ref = None
    for name, la in (("serial", 0), ("look-ahead", 1), ("serial", 0), ("look-ahead", 1)):
        _lib.set_option("IDIFF_SBR_LOOKAHEAD", la)
        A = [G.clone() for _ in range(3)]
        ev = _lib.sym_eigvals(A[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in (1, 2):
            ev = _lib.sym_eigvals(A[k])
        e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = ev.clone()
        print(f"D = {D:6d} {name:10s}: {e0.elapsed_time(e1) / 2:8.2f} ms per eigensolve; max |d eig| / max eig vs first run "
              f"{((ev - ref).abs().max() / ref.abs().max()).item():.2e}", flush=True)
    _lib.set_option("IDIFF_SBR_LOOKAHEAD", 0)
