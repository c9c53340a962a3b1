"""fp16-pair GEMM against the six-product form on the NIN projections of one nf = 128 NCSN++ forward at B = 2240 (one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
def t_of(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
for M, N, K in [(573440, 256, 256), (573440, 512, 256), (573440, 768, 256), (2293760, 128, 128), (143360, 256, 256)]:
    a = F.silu(torch.randn(M, K, device=dev)); w = torch.randn(N, K, device=dev) / K ** 0.5; o = torch.empty(M, N, device=dev)
    ep = _lib.make_epilogue(bias=torch.randn(N, device=dev))
    sc = _lib.gemm_pairs_scale(w)
    t6 = t_of(lambda: _lib.gemm(a, w, o, epilogue=ep))
    ref = o[:4096].double().cpu()
    if not _lib.gemm_pairs_ok(M, N, K):
        print(f"M{M} N{N} K{K}: six products {t6:.0f} us, pairs not served"); continue
    t3 = t_of(lambda: _lib.gemm_pairs(a, w, sc, o, epilogue=ep))
    exact = a[:4096].double().cpu() @ w.double().cpu().T + ep._keepalive[0].double().cpu()
    e3 = float((o[:4096].double().cpu() - exact).norm() / exact.norm()); e6 = float((ref - exact).norm() / exact.norm())
    print(f"M{M} N{N} K{K}: six products {t6:.0f} us (err {e6:.2e}), pairs {t3:.0f} us (err {e3:.2e})  x{t6/t3:.2f}", flush=True)
