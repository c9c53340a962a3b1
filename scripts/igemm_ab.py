"""Builds of the split-precision igemm compared on ONE box: python scripts/igemm_ab.py "<flags 1>" "<flags 2>" ...
Each entry builds a variant library (scripts/_variant.py) with the flags ("" = as committed; IDIFF_IGEMM_DIAG_* make timing-only kernels whose results are
wrong by construction) and times the 1x1 / NIN / attention contractions of one nf = 128 NCSN++ forward at B = 2240."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "igemm_ab"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = torch.device("cuda:0")
    B = 2240
    def t_of(fn):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5
    tot, detail = 0.0, []
    # one-source [M, K] x [N, K]^T with bias: (M, N, K, calls per forward)
    for M, N, K, calls in [(573440, 256, 256, 7), (573440, 512, 256, 5), (2293760, 256, 256, 1), (2293760, 128, 256, 1),
                           (2293760, 128, 128, 1), (573440, 256, 128, 2), (143360, 256, 256, 2)]:
        a = torch.randn(M, K, device=dev); bt = torch.randn(N, K, device=dev) / K ** 0.5; o = torch.empty(M, N, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(N, device=dev))
        t = t_of(lambda: _lib.gemm(a, bt, o, epilogue=ep))
        tot += t * calls
        if (M, N, K) in ((573440, 256, 256), (573440, 512, 256)): detail.append(f"M{M} N{N} K{K} {t*1e3:.0f} us")
        del a, bt, o
    # attention products, batched per image: [256, 256] x [256, 256]^T
    a = torch.randn(B, 256, 256, device=dev); bt = torch.randn(B, 256, 256, device=dev); o = torch.empty(B, 256, 256, device=dev)
    t = t_of(lambda: _lib.gemm(a, bt, o, M=256, N=256, K=256, lda=256, ldb=256, ldc=256, batch=B, stride_a=65536, stride_b=65536, stride_c=65536))
    tot += t * 10; detail.append(f"b{B} 256^3 {t*1e3:.0f} us")
    del a, bt, o
    # two-source shortcuts
    for M, N, K1, calls in [(2293760, 128, 128, 4), (573440, 256, 256, 4), (143360, 256, 256, 5)]:
        a1 = torch.randn(M, K1, device=dev); a2 = torch.randn(M, K1, device=dev); bt = torch.randn(N, 2 * K1, device=dev); o = torch.empty(M, N, device=dev)
        ep = _lib.make_epilogue(bias=torch.randn(N, device=dev))
        t = t_of(lambda: _lib.gemm_2src(a1, a2, bt, o, epilogue=ep))
        tot += t * calls
        if M == 573440: detail.append(f"2src M{M} N{N} K{2*K1} {t*1e3:.0f} us")
        del a1, a2, bt, o
    print(f"{sys.argv[2]!r:60s} split GEMMs {tot:7.2f} ms per forward   [{', '.join(detail)}]", flush=True)
    sys.exit(0)

args = sys.argv[1:]
try:
    for flags in args:
        build_variant(VARIANT, flags, scratch_limit=100000)
        run_child(__file__, VARIANT, flags or "(as committed)")
finally:
    remove_variant(VARIANT)
