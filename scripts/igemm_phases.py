"""Where a k-tile of the split-precision igemm_pipe_kernel spends its cycles: a diagnostic build (-DIDIFF_IGEMM_PHASES) stamps
s_memtime between the phases of every k-tile (each stamp waits for the wave's LDS traffic: it perturbs the schedule).
Run on the GPU box:  python scripts/igemm_phases.py   (builds a separate diagnostic library, libidiff_hip.<variant>.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child   # builds go to libidiff_hip.<name>.so, never to the product library
VARIANT = "igemm_phases"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = "cuda"
    names = ["LDS reads + MFMAs", "barrier 1", "stage (wait, cut, LDS writes)", "fetch issue", "barrier 2"]
    for (M, N, K) in ((2293760, 256, 256), (573440, 256, 256), (2293760, 128, 128)):
        a = torch.randn(M, K, device=dev); bt = torch.randn(N, K, device=dev) / K ** 0.5
        out = torch.empty(M, N, device=dev)
        nwg = ((M + 127) // 128) * ((N + 127) // 128)
        st = torch.zeros(nwg * 4 * 8, device=dev, dtype=torch.int32)
        ep = _lib.make_epilogue(bias=torch.randn(N, device=dev), colstats=st.view(torch.float64))
        for _ in range(3):
            _lib.gemm(a, bt, out=out, epilogue=ep)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): _lib.gemm(a, bt, out=out, epilogue=ep)
        e1.record(); torch.cuda.synchronize()
        s = st.view(nwg, 4, 8).double()
        nkt = s[0, 0, 7].item()
        m = s.mean(dim=(0, 1))
        per = "  ".join(f"{names[k]} {m[k].item() / nkt:6.0f}" for k in range(5))
        print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per launch (stamped build), {int(nkt)} k-tiles; cycles per k-tile and wave: {per} | "
              f"prologue {m[5].item():.0f}  epilogue {m[6].item():.0f}", flush=True)
    sys.exit(0)

try:
    build_variant(VARIANT, "-DIDIFF_IGEMM_PHASES")
    run_child(__file__, VARIANT)
finally:
    remove_variant(VARIANT)
