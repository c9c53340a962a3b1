"""Builds of attention256_kernel compared on ONE box: python scripts/attn_ab.py "<flags 1>" "<flags 2>" ...   ("" = as committed).
Each entry builds a variant library (scripts/_variant.py) and times the kernel at B = 2240, C = 256 and 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child
VARIANT = "attn_ab"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = torch.device("cuda:0")
    B, HW = 2240, 256
    out_line = []
    for C in (256, 128):
        qk, vt = torch.randn(B * HW, 2 * C, device=dev), torch.randn(B, C, HW, device=dev)
        out, one, bv = torch.empty(B * HW, C, device=dev), torch.tensor([1.0, 1.0], device=dev), torch.randn(C, device=dev)
        f = lambda: _lib.attention256(qk, vt, out, B, C, one, one, C ** -0.5, bias_v=bv)
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): f()
        e1.record(); torch.cuda.synchronize()
        # sanity: a reference on a few samples
        ref = torch.softmax(torch.einsum("bic,bjc->bij", qk[:2 * HW, :C].reshape(2, HW, C).double(), qk[:2 * HW, C:].reshape(2, HW, C).double()) * C ** -0.5, -1)
        ref = torch.einsum("bij,bcj->bic", ref, vt[:2].double()) + bv.double()
        err = float((out[:2 * HW].reshape(2, HW, C).double() - ref).norm() / ref.norm())
        out_line.append(f"C={C} {e0.elapsed_time(e1) / 6 * 1e3:6.0f} us (err {err:.1e})")
        del qk, vt, out
    print(f"{sys.argv[2]!r:44s} " + "   ".join(out_line), flush=True)
    sys.exit(0)

try:
    for flags in sys.argv[1:]:
        build_variant(VARIANT, flags)
        run_child(__file__, VARIANT, flags or "(as committed)")
finally:
    remove_variant(VARIANT)
