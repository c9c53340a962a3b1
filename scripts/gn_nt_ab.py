"""GroupNorm-apply with non-temporal loads / stores against the default cache policy, ONE box: python scripts/gn_nt_ab.py "<flags>" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _variant import build_variant, remove_variant, run_child
VARIANT = "gn_nt"
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import id_diff_amd
    from id_diff_amd import _lib
    dev = "cuda"
    out = []
    for (B, HW, C) in [(2240, 1024, 128), (2240, 256, 256), (2240, 1024, 256)]:
        x = torch.randn(B, HW, C, device=dev); y = torch.empty_like(x)
        G = 32
        st = torch.zeros(B * G * 2, device=dev); st[1::2] = 1.0
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        f = lambda: _lib.groupnorm_apply(x, C, None, 0, B, HW, G, st, ga, be, "silu", y)
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        out.append(f"{B}x{HW}x{C}: {us:6.0f} us {8.0 * x.numel() / us / 1e6:5.2f} TB/s")
        del x, y
    print(f"{sys.argv[2]!r:48s} " + "   ".join(out), flush=True)
    sys.exit(0)
try:
    for flags in sys.argv[1:]:
        build_variant(VARIANT, flags)
        run_child(__file__, VARIANT, flags or "(as committed)")
finally:
    remove_variant(VARIANT)
