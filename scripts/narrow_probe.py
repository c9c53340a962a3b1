"""The 128 -> 3 head conv: vector-ALU kernel vs the padded MFMA column (IDIFF_NO_PIPE=0/1 does not separate them; the
narrow kernel is bypassed by asking for 5 output channels... so time both entry conditions explicitly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = "cuda"
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for B, H in ((2240, 32), (256, 64)):
    x = torch.randn(B, H * H, 128, device=dev)
    rsc = torch.rand(B, device=dev)
    for cout in (3, 8):          # 8: the implicit-GEMM path on a 32-wide column, as the head ran before
        w = torch.randn(cout, 3, 3, 128, device=dev) / 34; b = torch.randn(cout, device=dev)
        out = torch.empty(B, H * H, cout, device=dev)
        ep = _lib.make_epilogue(bias=b, rowscale=rsc, rows_per_group=H * H)
        t = timeit(lambda: _lib.conv2d_nhwc(x, w, out, B, H, H, 128, cout, 3, 3, 1, 1, epilogue=ep))
        print(f"B={B} {H}x{H} 128->{cout}: {t*1e3:8.1f} us  {4.0 * x.numel() / t / 1e6:7.0f} GB/s of input", flush=True)
