import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
dev = "cuda"
BB = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shapes = [(BB, 32, 128, 128), (BB, 16, 256, 256), (BB, 32, 256, 128), (BB, 32, 384, 128), (BB, 8, 256, 256)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (B, H, Cin, Cout) in shapes + shapes:
    xx = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02
    out = torch.empty(B, H * H, Cout, device=dev)
    _lib.conv2d_nhwc(xx, w, out, B, H, H, Cin, Cout, 3, 3, 1, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.conv2d_nhwc(xx, w, out, B, H, H, Cin, Cout, 3, 3, 1, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"conv3x3 B={B} {H}x{H} {Cin}->{Cout}: {dt*1e6:.0f} us  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
a = torch.randn(8192, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
_lib.gemm(a, b); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    _lib.gemm(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"gemm 8192x4096x4096: {dt*1e6:.0f} us {2.0*8192*4096*4096/dt/1e12:.1f} TFLOP/s", flush=True)
