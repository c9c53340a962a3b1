"""Achieved HBM GB/s of the memory-bound kernels at BASELINE sizes (algorithmic bytes / time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import id_diff_amd
from id_diff_amd import _lib, op

dev = "cuda"
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def graph_time(fn, n=20, reps=5):
    """Device time per launch with the host out of the loop: n launches captured into one hipGraph, replayed `reps` times (the C-ABI
    launches on the stream it is handed, so stream capture records it).  None if capture is not possible."""
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n): fn()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (reps * n)
    except Exception as exc:                      # noqa: BLE001 -- a probe: say so and go on
        print(f"   (graph capture failed: {exc})", flush=True)
        return None
FLOOR = None   # seconds per call of the same entry point on a 4 KB tensor: host issue + launch + drain, no data
def say(name, nbytes, dt, dt_graph=None):
    if dt_graph is not None:
        print(f"{name:58s} {dt_graph*1e6:8.1f} us  {nbytes/dt_graph/1e9:7.0f} GB/s  ({nbytes/dt_graph/8e12*100:4.1f}% of 8 TB/s)  "
              f"[device time per launch, 20 launches replayed from a hipGraph: no host in the loop]", flush=True)
    raw = nbytes / dt
    note = ""
    if FLOOR is not None:
        corrected = nbytes / max(dt - FLOOR, 1e-7)
        note = f"  floor {FLOOR*1e6:4.1f} us -> {corrected/1e9:6.0f} GB/s net"
        if dt < 2.0 * FLOOR:
            note += "  LAUNCH-BOUND"
    print(f"{name:58s} {dt*1e6:8.1f} us  {raw/1e9:7.0f} GB/s  ({raw/8e12*100:4.1f}% of 8 TB/s){note}", flush=True)

k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dtype=torch.float32, device=dev)
# launch floor of a raw C-ABI call (what every row below pays before the first byte moves)
_x0 = torch.randn(1, 1, 32, 32, device=dev); _o0 = torch.empty(1, 1, 33, 33, device=dev)
FLOOR = timeit(lambda: _lib.upfirdn2d_raw(_x0, k, _o0, 1, 32, 32, 1, 1, 1, 1, 1, 2, 2, 2, 2), reps=200)
print(f"launch floor (upfirdn2d on one 32x32 plane, back-to-back calls): {FLOOR*1e6:.1f} us per call", flush=True)
# op.upfirdn2d (NCHW, minor = 1) at the three ncsnpp families, B = 128
for shape, up, down, pad in [((128, 128, 32, 32), 1, 2, (1, 1)), ((128, 256, 16, 16), 1, 2, (1, 1)), ((128, 256, 8, 8), 1, 2, (1, 1)),
                             ((128, 256, 16, 16), 2, 1, (2, 1)), ((128, 256, 8, 8), 2, 1, (2, 1)), ((128, 256, 4, 4), 2, 1, (2, 1)),
                             ((128, 3, 32, 32), 1, 1, (2, 2)), ((128, 128, 16, 16), 1, 1, (2, 2)), ((128, 256, 8, 8), 1, 1, (2, 2))]:
    x = torch.randn(*shape, device=dev)
    kk = k * (4 if up == 2 else 1)
    y = op.upfirdn2d(x, kk, up=up, down=down, pad=pad)
    out = torch.empty_like(y)
    n, c, h, w = shape
    def f(): _lib.upfirdn2d_raw(x, kk, out, n * c, h, w, 1, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    say(f"upfirdn2d NCHW {shape} up{up} down{down}", 4 * (x.numel() + y.numel()) + 64, timeit(f), graph_time(f))
# NHWC (minor = C) as the networks call it, 512 rows
for (B, H, C, up, down, pad) in [(512, 32, 128, 1, 2, (1, 1)), (512, 16, 256, 2, 1, (2, 1)), (512, 16, 256, 1, 2, (1, 1))]:
    x = torch.randn(B, H * H, C, device=dev)
    OH = _lib.upfirdn2d_out_size(H, up, down, pad[0], pad[1], 4)
    out = torch.empty(B, OH * OH, C, device=dev)
    kk = k * (4 if up == 2 else 1)
    def f(): _lib.upfirdn2d_raw(x, kk, out, B, H, H, C, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    say(f"upfirdn2d NHWC B={B} {H}x{H}x{C} up{up} down{down}", 4 * (x.numel() + out.numel()) + 64, timeit(f))
_xs = torch.randn(1, 4, 4, 4, device=dev); _bs = torch.randn(4, device=dev)
FLOOR = timeit(lambda: op.fused_leaky_relu(_xs, _bs), reps=200)     # the op API: autograd Function + torch.empty_like + launch
print(f"launch floor (op.fused_leaky_relu on 64 elements): {FLOOR*1e6:.1f} us per call", flush=True)
for shape in [(128, 128, 32, 32), (128, 256, 16, 16), (128, 256, 8, 8), (128, 256, 4, 4)]:
    x = torch.randn(*shape, device=dev); b = torch.randn(shape[1], device=dev)
    say(f"fused_leaky_relu {shape}", 4 * (2 * x.numel() + shape[1]), timeit(lambda: op.fused_leaky_relu(x, b)))
FLOOR = None
for (B, HW, C, G) in [(512, 1024, 128, 32), (512, 256, 256, 32), (512, 1024, 256, 32)]:
    x = torch.randn(B, HW, C, device=dev)
    ns = _lib.groupnorm_nsplit(B, HW, C)
    ws = torch.empty(B * ns * C * 2, device=dev, dtype=torch.float64); st = torch.empty(B * G * 2, device=dev)
    ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev); y = torch.empty_like(x)
    say(f"groupnorm_stats B={B} HW={HW} C={C}", 4 * x.numel(), timeit(lambda: _lib.groupnorm_stats(x, C, None, 0, B, HW, G, 1e-6, ws, st)))
    say(f"groupnorm_apply+silu B={B} HW={HW} C={C}", 8 * x.numel(), timeit(lambda: _lib.groupnorm_apply(x, C, None, 0, B, HW, G, st, ga, be, "silu", y)))
x = torch.randn(512 * 256, 256, device=dev)
say("softmax_rows 131072 x 256", 8 * x.numel(), timeit(lambda: _lib.softmax_rows(x, x, x.shape[0], 256, 0.0625)))
S = torch.randn(4480, 3072, device=dev)
mean = torch.empty(3072, device=dev, dtype=torch.float64); scr = torch.empty(32 * 3072, device=dev, dtype=torch.float64)
G = torch.empty(3072, 3072, device=dev, dtype=torch.float64)
lib = _lib.lib(); stq = torch.cuda.current_stream().cuda_stream
dt = timeit(lambda: lib.idiff_centered_gram_f64(S.data_ptr(), mean.data_ptr(), 1, 4480, 3072, G.data_ptr(), stq), reps=5)
print(f"centered_gram_f64 4480x3072: {dt*1e3:.2f} ms  {2*4480*3072*3072/2/dt/1e12:.1f} TFLOP/s fp64 (upper-triangular tiles)", flush=True)
