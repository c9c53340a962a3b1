"""Two launch sets of 2240 rows: one after the other on one stream vs side by side on two streams (do the tails and the
small-map layers of one forward fill up with the other's work?).  python scripts/two_stream_probe.py"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
args = argparse.Namespace(gpus=1, steps=1, warmup=0, inflight=2240, no_cpu_baseline=True, no_overlap=True, no_probe=True,
                          no_extras=True, device=None)
work = bench.Workload(args, 0, dev)
score_fn = work.builder.score_fn
n = 2240
xa = torch.rand(n, 3, 32, 32, device=dev); xb = torch.rand(n, 3, 32, 32, device=dev)
t = torch.full((n,), 1e-5, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def sequential():
    with torch.no_grad():
        score_fn(xa, t); score_fn(xb, t)

def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            a = score_fn(xa, t)
        with torch.cuda.stream(s2):
            b = score_fn(xb, t)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for _ in range(2):
    print(f"two forwards of {n} rows: one stream {timed(sequential):.1f} ms, two streams {timed(concurrent):.1f} ms", flush=True)
