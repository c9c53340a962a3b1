"""What the per-point spectrum costs the headline loop: bench.py's workload with (a) forwards only, (b) forwards + spectrum overlapped on the
side stream (the bench's timed region), (c) spectrum on the main stream.   python scripts/overlap_gap.py [points]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
args = types.SimpleNamespace(steps=P, warmup=1, inflight=2240, no_overlap=False, concurrent_sets=1)
work = bench.Workload(args, 0, dev)

def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(1, 1 + P):
            if mode == "forwards":
                work.last_S = work.builder.build(work.images[i], work.B, seed=1234 + 1000003 * (i + 1))
            else:
                work.point(i)
        if mode != "forwards":
            work.collect()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / P * 1e3

with torch.no_grad():
    work.point(0); work.collect()
for mode in ("forwards", "overlap", "forwards", "overlap"):
    print(f"{mode:10s} {run(mode):7.1f} ms per point", flush=True)
work.pipe = bench.dim_reduction.SpectrumPipeline(dev, overlap=False)
print(f"{'serial':10s} {run('serial'):7.1f} ms per point", flush=True)
