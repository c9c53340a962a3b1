"""The band reduction's look-ahead (IDIFF_SBR_LOOKAHEAD) at D = 12288 in different process states: 150 ms (serial 164) in a
fresh process, 260 ms once bench.py's config-2 leg has made its streams -- the helper stream then shares the caller's
hardware queue.  Modes: plain | stream | cfg5 | bench | bench_late | late_probe | late_cfg2 | bench_full."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from id_diff_amd import _lib
dev = torch.device("cuda:0")

def report(tag):
    for serial in (0, 1):
        _lib.set_option("IDIFF_SBR_LOOKAHEAD", 1 - serial)
        whole, stages = bench.spectrum_stage_report(16768, 12288, dev, reps=1)
        print(f"{tag} serial={serial}: whole {whole:.1f} ms", [(s["kernel"][:12], round(s["ms"], 1)) for s in stages], flush=True)
    _lib.set_option("IDIFF_SBR_LOOKAHEAD", 0)

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "plain":
    report("fresh process")
elif mode == "stream":
    s = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s):
        x = torch.randn(4096, 4096, device=dev); y = x @ x
    torch.cuda.synchronize()
    report("after a priority -1 side stream was used")
elif mode == "cfg5":
    d = bench.extra_cfg5(dev)
    print("extra_cfg5:", d["svd_wall_clock_ms_per_point"], [(k["kernel"][:12], round(k["ms"], 1)) for k in d["kernels"]], flush=True)
    report("after extra_cfg5")
elif mode == "bench":
    import argparse
    args = argparse.Namespace(gpus=1, steps=2, warmup=1, inflight=2240, no_cpu_baseline=True, no_overlap=False, no_probe=True,
                              no_extras=True, device=None)
    report("before anything")
    work = bench.Workload(args, 0, dev)
    report("after building the workload")
    with torch.no_grad():
        bench.timed_region(work, 1, 2, dev)
    report("after the timed region (side-stream spectra)")
    del work
    torch.cuda.empty_cache()
    report("after dropping the workload")
    with torch.no_grad():
        bench.extra_cfg2(dev)
    report("after extra_cfg2")
elif mode == "bench_late":
    import argparse
    args = argparse.Namespace(gpus=1, steps=2, warmup=1, inflight=2240, no_cpu_baseline=True, no_overlap=False, no_probe=True,
                              no_extras=True, device=None)
    work = bench.Workload(args, 0, dev)
    with torch.no_grad():
        bench.timed_region(work, 1, 2, dev)
    if len(sys.argv) > 2:
        with torch.no_grad():
            bench.spectrum_stage_report(work.rows, work.D, dev)
        print("stage report at D = 3072 done", flush=True)
    report("helper stream first made AFTER the timed region")
elif mode == "bench_full":
    # the real thing: bench.main with its extras, then the two forms again in the state it leaves behind
    orig = bench.extra_cfg5
    def wrapped(dev_):
        report("inside bench.main, right before extra_cfg5")
        return orig(dev_)
    bench.extra_cfg5 = wrapped
    bench.main(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[2:])
    report("after bench.main")
elif mode in ("late_probe", "late_cfg2"):
    import argparse
    args = argparse.Namespace(gpus=1, steps=2, warmup=1, inflight=2240, no_cpu_baseline=True, no_overlap=False, no_probe=True,
                              no_extras=True, device=None)
    work = bench.Workload(args, 0, dev)
    with torch.no_grad():
        bench.timed_region(work, 1, 2, dev)
        if mode == "late_probe":
            probe = bench.KernelProbe(); probe.install(); probe.active = True
            for i in range(1, 3):
                work.point(i)
            work.collect()
            torch.cuda.synchronize()
            probe.active = False; probe.uninstall()
            bench.roofline_report(probe)
        else:
            del work.model, work.builder, work.pipe
            torch.cuda.empty_cache()
            bench.extra_cfg2(dev)
    report("helper stream first made after " + mode)
