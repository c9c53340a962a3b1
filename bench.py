"""Headline benchmark: score-vector evals/sec (+ SVD wall-clock) of the manifold_dimension path, 32x32x3 NCSN++.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one data point of BASELINE config 3: 4480 score rows (B=128 -> (1024//128+1)*4 batches, last one
empty; dim_reduction.py:166-171) of the nf=128 NCSN++ on a synthetic 32x32x3 image, written into the
device-resident S [4480, 3072], then its centred singular spectrum and the integer ID.  Inputs (image, weights)
are resident in HBM before the timed region.  With N ranks every rank processes K points of its own (weak
scaling) and the region ends with the one all-gather of spectra the path has.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import id_diff_amd
from id_diff_amd import _lib, dim_reduction, parallel, plot_utils, sde_lib
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils
from id_diff_amd.lightning_data_modules.SyntheticImages import smooth_decoder_images

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
WORKLOAD = "ncsnpp nf128 ch(1,2,2,2) 4 resblocks attn@16 FIR, 32x32x3, VE-SDE t=1e-5, B=128 -> S 4480x3072 + centred spectrum + ID"


class ConvProbe:
    """HIP events around a sample of the dominant kernel's launches (the F(2x2,3x3) Winograd convs of csrc/winograd.hip:
    71 % of the kernel time of a forward pass) on the launch stream."""

    TRAFFIC = os.path.join("profiles", "r01_wino_traffic.json")

    def __init__(self):
        self.active = False
        self.records = []
        self._orig = _lib.conv2d_winograd

    def install(self):
        probe = self

        def timed(x, u, out, B, H, W, Cin, Cout, epilogue=None):
            if not probe.active:
                return probe._orig(x, u, out, B, H, W, Cin, Cout, epilogue)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = probe._orig(x, u, out, B, H, W, Cin, Cout, epilogue)
            b.record()
            # executed (algorithmic) flops of the Winograd form: 16 positions x [tiles x Cin] x [Cin x Cout] multiply-adds,
            # tiles = output pixels / 4; the implicit GEMM it replaces would be 2.25x that (9 taps per pixel)
            flops = 2.0 * 16 * (out.numel() // Cout // 4) * Cin * Cout
            probe.records.append((a, b, flops, f"{B}x{H}x{W}x{Cin}->{Cout}"))
            return r

        _lib.conv2d_winograd = timed

    def summary(self):
        if not self.records:
            return None
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        return {"launches": len(self.records), "avg_us": ms * 1e3 / len(self.records), "tflops": fl / (ms * 1e-3) / 1e12,
                "traffic": self.traffic()}

    def traffic(self):
        """HBM bytes per launch of the sampled shapes, from the committed PMC table (FETCH_SIZE / WRITE_SIZE passes,
        gfx950 corrections applied; profiles/r01_wino_traffic.json); None if a sampled shape is not in the table."""
        path = os.path.join(ROOT, self.TRAFFIC)
        if not os.path.exists(path):
            return None
        table = json.load(open(path))["shapes"]
        keys = [r[3] for r in self.records]
        if any(k not in table for k in keys):
            return None
        return {"bytes_per_launch": sum(table[k]["total_bytes"] for k in keys) / len(keys),
                "algorithmic_bytes_per_launch": sum(table[k]["algorithmic_bytes"] for k in keys) / len(keys),
                "source": self.TRAFFIC}


def cpu_baseline(cfg, rows_per_point, D):
    """The oracle (CPU restatement of the reference path) on this box's host cores, bounded sample."""
    from oracle import models as omodels, sde as osde
    # the GPU box exposes every host core (os.cpu_count() = 256) but grants a 16-core share per GPU:
    # oversubscribing the share makes the CPU run arbitrarily slow, so use the share
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = omodels.create_model(cfg)
    score_fn = osde.get_score_fn(osde.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales), model)
    n = 32
    x = torch.rand(n, 3, 32, 32)
    t = torch.full((n,), 1e-5)
    print(f"[bench] cpu_baseline: oracle score_fn on {cores} threads ...", file=sys.stderr, flush=True)
    with torch.no_grad():
        score_fn(x[:4], t[:4])
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 12.0:
            score_fn(x, t)
            reps += 1
            print(f"[bench] cpu_baseline: batch {reps} done at {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    evals_per_s = reps * n / (time.perf_counter() - t0)
    print("[bench] cpu_baseline: full SVD ...", file=sys.stderr, flush=True)
    S = torch.randn(rows_per_point, D)
    t0 = time.perf_counter()
    c = S - S.mean(0, keepdim=True)
    torch.linalg.svd(c)                      # full_matrices=True, as dim_reduction.py:197
    svd_s = time.perf_counter() - t0
    per_point = rows_per_point / evals_per_s + svd_s
    return {"value": rows_per_point / per_point, "unit": "score-vector evals/s", "cores": cores, "kind": "port",
            "sample": f"oracle NCSN++ score_fn under no_grad, {reps} batches of {n} rows ({evals_per_s:.2f} evals/s) + one "
                      f"full torch.linalg.svd of {rows_per_point}x{D} ({svd_s:.2f} s); per-point rate extrapolated",
            "score_evals_per_s": evals_per_s, "svd_s_per_point": svd_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=2240, help="score rows per launch set (measured best of 512..4480)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run each point's spectrum on the main stream")
    args = ap.parse_args()

    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
    cfg.model.init_scale = 1.0      # random weights with every branch numerically active (SURVEY 8-d cfg 3)
    torch.manual_seed(0)
    model = mutils.create_model(cfg).to(dev).eval()
    sde, eps = sde_lib.configure_sde(cfg)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    B = cfg.training.batch_size
    images = smooth_decoder_images(args.steps + args.warmup, [3, 32, 32], 64, seed=100 + rank).to(dev)
    _, _, rows = dim_reduction.batching((3, 32, 32), B)
    D = 3 * 32 * 32
    builder = dim_reduction.ScoreMatrixBuilder(score_fn, sde, eps, dev, inflight_rows=args.inflight)
    probe = ConvProbe()
    probe.install()

    pipe = dim_reduction.SpectrumPipeline(dev, overlap=not args.no_overlap)

    def one_point(i, timed):
        # sample the dominant kernel's launches of every timed step
        if timed:
            probe.active = True
        S = builder.build(images[i], B, seed=1234 + 1000003 * (i + 1) + rank)
        probe.active = False
        pipe.submit(S)   # spectrum of this point overlaps the score evaluations of the next one

    # probe only one inflight chunk per step: wrap builder.score_fn
    orig_score_fn = builder.score_fn
    state = {"calls": 0}

    def sampled_score_fn(x, t):
        # every launch set of a timed step is sampled (the first runs beside the previous point's spectrum, the last
        # mostly alone), so that the average is over the same launches rocprofv3 --stats averages
        state["calls"] += 1
        return orig_score_fn(x, t)

    builder.score_fn = sampled_score_fn
    n_chunks = (rows + builder.rows_per_launch(rows, D) - 1) // builder.rows_per_launch(rows, D)

    with torch.no_grad():
        for i in range(args.warmup):
            state["calls"] = 0
            one_point(i, False)
        pipe.results()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            state["calls"] = 0
            one_point(i, True)
        local = torch.stack(pipe.results())
        if world > 1:
            # the path's one exchange step: all ranks own `steps` points
            allsv = torch.empty(world * args.steps, D, device=dev)
            dist.all_gather_into_tensor(allsv, local)
        else:
            allsv = local
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # SVD wall-clock of one point, measured alone (outside the timed region) so that it is not a concurrency artefact
    with torch.no_grad():
        S_probe = torch.randn(rows, D, device=dev)
        _lib.spectrum(S_probe)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _lib.spectrum(S_probe)
        e1.record()
        torch.cuda.synchronize()
        svd_ms = e0.elapsed_time(e1) / 3
    if rank == 0:
        ids = [plot_utils.estimate_dim(s.tolist()) for s in allsv[: args.steps].cpu()]
        conv = probe.summary()
        roofline = None
        if conv:
            roofline = {"bound": "mfma", "kernel": "winograd_kernel (3x3 conv as F(2x2,3x3): 16 [tiles x Cin] x [Cin x Cout] contractions "
                                                   "per launch, v_mfma_f32_32x32x2_f32)",
                        "achieved": conv["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": conv["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                        "flops_counted": "executed Winograd-domain multiply-adds (the implicit GEMM of the same conv is 2.25x more)",
                        "direct_conv_equivalent_tflops": conv["tflops"] * 2.25,
                        "traffic": conv["traffic"]["bytes_per_launch"] if conv["traffic"] else None,
                        "traffic_detail": conv["traffic"],
                        "launches_sampled": conv["launches"], "avg_launch_us": conv["avg_us"]}
        line = {
            "metric": "score-vector evals/sec (rows of S per second incl. the per-point spectrum), 32x32 ncsnpp",
            "value": world * args.steps * rows / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD, "rows_per_point": rows, "cols": D, "batch_size": B,
                       "inflight_rows": args.inflight, "points_per_gpu": args.steps,
                       "parallelism": f"points sharded over {world} rank(s), one all-gather of spectra"},
            "svd_wall_clock_ms_per_point": svd_ms, "id_estimates": ids,
            "model_tflops_per_gpu": rows * args.steps * 21.79e9 / elapsed / 1e12,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, rows, D)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
