"""Headline benchmark: score-vector evals/sec (+ SVD wall-clock) of the manifold_dimension path, 32x32x3 NCSN++.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one data point of BASELINE config 3: 4480 score rows (B=128 -> (1024//128+1)*4 batches, last one
empty; dim_reduction.py:166-171) of the nf=128 NCSN++ on a synthetic 32x32x3 image, written into the
device-resident S [4480, 3072], then its centred singular spectrum and the integer ID.  Inputs (image, weights)
are resident in HBM before the timed region.  With N ranks every rank processes K points of its own (weak
scaling) and the region ends with the one exchange step the path has (parallel.gather_spectra).

Passes, in order:
  1. warm-up (W points), then the TIMED region (K points, no instrumentation)            -> value, ms_per_step
  2. the same K points again with HIP events around the instrumented launches (rank 0)   -> roofline, roofline.kernels
  3. the spectrum stages of one point, alone on the device                                -> svd_wall_clock_ms_per_point
  4. extra.cfg2 / extra.cfg5: KSphere + fcn and 64x64 BeatGANs side measurements (rank 0, N = 1, seconds each)
  5. cpu_baseline: the oracle on this box's host cores, bounded sample (rank 0, N = 1)
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import statistics
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

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters; fp64 matrix peak: public MI355X figure (= fp64 vector peak)
FP32_MFMA_PEAK_TFLOPS = 157.3
FP64_MFMA_PEAK_TFLOPS = 78.6
# split-precision contractions (igemm.hip, SPLIT): six bf16 MFMA products per fp32 product -> the fp32-equivalent ceiling is
# the dense bf16 peak (2.5 PFLOP/s) / 6
BF16_MFMA_PEAK_TFLOPS = 2500.0
F16_MFMA_PEAK_TFLOPS = 2500.0        # the fp16 forms take the same cycles as the bf16 ones (MI355X_MICROARCH.md, matrix-core table)
L2_READ_PEAK_GBS = 18800.0           # upper end of the guide's measured 16.8-18.8 TB/s for rows every workgroup reads from L2
SPLIT_EQUIV_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
HBM_PEAK_GBS = 8000.0
# fp32 tensors end to end.  3x3 convolutions: exact-fp32 matrix cores (Winograd F(2x2,3x3)).  1x1 / NIN / attention / dense
# contractions: every fp32 operand is cut EXACTLY into three bf16 pieces and six of the nine partial products (all of weight
# >= 2^-16) run on the bf16 matrix cores with fp32 accumulation -- 1.7e-7 against an fp64 contraction where the fp32 MFMA
# chain gives 2.0e-7 (scripts/bf16x6_probe.hip); every parity bar of tests/ is unchanged.  IDIFF_NO_SPLIT=1 selects fp32 MFMAs.
_CONV_ARITH = ("3x3 convs: Winograd F(4x4,3x3), fp32 transforms, fp32 MFMA contraction" if (os.environ.get("IDIFF_NO_WINO43H") or os.environ.get("IDIFF_NO_WINO43"))
               else ("3x3 convs: Winograd, fp32 transforms, contraction on pairs of fp16 values -- 22 significand bits, 3 products, fp32 accumulate"
                     + (" (F(4x4,3x3))" if os.environ.get("IDIFF_NO_WINO1D") else " (F(4,3) along the rows x 3 filter rows)")))
_GEMM_ARITH = ("1x1/attention/dense contractions: fp32 MFMA" if os.environ.get("IDIFF_NO_SPLIT")
               else "1x1/attention/dense contractions: fp32 operands split exactly into 3 bf16, 6 partial products, fp32 accumulate"
               + ("" if os.environ.get("IDIFF_NO_PAIRS") else "; the q/k/v projections of a GroupNorm's output: pairs of fp16 values, 3 products")
               + ("" if (os.environ.get("IDIFF_NO_PAIRS") or os.environ.get("IDIFF_NO_FUSED_ATTN")) else "; QK^T / softmax / PV of the 256-token attention blocks: one launch on fp16 pairs, softmax in fp32"))
DTYPE = f"f32 ({_CONV_ARITH}; {_GEMM_ARITH})"
WORKLOAD = "ncsnpp nf128 ch(1,2,2,2) 4 resblocks attn@16 FIR, 32x32x3, VE-SDE t=1e-5, B=128 -> S 4480x3072 + centred spectrum + ID"
# PMC traffic tables (separate FETCH_SIZE / WRITE_SIZE passes, scripts/profile_round.sh), each stamped with the sha256 of the kernel
# source it was measured on: F(4x4,3x3) (the dominant kernel) and F(2x2,3x3)
_CSRC = os.path.join("id-diff_amd", "csrc")
TRAFFIC_TABLES = {"wino1d_kernel": (os.path.join("profiles", "r05_wino1d_traffic.json"), (os.path.join(_CSRC, "wino1d.hip"),)),
                  "winograd43h_kernel": (os.path.join("profiles", "r05_wino43h_traffic.json"),
                                         (os.path.join(_CSRC, "winograd43h.hip"), os.path.join(_CSRC, "winograd43_shared.h"))),
                  "winograd43_kernel": (os.path.join("profiles", "r05_wino43_traffic.json"),
                                        (os.path.join(_CSRC, "winograd43.hip"), os.path.join(_CSRC, "winograd43_shared.h"))),
                  "winograd_kernel": (os.path.join("profiles", "r03_wino_traffic.json"), (os.path.join(_CSRC, "winograd.hip"),))}


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def source_sha256(rels):
    """sha256 over the concatenated sources of a kernel (scripts/parse_wino_traffic.py stamps its tables the same way)"""
    h = hashlib.sha256()
    for rel in rels:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


PAIRS_GEMM = "igemm_pipe_kernel on fp16 pairs (q / k / v projections of a GroupNorm's output)"
FUSED_ATTN = "attention256_kernel (QK^T -> softmax -> PV of a 256-token block in one launch, logits on chip, both contractions on fp16 pairs)"


class KernelProbe:
    """HIP events (on the launch stream) around every launch of the instrumented ``_lib`` entry points.

    Each entry is classified by the roofline that bounds it and carries its ALGORITHMIC work: executed flops for the
    contractions, compulsory HBM bytes (each operand once) for the streaming kernels -- SURVEY.md 8(d) per-unit figures
    times the units of the launch."""

    def __init__(self):
        self.active = False
        self.records = {}          # group -> list of (start, end, work, key)
        self._orig = {}

    def _wrap(self, name, account):
        orig = getattr(_lib, name)
        self._orig[name] = orig
        probe = self

        def timed(*a, **k):
            if not probe.active:
                return orig(*a, **k)
            acc = account(*a, **k)
            if acc is None:
                return orig(*a, **k)
            group, work, key = acc
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            probe.records.setdefault(group, []).append((e0, e1, work, key))
            return r

        setattr(_lib, name, timed)

    def install(self):
        def wino(x, u, out, B, H, W, Cin, Cout, epilogue=None, split=False):
            # executed flops of F(2x2,3x3): 16 positions x [tiles x Cin] x [Cin x Cout]; the implicit GEMM would be 2.25x that
            return "winograd_kernel", 2.0 * 16 * (B * H * W // 4) * Cin * Cout, f"{B}x{H}x{W}x{Cin}->{Cout}"

        def wino43(x, u, out, B, H, W, Cin, Cout, epilogue=None, pairs=False):
            # executed flops of F(4x4,3x3): 36 positions x [tiles x Cin] x [Cin x Cout]; the implicit GEMM would be 4x that.  On fp16
            # pairs (winograd43h_kernel) each of them is three fp16 products: counted once, as one fp32-equivalent multiply-add
            return ("winograd43h_kernel" if pairs else "winograd43_kernel"), 2.0 * 36 * (B * H * W // 16) * Cin * Cout, f"{B}x{H}x{W}x{Cin}->{Cout}"

        def wino1d(x, u, out, B, H, W, Cin, Cout, epilogue=None):
            # executed flops of the row-wise F(4,3): 6 positions x 3 filter rows of [row-tiles x Cin] x [Cin x Cout]; on fp16 pairs each multiply-add
            # is three fp16 products: counted once, as one fp32-equivalent multiply-add.  The implicit GEMM would be 2x that.
            return "wino1d_kernel", 2.0 * 18 * (B * H * W // 4) * Cin * Cout, f"{B}x{H}x{W}x{Cin}->{Cout}"

        def gn_apply(x, C, x2, C2, B, HW, G, stats, gamma, beta, act, y, mod=None):
            return "gn_apply_rows", 8.0 * B * HW * (C + (C2 or 0)), f"{B}x{HW}x{C + (C2 or 0)}"

        def gn_apply_cs(x, C, x2, C2, B, HW, G, ws1, ns1, ws2, ns2, eps, gamma, beta, act, y, mod=None):
            return "gn_apply_rows", 8.0 * B * HW * (C + (C2 or 0)), f"{B}x{HW}x{C + (C2 or 0)}"

        def gemm(a, bt, out=None, epilogue=None, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, batch=1,
                 stride_a=0, stride_b=0, stride_c=0):
            if M is None:
                M, K = a.shape
                N = bt.shape[0]
            if M * batch < 65536:                               # the network's 1x1 / NIN / attention contractions, not the embeddings
                return None
            res = 4.0 * M * N * batch if (epilogue is not None and epilogue.residual) else 0.0
            group = "igemm_pipe_kernel K<=128 (1x1 / NIN)" if K <= 128 else "igemm_pipe_kernel K>=256 (1x1 / NIN / attention products)"
            return group, (2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N) + res), f"{M}x{N}x{K}"

        def gemm_pairs_2src(a1, a2, act_scale, bt, w_scale, out, epilogue=None):
            M, K1 = a1.shape
            K, N = K1 + a2.shape[1], bt.shape[0]
            return PAIRS_GEMM, (2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)), f"2src {M}x{N}x{K}"

        def gemm_pairs(a, bt, w_scale, out, epilogue=None, weight_is_a=False, M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
                       batch=1, stride_a=0, stride_b=0, stride_c=0, act_scale=None):
            if M is None:
                M, K = a.shape
                N = bt.shape[0]
            return (PAIRS_GEMM,
                    (2.0 * batch * M * N * K, 4.0 * (batch * (M * N + (N * K if weight_is_a else M * K)) + (M * K if weight_is_a else N * K))), f"{M}x{N}x{K}")

        def attn(qk, vt, out, B, C, s_qk, s_v, scale, bias_v=None):
            # two contractions of 2 * 256 * 256 * C flops per sample; compulsory bytes: q, k, v in and the mixed values out, each once
            return FUSED_ATTN, (2.0 * 2.0 * B * 256 * 256 * C, 4.0 * 4.0 * B * 256 * C), f"{B}x256x{C}"

        def gemm_2src(a1, a2, bt, out, epilogue=None):
            M, K1 = a1.shape
            K, N = K1 + a2.shape[1], bt.shape[0]
            res = 4.0 * M * N if (epilogue is not None and epilogue.residual) else 0.0
            return "igemm_pipe_kernel two-source shortcut", (2.0 * M * N * K, 4.0 * (M * K + N * K + M * N) + res), f"{M}x{N}x{K}"

        def ufd(x, k, out, major, in_h, in_w, minor, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
            return "upfirdn2d_nhwc", 4.0 * (x.numel() + out.numel()) + 4.0 * k.numel(), f"{major}x{in_h}x{in_w}x{minor} up{up_x} down{down_x}"

        def softmax(x, y, rows, cols, scale):
            return "softmax_rows", 8.0 * rows * cols, f"{rows}x{cols}"

        for name, fn in (("conv2d_winograd", wino), ("conv2d_winograd43", wino43), ("conv2d_wino1d", wino1d), ("groupnorm_apply", gn_apply), ("groupnorm_apply_colstats", gn_apply_cs),
                         ("gemm", gemm), ("gemm_2src", gemm_2src), ("gemm_pairs", gemm_pairs), ("gemm_pairs_2src", gemm_pairs_2src),
                         ("upfirdn2d_raw", ufd), ("softmax_rows", softmax), ("attention256", attn)):
            self._wrap(name, fn)

    def uninstall(self):
        for name, orig in self._orig.items():
            setattr(_lib, name, orig)

    def group(self, name):
        recs = self.records.get(name, [])
        if not recs:
            return None
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        out = {"launches": len(recs), "avg_us": ms * 1e3 / len(recs), "keys": [r[3] for r in recs]}
        if isinstance(recs[0][2], tuple):       # contractions carry (flops, compulsory bytes): the roofline picks the bound
            out["flops_rate"] = sum(r[2][0] for r in recs) / (ms * 1e-3)
            out["rate"] = sum(r[2][1] for r in recs) / (ms * 1e-3)
        else:
            out["rate"] = sum(r[2] for r in recs) / (ms * 1e-3)
        return out


def winograd_traffic(keys, kernel="winograd_kernel"):
    """HBM bytes per launch of the sampled shapes from the committed PMC table (separate FETCH_SIZE / WRITE_SIZE passes,
    gfx950 corrections applied) -- ONLY if the table was measured on the kernel source that is in the tree now."""
    table_rel, source_rel = TRAFFIC_TABLES[kernel]
    path = os.path.join(ROOT, table_rel)
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if doc.get("kernel_source_sha256") != source_sha256(source_rel):
        return None
    table = doc["shapes"]
    if any(k not in table for k in keys):
        return None
    return {"bytes_per_launch": sum(table[k]["total_bytes"] for k in keys) / len(keys),
            "algorithmic_bytes_per_launch": sum(table[k]["algorithmic_bytes"] for k in keys) / len(keys),
            "source": table_rel, "kernel_source_sha256": doc["kernel_source_sha256"]}


def winograd43h_l2_bytes(keys):
    """Bytes winograd43h_kernel pulls from L2 per launch, by construction: per workgroup (32 tiles x 64 output channels) and K step
    (16 channels) the filter slab 36 x 64 x 16 x 4 B (fp16 pairs) and the 32 input patches 36 x 16 x 4 B each."""
    tot = 0.0
    for k in keys:
        B, H, W, rest = k.split("x")
        Cin, Cout = rest.split("->")
        B, H, W, Cin, Cout = map(int, (B, H, W, Cin, Cout))
        wgs = ((B * (H // 4) * (W // 4) + 31) // 32) * (Cout // 64)
        tot += wgs * (Cin // 16) * (36 * 64 * 16 * 4 + 32 * 36 * 16 * 4)
    return tot / len(keys)


def wino1d_l2_bytes(keys):
    """Bytes wino1d_kernel pulls from L2 per launch, by construction: per workgroup (512 pixels x 64 output channels) and K step (16
    channels) the filter slab 18 x 64 x 16 x 4 B (fp16 pairs) and the block's image rows plus one halo row where the block is part of an image."""
    tot = 0.0
    for k in keys:
        B, H, W, rest = k.split("x")
        Cin, Cout = rest.split("->")
        B, H, W, Cin, Cout = map(int, (B, H, W, Cin, Cout))
        rb = 512 // W
        wgs = ((B * H + rb - 1) // rb) * (Cout // 64)
        rows = rb + (1 if H > rb else 0)
        tot += wgs * (Cin // 16) * (18 * 64 * 16 * 4 + rows * W * 16 * 4)
    return tot / len(keys)


def roofline_report(probe):
    dom1d = probe.group("wino1d_kernel")
    dom43h, dom43, dom22 = probe.group("winograd43h_kernel"), probe.group("winograd43_kernel"), probe.group("winograd_kernel")
    dom = dom1d or dom43h or dom43 or dom22
    if dom is None:
        return None
    tfl = dom["rate"] / 1e12
    dom_name = ("wino1d_kernel" if dom1d is not None else
                "winograd43h_kernel" if dom43h is not None else ("winograd43_kernel" if dom43 is not None else "winograd_kernel"))
    traffic = winograd_traffic(dom["keys"], dom_name)
    kernels = []
    if dom1d is not None and dom43h is not None:
        t = dom43h["rate"] / 1e12
        kernels.append({"kernel": "winograd43h_kernel (F(4x4,3x3) on fp16 pairs: GroupNorm-fed 3x3 convs the row-wise kernel does not serve) [3 fp16 products per "
                                  "fp32 multiply-add]", "bound": "mfma", "achieved": t, "peak": F16_MFMA_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s (fp32-equivalent)",
                        "frac": t / (F16_MFMA_PEAK_TFLOPS / 3.0), "launches_sampled": dom43h["launches"], "avg_launch_us": dom43h["avg_us"]})
        dom43h = None
    if dom43h is not None and dom43 is not None:
        t43 = dom43["rate"] / 1e12
        kernels.append({"kernel": "winograd43_kernel (F(4x4,3x3), fp32 contraction: inputs not fed by a GroupNorm, Cin % 16 != 0)", "bound": "mfma",
                        "achieved": t43, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": t43 / FP32_MFMA_PEAK_TFLOPS,
                        "launches_sampled": dom43["launches"], "avg_launch_us": dom43["avg_us"]})
    if dom_name != "winograd_kernel" and dom22 is not None:
        t22 = dom22["rate"] / 1e12
        kernels.append({"kernel": "winograd_kernel (F(2x2,3x3): the 4x4 maps and launches too small for the 4x4 form)", "bound": "mfma",
                        "achieved": t22, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": t22 / FP32_MFMA_PEAK_TFLOPS,
                        "launches_sampled": dom22["launches"], "avg_launch_us": dom22["avg_us"]})
    for name in ("gn_apply_rows", "igemm_pipe_kernel K<=128 (1x1 / NIN)", "igemm_pipe_kernel K>=256 (1x1 / NIN / attention products)",
                 "igemm_pipe_kernel two-source shortcut", PAIRS_GEMM, FUSED_ATTN, "upfirdn2d_nhwc", "softmax_rows"):
        g = probe.group(name)
        if not g:
            continue
        if name == FUSED_ATTN:
            # 64 flop per compulsory byte: at 8 TB/s the memory roof (512 TFLOP/s fp32-equivalent) lies below the fp16 peak / 3 (833)
            tf, gbs = g["flops_rate"] / 1e12, g["rate"] / 1e9
            kernels.append({"kernel": name + " [3 fp16 products per fp32 multiply-add]", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "flop_per_byte": g["flops_rate"] / g["rate"], "tflops": tf, "gbs": gbs,
                            "frac_of_f16_mfma_peak_over_3": tf / (F16_MFMA_PEAK_TFLOPS / 3.0),
                            "launches_sampled": g["launches"], "avg_launch_us": g["avg_us"]})
            continue
        if name == PAIRS_GEMM:
            tf, gbs = g["flops_rate"] / 1e12, g["rate"] / 1e9
            kernels.append({"kernel": name + " [3 fp16 products per fp32 multiply-add]", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "flop_per_byte": g["flops_rate"] / g["rate"], "tflops": tf, "gbs": gbs,
                            "launches_sampled": g["launches"], "avg_launch_us": g["avg_us"]})
            continue
        if "flops_rate" in g:
            # roofline of a contraction: attainable = min(MFMA peak, arithmetic intensity x HBM peak) over the sampled launches
            tf, gbs = g["flops_rate"] / 1e12, g["rate"] / 1e9
            intensity = g["flops_rate"] / g["rate"]                    # flop per compulsory byte
            # these contractions run as split-precision products on the bf16 matrix cores unless IDIFF_NO_SPLIT is set:
            # fp32-equivalent flops against the bf16 peak / 6
            split = not os.environ.get("IDIFF_NO_SPLIT")
            peak = SPLIT_EQUIV_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
            mfma_bound = intensity * HBM_PEAK_GBS / 1e3 >= peak
            kernels.append({"kernel": name + (" [3 x bf16 split operands, 6 partial products]" if split else ""),
                            "bound": "mfma" if mfma_bound else "hbm",
                            "achieved": tf if mfma_bound else gbs, "peak": peak if mfma_bound else HBM_PEAK_GBS,
                            "unit": "TFLOP/s (fp32-equivalent)" if mfma_bound else "GB/s",
                            "frac": tf / peak if mfma_bound else gbs / HBM_PEAK_GBS,
                            "flop_per_byte": intensity, "tflops": tf, "gbs": gbs,
                            "launches_sampled": g["launches"], "avg_launch_us": g["avg_us"]})
        else:
            kernels.append({"kernel": name, "bound": "hbm", "achieved": g["rate"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": g["rate"] / 1e9 / HBM_PEAK_GBS, "launches_sampled": g["launches"], "avg_launch_us": g["avg_us"]})
    peak, extra = FP32_MFMA_PEAK_TFLOPS, {}
    if dom1d is not None:
        name = ("wino1d_kernel (3x3 conv as F(4,3) along the rows, the three filter rows a direct sum: 18 [row-tiles x Cin] x [Cin x Cout] contractions "
                "per launch on v_mfma_f32_32x32x16_f16, each fp32 operand a pair of fp16 values, 3 fp16 products per fp32 multiply-add)")
        counted, ratio = ("executed Winograd-domain multiply-adds, each counted ONCE (fp32-equivalent; the matrix cores execute three fp16 products for "
                          "it, so the peak is the dense fp16 peak / 3; 4.5 per output pixel and channel pair -- F(4x4,3x3) would execute 2.25, the implicit "
                          "GEMM 9)"), 2.0
        peak = F16_MFMA_PEAK_TFLOPS / 3.0
        l2 = wino1d_l2_bytes(dom["keys"]) / (dom["avg_us"] * 1e-6) / 1e9
        extra = {"l2_read": {"achieved": l2, "peak": L2_READ_PEAK_GBS, "unit": "GB/s", "frac": l2 / L2_READ_PEAK_GBS,
                             "bytes": "per workgroup (512 pixels x 64 channels) and 16-channel step: 73,728 B of filter pairs + 34,816 - 36,864 B of input "
                                      "rows (every pixel once, one halo row), by construction"}}
    elif dom43h is not None:
        name = ("winograd43h_kernel (3x3 conv as F(4x4,3x3): 36 [tiles x Cin] x [Cin x Cout] contractions per launch on "
                "v_mfma_f32_32x32x16_f16, each fp32 operand a pair of fp16 values, 3 fp16 products per fp32 multiply-add)")
        counted, ratio = ("executed Winograd-domain multiply-adds, each counted ONCE (fp32-equivalent; the matrix cores execute three fp16 "
                          "products for it, so the peak is the dense fp16 peak / 3; the implicit GEMM of the same conv is 4x more)"), 4.0
        peak = F16_MFMA_PEAK_TFLOPS / 3.0
        # the kernel's actual limiter: every CU pulls its filter slabs and input patches through L2 (no reuse within a workgroup beyond
        # its 32 tiles x 64 channels); MI355X_MICROARCH.md measures 16.8-18.8 TB/s for rows every workgroup reads from its XCD's L2
        l2 = winograd43h_l2_bytes(dom["keys"]) / (dom["avg_us"] * 1e-6) / 1e9
        extra = {"l2_read": {"achieved": l2, "peak": L2_READ_PEAK_GBS, "unit": "GB/s", "frac": l2 / L2_READ_PEAK_GBS,
                             "bytes": "per workgroup and 16-channel step: 147,456 B of filter pairs + 73,728 B of input patches, by construction"}}
    elif dom43 is not None:
        name = ("winograd43_kernel (3x3 conv as F(4x4,3x3): 36 [tiles x Cin] x [Cin x Cout] contractions per launch, "
                "v_mfma_f32_32x32x2_f32)")
        counted, ratio = "executed Winograd-domain multiply-adds (the implicit GEMM of the same conv is 4x more)", 4.0
    else:
        name = ("winograd_kernel (3x3 conv as F(2x2,3x3): 16 [tiles x Cin] x [Cin x Cout] contractions per launch, "
                "v_mfma_f32_32x32x2_f32)")
        counted, ratio = "executed Winograd-domain multiply-adds (the implicit GEMM of the same conv is 2.25x more)", 2.25
    out = {"bound": "mfma", "kernel": name,
           "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak,
           "flops_counted": counted,
           "direct_conv_equivalent_tflops": tfl * ratio,
           "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
           "launches_sampled": dom["launches"], "avg_launch_us": dom["avg_us"],
           "sampled_in": "a second, untimed pass over the same points (spectrum overlap on, as in the timed region)",
           "kernels": kernels}
    out.update(extra)
    return out


def spectrum_stage_report(rows, D, dev, reps=3):
    """The spectrum of one point alone on the device, stage by stage (events on the launch stream)."""
    S = torch.randn(rows, D, device=dev)

    def timed(fn, reps=reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    whole = timed(lambda: _lib.spectrum(S))
    mean = _lib.column_sums(S) / rows
    gram_ms = timed(lambda: _lib.centered_gram(S, mean))
    G = _lib.centered_gram(S, mean)
    band_ms = timed(lambda: _lib.sym_band(G.clone(), dense=False)) - timed(lambda: G.clone())
    eig_ms = timed(lambda: _lib.sym_eigvals(G.clone())) - timed(lambda: G.clone())
    # algorithmic work: Gram = 2 M D^2 / 2 flops (upper-triangular tiles); band reduction streams the LOWER TRIANGLE of the
    # trailing block three times per 32-column panel (read for Y = A'V, read + write for the rank-64 update): 12 bytes x
    # sum_k m_k^2 (24 in round 2, when both triangles were kept)
    m2 = sum((D - 32 * (k + 1)) ** 2 for k in range(max(0, (D - 128 + 31) // 32)))
    stages = [
        {"kernel": "gram_big_kernel + mirror pass (fp64 centred Gram, upper-triangular 128x128 tiles, v_mfma_f64_16x16x4_f64)", "bound": "mfma", "achieved": rows * D * D / (gram_ms * 1e-3) / 1e12,
         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rows * D * D / (gram_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms": gram_ms},
        {"kernel": "band reduction, stage 1 of the eigensolver (launches per 32-column panel: see DESIGN 4.2; latency-bound at D = 3072)", "bound": "hbm",
         "achieved": 12.0 * m2 / (band_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": 12.0 * m2 / (band_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": band_ms},
        {"kernel": "systolic bulge chasing + Sturm bisection (one persistent launch; a chain of 2D dependent steps)", "bound": "latency",
         "ms": eig_ms - band_ms, "us_per_sweep": (eig_ms - band_ms) * 1e3 / D},
    ]
    return whole, stages


def _events_ms(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def extra_cfg2(dev):
    """BASELINE config 2 (north_star: "throughput on synthetic KSphere"): the 50-sphere in R^100, random-weight fcn 2048 x 5,
    B = 500 -> S 1501 x 100 per point.  (i) the drop-in driver end to end on P = 512 points (points batched per launch set,
    batched spectrum kernel), (ii) 4096 batched spectra of 1501 x 100, (iii) the LDS-resident tridiagonalisation alone."""
    cfg = read_config('configs/dimension_estimation/paper/euclidean_data/ksphere/50dim.py')
    cfg.model.allow_random_init = True
    cfg.device = str(dev)
    cfg.data.data_samples = 8000
    cfg.dim_estimation.num_datapoints = 513                       # the reference's loop stops one short: 512 points
    dim_reduction.get_manifold_dimension(cfg, return_svd=True)    # first call: data / model set-up, allocator warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    svd = dim_reduction.get_manifold_dimension(cfg, return_svd=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    P = len(svd['singular_values'])
    M, D, NP = 1501, 100, 4096
    S = torch.randn(NP, M, D, device=dev)
    spectra_ms = _events_ms(lambda: _lib.spectrum(S))
    mean = torch.empty(NP, D, dtype=torch.float64, device=dev)
    scratch = torch.empty(NP * 32 * D, dtype=torch.float64, device=dev)
    G = torch.empty(NP, D, D, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    lib.idiff_colmean_f64(S.data_ptr(), NP, M, D, mean.data_ptr(), scratch.data_ptr(), st)
    gram_ms = _events_ms(lambda: lib.idiff_centered_gram_f64(S.data_ptr(), mean.data_ptr(), NP, M, D, G.data_ptr(), st))
    diag, offd = torch.empty(NP, D, dtype=torch.float64, device=dev), torch.empty(NP, D, dtype=torch.float64, device=dev)
    G2 = G.clone()
    clone_ms = _events_ms(lambda: G2.copy_(G))

    def tri():
        G2.copy_(G)                                               # the kernel overwrites its input
        lib.idiff_symtridiag_f64(G2.data_ptr(), NP, D, diag.data_ptr(), offd.data_ptr(), None, st)
    tri_ms = _events_ms(tri) - clone_ms
    # Householder tridiagonalisation: (4/3) D^3 flops per matrix (a symmetric matrix-vector product and a symmetric rank-2 update per
    # step); the register-resident kernel spends 5 n FMAs per row and step instead of the symmetric minimum (it keeps both
    # triangles and recomputes the step's scalars in every lane pair), so the fraction is of ALGORITHMIC flops
    tri_flops = NP * (4.0 / 3.0) * D ** 3
    del S, G, G2, scratch
    # north_star: "correct ID recovered": the same driver on the exact score of the noised k-sphere (models/ksphere_exact.py)
    cfg_e = read_config('configs/dimension_estimation/paper/euclidean_data/ksphere/50dim.py')
    cfg_e.model.name = 'ksphere_exact'
    cfg_e.device = str(dev)
    cfg_e.data.data_samples = 8000
    cfg_e.dim_estimation.num_datapoints = 65                      # 64 points
    dims_e = plot_utils.plot_dims(dim_reduction.get_manifold_dimension(cfg_e, return_svd=True))[1]
    return {"workload": "KSphere 50-sphere in R^100, random-weight fcn 2048x5, VE-SDE t=1e-5, B=500 -> S 1501x100 per point",
            "points": P, "driver_seconds": dt, "evals_per_s_end_to_end": P * M / dt,
            "id_estimates_note": "random-weight fcn: throughput only (its ID means nothing and is not reported); the ID acceptance is ksphere_exact below",
            "ksphere_exact": {"model": "exact score of the noised 50-sphere (models/ksphere_exact.py), same driver and recipe",
                              "points": len(dims_e), "id_estimates_min_max": [int(min(dims_e)), int(max(dims_e))], "true_dimension": 50},
            "batched_spectra": {"matrices": NP, "shape": [M, D], "ms": spectra_ms},
            "kernels": [
                {"kernel": "gram_small_batched_kernel (fp64 centred Gram, one workgroup per 1501x100 matrix, 28 upper-triangular 16x16 "
                           "blocks on v_mfma_f64_16x16x4_f64, operands centred once)", "bound": "mfma",
                 "achieved": NP * M * D * D / (gram_ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": NP * M * D * D / (gram_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms": gram_ms,
                 "hbm_gbs": 4.0 * NP * M * D / (gram_ms * 1e-3) / 1e9},
                {"kernel": "tridiag_reg_kernel (one 100x100 matrix per 256 threads, rows in registers, a lane pair per row, vectors "
                           "exchanged as LDS broadcasts)", "bound": "valu-fp64",
                 "achieved": tri_flops / (tri_ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (algorithmic 4/3 D^3)",
                 "frac": tri_flops / (tri_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms": tri_ms,
                 "hbm_gbs": 8.0 * NP * D * D / (tri_ms * 1e-3) / 1e9}]}


def extra_cfg5(dev):
    """BASELINE config 5 on one GPU: the 64x64x3 BeatGANs U-Net (87.5 M parameters, random weights with the zero-initialised
    convs randomised) score evaluations, and the spectrum of one 16768 x 12288 score matrix, stage by stage."""
    cfg = read_config('configs/dimension_estimation/extra_experiments/styleGAN/style_gan_64d_BeatGAN.py')
    torch.manual_seed(0)
    model = mutils.create_model(cfg)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for prm in model.parameters():
            if float(prm.abs().sum()) == 0.0:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.02)
    model = model.to(dev).eval()
    sde, eps = sde_lib.configure_sde(cfg)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    rows_launch = (2240 * 3072) // (3 * 64 * 64)                  # the driver's launch-set size for this sample size (560)
    x = torch.rand(rows_launch, 3, 64, 64, device=dev)
    t = torch.full((rows_launch,), float(eps), device=dev)
    with torch.no_grad():
        fwd_ms = _events_ms(lambda: score_fn(x, t), reps=2)
        del model, score_fn, x
        torch.cuda.empty_cache()
        rows, D = dim_reduction.batching((3, 64, 64), cfg.training.batch_size)[2], 3 * 64 * 64
        whole_ms, stages = spectrum_stage_report(rows, D, dev, reps=1)
    evals = rows_launch / (fwd_ms * 1e-3)
    return {"workload": "StyleGAN-64d-shaped 64x64x3, BeatGANsUNetModel ch128 mult(1,1,2,3,4), VE-SDE t=1e-5, B=128 -> S 16768x12288",
            "score_evals_per_s": evals, "rows_per_launch": rows_launch, "ms_per_launch_set": fwd_ms,
            "model_tflops_direct_conv_equivalent": evals * 37.43e9 / 1e12,
            "svd_wall_clock_ms_per_point": whole_ms,
            "evals_per_s_one_point_no_overlap": rows / (rows / evals + whole_ms * 1e-3),
            "kernels": stages}


def cpu_baseline(cfg, rows_per_point, D, S_gpu):
    """the oracle (CPU restatement of the reference path) on this box's host cores, bounded sample.

    score_fn both under no_grad and with autograd enabled (the reference never disables it, dim_reduction.py:183);
    ``torch.linalg.svd`` with full matrices exactly as dim_reduction.py:197 and ``svdvals`` as the fair floor, both on
    the S the GPU path consumed for its last timed point; medians after one warm-up each."""
    from oracle import models as omodels, sde as osde
    # the GPU box exposes every host core (os.cpu_count() = 256) but grants a 16-core share per GPU:
    # oversubscribing the share makes the CPU run arbitrarily slow, so use the share
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    model_name = "unknown"
    try:
        model_name = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    torch.manual_seed(0)
    model = omodels.create_model(cfg)
    score_fn = osde.get_score_fn(osde.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales), model)

    # the reference evaluates B = 128 rows per call (dim_reduction.py:167-183): time that batch shape.  The two modes are
    # INTERLEAVED (no_grad, autograd, no_grad, ...) after one warm-up batch of each, median of 3 per mode: timed one mode after the
    # other, the first series was still warming the allocator and the thread pool and autograd came out faster than no_grad.
    B = int(cfg.training.batch_size)
    x, t = torch.rand(B, 3, 32, 32), torch.full((B,), 1e-5)

    def one(grad):
        t0 = time.perf_counter()
        if grad:
            score_fn(x, t).detach()
        else:
            with torch.no_grad():
                score_fn(x, t)
        return time.perf_counter() - t0

    times = {False: [], True: []}
    for rep in range(4):
        for grad in (False, True):
            dt = one(grad)
            if rep:
                times[grad].append(dt)
            print(f"[bench] cpu_baseline: score_fn {'autograd' if grad else 'no_grad'} batch {rep} of {B} rows: {dt:.2f}s"
                  f"{' (warm-up)' if not rep else ''}", file=sys.stderr, flush=True)
    nograd, withgrad = B / statistics.median(times[False]), B / statistics.median(times[True])
    S = S_gpu.cpu()
    c = S - S.mean(0, keepdim=True)

    def median_time(fn, reps):
        times = []
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline: svd rep {rep}: {times[-1]:.2f}s", file=sys.stderr, flush=True)
        return statistics.median(times[1:])

    svd_full = median_time(lambda: torch.linalg.svd(c), 3)           # full_matrices=True, as dim_reduction.py:197
    svd_vals = median_time(lambda: torch.linalg.svdvals(c), 3)
    as_reference = rows_per_point / (rows_per_point / withgrad + svd_full)
    floor = rows_per_point / (rows_per_point / nograd + svd_vals)
    return {"value": as_reference, "unit": "score-vector evals/s", "cores": cores, "cpu_model": model_name, "kind": "port",
            "sample": f"oracle NCSN++ score_fn on batches of B = {B} rows (the reference's batch shape), the two modes interleaved, median of 3 each after 1 warm-up: "
                      f"autograd on (as the reference runs, dim_reduction.py:183) {withgrad:.2f} evals/s, no_grad {nograd:.2f} "
                      f"evals/s; torch.linalg.svd "
                      f"(full matrices, dim_reduction.py:197) {svd_full:.2f} s and svdvals {svd_vals:.2f} s, median of 3, on the "
                      f"{rows_per_point}x{D} S of the last timed GPU point; per-point rates extrapolated",
            "score_evals_per_s_autograd": withgrad, "score_evals_per_s_no_grad": nograd, "svd_full_s": svd_full,
            "svdvals_s": svd_vals, "value_no_grad_svdvals": floor}


class Workload:
    """BASELINE config 3 on one rank."""

    def __init__(self, args, rank, dev):
        cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
        cfg.model.init_scale = 1.0      # random weights with every branch numerically active (SURVEY 8-d cfg 3)
        torch.manual_seed(0)
        self.cfg = cfg
        self.model = mutils.create_model(cfg).to(dev).eval()
        sde, eps = sde_lib.configure_sde(cfg)
        score_fn = mutils.get_score_fn(sde, self.model, conditional=False, train=False, continuous=True)
        self.B = cfg.training.batch_size
        self.images = smooth_decoder_images(args.steps + args.warmup, [3, 32, 32], 64, seed=100 + rank).to(dev)
        _, _, self.rows = dim_reduction.batching((3, 32, 32), self.B)
        self.D = 3 * 32 * 32
        self.builder = dim_reduction.ScoreMatrixBuilder(score_fn, sde, eps, dev, inflight_rows=args.inflight,
                                                        concurrent_sets=getattr(args, "concurrent_sets", 1))
        self.rank = rank
        self.pipe = dim_reduction.SpectrumPipeline(dev, overlap=not args.no_overlap)
        self.last_S = None

    def point(self, i):
        self.last_S = self.builder.build(self.images[i], self.B, seed=1234 + 1000003 * (i + 1) + self.rank)
        self.pipe.submit(self.last_S)   # spectrum of this point overlaps the score evaluations of the next one

    def collect(self):
        return torch.stack(self.pipe.results())


class HostRehearsal:
    """`--device cpu`: NOT a measurement.  A stand-in with no GPU work (12 'rows' per point, a fixed 5-value spectrum that encodes
    rank and point) so that the launch / process-group / barrier / exchange / JSON logic of this file can be run as plain
    processes on a box without GPUs (tests/test_parallel_gloo.py); the line says so in `data` and `config.workload`."""
    rows, D, B = 12, 5, 4
    data = "none: host rehearsal of the launch and exchange logic, no GPU work"

    def __init__(self, args, rank, dev):
        self.rank, self.out, self.cfg, self.last_S = rank, [], None, None

    def point(self, i):
        self.out.append(torch.tensor([50., 40., 30., 2., 1.]) + 0.01 * i + 0.001 * self.rank)

    def collect(self):
        out, self.out = torch.stack(self.out), []
        return out


def timed_region(work, warmup, steps, dev):
    """W untimed points, then EXACTLY `steps` points between barrier + synchronize on both sides; the region includes the
    path's one exchange step.  Returns (max-over-ranks seconds, spectra and int32 IDs of ALL ranks' points in point order)."""
    rank, world = parallel.rank_world()
    grouped = dist.is_available() and dist.is_initialized()
    for i in range(warmup):
        work.point(i)
    if warmup:
        warm = work.collect()
        # the exchange step is part of the warm-up too: the first all-gather of a communicator sets up its channels
        # (tens of ms once), which is start-up cost, not throughput
        parallel.gather_spectra(warm, world * warmup, warm.shape[1], dev)
    _sync(dev)
    if grouped:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        work.point(i)
    local = work.collect()
    # the integer ID of every point, by the reference's rule (float64 numpy on the host), on the rank that owns the point
    dims = [plot_utils.estimate_dim(s.tolist()) for s in local.cpu()]
    # point p of the weak-scaling job = (rank p % world, its step p // world): parallel.gather_spectra's round-robin layout
    allsv, alldims = parallel.gather_spectra(local, world * steps, local.shape[1], dev, dims=dims)
    _sync(dev)
    if grouped:
        dist.barrier()
    _sync(dev)
    elapsed = time.perf_counter() - t0
    if grouped:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed, allsv, alldims


def main(argv=None, workload_factory=Workload):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=2240, help="score rows per launch set (measured best of 512..4480)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run each point's spectrum on the main stream")
    ap.add_argument("--no-probe", action="store_true", help="skip the instrumented second pass (roofline = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-2 / config-5 side measurements (extra = null)")
    ap.add_argument("--concurrent-sets", type=int, default=1,
                    help="launch sets of a point on this many worker streams (2: -1.8 %% per point; per-kernel events then overlap)")
    ap.add_argument("--device", default=None, help="(tests) 'cpu' with a stand-in workload")
    args = ap.parse_args(argv)

    if args.gpus > 1 and not parallel.launched():
        # a plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (importing torch
        # does not; device_count only enumerates) and never will -- it starts N fresh rank processes of this file, relays
        # rank 0's one JSON line and exits with the first non-zero code of any rank.
        if workload_factory is not Workload:
            raise RuntimeError("bench.main(--gpus N > 1) without RANK / WORLD_SIZE starts rank PROCESSES of bench.py; a "
                               "stand-in workload_factory cannot travel to them -- set the launcher environment instead")
        rc = parallel.launch_local_ranks(os.path.abspath(__file__), list(sys.argv[1:] if argv is None else argv), args.gpus,
                                         need_devices=args.device != "cpu")
        if rc:
            raise SystemExit(rc)
        return None

    # under a launcher this initialises RCCL (also at world size 1) BEFORE anything else touches the GPU
    if parallel.launched():
        parallel.check_world(args.gpus, int(os.environ["WORLD_SIZE"]), need_devices=args.device != "cpu")
    rank, world, local_rank = parallel.init_from_env(backend="gloo" if args.device == "cpu" else None)
    dev = torch.device(args.device) if args.device else torch.device(f"cuda:{parallel.device_ordinal(local_rank)}")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
        # one process per GPU: every rank launches from its own host thread; the few CPU-side tensor ops (ID rule, pinned
        # flags) must not fan out over every core of the node in each of N processes
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, ncores // max(1, world))))
    elif workload_factory is Workload:
        workload_factory = HostRehearsal
    devices = parallel.rank_devices(dev)

    work = workload_factory(args, rank, dev)
    with torch.no_grad():
        elapsed, allsv, alldims = timed_region(work, args.warmup, args.steps, dev)
    rows, D = work.rows, work.D
    line = None
    if dev.type == "cuda":
        roofline, svd_ms, stages = None, None, None
        if rank == 0 and not args.no_probe:
            probe = KernelProbe()
            probe.install()
            probe.active = True
            with torch.no_grad():
                for i in range(args.warmup, args.warmup + args.steps):
                    work.point(i)
                work.collect()
            torch.cuda.synchronize()
            probe.active = False
            probe.uninstall()
            roofline = roofline_report(probe)
        if rank == 0:
            with torch.no_grad():
                svd_ms, stages = spectrum_stage_report(rows, D, dev)
            if roofline is not None:
                roofline["kernels"] += stages
    if rank == 0:
        sv_mine = allsv[0::world][: args.steps].cpu()         # rank 0's own points
        ids = [plot_utils.estimate_dim(s.tolist()) for s in sv_mine]
        line = {
            "metric": "score-vector evals/sec (rows of S per second incl. the per-point spectrum), 32x32 ncsnpp",
            "value": world * args.steps * rows / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE if dev.type == "cuda" else "none (host rehearsal)", "data": getattr(work, "data", "synthetic"),
            "config": {"workload": WORKLOAD if dev.type == "cuda" else HostRehearsal.data, "rows_per_point": rows, "cols": D, "batch_size": getattr(work, "B", None),
                       "inflight_rows": args.inflight, "concurrent_sets": getattr(args, "concurrent_sets", 1), "points_per_gpu": args.steps,
                       "parallelism": f"points sharded over {world} rank(s), one all-gather of spectra",
                       "process_group": dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None,
                       "world_size": world, "rank_devices": devices,
                       "launched_by": ("bench.py (self-launched rank processes)" if os.environ.get("IDIFF_SELF_LAUNCHED")
                                       else "external launcher" if parallel.launched() else "single process")},
            "id_estimates": ids, "id_estimates_all_ranks": alldims.tolist(),
        }
        if dev.type == "cuda":
            line["model_tflops_per_gpu_direct_conv_equivalent"] = rows * args.steps * 21.79e9 / elapsed / 1e12
            line["svd_wall_clock_ms_per_point"] = svd_ms
            line["roofline"] = roofline
            if world == 1 and not args.no_extras:
                # the other workloads north_star names, each bounded to a few seconds; not part of `value`
                last_S, work.last_S = work.last_S, None
                del work.model, work.builder, work.pipe
                torch.cuda.empty_cache()
                with torch.no_grad():
                    line["extra"] = {"cfg2": extra_cfg2(dev), "cfg5": extra_cfg5(dev)}
                work.last_S = last_S
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(work.cfg, rows, D, work.last_S)
        print(json.dumps(line))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
