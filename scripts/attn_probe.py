"""One-launch attention (csrc/attention.hip) against the three-launch form at the bench's launch-set size: time of QK^T + softmax + PV per
attention block of the nf = 128 NCSN++ (B = 2240 samples of 256 tokens x 256 channels).   python scripts/attn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib

dev = torch.device("cuda:0")
B, HW = int(os.environ.get("ROWS", 2240)), 256

def t_of(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for C in (256, 128):
    qk = torch.randn(B * HW, 2 * C, device=dev)
    vt = torch.randn(B, C, HW, device=dev)
    bv = torch.randn(C, device=dev)
    out = torch.empty(B * HW, C, device=dev)
    one = torch.tensor([1.0, 1.0], device=dev)
    scale = C ** -0.5
    fused = t_of(lambda: _lib.attention256(qk, vt, out, B, C, one, one, scale, bias_v=bv))
    lg = torch.empty(B, HW, HW, device=dev)
    mixed = torch.empty(B, HW, C, device=dev)
    ep = _lib.make_epilogue(bias=bv)
    t_qk = t_of(lambda: _lib.gemm(qk, qk[:, C:], out=lg, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B, stride_a=HW * 2 * C, stride_b=HW * 2 * C, stride_c=HW * HW))
    t_sm = t_of(lambda: _lib.softmax_rows(lg, lg, B * HW, HW, scale))
    t_pv = t_of(lambda: _lib.gemm(lg, vt, out=mixed, M=HW, N=C, K=HW, lda=HW, ldb=HW, ldc=C, batch=B, stride_a=HW * HW, stride_b=C * HW, stride_c=HW * C, epilogue=ep))
    flops = 2.0 * B * HW * HW * C * 2
    byt = 4.0 * B * HW * C * 4
    print(f"C = {C}: one launch {fused:7.0f} us ({flops / fused / 1e6:6.1f} TFLOP/s fp32-equivalent, {byt / fused / 1e3:6.0f} GB/s of q, k, v, out)   "
          f"three launches {t_qk + t_sm + t_pv:7.0f} us (QK^T {t_qk:.0f} + softmax {t_sm:.0f} + PV {t_pv:.0f})   {(t_qk + t_sm + t_pv) / fused:.2f}x", flush=True)
    del qk, vt, out, lg, mixed
