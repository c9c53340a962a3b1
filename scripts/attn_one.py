"""A few launches of the one-launch attention alone (for rocprofv3 --pmc passes): python scripts/attn_one.py [C] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2240
dev = torch.device("cuda:0")
qk, vt = torch.randn(B * 256, 2 * C, device=dev), torch.randn(B, C, 256, device=dev)
out, one, bv = torch.empty(B * 256, C, device=dev), torch.tensor([1.0, 1.0], device=dev), torch.randn(C, device=dev)
for _ in range(4):
    _lib.attention256(qk, vt, out, B, C, one, one, C ** -0.5, bias_v=bv)
torch.cuda.synchronize()
