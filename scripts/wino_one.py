"""One Winograd conv shape, a few launches (for rocprofv3 --pmc runs): wino_one.py H Cin Cout [B] [split | f43 | f43h | w1d]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import id_diff_amd
from id_diff_amd import _lib
H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2240
dev = torch.device("cuda:0")
x = torch.randn(B, H * H, Cin, device=dev)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
split = len(sys.argv) > 5 and sys.argv[5] == 'split'
f43 = len(sys.argv) > 5 and sys.argv[5] in ('f43', 'f43h')
pairs = len(sys.argv) > 5 and sys.argv[5] == 'f43h'
w1d = len(sys.argv) > 5 and sys.argv[5] == 'w1d'
u = _lib.wino1d_pack(w, Cin, Cout) if w1d else (_lib.winograd43_pack(w, Cin, Cout, pairs=pairs) if f43 else _lib.winograd_pack(w, Cin, Cout, split=bool(split)))
o = torch.empty(B, H * H, Cout, device=dev)
ep = _lib.make_epilogue(bias=torch.randn(Cout, device=dev))
for _ in range(4):
    if w1d:
        _lib.conv2d_wino1d(x, u, o, B, H, H, Cin, Cout, epilogue=ep)
    elif f43:
        _lib.conv2d_winograd43(x, u, o, B, H, H, Cin, Cout, epilogue=ep, pairs=pairs)
    else:
        _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=ep, split=bool(split))
torch.cuda.synchronize()
print("done", flush=True)
