import os, sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout) in [(1, 16, 32, 8, 64), (1, 16, 32, 16, 64), (1, 16, 32, 64, 64), (1, 16, 16, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev); wt = w.permute(0, 2, 3, 1).contiguous().to(dev)
    u = _lib.winograd43_pack(wt, Cin, Cout)
    out = torch.full((B, H, W, Cout), 777.0, device=dev)
    _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout)
    ref = F.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    o = out.cpu().double()
    bad = (o - ref).abs() > 1e-3
    print((B, H, W, Cin, Cout), "bad:", int(bad.sum()), "of", o.numel(), "rel err", float((o - ref).norm() / ref.norm()))
    t = bad.reshape(B, H // 4, 4, W // 4, 4, Cout // 4, 4).permute(0, 1, 3, 2, 4, 5, 6).reshape(-1, 4, 4, Cout // 4, 4)   # [tile][a][b][cq][e]
    print("  bad by tile:", t.sum((1, 2, 3, 4)).tolist())
    print("  bad by a:", t.sum((0, 2, 3, 4)).tolist(), " by b:", t.sum((0, 1, 3, 4)).tolist())
    print("  bad by cq:", t.sum((0, 1, 2, 4)).tolist(), " by e:", t.sum((0, 1, 2, 3)).tolist())
