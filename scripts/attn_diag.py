"""Where the one-launch attention's error comes from: (a) V = identity -> the output IS the probability matrix (QK^T + softmax alone);
(b) q = k = 0 -> uniform rows, the output is the mean of v over the keys (PV alone); (c) the full product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
B, C, HW = 3, 256, 256
g = torch.Generator().manual_seed(1)
for gain in (1.0, 6.0, 0.05):
    n = F.group_norm(torch.randn(B, C, HW, generator=g) * 3 + 1, 32).permute(0, 2, 1).contiguous()
    wqk = torch.randn(2 * C, C, generator=g) * (gain / C ** 0.5)
    wv = torch.randn(C, C, generator=g) * (0.7 / C ** 0.5)
    qk = (n.reshape(-1, C) @ wqk.T).contiguous()
    vt = torch.einsum("oc,bpc->bop", wv, n).contiguous()
    s_qk, s_v = _lib.pairs_scale_from_rows(wqk.to(dev)), _lib.pairs_scale_from_rows(wv.to(dev))
    one = torch.tensor([1.0, 1.0], device=dev)
    scale = C ** -0.5
    q64, k64 = qk[:, :C].double().reshape(B, HW, C), qk[:, C:].double().reshape(B, HW, C)
    P = torch.softmax(torch.einsum("bic,bjc->bij", q64, k64) * scale, dim=-1)
    out = torch.empty(B * HW, C, device=dev)
    eye = torch.eye(HW).expand(B, HW, HW).contiguous().to(dev)          # vt[b, c, j] = delta(c, j): out[b, i, c] = P[b, i, c]
    _lib.attention256(qk.to(dev), eye, out, B, C, s_qk, one, scale)
    e_p = rel(out.cpu().reshape(B, HW, C), P)
    rowsum = out.reshape(B, HW, C).sum(-1)
    _lib.attention256(torch.zeros_like(qk).to(dev), vt.to(dev), out, B, C, s_qk, s_v, scale)
    e_v = rel(out.cpu().reshape(B, HW, C), vt.double().mean(-1)[:, None, :].expand(B, HW, C))
    _lib.attention256(qk.to(dev), vt.to(dev), out, B, C, s_qk, s_v, scale)
    e_all = rel(out.cpu().reshape(B, HW, C), torch.einsum("bij,bcj->bic", P, vt.double()))
    # probabilities of the three-launch form
    lg = torch.empty(B, HW, HW, device=dev)
    qd = qk.to(dev)
    _lib.gemm(qd, qd[:, C:], out=lg, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B, stride_a=HW * 2 * C, stride_b=HW * 2 * C, stride_c=HW * HW)
    _lib.softmax_rows(lg, lg, B * HW, HW, scale)
    print(f"gain {gain}: P alone {e_p:.2e} (three-launch P {rel(lg.cpu(), P):.2e}; row sums off by {float((rowsum - 1).abs().max()):.1e}), PV alone {e_v:.2e}, full {e_all:.2e}; "
          f"scales s_qk {s_qk[0].item()} s_v {s_v[0].item()}, rms(q,k) {float(qk.pow(2).mean().sqrt()):.3f}", flush=True)
