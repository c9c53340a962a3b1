"""Config-2 batched kernels alone: gram_small_batched_kernel and tridiag_reg_kernel against fp64 references and against the
forms they replace (IDIFF_GRAM_SMALL_TILES / IDIFF_TRIDIAG_ONESTAGE), with HIP-event timings.  python scripts/cfg2_kernels.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import id_diff_amd
from id_diff_amd import _lib

def say(*a): print(*a, flush=True)

def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

dev = torch.device("cuda:0")
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for (M, D) in ((1501, 100), (333, 64), (700, 112), (90, 52), (257, 128), (50, 20)):
    P = NP if (M, D) == (1501, 100) else 37
    S = torch.randn(P, M, D, device=dev) * torch.linspace(0.2, 3.0, D, device=dev) + 0.7
    mean = torch.empty(P, D, dtype=torch.float64, device=dev)
    scratch = torch.empty(P * 32 * D, dtype=torch.float64, device=dev)
    lib.idiff_colmean_f64(S.data_ptr(), P, M, D, mean.data_ptr(), scratch.data_ptr(), st)
    G = torch.empty(P, D, D, dtype=torch.float64, device=dev)
    gram = lambda: lib.idiff_centered_gram_f64(S.data_ptr(), mean.data_ptr(), P, M, D, G.data_ptr(), st)
    t_new = ev(gram)
    G_new = G.clone()
    with _lib.thread_option("IDIFF_GRAM_SMALL_TILES", 1):
        t_old = ev(gram)
    same = torch.equal(G, G_new)
    c = S[:3].double() - S[:3].double().mean(1, keepdim=True)
    ref = c.transpose(1, 2) @ c
    err = float(((G_new[:3] - ref).abs().max() / ref.abs().max()))
    say(f"gram  P={P} {M}x{D}: new {t_new:.3f} ms ({P*M*D*D/t_new/1e9:.1f} TFLOP/s), 64x64 tiles {t_old:.3f} ms, bit-identical {same}, "
        f"rel err vs fp64 torch {err:.1e}, symmetric {bool(torch.equal(G_new, G_new.transpose(1, 2)))}")
    diag, offd = torch.empty(P, D, dtype=torch.float64, device=dev), torch.empty(P, D, dtype=torch.float64, device=dev)
    G2 = G_new.clone()
    def tri():
        G2.copy_(G_new)
        lib.idiff_symtridiag_f64(G2.data_ptr(), P, D, diag.data_ptr(), offd.data_ptr(), None, st)
    t_copy = ev(lambda: G2.copy_(G_new))
    t_tri = ev(tri) - t_copy
    d_new, o_new = diag.clone(), offd.clone()
    with _lib.thread_option("IDIFF_TRIDIAG_ONESTAGE", 1):
        t_tri_old = ev(tri) - t_copy
    # eigenvalues of the tridiagonal vs eigvalsh of G (fp64, CPU)
    errs = []
    for p in range(min(P, 3)):
        d, o = d_new[p].cpu().numpy(), o_new[p].cpu().numpy()
        T = np.diag(d) + np.diag(o[:D - 1], 1) + np.diag(o[:D - 1], -1)
        lam, ref_l = np.linalg.eigvalsh(T), np.linalg.eigvalsh(G_new[p].cpu().numpy())
        errs.append(float(np.abs(lam - ref_l).max() / np.abs(ref_l).max()))
    d_old = diag.cpu().numpy()
    say(f"tridiag P={P} D={D}: registers {t_tri:.3f} ms, LDS form {t_tri_old:.3f} ms, eig err vs fp64 eigvalsh {max(errs):.1e}, "
        f"|diag| sums new/old {float(d_new.abs().sum()):.6e} / {float(np.abs(d_old).sum()):.6e}, finite {bool(torch.isfinite(d_new).all() and torch.isfinite(o_new).all())}")
S = torch.randn(NP, 1501, 100, device=dev)
say(f"spectrum of {NP} x 1501x100: {ev(lambda: _lib.spectrum(S)):.3f} ms")
sv = _lib.spectrum(S[:4]).cpu().double().numpy()
c = S[:4].cpu().double(); c = c - c.mean(1, keepdim=True)
ref = np.linalg.svd(c.numpy(), compute_uv=False)
say(f"singular values vs fp64 SVD: max rel err {float(np.abs(sv / ref - 1).max()):.2e}")
