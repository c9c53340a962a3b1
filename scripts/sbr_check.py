"""Two-stage eigensolver on the MI355X against numpy fp64: stage 1 alone (band), then the whole path, then timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import id_diff_amd
from id_diff_amd import _lib

dev = "cuda"
rng = np.random.default_rng(0)


def make(kind, D):
    if kind == "gauss":
        S = rng.standard_normal((D + 64, D))
    elif kind == "cliff":
        S = rng.standard_normal((D + 100, D)); S[:, D - 40:] /= 70
        S = S @ np.linalg.qr(rng.standard_normal((D, D)))[0]
    elif kind == "geo1e8":
        u, _ = np.linalg.qr(rng.standard_normal((D + 50, D))); v, _ = np.linalg.qr(rng.standard_normal((D, D)))
        S = (u * np.logspace(0, -4, D)) @ v.T
    elif kind == "zero":
        S = np.zeros((D + 10, D))
    elif kind == "lowrank":
        S = rng.standard_normal((D + 50, 20)) @ rng.standard_normal((20, D))
    return S.T @ S


ok = True
for kind, D in [("gauss", 129), ("gauss", 200), ("cliff", 517), ("geo1e8", 400), ("zero", 260), ("lowrank", 300), ("gauss", 1024),
                ("cliff", 1501)]:
    G = make(kind, D)
    ref = np.linalg.eigvalsh(G)
    scale = max(np.abs(ref).max(), 1e-300)
    Gd = torch.from_numpy(G).to(dev)
    B = _lib.sym_band(Gd.clone()).cpu().numpy()
    evb = np.linalg.eigvalsh(B)
    ev = _lib.sym_eigvals(Gd.clone()).cpu().numpy()
    _lib.set_option("IDIFF_TRIDIAG_ONESTAGE", True)
    ev1 = _lib.sym_eigvals(Gd.clone()).cpu().numpy() if D % 2 == 0 else ev
    _lib.set_option("IDIFF_TRIDIAG_ONESTAGE", False)
    e_band, e_full, e_one = (np.abs(x - ref).max() / scale for x in (evb, ev, ev1))
    flag = e_band < 1e-13 and e_full < 1e-13
    ok &= bool(flag)
    print(f"{kind:8s} D={D:5d}: band {e_band:.2e}  two-stage {e_full:.2e}  one-stage {e_one:.2e}  sym {np.abs(B - B.T).max():.1e} {'ok' if flag else 'FAIL'}", flush=True)

print("ALL OK" if ok else "FAILURES")
if ok and "--time" in sys.argv:
    for D in (3072, 12288):
        G = torch.randn(D + 64, D, device=dev, dtype=torch.float64)
        G = G.T @ G
        for name, one in (("two-stage", False), ("one-stage", True)):
            _lib.set_option("IDIFF_TRIDIAG_ONESTAGE", one)
            _lib.sym_eigvals(G.clone()); torch.cuda.synchronize()
            A = G.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); ev = _lib.sym_eigvals(A); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"D={D}: {name} tridiag+bisect {dt * 1e3:.1f} ms, trace err {abs(float(ev.sum()) / float(G.diagonal().sum()) - 1):.1e}", flush=True)
        _lib.set_option("IDIFF_TRIDIAG_ONESTAGE", False)
