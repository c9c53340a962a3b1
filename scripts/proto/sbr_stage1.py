"""numpy prototype of stage 1 (full -> band) of the two-stage tridiagonalisation: CholeskyQR2 + Householder
reconstruction panels, compact-WY two-sided update.  Validates the formulas the HIP kernels implement."""
import numpy as np


def chol_clamped(G):
    """Upper Cholesky factor of the Jacobi-scaled G with clamped pivots; returns R (G ~= R^T R) and R^-1."""
    b = G.shape[0]
    d = np.sqrt(np.where(np.diag(G) > 0, np.diag(G), 1.0))
    Gs = G / d[:, None] / d[None, :]
    R = np.zeros_like(G)
    A = Gs.copy()
    for j in range(b):
        piv = A[j, j]
        if not (piv > 1e-30):
            R[j, j] = 1.0            # dependent / zero column: leave it (Q column ~ 0)
            A[j, j + 1:] = 0.0
            continue
        r = np.sqrt(piv)
        R[j, j] = r
        R[j, j + 1:] = A[j, j + 1:] / r
        A[j + 1:, j + 1:] -= np.outer(R[j, j + 1:], R[j, j + 1:])
    R = R * d[None, :]
    Rinv = np.linalg.solve(R, np.eye(b))   # HIP: triangular back substitution
    return R, Rinv


def panel(P):
    """V (m x b, unit lower trapezoidal), T (b x b upper), R_actual (b x b) with (I - V T V^T)^T P = [R; 0]."""
    m, b = P.shape
    R1, R1inv = chol_clamped(P.T @ P)
    Q = P @ R1inv
    R2, R2inv = chol_clamped(Q.T @ Q)
    Q1top = Q[:b] @ R2inv
    s = -np.where(np.diag(Q1top) >= 0, 1.0, -1.0)            # S' = -sign(diag)
    Btop = np.eye(b) - Q1top * s[None, :]
    # LU without pivoting of Btop
    L = np.eye(b); U = Btop.copy()
    for j in range(b):
        L[j + 1:, j] = U[j + 1:, j] / U[j, j]
        U[j + 1:, j:] -= np.outer(L[j + 1:, j], U[j, j:])
    U = np.triu(U)
    Uinv = np.linalg.solve(U, np.eye(b))
    Mq = R2inv @ (-(s[:, None]) * Uinv)                       # V_low = Q_low Mq
    V = np.vstack([L, Q[b:] @ Mq])
    GV = V.T @ V
    Tinv = np.triu(GV, 1) + 0.5 * np.diag(np.diag(GV))
    T = np.linalg.solve(Tinv, np.eye(b))
    C = T.T @ (V.T @ P)
    Pnew = P - V @ C
    return V, T, Pnew


def to_band(A, b, corner):
    A = A.copy()
    D = A.shape[0]
    resid2 = 0.0
    j0 = 0
    while D - j0 > corner:
        lo = j0 + b
        P = A[lo:, j0:j0 + b]
        V, T, Pnew = panel(P)
        resid2 += float((Pnew[b:] ** 2).sum())
        Rn = np.triu(Pnew[:b])
        A[lo:, j0:j0 + b] = 0.0
        A[lo:lo + b, j0:j0 + b] = Rn
        A[j0:j0 + b, lo:] = A[lo:, j0:j0 + b].T
        Ap = A[lo:, lo:]
        Y = Ap @ V
        K = V.T @ Y
        Z = Y @ T - 0.5 * V @ (T.T @ K @ T)
        A[lo:, lo:] = Ap - V @ Z.T - Z @ V.T
        j0 += b
    # corner: unblocked Householder to bandwidth b inside the dense trailing block
    n = D - j0
    Cn = A[j0:, j0:]
    for c in range(0, n - b - 1):
        x = Cn[c + b:, c].copy()
        tail = float((x[1:] ** 2).sum())
        if tail == 0.0:
            continue
        norm = np.sqrt(x[0] ** 2 + tail)
        alpha = -norm if x[0] > 0 else norm
        v = x.copy(); v[0] -= alpha
        tau = 2.0 / float(v @ v)
        sub = Cn[c + b:, :]
        sub -= tau * np.outer(v, v @ sub)
        sub2 = Cn[:, c + b:]
        sub2 -= tau * np.outer(sub2 @ v, v)
    return A, np.sqrt(resid2)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for D, b, kind in [(300, 32, "gauss"), (517, 32, "cliff"), (400, 16, "geo1e12"), (260, 32, "zero"), (300, 32, "lowrank")]:
        if kind == "gauss":
            S = rng.standard_normal((D + 50, D))
        elif kind == "cliff":
            S = rng.standard_normal((D + 100, D)); S[:, D - 40:] /= 70
            S = S @ np.linalg.qr(rng.standard_normal((D, D)))[0]
        elif kind == "geo1e12":
            u, _ = np.linalg.qr(rng.standard_normal((D + 50, D))); v, _ = np.linalg.qr(rng.standard_normal((D, D)))
            S = (u * np.logspace(0, -6, D)) @ v.T
        elif kind == "zero":
            S = np.zeros((D + 10, D))
        else:
            S = rng.standard_normal((D + 50, 20)) @ rng.standard_normal((20, D))
        G = S.T @ S
        B, resid = to_band(G, b, 4 * b)
        i, j = np.indices(B.shape)
        outside = np.abs(B[np.abs(i - j) > b]).max() if D > b + 1 else 0.0
        ev_ref = np.linalg.eigvalsh(G)
        ev = np.linalg.eigvalsh(np.where(np.abs(i - j) <= b, B, 0.0))
        scale = max(abs(ev_ref).max(), 1e-300)
        print(f"{kind:8s} D={D} b={b}: outside-band max {outside:.2e}  panel resid {resid / scale:.2e}  "
              f"eig err / ||A|| {np.abs(ev - ev_ref).max() / scale:.2e}  sym err {np.abs(B - B.T).max() / scale:.1e}")
