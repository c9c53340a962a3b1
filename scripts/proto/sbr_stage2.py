"""numpy prototype of stage 2 (band -> tridiagonal) by bulge chasing, task-decomposed as the HIP kernel does it.

Task (s, t): sweep s (eliminates column s below the first subdiagonal), chase step t; reflector range [a, e) with
a = s + 1 + t*b.  A task (1) for t = 0 builds the reflector from column s, (2) applies it two-sided to the diagonal
block, (3) applies it from the right to the block below (creating the bulge), builds the next reflector from that
block's first column and applies it from the left to the block's other columns.
Dependency: (s+1, t) needs (s, t+2) [last row of its lower block is the first row of (s, t+2)'s diagonal block]:
sweeps run LAG = 3 steps apart.  The driver below executes tasks in wavefront order k = LAG*s + t."""
import numpy as np


def house(x):
    """(v, tau, alpha): (I - tau v v^T) x = alpha e0."""
    tail = float(x[1:] @ x[1:])
    if tail == 0.0:
        return np.zeros_like(x), 0.0, float(x[0])
    norm = np.sqrt(x[0] * x[0] + tail)
    alpha = -norm if x[0] > 0 else norm
    v = x.copy(); v[0] -= alpha
    return v, 2.0 / float(v @ v), alpha


class Chase:
    def __init__(self, B, b):
        self.B = B.copy()      # dense symmetric with half-bandwidth b (prototype keeps both triangles in step)
        self.b = b
        self.D = B.shape[0]
        self.refl = {}         # sweep -> (v, tau) carried to its next step

    def steps(self, s):
        """number of tasks of sweep s"""
        D, b = self.D, self.b
        n, a = 0, s + 1
        while a < D and (D - a >= 2 or n == 0 and False):
            n += 1
            a += b
        return n

    def task(self, s, t):
        B, b, D = self.B, self.b, self.D
        a = s + 1 + t * b
        e = min(a + b, D)
        if e - a < 2 and t == 0:
            return
        if t == 0:
            v, tau, alpha = house(B[a:e, s].copy())
            B[a:e, s] = 0.0; B[a, s] = alpha
            B[s, a:e] = B[a:e, s]
        else:
            v, tau = self.refl.pop(s)
        if tau != 0.0:
            # two-sided on the diagonal block
            Dg = B[a:e, a:e]
            p = tau * (Dg @ v)
            w = p - 0.5 * tau * float(p @ v) * v
            Dg -= np.outer(v, w) + np.outer(w, v)
        e2 = min(e + b, D)
        if e2 > e:
            Ob = B[e:e2, a:e]
            if tau != 0.0:
                Ob -= tau * np.outer(Ob @ v, v)
            if e2 - e >= 2:
                v2, tau2, alpha2 = house(Ob[:, 0].copy())
                Ob[:, 0] = 0.0; Ob[0, 0] = alpha2
                if tau2 != 0.0:
                    Ob[:, 1:] -= tau2 * np.outer(v2, v2 @ Ob[:, 1:])
                self.refl[s] = (v2, tau2)
            B[a:e, e:e2] = Ob.T


def run(B, b, lag):
    ch = Chase(B, b)
    D = B.shape[0]
    tasks = []
    for s in range(D - 2):
        t, a = 0, s + 1
        while a < D:
            e = min(a + b, D)
            if e - a < 2 and t > 0 and s not in [x[1] for x in []]:
                pass
            tasks.append((lag * s + t, s, t))
            e2 = min(e + b, D)
            if e2 - e < 2:       # no next reflector
                break
            t += 1; a += b
    tasks.sort()
    for _, s, t in tasks:
        ch.task(s, t)
    return ch.B


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for D, b in [(97, 8), (200, 16), (260, 32), (65, 32), (40, 32)]:
        A = rng.standard_normal((D, D)); A = A + A.T
        i, j = np.indices(A.shape)
        A[np.abs(i - j) > b] = 0.0
        ref = np.linalg.eigvalsh(A)
        for lag in (3, 2, 1):
            T = run(A, b, lag)
            off = np.abs(T[np.abs(i - j) > 1]).max()
            ev = np.linalg.eigvalsh(np.where(np.abs(i - j) <= 1, T, 0.0))
            print(f"D={D} b={b} lag={lag}: off-tridiagonal max {off:.2e}, eig err {np.abs(ev - ref).max() / np.abs(ref).max():.2e}")


def touched(s, t, b, D):
    """set of (row, col) lower-triangle elements task (s, t) reads or writes"""
    a = s + 1 + t * b
    e = min(a + b, D)
    e2 = min(e + b, D)
    el = set()
    if t == 0:
        el |= {(r, s) for r in range(a, e)}
    el |= {(r, c) for r in range(a, e) for c in range(a, r + 1)}
    el |= {(r, c) for r in range(e, e2) for c in range(a, e)}
    return el


def check_lag(D, b, lag):
    tasks = {}
    for s in range(D - 2):
        t, a = 0, s + 1
        while a < D:
            e = min(a + b, D); e2 = min(e + b, D)
            tasks.setdefault(lag * s + t, []).append((s, t))
            if e2 - e < 2:
                break
            t += 1; a += b
    # (1) tasks of one wavefront are pairwise disjoint; (2) any two conflicting tasks keep the sequential (s, t) order
    last_writer = {}
    ok = True
    for k in sorted(tasks):
        seen = {}
        for (s, t) in tasks[k]:
            for el in touched(s, t, b, D):
                if el in seen:
                    ok = False
                seen[el] = (s, t)
        for el, st in seen.items():
            if el in last_writer and last_writer[el] > st:
                ok = False
            last_writer[el] = st
    return ok


if __name__ == "__main__":
    for lag in (3, 2, 1):
        print("lag", lag, "conflict-free:", all(check_lag(D, b, lag) for D, b in [(60, 4), (97, 8), (75, 16)]))
