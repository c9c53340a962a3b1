"""Oracle restatement of the counter-based noise stream of the HIP driver (id-diff_amd/csrc/rng.hip).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference draws ``z = torch.randn_like(batch)`` from the unseeded global generator (dim_reduction.py:180), so there
is nothing of the reference's to restate here: this file restates OUR stream so that the CPU oracle can be fed the draws
the device path consumes without a GPU in the room (tests/golden/make_cfg3_point.py runs in the build container).

Philox4x32-10 (Salmon et al., SC'11): key = the 64-bit point seed, counter = index of the 4-element group inside the
point's logical [rows, D] noise matrix; four uniforms in (0, 1] -> two Box-Muller pairs -> four normals.  Integer part
bit-exact by construction; the float part (logf / sqrtf / sincosf on the device, numpy float32 here) agrees to a few
ulp -- tests/test_hip_pipeline.py pins this file against the device's ``z_out`` on the GPU box.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr_lo, ctr_hi, seed):
    """Four uint32 words per 64-bit counter (arrays ``ctr_lo`` / ``ctr_hi`` of uint32 halves), key = 64-bit ``seed``."""
    x = ctr_lo.astype(np.uint64)
    y = ctr_hi.astype(np.uint64)
    z = np.zeros_like(x)
    w = np.zeros_like(x)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * x, _M1 * z                    # 32 x 32 -> 64 bit products
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        x, y, z, w = hi1 ^ y ^ np.uint64(k0), lo1, hi0 ^ w ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return x.astype(np.uint32), y.astype(np.uint32), z.astype(np.uint32), w.astype(np.uint32)


def _u01(v):
    """(0, 1]: 24 bits, never feeds log(0) (rng.hip: u01)."""
    return ((v >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)


def normal_rows(seed, D, row0, n):
    """The N(0, 1) draws of rows [row0, row0 + n) of a point's [rows, D] noise matrix (D % 4 == 0), float32 [n, D]."""
    if D % 4:
        raise ValueError("the in-kernel stream writes 16-byte groups: D % 4 == 0")
    ctr = (np.arange(row0 * D, (row0 + n) * D, 4, dtype=np.uint64)) >> np.uint64(2)
    x, y, z, w = philox4x32_10((ctr & _MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), seed)
    two_pi = np.float32(6.2831853071795864)
    r0 = np.sqrt(np.float32(-2.0) * np.log(_u01(x)), dtype=np.float32)
    r1 = np.sqrt(np.float32(-2.0) * np.log(_u01(z)), dtype=np.float32)
    a0, a1 = two_pi * _u01(y), two_pi * _u01(w)
    out = np.stack([r0 * np.cos(a0, dtype=np.float32), r0 * np.sin(a0, dtype=np.float32),
                    r1 * np.cos(a1, dtype=np.float32), r1 * np.sin(a1, dtype=np.float32)], axis=1)
    return out.reshape(n, D).astype(np.float32)
