// Two-stage tridiagonalisation of the fp64 Gram matrix on gfx950 (successive band reduction):
//
//   stage 1  full -> band (half-bandwidth BW = 32), one panel of BW columns at a time.  The panel P (m x BW, the part of
//            the block column below the band) is factored by CholeskyQR2 (two Gram products + two BW x BW Cholesky
//            factorisations) and turned into a compact-WY block reflector Q = I - V T V^T by Householder
//            reconstruction (LU of the leading BW x BW block); T^-1 = striu(V^T V) + diag(V^T V)/2 is formed from the
//            V that was actually stored, so Q is orthogonal to rounding whatever the panel looked like.  The trailing
//            matrix gets the two-sided update A' -= V Z^T + Z V^T with Y = A' V, Z = Y T - V (T^T V^T Y T)/2: two passes
//            over A' per BW columns (one read, one read-modify-write) on v_mfma_f64_16x16x4_f64 instead of one 16-byte
//            pass per COLUMN of the unblocked Householder sweep (spectrum.hip), i.e. 16/BW of its HBM traffic.
//            The last <= CORNER columns are reduced by one workgroup in LDS (plain Householder, no rank assumptions).
//   stage 2  band -> tridiagonal by bulge chasing on the compact band (D x 2BW doubles, L2 resident): sweep s removes
//            column s below the subdiagonal with a length-BW reflector and chases the bulge down the band in steps of
//            BW.  Task (s, t) depends on (s, t-1) and (s-1, t+1): the tasks with 2s + t = k are independent and form
//            launch k (2D - 5 launches of <= D / (2 BW) single-wave workgroups).
//
// What the reference computes at this point is torch.linalg.svd on the CPU (dim_reduction.py:197); only the singular
// values are kept, so no transformation is ever accumulated.
#include "common.h"

namespace {

typedef double doublex4 __attribute__((ext_vector_type(4)));

constexpr int BW = 32;             // half-bandwidth after stage 1 = panel width
constexpr int CORNER = 128;        // trailing block reduced in LDS by one workgroup (CORNER >= 4 BW keeps panels tall)
constexpr int CHUNK = 128;         // minimum rows of a panel per workgroup in the tall-skinny kernels
constexpr int LDB = 2 * BW;        // leading dimension of the compact lower band: diagonals 0 .. 2BW-1 (bulge room)
constexpr int PAD = 2;             // zero columns appended to the band (indices beyond it read as zero too): node 0 runs one sweep per column
constexpr int KSPLIT_COLS = 128;   // columns of A' per LDS block of the Y = A' V kernel (its V slice, 32 KB: three workgroups per CU by registers)

// ---------------------------------------------------------------------------------------------- MFMA helpers
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; the accumulator
// register r of lane l is C[(l >> 4) + 4 r][l & 15].  The k order of a reduction is free, so a lane may fetch FOUR
// consecutive k of its row with two 16-byte loads and feed them to four consecutive MFMAs (both operands permuted alike).
__device__ __forceinline__ doublex4 mfma(double a, double b, doublex4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// v_rsq_f64 / v_rcp_f64 seeds (~2^-26 relative) + two Newton steps: full double accuracy without the ~150-instruction
// IEEE sqrt / division sequences, which dominated the single-wave factorisations of the panel kernels.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

// One wave exchanging data through LDS with itself: DS operations of a wave execute in order, so all that is needed is
// that the compiler neither reorders nor caches across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// acc[16 x 16] += X[r0 .. r1)^T[:, i0 .. i0+16) * Y[r0 .. r1)[:, j0 .. j0+16)   (row-major X, Y).  Thirty-two rows (eight MFMA
// k-steps) are fetched before the first MFMA is issued: the loop is a chain of L2 round trips otherwise.
__device__ __forceinline__ void gram_tile(const double *__restrict__ X, int ldx, int i0, const double *__restrict__ Y, int ldy,
                                          int j0, int r0, int r1, doublex4 &acc) {
  const int lane = threadIdx.x & 63, fl = lane & 15, fk = lane >> 4;
  doublex4 acc2 = {0.0, 0.0, 0.0, 0.0};
  for (int r = r0; r < r1; r += 32) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = r + 4 * u + fk;
      a[u] = rr < r1 ? X[(int64_t)rr * ldx + i0 + fl] : 0.0;
      b[u] = rr < r1 ? Y[(int64_t)rr * ldy + j0 + fl] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc = mfma(a[u], b[u], acc);
      acc2 = mfma(a[u + 1], b[u + 1], acc2);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] += acc2[q];
}

__device__ __forceinline__ void store_tile_partial(double *__restrict__ dst /* [BW][BW] */, int i0, int j0, const doublex4 &acc) {
  const int lane = threadIdx.x & 63, fl = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(i0 + fk + 4 * r) * BW + j0 + fl] = acc[r];
}

// The four waves of a 256-thread workgroup own the four 16 x 16 tiles of a BW x BW product (BW = 32).
__device__ __forceinline__ void wave_tile(int &i0, int &j0) {
  const int wave = threadIdx.x >> 6;
  i0 = (wave >> 1) * 16;
  j0 = (wave & 1) * 16;
}

// ---------------------------------------------------------------------------------------------- stage 1: panel kernels
struct PanelGeom {
  double *A;        // [D][D] row-major, both triangles
  int D, j0, lo, m; // panel columns [j0, j0+BW), rows [lo, D) with lo = j0 + BW, m = D - lo
  int chunk_rows;   // rows of the panel per workgroup of the tall-skinny kernels (= CHUNK <= 256: one row per thread)
  int nchunk;       // ceil(m / chunk_rows)
};

// k1: Gpart[chunk] = P_chunk^T P_chunk
__global__ void __launch_bounds__(256) panel_gram_kernel(PanelGeom g, double *__restrict__ Gpart) {
  const int chunk = blockIdx.x;
  const int r0 = chunk * g.chunk_rows, r1 = min(g.m, r0 + g.chunk_rows);
  const double *P = g.A + (int64_t)g.lo * g.D + g.j0;
  int i0, j0;
  wave_tile(i0, j0);
  doublex4 acc = {0.0, 0.0, 0.0, 0.0};
  gram_tile(P, g.D, i0, P, g.D, j0, r0, r1, acc);
  store_tile_partial(Gpart + (int64_t)chunk * BW * BW, i0, j0, acc);
}

// Sum of `n` BW x BW partials into LDS (pitch BW + 1), symmetrised from the upper triangle.  Every thread of the
// workgroup takes BW*BW / blockDim elements; the loop over the partials is unrolled so that a thread keeps 16 loads in
// flight (one wave walking the partials one at a time spent 50 us of L2 round trips here).
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, int n, double (*M)[BW + 1], bool symmetric) {
  // 256 threads x 4 consecutive elements = one BW x BW partial per pass; 16-byte loads, eight partials in flight
  static_assert(BW * BW == 1024, "reduce_partials assumes 256 threads x 4 elements");
  const int e0 = threadIdx.x * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const double2 *src = reinterpret_cast<const double2 *>(part + e0);
#pragma unroll 8
  for (int c = 0; c < n; ++c) {
    const double2 lo = src[(int64_t)c * (BW * BW / 2)], hi = src[(int64_t)c * (BW * BW / 2) + 1];
    s0 += lo.x; s1 += lo.y; s2 += hi.x; s3 += hi.y;
  }
  if (threadIdx.x < 256) {
    const int i = e0 / BW, j = e0 % BW;
    M[i][j] = s0; M[i][j + 1] = s1; M[i][j + 2] = s2; M[i][j + 3] = s3;
  }
  __syncthreads();
  if (symmetric) {
    for (int e = threadIdx.x; e < BW * BW; e += blockDim.x) {
      const int i = e / BW, j = e % BW;
      if (i > j) M[i][j] = M[j][i];
    }
    __syncthreads();
  }
}

// Upper Cholesky factor R (G = R^T R) of the Jacobi-scaled symmetric matrix M (LDS, pitch BW + 1), pivots clamped: a
// column whose pivot is not positive keeps R[j][j] = 1 and a zero row (its Q column comes out ~0; the T factor built from
// the stored V keeps the block reflector orthogonal whatever happens here).  One wave, TWO lanes per row: lane (i, h) =
// (tid & 31, tid >> 5) keeps columns 16 h .. 16 h + 15 of row i of the working matrix in registers (the matrix is
// symmetric, so a[j] is also the lane's element of column j); per step one LDS word per row broadcasts column j.  89
// registers and no scratch -- the first form, one lane per row with all 32 columns, needed more than 512 registers fully
// unrolled and spilled 176 of them; every element sees the same operations in the same order as there.
// Every workgroup of panel_q_kernel / panel_v_kernel redoes the factorisation it needs.  On exit M holds R (upper,
// scaling folded back in), zeros below.
__device__ __forceinline__ void cholesky_upper2(double (*M)[BW + 1], double *dsc /* [BW] */, double (*bc)[BW] /* [2][BW] */) {
  static_assert(BW == 32, "lane = (row, column half)");
  const int tid = threadIdx.x, i = tid & 31, h = (tid >> 5) & 1;
  const double dii = M[i][i];
  const double rdi = dii > 0.0 ? fast_rsqrt(dii) : 1.0;          // 1 / d_i
  if (h == 0) dsc[i] = rdi;
  wave_lds_sync();
  double a[16];                   // a[kk] = element (i, 16 h + kk); turns into R[16 h + kk][i] once that step is done
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) a[kk] = M[i][16 * h + kk] * (rdi * dsc[16 * h + kk]);
#pragma unroll
  for (int j = 0; j < BW; ++j) {
    double (*buf) = bc[j & 1];
    if (h == (j >> 4)) buf[i] = a[j & 15];
    wave_lds_sync();
    const double piv = buf[j];
    const bool ok = piv > 1e-30;
    const double rinv = ok ? fast_rsqrt(piv) : 0.0;
    const double rji = buf[i] * rinv;                  // R[j][i], this row's multiplier
    if (h == (j >> 4)) a[j & 15] = i == j ? (ok ? piv * rinv : 1.0) : (i > j ? rji : 0.0);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const double upd = a[kk] - rji * (buf[(16 * h + kk) & 31] * rinv);
      a[kk] = 16 * h + kk > j ? upd : a[kk];
    }
  }
  const double di = dii > 0.0 ? dii * rdi : 1.0;                  // d_i = sqrt(G_ii)
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) M[16 * h + kk][i] = a[kk] * di;  // R[j][i] (zero for j > i), column scaling folded back
  wave_lds_sync();
}

// LU without pivoting of the matrix in LDS (in place: unit-lower multipliers below the diagonal, U on and above it).
// Lane i keeps row i in registers; the pivot row is passed through LDS by its owner.  Wave 0 only.
__device__ __forceinline__ void lu_nopivot(double (*Bm)[BW + 1], double (*bc)[BW] /* [2][BW] */) {
  const int tid = threadIdx.x, i = tid < BW ? tid : 0;
  double a[BW];
#pragma unroll
  for (int k = 0; k < BW; ++k) a[k] = Bm[i][k];
#pragma unroll
  for (int j = 0; j < BW - 1; ++j) {
    double (*buf) = bc[j & 1];
    if (tid == j) {
#pragma unroll
      for (int k = j; k < BW; ++k) buf[k] = a[k];
    }
    wave_lds_sync();
    if (i > j) {
      const double l = a[j] * fast_rcp(buf[j]);
      a[j] = l;
#pragma unroll
      for (int k = j + 1; k < BW; ++k) a[k] -= l * buf[k];
    }
  }
  if (tid < BW) {
#pragma unroll
    for (int k = 0; k < BW; ++k) Bm[i][k] = a[k];
  }
  wave_lds_sync();
}

// x <- x R^-1 for one row x (forward substitution over the columns of the upper-triangular R held in LDS).
// Right-looking: as soon as x[j] is final it is taken out of every later column, so the 31 - j updates of a step are
// independent of each other and only one multiply-add per column sits on the dependency chain (the left-looking loop
// `s -= x[i] R[i][j]` was a chain of 496 dependent fp64 fmas, ~2 us for a wave on its own).  Every x[k] still receives
// its updates in the order j = 0, 1, ..., k - 1: the result is bit-identical to the left-looking form.
__device__ __forceinline__ void row_solve_upper(double *x, const double (*R)[BW + 1]) {
#pragma unroll
  for (int j = 0; j < BW; ++j) {      // fully unrolled: x[] stays in registers
    const double xj = x[j] * fast_rcp(R[j][j]);
    x[j] = xj;
#pragma unroll
    for (int k = j + 1; k < BW; ++k) x[k] -= xj * R[j][k];
  }
}

// k2 + k3: R1 = chol(sum Gpart) redone by EVERY workgroup in its own LDS (two lanes per row: no scratch), then
// Q_chunk = P_chunk R1^-1 (one thread per row), stored; Gpart2[chunk] = Q_chunk^T Q_chunk (a buffer of its own: other
// workgroups may still be adding up Gpart).  One launch instead of a one-workgroup launch followed by this one.
__global__ void __launch_bounds__(256) panel_q_kernel(PanelGeom g, const double *__restrict__ Gpart, double *__restrict__ Q,
                                                      double *__restrict__ Gpart2) {
  __shared__ double R[BW][BW + 1];
  __shared__ double dsc[BW];
  __shared__ double bc[2][BW];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  reduce_partials(Gpart, g.nchunk, R, true);
  if (tid < 64) cholesky_upper2(R, dsc, bc);
  __syncthreads();
  const int r0 = chunk * g.chunk_rows, r1 = min(g.m, r0 + g.chunk_rows);
  const int row = r0 + tid;
  if (row < r1) {                                   // chunk_rows <= 256: one row per thread
    double x[BW];
    const double *p = g.A + (int64_t)(g.lo + row) * g.D + g.j0;
#pragma unroll
    for (int j = 0; j < BW; ++j) x[j] = p[j];
    row_solve_upper(x, R);
    double *q = Q + (int64_t)row * BW;
#pragma unroll
    for (int j = 0; j < BW; ++j) q[j] = x[j];
  }
  __threadfence_block();
  __syncthreads();
  int i0, j0;
  wave_tile(i0, j0);
  doublex4 acc = {0.0, 0.0, 0.0, 0.0};
  gram_tile(Q, BW, i0, Q, BW, j0, r0, r1, acc);
  store_tile_partial(Gpart2 + (int64_t)chunk * BW * BW, i0, j0, acc);
}

// k4 + k5: the Householder reconstruction redone by EVERY workgroup in its own LDS, then the V rows.
//   R2 = chol(sum Gpart2); Q1_top = Q_top R2^-1; S' = -sign(diag Q1_top); LU (no pivoting) of I - Q1_top S' = V_top U'
//   (Ballard et al., "Reconstructing Householder vectors from TSQR": diagonal entries start at 1 + |q_ii| >= 1, the
//   multipliers stay bounded by 1 for an orthonormal Q1); rows < BW of V are V_top (unit lower), rows >= BW:
//   v = ((q R2^-1) * (-S')) U'^-1; then the partials of V^T V and V^T P.  Chunk 0 stores V_top for the R write-back of
//   trailing_tz_kernel.  One launch instead of a one-workgroup launch followed by the row kernel: 7 launches per panel.
//   (Possible since the two-lane Cholesky: with it neither factorisation spills, 126 registers in the one-workgroup form.)
__global__ void __launch_bounds__(256) panel_v_kernel(PanelGeom g, const double *__restrict__ Gpart2, const double *__restrict__ Q,
                                                      double *__restrict__ Vtop, double *__restrict__ V,
                                                      double *__restrict__ VtVpart, double *__restrict__ VtPpart) {
  __shared__ double Ra[BW][BW + 1];       // R2
  __shared__ double Ub[BW][BW + 1];       // U' on and above the diagonal, the multipliers of V_top below it
  __shared__ double sg[BW];               // S'
  __shared__ double bc[2][BW];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  reduce_partials(Gpart2, g.nchunk, Ra, true);
  if (tid < 64) cholesky_upper2(Ra, sg, bc);          // sg doubles as the factorisation's scaling scratch here
  __syncthreads();
  if (tid < BW) {                                     // Q1_top rows
    double x[BW];
#pragma unroll
    for (int j = 0; j < BW; ++j) x[j] = Q[(int64_t)tid * BW + j];
    row_solve_upper(x, Ra);
#pragma unroll
    for (int j = 0; j < BW; ++j) Ub[tid][j] = x[j];
  }
  __syncthreads();
  if (tid < BW) sg[tid] = Ub[tid][tid] >= 0.0 ? -1.0 : 1.0;      // S'
  __syncthreads();
  for (int e = tid; e < BW * BW; e += 256) {
    const int i = e / BW, j = e % BW;
    Ub[i][j] = (i == j ? 1.0 : 0.0) - Ub[i][j] * sg[j];
  }
  __syncthreads();
  if (tid < 64) lu_nopivot(Ub, bc);
  __syncthreads();
  if (chunk == 0) {
    for (int e = tid; e < BW * BW; e += 256) {
      const int i = e / BW, j = e % BW;
      Vtop[e] = i > j ? Ub[i][j] : (i == j ? 1.0 : 0.0);
    }
  }
  const int r0 = chunk * g.chunk_rows, r1 = min(g.m, r0 + g.chunk_rows);
  const int row = r0 + tid;
  if (row < r1) {
    double x[BW];
    if (row < BW) {
#pragma unroll
      for (int j = 0; j < BW; ++j) x[j] = j < row ? Ub[row][j] : (j == row ? 1.0 : 0.0);
    } else {
#pragma unroll
      for (int j = 0; j < BW; ++j) x[j] = Q[(int64_t)row * BW + j];
      row_solve_upper(x, Ra);
#pragma unroll
      for (int j = 0; j < BW; ++j) x[j] *= -sg[j];
      row_solve_upper(x, Ub);                         // reads U' only: the diagonal and what is above it
    }
    double *v = V + (int64_t)row * BW;
#pragma unroll
    for (int j = 0; j < BW; ++j) v[j] = x[j];
  }
  __threadfence_block();
  __syncthreads();
  int i0, j0;
  wave_tile(i0, j0);
  doublex4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
  const double *P = g.A + (int64_t)g.lo * g.D + g.j0;
  gram_tile(V, BW, i0, V, BW, j0, r0, r1, acc);
  gram_tile(V, BW, i0, P, g.D, j0, r0, r1, acc2);
  store_tile_partial(VtVpart + (int64_t)chunk * BW * BW, i0, j0, acc);
  store_tile_partial(VtPpart + (int64_t)chunk * BW * BW, i0, j0, acc2);
}

// k7: Ypart[ks][m][BW] = A'[:, ks-th column slice] V[slice]; workgroup = 128 rows x BW columns x one slice of KSPLIT_COLS
// columns.  The V slice (KSPLIT_COLS x BW doubles) is staged in LDS once per workgroup; wave w owns rows [32 w, 32 w + 32)
// as two 16-row MFMA tiles x two 16-column tiles.  A lane fetches four consecutive k of each of its two rows (32 bytes
// each) and spends them on four MFMAs per tile.
constexpr int YROWS = 128;
// BUF: A' is addressed through one buffer descriptor with 32-bit byte offsets (per-lane part in one register, per-chunk
// part in the scalar offset): no vector instruction per load.  PMC at D = 12288 had shown 4.8 vector instructions per MFMA
// in this kernel (64-bit address arithmetic per element, LDS index arithmetic per operand), and on gfx950 they do not
// overlap the fp64 MFMAs of the same SIMD.  Matrices of 4 GB and more (D >= 23171) keep the pointer form.
typedef unsigned int sbr_uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int sbr_uintx2 __attribute__((ext_vector_type(2)));
template <bool BUF>
__global__ void __launch_bounds__(256) trailing_y_kernel(PanelGeom g, const double *__restrict__ V, double *__restrict__ Ypart, int krange,
                                                         int off /* lo & 63, or -1: both triangles of A' are valid */) {
  __shared__ double Vs[KSPLIT_COLS][BW];
  const int rb = blockIdx.x, ks = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fl = lane & 15, fk = lane >> 4;
  const int k_lo = ks * krange, k_hi = min(g.m, k_lo + krange);
  const int row0 = rb * YROWS + wave * 32 + fl, row1 = row0 + 16;           // local rows of A'
  const int r0c = min(row0, g.m - 1), r1c = min(row1, g.m - 1);
  const double *Ap = g.A + (int64_t)g.lo * g.D + g.lo;
  const double *ar0 = Ap + (int64_t)r0c * g.D;
  const double *ar1 = Ap + (int64_t)r1c * g.D;
  // Only the tiles of the absolute 64-grid at or below the diagonal are kept up to date (k11s).  A 16 x 16 block whose
  // column tile lies to the right of its row tile is read as the transpose of its mirror image: lane (fl, fk) then takes
  // A'[column][row fl] -- for a fixed column the sixteen lanes of a row group read 128 contiguous bytes.
  const int t0 = off < 0 ? 0x3fffffff : (rb * YROWS + wave * 32 + off) >> 6;          // the tile row of the wave's 32 rows
  doublex4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (doublex4){0.0, 0.0, 0.0, 0.0};
  // The 32-column chunks of the slice are software-pipelined: chunk i + 1 is requested before the MFMAs of chunk i (the
  // plain load -> use loop left every wave waiting out a full memory latency per chunk at two waves per SIMD: 146 us per
  // panel at D = 12288).  The compiler's wait counts are static, so the loop is written for them: the prefetch is
  // unconditional (the last chunk of a range re-requests itself) and there is no branch between a request and its use -- a
  // conditional prefetch, or per-chunk mode tests, merge into `s_waitcnt vmcnt(0)` in front of the MFMAs, which waits for
  // the chunk just requested.  A wave's 32 rows lie in ONE tile row (32-row blocks, 64-row tiles), so its chunks split
  // into a run read directly (tile column <= tile row) and a run read as mirrored transposes: one loop each.
  typedef double double4u __attribute__((ext_vector_type(4), aligned(8)));     // 32 bytes of a row, 8-byte aligned
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, BUF ? (int)((int64_t)g.D * g.D * 8) : 0, 0x00020000);
  // byte offsets of this lane's share of a chunk at column 0: rows r0c / r1c, four columns from 4 fk (direct), and the
  // transposed positions (mirror)
  const unsigned vd0 = (unsigned)((((int64_t)(g.lo + r0c) * g.D) + g.lo + 4 * fk) * 8), vd1 = (unsigned)((((int64_t)(g.lo + r1c) * g.D) + g.lo + 4 * fk) * 8);
  const unsigned vm0 = (unsigned)((((int64_t)(g.lo + 4 * fk) * g.D) + g.lo + r0c) * 8), vm1 = (unsigned)((((int64_t)(g.lo + 4 * fk) * g.D) + g.lo + r1c) * 8);
  auto as2 = [](sbr_uintx4 v, double &x, double &y) {
    x = __builtin_bit_cast(double, (sbr_uintx2){v.x, v.y});
    y = __builtin_bit_cast(double, (sbr_uintx2){v.z, v.w});
  };
  auto load_direct = [&](int c, double (&a0)[2][4], double (&a1)[2][4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (BUF) {
        const int so = (c + 16 * h) * 8;               // wave-uniform: rides in the scalar offset
        as2(__builtin_amdgcn_raw_buffer_load_b128(rA, (int)vd0, so, 0), a0[h][0], a0[h][1]);
        as2(__builtin_amdgcn_raw_buffer_load_b128(rA, (int)vd0, so + 16, 0), a0[h][2], a0[h][3]);
        as2(__builtin_amdgcn_raw_buffer_load_b128(rA, (int)vd1, so, 0), a1[h][0], a1[h][1]);
        as2(__builtin_amdgcn_raw_buffer_load_b128(rA, (int)vd1, so + 16, 0), a1[h][2], a1[h][3]);
      } else {
        const int kb = c + 16 * h + 4 * fk;
        const double4u v = *reinterpret_cast<const double4u *>(ar0 + kb), w = *reinterpret_cast<const double4u *>(ar1 + kb);
        a0[h][0] = v.x; a0[h][1] = v.y; a0[h][2] = v.z; a0[h][3] = v.w;
        a1[h][0] = w.x; a1[h][1] = w.y; a1[h][2] = w.z; a1[h][3] = w.w;
      }
    }
  };
  auto load_mirror = [&](int c, double (&a0)[2][4], double (&a1)[2][4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (BUF) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int so = (c + 16 * h + u) * g.D * 8;   // wave-uniform (D^2 x 8 < 2^32 on this path)
          a0[h][u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)vm0, so, 0));
          a1[h][u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)vm1, so, 0));
        }
      } else {
        const double *q = Ap + (int64_t)(c + 16 * h + 4 * fk) * g.D;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0[h][u] = q[(int64_t)u * g.D + r0c];
          a1[h][u] = q[(int64_t)u * g.D + r1c];
        }
      }
    }
  };
  auto load_clamped = [&](int c, double (&a0)[2][4], double (&a1)[2][4]) {     // the one partial chunk at the end of A'
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kb = c + 16 * h + 4 * fk;
      const bool up = ((c + 16 * h + off) >> 6) > t0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = min(kb + u, g.m - 1);          // columns beyond the slice meet zero rows of Vs
        a0[h][u] = up ? Ap[(int64_t)k * g.D + r0c] : ar0[k];
        a1[h][u] = up ? Ap[(int64_t)k * g.D + r1c] : ar1[k];
      }
    }
  };
  auto fill = [&](int c) {                             // uniform over the workgroup: the next KSPLIT_COLS rows of V
    const int c1 = min(k_hi, c + KSPLIT_COLS);
    __syncthreads();
    for (int e = tid; e < KSPLIT_COLS * BW; e += 256) {
      const int kl = e / BW, k = c + kl;
      // rows 4 apart would sit on the same banks (1 KB stride): rotate every other group of four rows by 16 columns
      Vs[kl][(e % BW + 16 * ((kl >> 2) & 1)) & 31] = k < c1 ? V[(int64_t)k * BW + e % BW] : 0.0;
    }
    __syncthreads();
  };
  // a lane's two operand columns of Vs are fixed: row kl = inblk + 16 h + 4 fk + u has (kl >> 2) & 1 = fk & 1 (inblk is a
  // multiple of 32), so the rotation is 16 (fk & 1) for every read of the lane, and the row index is one per-lane base plus
  // a uniform part -- one register and an immediate per LDS read instead of index arithmetic per operand
  const double *vb0 = &Vs[4 * fk][(fl + 16 * (fk & 1)) & 31], *vb1 = &Vs[4 * fk][(16 + fl + 16 * (fk & 1)) & 31];
  auto mma = [&](int inblk, const double (&a0)[2][4], const double (&a1)[2][4]) {
    const double *p0 = vb0 + inblk * BW, *p1 = vb1 + inblk * BW;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double b0 = p0[(16 * h + u) * BW], b1 = p1[(16 * h + u) * BW];
        acc[0][0] = mfma(a0[h][u], b0, acc[0][0]);
        acc[0][1] = mfma(a0[h][u], b1, acc[0][1]);
        acc[1][0] = mfma(a1[h][u], b0, acc[1][0]);
        acc[1][1] = mfma(a1[h][u], b1, acc[1][1]);
      }
  };
  double A0[3][2][4], A1[3][2][4];
  // chunks cb, cb + 32, ... < ce, every one of them whole; every wave of the workgroup passes the same chunks in the same
  // order (only the loader differs), so the barriers of fill() meet.  Requests run TWO chunks ahead of the MFMAs (three
  // register sets in rotation): a chunk is 2048-4096 cycles of matrix work at two waves per SIMD, less than a loaded
  // memory round trip.
  auto run = [&](auto loader, int cb, int ce) {
    if (cb >= ce) return;
    loader(cb, A0[0], A1[0]);
    loader(min(cb + 32, ce - 32), A0[1], A1[1]);
    auto step = [&](int c, double (&a0)[2][4], double (&a1)[2][4], double (&n0)[2][4], double (&n1)[2][4]) {
      const int inblk = (c - k_lo) & (KSPLIT_COLS - 1);
      if (inblk == 0) fill(c);
      loader(min(c + 64, ce - 32), n0, n1);
      mma(inblk, a0, a1);
    };
    for (int c = cb; c < ce; c += 96) {
      step(c, A0[0], A1[0], A0[2], A1[2]);
      if (c + 32 < ce) step(c + 32, A0[1], A1[1], A0[0], A1[0]);
      if (c + 64 < ce) step(c + 64, A0[2], A1[2], A0[1], A1[1]);
    }
  };
  const int k_full = k_lo + ((k_hi - k_lo) & ~31);                       // k_hi <= g.m: chunks below k_full are whole
  const int cm = off < 0 ? k_full : min(max(64 * (t0 + 1) - off, k_lo), k_full);     // first mirrored column of this wave
  run(load_direct, k_lo, cm);
  run(load_mirror, cm, k_full);
  if (k_full < k_hi) {
    const int inblk = (k_full - k_lo) & (KSPLIT_COLS - 1);
    if (inblk == 0) fill(k_full);
    load_clamped(k_full, A0[0], A1[0]);
    mma(inblk, A0[0], A1[0]);
  }
  double *yp = Ypart + ((int64_t)ks * g.m) * BW;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rb * YROWS + wave * 32 + a * 16 + fk + 4 * r;
      if (orow < g.m) {
        yp[(int64_t)orow * BW + fl] = acc[a][0][r];
        yp[(int64_t)orow * BW + 16 + fl] = acc[a][1][r];
      }
    }
}

// k8: Y = sum_ks Ypart (row chunk), Kpart[chunk] = V_chunk^T Y_chunk
__global__ void __launch_bounds__(256) trailing_yk_kernel(PanelGeom g, const double *__restrict__ Ypart, int nks, const double *__restrict__ V,
                                                          double *__restrict__ Y, double *__restrict__ Kpart) {
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int r0 = chunk * g.chunk_rows, r1 = min(g.m, r0 + g.chunk_rows);
  for (int e = tid; e < (r1 - r0) * BW; e += 256) {
    const int64_t idx = (int64_t)r0 * BW + e;
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < nks; ++k) s += Ypart[(int64_t)k * g.m * BW + idx];
    Y[idx] = s;
  }
  __threadfence_block();
  __syncthreads();
  int i0, j0;
  wave_tile(i0, j0);
  doublex4 acc = {0.0, 0.0, 0.0, 0.0};
  gram_tile(V, BW, i0, Y, BW, j0, r0, r1, acc);
  store_tile_partial(Kpart + (int64_t)chunk * BW * BW, i0, j0, acc);
}

// k6 + k9 + k10 in one launch: every workgroup REDOES the two small single-workgroup steps (T^-1 and C from the V^T V / V^T P
// partials; W2 from the K partials) in its own LDS instead of waiting for two one-workgroup launches in between -- the
// partial sums are a few hundred KB of L2-resident data and the three 32 x 32 triangular solves take two waves a few
// microseconds, against 16 + 13 us for the launches (9 launches per panel instead of 11).  Same operations in the same order
// as the three kernels it replaces (round 2's panel_t / trailing_w2 / trailing_z), so the same bits; chunk 0 also writes the panel's surviving
// block R back into A.
__global__ void __launch_bounds__(256) trailing_tz_kernel(PanelGeom g, const double *__restrict__ VtVpart, const double *__restrict__ VtPpart,
                                                          const double *__restrict__ Vtop, const double *__restrict__ Kpart,
                                                          const double *__restrict__ Y, const double *__restrict__ V,
                                                          double *__restrict__ Z, double *__restrict__ resid2) {
  __shared__ double Ti[BW][BW + 1];
  __shared__ double Wm[BW][BW + 1];
  __shared__ double Cm[BW][BW + 1];
  __shared__ double red[4];
  const int chunk = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  // ---- panel_t: Tinv = striu(V^T V) + diag(V^T V) / 2, C = Tinv^-T (V^T P)
  reduce_partials(VtVpart, g.nchunk, Ti, true);
  for (int e = tid; e < BW * BW; e += 256) { const int i = e / BW, j = e % BW; Ti[i][j] = i < j ? Ti[i][j] : (i == j ? 0.5 * Ti[i][i] : 0.0); }
  reduce_partials(VtPpart, g.nchunk, Cm, false);
  reduce_partials(Kpart, g.nchunk, Wm, false);
  // symmetrise K (V^T A' V of a symmetric A' up to rounding)
  for (int e = tid; e < BW * BW; e += 256) { const int i = e / BW, j = e % BW; if (i < j) { const double s = 0.5 * (Wm[i][j] + Wm[j][i]); Wm[i][j] = s; Wm[j][i] = s; } }
  __syncthreads();
  if (wave == 0 && tid < BW) {          // C: column c of C solves Tinv^T x = W[:, c]
    double x[BW];
#pragma unroll
    for (int i = 0; i < BW; ++i) x[i] = Cm[i][tid];
    row_solve_upper(x, Ti);
#pragma unroll
    for (int i = 0; i < BW; ++i) Cm[i][tid] = x[i];
  } else if (wave == 1 && tid - 64 < BW) {   // meanwhile: K <- K Tinv^-1 (thread i owns row i)
    const int i = tid - 64;
    double x[BW];
#pragma unroll
    for (int j = 0; j < BW; ++j) x[j] = Wm[i][j];
    row_solve_upper(x, Ti);
#pragma unroll
    for (int j = 0; j < BW; ++j) Wm[i][j] = x[j];
  }
  __syncthreads();
  if (wave == 1 && tid - 64 < BW) {     // K <- Tinv^-T K (thread c owns column c), W2 = -K / 2
    const int c = tid - 64;
    double x[BW];
#pragma unroll
    for (int i = 0; i < BW; ++i) x[i] = Wm[i][c];
    row_solve_upper(x, Ti);
#pragma unroll
    for (int i = 0; i < BW; ++i) Wm[i][c] = -0.5 * x[i];
  }
  if (chunk == 0) {
    // R = P_top - V_top C, upper triangle kept (what is below is rounding noise of an exact annihilation), mirrored
    for (int e = tid; e < BW * BW; e += 256) {
      const int i = e / BW, j = e % BW;
      double r = 0.0;
      if (i <= j) {
        r = g.A[(int64_t)(g.lo + i) * g.D + g.j0 + j];
        for (int k = 0; k <= i; ++k) r -= Vtop[i * BW + k] * Cm[k][j];      // V_top is unit lower triangular
      }
      g.A[(int64_t)(g.lo + i) * g.D + g.j0 + j] = r;
      g.A[(int64_t)(g.j0 + j) * g.D + g.lo + i] = r;
    }
  }
  __syncthreads();
  // ---- trailing_z: Z = Y Tinv^-1 + V W2, residual of the rows below the band
  double res = 0.0;
  const int row = chunk * g.chunk_rows + tid;
  if (tid < g.chunk_rows && row < g.m) {
    double x[BW], v[BW];
#pragma unroll
    for (int j = 0; j < BW; ++j) { x[j] = Y[(int64_t)row * BW + j]; v[j] = V[(int64_t)row * BW + j]; }
    row_solve_upper(x, Ti);
#pragma unroll
    for (int j = 0; j < BW; ++j) {
      double s = x[j];
#pragma unroll
      for (int k = 0; k < BW; ++k) s += v[k] * Wm[k][j];
      Z[(int64_t)row * BW + j] = s;
    }
    if (row >= BW) {
      double *p = g.A + (int64_t)(g.lo + row) * g.D + g.j0;
#pragma unroll
      for (int j = 0; j < BW; ++j) {
        double s = p[j];
#pragma unroll
        for (int k = 0; k < BW; ++k) s -= v[k] * Cm[k][j];
        res += s * s;
        p[j] = 0.0;
        g.A[(int64_t)(g.j0 + j) * g.D + g.lo + row] = 0.0;
      }
    }
  }
  res = wave_sum(res);
  if ((tid & 63) == 0) red[tid >> 6] = res;
  __syncthreads();
  if (tid == 0) {
    const double s = red[0] + red[1] + red[2] + red[3];
    if (s != 0.0) atomicAdd(resid2, s);
  }
}

// k11: A' -= V Z^T + Z V^T on 64 x 64 tiles (wave = 32 x 32 = 2 x 2 MFMA tiles, K = 2 BW).  Tiles below the diagonal run
// the two products in the opposite order of the tiles above it, so that (i, j) and (j, i) add the same numbers in the
// same order: A' stays exactly symmetric; diagonal tiles write their lower half and its mirror.
__global__ void __launch_bounds__(256) trailing_update_kernel(PanelGeom g, const double *__restrict__ V, const double *__restrict__ Z) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fl = lane & 15, fk = lane >> 4;
  const int wi = ti * 64 + (wave >> 1) * 32, wj = tj * 64 + (wave & 1) * 32;
  const bool below = ti > tj;
  // operands: rows of V / Z, four consecutive k per lane
  doublex4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (doublex4){0.0, 0.0, 0.0, 0.0};
  // the tile of A' is requested first: its HBM / L2 latency runs under the operand loads and the 64 MFMAs
  double *Ap = g.A + (int64_t)g.lo * g.D + g.lo;
  double cold[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = wi + a * 16 + fk + 4 * r, gj = wj + b * 16 + fl;
        cold[a][b][r] = (gi < g.m && gj < g.m) ? Ap[(int64_t)gi * g.D + gj] : 0.0;
      }
  const double *first = below ? Z : V, *second = below ? V : Z;      // acc += first_i second_j^T + second_i first_j^T
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const double *Ai = pass == 0 ? first : second, *Bj = pass == 0 ? second : first;
#pragma unroll
    for (int kc = 0; kc < BW; kc += 16) {
      double av[2][4], bv[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ri = min(wi + q * 16 + fl, g.m - 1), rj = min(wj + q * 16 + fl, g.m - 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[q][u] = Ai[(int64_t)ri * BW + kc + 4 * fk + u];
          bv[q][u] = Bj[(int64_t)rj * BW + kc + 4 * fk + u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma(av[a][u], bv[b][u], acc[a][b]);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = wi + a * 16 + fk + 4 * r, gj = wj + b * 16 + fl;
        if (gi < g.m && gj < g.m) {
          if (ti != tj) {
            Ap[(int64_t)gi * g.D + gj] = cold[a][b][r] - acc[a][b][r];
          } else if (gi >= gj) {
            const double val = cold[a][b][r] - acc[a][b][r];
            Ap[(int64_t)gi * g.D + gj] = val;
            Ap[(int64_t)gj * g.D + gi] = val;
          }
        }
      }
}


// ---------------------------------------------------------------------------------------------- stage 1: lower-triangle form
// From here on only the LOWER triangle of the trailing block A' is kept up to date, at the granularity of an ABSOLUTE
// grid of 64 x 64 tiles (tile index = matrix index >> 6, so a tile is the same set of elements for every panel although
// the trailing block starts 32 further each time): tiles below the diagonal and the diagonal tiles (both halves) are
// valid, tiles above it are stale.  trailing_update then reads and writes half the bytes (k11s); Y = A' V (k7) fetches a
// 16 x 16 block above the diagonal as the transpose of its mirror image.
//
// k11s: A' -= V Z^T + Z V^T on the tiles of the absolute 64-grid at or below the diagonal (wave = 32 x 32 = 2 x 2 MFMA
// tiles, K = 2 BW).  Diagonal tiles write their lower half and its mirror, so they stay valid in both halves.
// part 0: the whole block.  Look-ahead (sbr_to_band): part 1 = only the first 32 columns of A' (the next panel and the band
// block above it; launched over tile column 0, one workgroup per tile row), part 2 = everything else -- the two parts
// write disjoint elements (a wave owns 32 columns, and wj == 0 names exactly the waves of columns [0, 32)).
// sel 0: every tile (triangular launch); sel 1: tile column 0 (nt workgroups; with part 1); sel 2: only the tiles the
// pipelined kernel below leaves out -- the diagonal, tile column 0 when the block starts inside a tile, the last tile row
// when it ends inside one (3 nt workgroups, duplicates exit).
__global__ void __launch_bounds__(256) trailing_update_lower_kernel(PanelGeom g, int off, const double *__restrict__ V,
                                                                    const double *__restrict__ Z, int part, int sel, int nt) {
  // triangular launch: block b -> (ti, tj), tj <= ti
  const int b = blockIdx.x;
  int ti, tj;
  if (sel == 1) {
    ti = b; tj = 0;
  } else if (sel == 2) {
    const bool ragged = ((g.m + off) & 63) != 0;
    if (b < nt) { ti = b; tj = b; }
    else if (b < 2 * nt) { ti = b - nt; tj = 0; if (off == 0 || ti == 0 || (ragged && ti == nt - 1)) return; }
    else { ti = nt - 1; tj = b - 2 * nt; if (!ragged || tj == ti) return; }
  } else {
    ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    tj = b - ti * (ti + 1) / 2;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fl = lane & 15, fk = lane >> 4;
  const int wi = ti * 64 + (wave >> 1) * 32 - off, wj = tj * 64 + (wave & 1) * 32 - off;      // local indices (may start at -32)
  if ((part == 1 && wj != 0) || (part == 2 && wj == 0)) return;      // wave-uniform; no barrier in this kernel
  doublex4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (doublex4){0.0, 0.0, 0.0, 0.0};
  double *Ap = g.A + (int64_t)g.lo * g.D + g.lo;
  double cold[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = wi + a * 16 + fk + 4 * r, gj = wj + c * 16 + fl;
        cold[a][c][r] = (gi >= 0 && gj >= 0 && gi < g.m && gj < g.m) ? Ap[(int64_t)gi * g.D + gj] : 0.0;
      }
  // acc += Z_i V_j^T + V_i Z_j^T, in this order for every tile (the mirror of a diagonal tile is written, not recomputed)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const double *Ai = pass == 0 ? Z : V, *Bj = pass == 0 ? V : Z;
#pragma unroll
    for (int kc = 0; kc < BW; kc += 16) {
      double av[2][4], bv[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ri = min(max(wi + q * 16 + fl, 0), g.m - 1), rj = min(max(wj + q * 16 + fl, 0), g.m - 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[q][u] = Ai[(int64_t)ri * BW + kc + 4 * fk + u];
          bv[q][u] = Bj[(int64_t)rj * BW + kc + 4 * fk + u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[a][c] = mfma(av[a][u], bv[c][u], acc[a][c]);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = wi + a * 16 + fk + 4 * r, gj = wj + c * 16 + fl;
        if (gi >= 0 && gj >= 0 && gi < g.m && gj < g.m) {
          if (ti != tj) {
            Ap[(int64_t)gi * g.D + gj] = cold[a][c][r] - acc[a][c][r];
          } else if (gi >= gj) {
            const double val = cold[a][c][r] - acc[a][c][r];
            Ap[(int64_t)gi * g.D + gj] = val;
            Ap[(int64_t)gj * g.D + gi] = val;
          }
        }
      }
}

// k11p: the same update for large blocks, software-pipelined.  trailing_update_lower_kernel is load -> 64 MFMAs -> store per
// wave with three waves per SIMD, and the three phases of co-resident workgroups drift into step: 54 ms of a D = 12288
// band reduction against 26 ms for the same kernel with its MFMAs removed (5.9 TB/s) and a 16 ms fp64-MFMA floor.  Here a
// workgroup walks a STRIP of up to `strip` consecutive tiles of one tile row: the row operands (Z_i, V_i) stay in
// registers, the column operands (V_j, Z_j: 32 KB per tile) pass through a double-buffered LDS stage, and the next
// tile's 32 KB of A' and its column operands are requested before the MFMAs of the current tile.  Only tiles that lie
// wholly inside the block and strictly below the diagonal come here (no predicates, no mirror writes): tile rows
// ti_lo .. , tile columns tj_lo .. ti - 1; trailing_update_lower_kernel (sel 2) takes the rest.
// Grid (strips per row, rows): block (s, y) takes tiles tj = tj_lo + s strip ... of row ti = ti_lo + y.
constexpr int UPITCH = BW + 2;          // LDS pitch of an operand row (doubles): 272 bytes, 16 rows cover all 64 banks
__global__ void __launch_bounds__(256, 2) trailing_update_strip_kernel(PanelGeom g, int off, const double *__restrict__ V,
                                                                       const double *__restrict__ Z, int part, int strip,
                                                                       int ti_lo, int tj_lo) {
  __shared__ __attribute__((aligned(16))) double Bs[2][2][64][UPITCH];      // [stage][V | Z][column of the tile][k]
  const int ti = ti_lo + blockIdx.y, tj0 = tj_lo + blockIdx.x * strip;
  if (tj0 > ti - 1) return;
  const int tj1 = min(ti - 1, tj0 + strip - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fl = lane & 15, fk = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int wi = ti * 64 + wr * 32 - off;                                 // local row of this wave's 32 x 32 block
  double *Ap = g.A + (int64_t)g.lo * g.D + g.lo;
  typedef double double2u __attribute__((ext_vector_type(2), aligned(8)));

  // row operands: av[pass][kc / 16][q][u] = (pass == 0 ? Z : V)[row wi + 16 q + fl][kc + 4 fk + u]
  double av[2][2][2][4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const double *Ai = pass == 0 ? Z : V;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const double *row = Ai + (int64_t)(wi + q * 16 + fl) * BW + 4 * fk;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int u = 0; u < 4; ++u) av[pass][kc][q][u] = row[kc * 16 + u];
    }
  }
  // staging role: thread -> (matrix, column of the tile, half of the 32 k)
  const int s_mat = tid >> 7, s_col = (tid >> 1) & 63, s_half = tid & 1;
  const double *s_src = (s_mat == 0 ? V : Z) + (int64_t)(s_col - off) * BW + s_half * 16;
  double2u opr[8];
  auto fetch_operands = [&](int tj) {
    const double2u *src = reinterpret_cast<const double2u *>(s_src + (int64_t)tj * 64 * BW);
#pragma unroll
    for (int e = 0; e < 8; ++e) opr[e] = src[e];
  };
  auto stage_operands = [&](int st) {
    double2u *dst = reinterpret_cast<double2u *>(&Bs[st][s_mat][s_col][s_half * 16]);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = opr[e];
  };
  // element (a, c, r) of the wave's block = block origin (uniform) + lane offset (one register) + a uniform offset
  const int lane_off = fk * g.D + fl;
  auto fetch_cold = [&](int tj, double (&cold)[2][2][4]) {
    const int wj = tj * 64 + wc * 32 - off;
    if (part == 2 && wj == 0) return;
    const double *blk = Ap + ((int64_t)wi * g.D + wj);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[a][c][r] = blk[((int64_t)(a * 16 + 4 * r) * g.D + c * 16) + lane_off];
  };
  auto tile = [&](int tj, int st, const double (&cold)[2][2][4], int next_st) {
    const int wj = tj * 64 + wc * 32 - off;
    if (part == 2 && wj == 0) {                        // wave-uniform; the barriers are outside
      if (next_st >= 0) stage_operands(next_st);
      return;
    }
    doublex4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][c] = (doublex4){0.0, 0.0, 0.0, 0.0};
    // acc += Z_i V_j^T + V_i Z_j^T, in this order for every tile (as trailing_update_lower_kernel: bit-identical results)
    // the column operands of group g + 1 = (pass, kc) are read from LDS before the 16 MFMAs of group g are issued
    double bv[2][2][4];
    auto read_b = [&](int grp, double (&b)[2][4]) {
      const int pass = grp >> 1, kc = grp & 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double2u *q = reinterpret_cast<const double2u *>(&Bs[st][pass][wc * 32 + c * 16 + fl][kc * 16 + 4 * fk]);
        const double2u lo = q[0], hi = q[1];
        b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = hi.x; b[c][3] = hi.y;
      }
    };
    read_b(0, bv[0]);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (grp + 1 < 4) read_b(grp + 1, bv[(grp + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);              // or the scheduler sinks the reads back down to their use
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[a][c] = mfma(av[grp >> 1][grp & 1][a][u], bv[grp & 1][c][u], acc[a][c]);
    }
    // the next tile's column operands go to LDS BEFORE this tile's stores are issued: the wait for those loads would
    // otherwise also wait for the stores (one counter for both)
    if (next_st >= 0) stage_operands(next_st);
    double *blk = Ap + ((int64_t)wi * g.D + wj);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          blk[((int64_t)(a * 16 + 4 * r) * g.D + c * 16) + lane_off] = cold[a][c][r] - acc[a][c][r];
  };

  double coldA[2][2][4], coldB[2][2][4];
  fetch_operands(tj0);
  fetch_cold(tj0, coldA);
  stage_operands(0);
  // every load of the prologue (the row operands among them) is complete before the loop: the compiler's wait-count
  // bookkeeping otherwise carries "row operands pending" around the back edge and makes the MFMAs of EVERY tile wait for
  // the loads just issued for the next one (vmcnt(14) ... vmcnt(0) inside the MFMA sequence: the prefetch hid nothing)
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  __syncthreads();
  for (int tj = tj0; tj <= tj1; tj += 2) {
    // even position of the strip: stage 0 / coldA; odd: stage 1 / coldB
    const bool more1 = tj + 1 <= tj1;
    if (more1) { fetch_operands(tj + 1); __builtin_amdgcn_sched_barrier(0); fetch_cold(tj + 1, coldB); }
    tile(tj, 0, coldA, more1 ? 1 : -1);
    __syncthreads();
    if (!more1) break;
    const bool more2 = tj + 2 <= tj1;
    if (more2) { fetch_operands(tj + 2); __builtin_amdgcn_sched_barrier(0); fetch_cold(tj + 2, coldA); }
    tile(tj + 1, 1, coldB, more2 ? 0 : -1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- stage 1: corner in LDS
// The trailing n x n block (n <= CORNER) is brought to half-bandwidth BW by plain Householder reflections, one
// workgroup, the block in LDS (pitch n + 1): column c keeps rows <= c + BW.
__device__ __forceinline__ double block_sum256(double v, double *buf) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
  __syncthreads();
  return buf[0] + buf[1] + buf[2] + buf[3];
}

__global__ void __launch_bounds__(256) corner_kernel(double *__restrict__ A, int D, int j0) {
  extern __shared__ double sm[];
  const int n = D - j0, pitch = n + 1, tid = threadIdx.x;
  double *C = sm, *v = sm + n * pitch, *w = v + n, *red = w + n;
  // (only the lower triangle of the trailing block is kept up to date: the upper half is its mirror)
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n, hi = max(i, j), lo2 = min(i, j);
    C[i * pitch + j] = A[(int64_t)(j0 + hi) * D + j0 + lo2];
  }
  __syncthreads();
  for (int c = 0; c + BW + 1 < n; ++c) {
    const int base = c + BW, len = n - base;            // reflector acts on local rows/cols [base, n)
    double part = 0.0;
    for (int r = 1 + tid; r < len; r += 256) { const double x = C[(base + r) * pitch + c]; part += x * x; }
    const double tail = block_sum256(part, red);
    if (tail == 0.0) continue;                          // uniform
    const double x0 = C[base * pitch + c];
    const double norm = sqrt(x0 * x0 + tail);
    const double alpha = x0 > 0.0 ? -norm : norm;
    const double v0 = x0 - alpha;
    const double tau = 2.0 / (v0 * v0 + tail);
    for (int r = tid; r < len; r += 256) v[r] = r == 0 ? v0 : C[(base + r) * pitch + c];
    __syncthreads();
    // column c (and its mirror) become alpha e_0
    for (int r = tid; r < len; r += 256) {
      const double val = r == 0 ? alpha : 0.0;
      C[(base + r) * pitch + c] = val;
      C[c * pitch + base + r] = val;
    }
    // columns c+1 .. base-1 (left of the diagonal block): rows [base, n) <- H rows, mirrored
    for (int col = c + 1 + (tid >> 6); col < base; col += 4) {
      double d = 0.0;
      for (int r = (tid & 63); r < len; r += 64) d += v[r] * C[(base + r) * pitch + col];
      d = wave_sum(d) * tau;
      for (int r = (tid & 63); r < len; r += 64) {
        const double val = C[(base + r) * pitch + col] - d * v[r];
        C[(base + r) * pitch + col] = val;
        C[col * pitch + base + r] = val;
      }
    }
    __syncthreads();
    // two-sided on the diagonal block [base, n)^2: p = tau * B v, w = p - (tau/2)(p.v) v, B -= v w^T + w v^T
    for (int r = tid; r < len; r += 256) {
      double s = 0.0;
      for (int k = 0; k < len; ++k) s += C[(base + r) * pitch + base + k] * v[k];
      w[r] = tau * s;
    }
    __syncthreads();
    double pv = 0.0;
    for (int r = tid; r < len; r += 256) pv += w[r] * v[r];
    const double Kc = 0.5 * tau * block_sum256(pv, red);
    for (int r = tid; r < len; r += 256) w[r] -= Kc * v[r];
    __syncthreads();
    for (int e = tid; e < len * len; e += 256) {
      const int r = e / len, k = e - r * len;
      C[(base + r) * pitch + base + k] -= v[r] * w[k] + w[r] * v[k];
    }
    __syncthreads();
  }
  for (int e = tid; e < n * n; e += 256) A[(int64_t)(j0 + e / n) * D + j0 + e % n] = C[(e / n) * pitch + e % n];
}

// ---------------------------------------------------------------------------------------------- band extraction
// AB[j][k] = A[j + k][j] for k <= BW (lower band, column-major), zero for BW < k < 2 BW (room for the bulges).
__global__ void __launch_bounds__(256) extract_band_kernel(const double *__restrict__ A, int D, double *__restrict__ AB) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)(D + PAD) * LDB) return;
  const int j = (int)(e / LDB), k = (int)(e % LDB);
  AB[e] = (k <= BW && j + k < D) ? A[(int64_t)(j + k) * D + j] : 0.0;
}

// ---------------------------------------------------------------------------------------------- stage 2: bulge chasing
// Task (s, t), one wave: reflector range [a, e) with a = s + 1 + t BW; window = rows [a, e2) x columns [a, e) of the lower
// band (diagonal block, then the block below it), held in LDS as W[row - a][col - a] (pitch BW + 1).
//   t = 0 : reflector from column s (rows [a, e)), column s becomes alpha e_0
//   t > 0 : reflector left by task (s, t-1) in vs[s]
//   (1) two-sided on the diagonal block  (2) from the right on the lower block  (3) next reflector from the lower block's
//   first column, applied from the left to its other columns, stored in vs[s] for task (s, t+1).
// Launch k runs every task with 2 s + t = k.
struct ChaseArgs {
  double *AB;      // [D][LDB]
  double *vs;      // [D][BW + 1]: reflector carried by sweep s (v[0..BW), tau)
  int D, k, s_hi;  // blockIdx.x -> s = s_hi - blockIdx.x
};

__device__ __forceinline__ void make_house(double x0, double tail, double &alpha, double &v0, double &tau) {
  if (tail == 0.0) { alpha = x0; v0 = 0.0; tau = 0.0; return; }
  const double n2 = x0 * x0 + tail;
  const double norm = n2 * fast_rsqrt(n2);          // on the critical path of every chase step: no IEEE sqrt / division
  alpha = x0 > 0.0 ? -norm : norm;
  v0 = x0 - alpha;
  tau = 2.0 * fast_rcp(v0 * v0 + tail);
}

// Thread (r = tid >> 2, part = tid & 3) keeps row r of the 2BW x BW window, columns [8 part, 8 part + 8), in registers
// (rows < BW: the diagonal block, symmetric, both triangles; rows >= BW: the block below).  Matrix-vector products are
// partial sums over the 8 columns + two shuffles; the rank-two / rank-one updates are local.
__global__ void __launch_bounds__(256) chase_kernel(ChaseArgs g) {
  __shared__ double v[BW], pq[2 * BW], x2[BW], v2[BW], dpart[2][BW];
  __shared__ double sc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = g.s_hi - blockIdx.x, t = g.k - 2 * s;
  const int D = g.D;
  const int a = s + 1 + t * BW;
  if (s < 0 || t < 0 || D - a < 2) return;
  const int e = min(a + BW, D), e2 = min(e + BW, D);
  const int len = e - a, len2 = e2 - e;          // diagonal block len x len, block below len2 x len
  double *AB = g.AB;
  const int r = tid >> 2, c0 = (tid & 3) * 8;
  const bool diag_row = r < BW;
  const int gr = diag_row ? a + r : e + (r - BW);                 // global row of this thread
  const bool row_ok = diag_row ? r < len : (r - BW) < len2;
  double x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j, gc = a + c;
    double val = 0.0;
    if (row_ok && c < len) {
      // lower band storage: (i, j) with i >= j sits at AB[j * LDB + i - j]
      val = gr >= gc ? AB[(int64_t)gc * LDB + (gr - gc)] : AB[(int64_t)gr * LDB + (gc - gr)];
    }
    x[j] = val;
  }
  double tau;
  if (t == 0) {
    if (wave == 0) {
      const double *col = AB + (int64_t)s * LDB;             // column s: rows [a, e) are diagonals 1 .. len
      const double xv = lane < len ? col[1 + lane] : 0.0;
      const double tail = wave_sum(lane >= 1 && lane < len ? xv * xv : 0.0);
      double alpha, v0, tl;
      make_house(__shfl(xv, 0, 64), tail, alpha, v0, tl);
      if (lane < BW) v[lane] = lane < len ? (lane == 0 ? v0 : xv) : 0.0;
      if (lane < len) AB[(int64_t)s * LDB + 1 + lane] = lane == 0 ? alpha : 0.0;
      if (lane == 0) sc[0] = tl;
    }
  } else {
    const double *src = g.vs + (int64_t)s * (BW + 1);
    if (tid < BW) v[tid] = tid < len ? src[tid] : 0.0;
    if (tid == 0) sc[0] = src[BW];
  }
  __syncthreads();
  tau = sc[0];
  if (tau != 0.0) {
    // rows < BW: p_r = tau (Dg v)_r ; rows >= BW: q_r = tau (Ob v)_r
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += x[j] * v[c0 + j];
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    if ((tid & 3) == 0) pq[r] = tau * part;
    __syncthreads();
    // K = (tau / 2) p.v, every wave for itself
    const double pv = wave_sum(lane < BW ? pq[lane] * v[lane] : 0.0);
    const double kk = 0.5 * tau * pv;
    if (diag_row) {
      const double vr = v[r], wr = pq[r] - kk * vr;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double vc = v[c0 + j], wc = pq[c0 + j] - kk * vc; x[j] -= vr * wc + wr * vc; }
    } else {
      const double qr = pq[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] -= qr * v[c0 + j];
    }
  }
  // next reflector from the first column of the block below
  const bool chase_on = len2 >= 2;
  if (chase_on) {
    if (!diag_row && (tid & 3) == 0) x2[r - BW] = x[0];
    __syncthreads();
    double alpha2, v20, tau2;
    {
      const double xv = lane < len2 ? x2[lane] : 0.0;
      const double tail = wave_sum(lane >= 1 && lane < len2 ? xv * xv : 0.0);
      make_house(x2[0], tail, alpha2, v20, tau2);          // every wave computes the same numbers
      if (wave == 0 && lane < BW) v2[lane] = lane < len2 ? (lane == 0 ? v20 : xv) : 0.0;
    }
    __syncthreads();
    if (!diag_row) {
      // d_c = tau2 * sum_r v2_r Ob[r][c] over the 32 rows of the block below = waves 2 and 3, 16 rows each
      const double vr = v2[r - BW];
      double dloc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double dv = vr * x[j];
        dv += __shfl_xor(dv, 4, 64); dv += __shfl_xor(dv, 8, 64); dv += __shfl_xor(dv, 16, 64); dv += __shfl_xor(dv, 32, 64);
        dloc[j] = dv;
      }
      if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart[wave - 2][lane * 8 + j] = dloc[j];
      }
    }
    __syncthreads();
    if (!diag_row) {
      const double vr = v2[r - BW];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const double dc = tau2 * (dpart[0][c] + dpart[1][c]);
        if (c == 0) x[j] = (r == BW) ? alpha2 : 0.0;
        else x[j] -= dc * vr;
      }
    }
    double *dst = g.vs + (int64_t)s * (BW + 1);
    if (tid < BW) dst[tid] = v2[tid];
    if (tid == 0) dst[BW] = tau2;
  }
  // store the lower triangle of the diagonal block and the block below
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j, gc = a + c;
    if (row_ok && c < len && gr >= gc) AB[(int64_t)gc * LDB + (gr - gc)] = x[j];
  }
}

// ---------------------------------------------------------------------------------------------- stage 2, systolic form
// ONE launch: workgroup t ("node") owns chase step t of EVERY sweep.  Its window -- rows [a, a + 2BW) x columns
// [a, a + BW), a = s + 1 + t BW -- lives in registers for the whole kernel (layout of chase_kernel) and slides by one
// row and column per sweep.  What crosses workgroups are two 33-double messages per sweep:
//   refl : node t -> t + 1, the reflector it made from the first column of its lower block (task (s, t + 1) needs it);
//   col  : node t -> t - 1, the first column of its window after task (s, t): the column that enters node t - 1's window
//          at sweep s + 1 (this IS the dependency (s + 1, t - 1) after (s, t)).
// Messages are data-tagged 8-byte granules {32 data bits, sweep tag}: one agent-scope (sc1) store per granule, polled by
// the consumer with agent-scope loads -- no flag, no fence, nothing cached on either side (MI355X_MICROARCH.md,
// hand-off by granules).  Two slots per mailbox suffice: a producer can be at most one sweep ahead of its consumer.
// The band is zero-padded by PAD columns, so every window is a full block (zero rows and columns are inert under the
// reflections) and a message that no longer comes (the neighbour has run out of sweeps) is a message of zeros.
// Node 0 emits the tridiagonal: e[s] when it makes the reflector of sweep s, d[s + 1] from its leaving column.
// Every spin is bounded; on overflow the abort word is set, everybody leaves and the caller reports NaN.
struct SysArgs {
  const double *AB;           // [Dp][LDB]
  unsigned long long *mbox;   // [T][2 kinds][2 slots][128] granules, zeroed before the launch
  double *diag, *offd;        // [D]
  int *abort_flag;
  int D, Dp;
  long spin_limit;            // polls before a waiting thread gives up (IDIFF_CHASE_SPIN_LIMIT; tests force the abort path with 1)
};
constexpr int MSG = BW + 1;           // doubles per message
constexpr int GRAN = 2 * MSG;         // granules per message
constexpr long SPIN_LIMIT = 1L << 24;  // default of SysArgs::spin_limit: tens of seconds

__device__ __forceinline__ unsigned long long *mailbox(const SysArgs &g, int node, int kind, int slot) {
  return g.mbox + (((int64_t)node * 2 + kind) * 2 + slot) * 128;
}

// threads [0, GRAN) publish `src` (LDS, MSG doubles) with tag `tag`
__device__ __forceinline__ void msg_send(unsigned long long *box, const double *src, unsigned tag) {
  const int tid = threadIdx.x;
  if (tid < GRAN) {
    const unsigned half = reinterpret_cast<const unsigned *>(src)[tid];
    __hip_atomic_store(box + tid, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// threads [base, base + GRAN) wait for their granule of tag `tag` and put it into `dst` (LDS); a thread that gives up (the
// abort word is set, or it sets it itself after spin_limit polls) returns true
__device__ __forceinline__ bool msg_recv(const unsigned long long *box, double *dst, unsigned tag, int *abort_flag, int base,
                                         long spin_limit) {
  const int i = (int)threadIdx.x - base;
  if (i >= 0 && i < GRAN) {
    long spins = 0;
    for (;;) {
      const unsigned long long u = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(u >> 32) == tag) { reinterpret_cast<unsigned *>(dst)[i] = (unsigned)u; return false; }
      __builtin_amdgcn_s_sleep(1);                            // (polling without it measured the same)
      if ((++spins & 255) == 0 || spin_limit < 256) {          // (a tiny limit is a test forcing the abort path: checked at every poll)
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        if (spins > spin_limit) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
      }
    }
  }
  return false;
}

// one thread publishes 32-bit half `h` (0 .. GRAN) of a message whose double h / 2 is `val`
__device__ __forceinline__ void msg_send_half(unsigned long long *box, int h, double val, unsigned tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
  const unsigned half = (h & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
  __hip_atomic_store(box + h, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) chase_systolic_kernel(SysArgs g) {
  // A sweep costs the two hand-offs of the cycle (s, t) -> (s, t + 1) -> (s + 1, t) plus twice the path from "messages
  // in" to "messages out", so that path is kept short: both messages are awaited at once by different threads; what the
  // outgoing messages need (row products, the lower block's first column, two sums over rows) takes ONE more workgroup
  // barrier; every double of a message is sent by the thread that holds it; the bulk of the two-sided update, the left
  // application of the new reflector and the window shift for the next sweep run after the sends.
  __shared__ double v[MSG], pq[2 * BW], dpart[2][BW], red[5];
  __shared__ double msg_in[MSG], rowbuf[BW], edge[4][BW], xcol[BW];
  __shared__ int ab;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x, Dp = g.Dp;
  const int S = Dp - 2 - t * BW;                 // sweeps this node takes part in: task (s, t) exists iff Dp - (s + 1 + t BW) >= 2
  if (S <= 0) return;
  const double *AB = g.AB;
  const int r = tid >> 2, part = tid & 3, c0 = part * 8;
  const bool diag_row = r < BW;
  double x[8];
  {
    const int a = 1 + t * BW, gr = a + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gc = a + c0 + j;
      double val = 0.0;
      if (gr < Dp && gc < Dp) val = gr >= gc ? AB[(int64_t)gc * LDB + (gr - gc)] : AB[(int64_t)gr * LDB + (gc - gr)];
      x[j] = val;
    }
    if (t == 0) {
      if (tid < BW) xcol[tid] = AB[1 + tid];       // column 0, rows 1 .. BW
      if (tid == 0) g.diag[0] = AB[0];
    }
    if (tid == 0) ab = 0;
  }
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    // ---- messages in: the column of sweep s - 1 from node t + 1 (threads 0 .. 65) and the reflector of sweep s from
    //      node t - 1 (threads 128 .. 193) are awaited side by side; node 0 makes its own reflector meanwhile
    if (s > 0) {
      const bool has = Dp - (s + (t + 1) * BW) >= 2;          // task (s - 1, t + 1) exists
      if (has) { if (msg_recv(mailbox(g, t, 1, (s - 1) & 1), msg_in, (unsigned)s, g.abort_flag, 0, g.spin_limit)) ab = 1; }
      else if (tid < MSG) msg_in[tid] = 0.0;
    }
    if (t == 0) {
      if (wave == 2) {                                         // (waves 0 and 1 hold the threads that poll the column message)
        const double xv = lane < BW ? xcol[lane] : 0.0;
        const double tail = wave_sum(lane >= 1 && lane < BW ? xv * xv : 0.0);
        double alpha, v0, tl;
        make_house(__shfl(xv, 0, 64), tail, alpha, v0, tl);
        if (lane < BW) v[lane] = lane == 0 ? v0 : xv;
        if (lane == 0) { v[BW] = tl; if (s < g.D) g.offd[s] = alpha; }
      }
    } else {
      if (msg_recv(mailbox(g, t, 0, s & 1), v, (unsigned)(s + 1), g.abort_flag, 128, g.spin_limit)) ab = 1;     // v[0 .. BW) and tau = v[BW]
    }
    __syncthreads();                                           // B1
    if (ab) break;                                             // uniform: written before the barrier
    if (s > 0 && part == 3 && r >= BW - 1) {
      // the rows of the new last column that come from node t + 1 (the shift at the end of the previous sweep left them open)
      x[7] = r == BW - 1 ? msg_in[0] : (r < 2 * BW - 1 ? msg_in[r - BW + 1] : msg_in[BW]);
    }
    // ---- what the two messages need, first: the row products pq (local to a row's four threads), the first column of the
    //      lower block after the right application (local to its rows), and two sums over rows -- pv over the upper block,
    //      the tail norm over the lower block's first column -- which meet in LDS behind ONE barrier
    const double tau = v[BW];
    double pqr = 0.0;
    if (tau != 0.0) {
      double ps = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) ps += x[j] * v[c0 + j];
      ps += __shfl_xor(ps, 1, 64);
      ps += __shfl_xor(ps, 2, 64);
      pqr = tau * ps;
    }
    if (part == 0) pq[r] = pqr;
    const double v0c = v[0];
    const double x2r = (!diag_row && part == 0) ? x[0] - pqr * v0c : 0.0;      // lower block, column 0, after the right application
    {
      // sums over the wave's sixteen rows: only the part-0 lanes (every fourth) carry a term
      double sum = part != 0 ? 0.0 : (diag_row ? pqr * v[r] : (r > BW ? x2r * x2r : 0.0));
      sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      if (lane == 0) red[wave] = sum;
      if (r == BW && part == 0) red[4] = x2r;
    }
    __syncthreads();                                           // B2
    const double pv = red[0] + red[1];
    const double kk = 0.5 * tau * pv;
    // ---- both messages leave here, each double from the thread that holds it (rows' part-0 threads), the two scalars
    //      from two threads of their own; the column's 32 entries do not wait for the new reflector
    if (part == 0 && diag_row) {
      const double vr = v[r], wr = pqr - kk * vr, w0 = pq[0] - kk * v0c;
      const double colval = x[0] - (vr * w0 + wr * v0c);       // upper block, column 0, after the two-sided application
      if (t > 0) {
        unsigned long long *box = mailbox(g, t - 1, 1, s & 1);
        msg_send_half(box, 2 * r, colval, (unsigned)(s + 1));
        msg_send_half(box, 2 * r + 1, colval, (unsigned)(s + 1));
      } else {
        if (r == 0) { if (s + 1 < g.D) g.diag[s + 1] = colval; }
        else xcol[r - 1] = colval;
      }
    }
    double alpha2, v20, tau2;
    make_house(red[4], red[2] + red[3], alpha2, v20, tau2);   // every thread computes the same numbers
    {
      const bool refl_to = Dp - (s + 1 + (t + 1) * BW) >= 2;  // task (s, t + 1) exists
      if (part == 0) {
        if (diag_row) {
        } else if (refl_to) {
          unsigned long long *box = mailbox(g, t + 1, 0, s & 1);
          const double val = r == BW ? v20 : x2r;
          msg_send_half(box, 2 * (r - BW), val, (unsigned)(s + 1));
          msg_send_half(box, 2 * (r - BW) + 1, val, (unsigned)(s + 1));
        }
      } else if (tid == 1) {
        if (t > 0) {
          unsigned long long *box = mailbox(g, t - 1, 1, s & 1);
          msg_send_half(box, 2 * BW, alpha2, (unsigned)(s + 1));
          msg_send_half(box, 2 * BW + 1, alpha2, (unsigned)(s + 1));
        } else {
          xcol[BW - 1] = alpha2;
        }
      } else if (tid == 2 && refl_to) {
        unsigned long long *box = mailbox(g, t + 1, 0, s & 1);
        msg_send_half(box, 2 * BW, tau2, (unsigned)(s + 1));
        msg_send_half(box, 2 * BW + 1, tau2, (unsigned)(s + 1));
      }
    }
    // ---- the bulk of the two-sided application
    if (tau != 0.0) {
      if (diag_row) {
        const double vr = v[r], wr = pqr - kk * vr;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double vc = v[c0 + j], wc = pq[c0 + j] - kk * vc; x[j] -= vr * wc + wr * vc; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] -= pqr * v[c0 + j];
      }
    }
    // ---- left application of the new reflector to the lower block
    const double vr2 = diag_row ? 0.0 : (r == BW ? v20 : __shfl(x2r, lane & ~3, 64));
    if (!diag_row) {
      double dloc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double dv = vr2 * x[j];
        dv += __shfl_xor(dv, 4, 64); dv += __shfl_xor(dv, 8, 64); dv += __shfl_xor(dv, 16, 64); dv += __shfl_xor(dv, 32, 64);
        dloc[j] = dv;
      }
      if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart[wave - 2][lane * 8 + j] = dloc[j];
      }
    }
    __syncthreads();                                           // B4
    if (!diag_row) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const double dc = tau2 * (dpart[0][c] + dpart[1][c]);
        if (c == 0) x[j] = (r == BW) ? alpha2 : 0.0;
        else x[j] -= dc * vr2;
      }
    }
    // ---- shift the window for the next sweep: W'[r][c] = W[r + 1][c + 1]; the new last column is the mirror of the lower
    //      block's first row for rows < BW - 1, the rest of it arrives with node t + 1's column message (filled in above)
    if (s + 1 < S) {
      if (r == BW) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rowbuf[c0 + j] = x[j];
      }
      if ((r & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) edge[wave][c0 + j] = x[j];
      }
      __syncthreads();                                         // E1
      double y[9];                                             // row r + 1, columns c0 .. c0 + 8
      const bool last_in_wave = (r & 15) == 15;
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = __shfl_down(x[j], 4, 64);
      y[8] = __shfl_down(x[0], 5, 64);                         // thread (r + 1, part + 1), meaningful for part < 3
      if (last_in_wave) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = wave < 3 ? edge[wave + 1][c0 + j] : 0.0;
        y[8] = (wave < 3 && part < 3) ? edge[wave + 1][c0 + 8] : 0.0;
      }
      if (part == 3) y[8] = r < BW - 1 ? rowbuf[r + 1] : 0.0;  // rows >= BW - 1: from the column message, next sweep
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = y[j + 1];
    }
  }
}

__global__ void __launch_bounds__(256) finish_de_kernel(int D, double *__restrict__ diag, double *__restrict__ offd,
                                                        const double *__restrict__ resid2, const double *__restrict__ fro2,
                                                        double tol2, const int *__restrict__ abort_flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  const bool bad = (*resid2 > tol2 * *fro2) || *abort_flag != 0;
  const double nanv = __longlong_as_double(0x7ff8000000000000ll);
  if (bad) { diag[i] = nanv; offd[i] = nanv; }
  else if (i == D - 1) offd[i] = 0.0;
}

__global__ void __launch_bounds__(256) band_to_de_kernel(const double *__restrict__ AB, int D, double *__restrict__ diag,
                                                         double *__restrict__ offd, const double *__restrict__ resid2,
                                                         const double *__restrict__ fro2, double tol2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  // fail loudly: if the panels' annihilation left more than tol * ||A||_F behind, the spectrum would be silently wrong
  const bool bad = resid2 && *resid2 > tol2 * *fro2;
  const double nanv = __longlong_as_double(0x7ff8000000000000ll);
  diag[i] = bad ? nanv : AB[(int64_t)i * LDB];
  offd[i] = bad ? nanv : (i + 1 < D ? AB[(int64_t)i * LDB + 1] : 0.0);
}

__global__ void __launch_bounds__(256) fro2_kernel(const double *__restrict__ A, int64_t n, double *__restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += A[i] * A[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

namespace idiff {

// Helper stream and the two events of the band reduction's look-ahead: one set per host thread and device, made on first
// use with the priority of the stream that asked, kept for the life of the thread.
constexpr int LOOKAHEAD_MIN_M = 4096;
struct Lookahead {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static Lookahead *lookahead_for(hipStream_t st) {
  thread_local Lookahead cache[64];
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  Lookahead &l = cache[dev & 63];
  if (!l.side) {
    // A priority class of its own: the runtime maps streams of one class onto a small pool of hardware queues, and a
    // helper that lands on the caller's queue runs its launches in line with the caller's (measured inside bench.py, where
    // a third stream exists: band reduction at D = 12288 253 ms with the helper in the caller's class, 231 ms serial).
    // The update is the bandwidth-bound half, so it takes the lower class where there is one.
    int prio = 0, least = 0, greatest = 0;
    if (hipStreamGetPriority(st, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = prio; }
    if (prio < least) ++prio;                          // numerically larger = lower priority
    else if (prio > greatest) --prio;
    hipStream_t s2 = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio) != hipSuccess ||
        hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (a) (void)hipEventDestroy(a);
      if (s2) (void)hipStreamDestroy(s2);
      return nullptr;                                   // no helper stream: the serial form is always available
    }
    l.side = s2; l.fork = a; l.join = b;
  }
  return &l;
}

int64_t sbr_scratch_doubles(int D) {
  const int64_t m = D;
  const int64_t nchunk = (m + CHUNK - 1) / CHUNK, nks = (m + KSPLIT_COLS - 1) / KSPLIT_COLS;
  return (int64_t)(D + PAD) * LDB             // band, zero-padded
         + (int64_t)D * (BW + 1)              // carried reflectors of stage 2
         + 5 * m * BW                         // Q, V (two: look-ahead), Y, Z
         + nks * m * BW                       // Ypart
         + 5 * nchunk * BW * BW               // Gram partials (G, G2, VtV, VtP, K)
         + 8 * BW * BW + BW + 16;             // R1, R2, U, Vtop, Tinv, C, W2, spare | sgn | scalars
}

// The read-modify-write pass over the lower triangle of A': strips of tiles through the pipelined kernel once there are
// enough tiles to fill the chip with strips, one tile per workgroup below that.
constexpr int STRIP_MIN_TILES = 2048;
static void launch_update(hipStream_t st, const PanelGeom &g, int off, const double *V, const double *Z, int part, int nt) {
  const int tiles = nt * (nt + 1) / 2;
  if (tiles < STRIP_MIN_TILES) {
    hipLaunchKernelGGL(trailing_update_lower_kernel, dim3(tiles), dim3(256), 0, st, g, off, V, Z, part, 0, nt);
    return;
  }
  const int strip = min(8, max(2, tiles / 1024));
  const bool ragged = ((g.m + off) & 63) != 0;
  const int ti_lo = 1, ti_hi = ragged ? nt - 2 : nt - 1, tj_lo = off ? 1 : 0;       // rows ti_lo .. ti_hi, columns tj_lo .. ti - 1
  if (ti_hi >= ti_lo && ti_hi - 1 >= tj_lo)
    hipLaunchKernelGGL(trailing_update_strip_kernel, dim3(ceil_div(ti_hi - tj_lo, strip), ti_hi - ti_lo + 1), dim3(256), 0, st, g, off, V, Z,
                       part, strip, ti_lo, tj_lo);
  hipLaunchKernelGGL(trailing_update_lower_kernel, dim3(3 * nt), dim3(256), 0, st, g, off, V, Z, part, 2, nt);
}

// Stage 1: G (D x D, both triangles, overwritten) -> compact lower band AB = scratch[0 .. D * LDB) (column-major band:
// AB[j * LDB + k] = B[j + k][j]).  Everything is enqueued on `st`.
int sbr_to_band(double *G, int D, double *scratch, hipStream_t st) {
  double *AB = scratch;
  double *vs = AB + (int64_t)(D + PAD) * LDB;
  double *Q = vs + (int64_t)D * (BW + 1);
  double *Vbuf[2] = {Q + (int64_t)D * BW, Q + 2 * (int64_t)D * BW};
  double *Y = Vbuf[1] + (int64_t)D * BW, *Z = Y + (int64_t)D * BW;
  const int64_t nchunk_max = (D + CHUNK - 1) / CHUNK, nks_max = (D + KSPLIT_COLS - 1) / KSPLIT_COLS;
  double *Ypart = Z + (int64_t)D * BW;
  double *Gp = Ypart + nks_max * D * BW;
  double *Gp2 = Gp + nchunk_max * BW * BW;
  double *VtVp = Gp2 + nchunk_max * BW * BW, *VtPp = VtVp + nchunk_max * BW * BW, *Kp = VtPp + nchunk_max * BW * BW;
  double *R1 = Kp + nchunk_max * BW * BW;
  double *R2 = R1 + BW * BW, *U = R2 + BW * BW, *Vtop = U + BW * BW, *Tinv = Vtop + BW * BW, *C = Tinv + BW * BW,
         *W2 = C + BW * BW;
  double *sgn = W2 + 2 * BW * BW;
  double *scal = sgn + BW;             // [0] = residual^2, [1] = ||G||_F^2

  hipError_t e = hipMemsetAsync(scal, 0, 2 * sizeof(double), st);
  if (e != hipSuccess) { set_error("sbr: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(fro2_kernel, dim3(512), dim3(256), 0, st, G, (int64_t)D * D, scal + 1);

  // IDIFF_SBR_SYNC (debugging): wait for every launch and say which one it was -- a fault then names its kernel
  const bool dbg = option(OPT_SBR_SYNC);
  auto after = [&](const char *what, int j) {
    if (!dbg) return;
    const hipError_t err = hipStreamSynchronize(st);
    fprintf(stderr, "[sbr] D=%d j0=%d %s: %s\n", D, j, what, hipGetErrorString(err));
    fflush(stderr);
  };
  after("fro2", 0);
  // IDIFF_SBR_FULL (A/B): the round-2 form that keeps both triangles of the trailing block up to date
  const bool full = option(OPT_SBR_FULL);
  // Look-ahead (OPT-IN, IDIFF_SBR_LOOKAHEAD): the six panel kernels of panel j + 1 need only the first 32 columns of the
  // updated block.  Those are updated first (part 1, m x 32), then the rest of the update (part 2: the launch that moves
  // the bytes) runs on a helper stream BESIDE the latency-bound panel chain and joins before Y = A' V needs the whole
  // block.  Only for blocks large enough that the update outlasts the cross-stream hand-off.  Measured at D = 12288:
  // band reduction 164 -> 151 ms in a process with two or three streams; 164 -> 262 ms once the process had made a few
  // more (bench.py after its config-2 leg): the runtime then maps the helper onto the caller's hardware queue, whatever
  // its priority class, and every fork / join becomes an in-queue barrier.  A library cannot see that mapping, so the
  // default is the serial form.
  Lookahead *la = (full || dbg || !option(OPT_SBR_LOOKAHEAD) || D - BW < LOOKAHEAD_MIN_M) ? nullptr : lookahead_for(st);
  bool join_pending = false;
  int j0 = 0, panel = 0;
  while (D - j0 > CORNER) {
    double *V = Vbuf[panel & 1];
    ++panel;
    PanelGeom g;
    g.A = G; g.D = D; g.j0 = j0; g.lo = j0 + BW; g.m = D - g.lo;
    g.chunk_rows = CHUNK;
    g.nchunk = ceil_div(g.m, g.chunk_rows);
    // column range per workgroup of the Y kernel: as many workgroups as fit the chip AT ONCE (three per CU: 160 registers,
    // 32 KB of LDS each -- with a 64 KB V block it was two, and the third resident wave per SIMD is worth 10 % here),
    // never a few more -- the kernel is bound by the fp64 matrix pipe, and 576 equal workgroups on 512 places took two
    // rounds (435 us at m = 12256 against 250 us at m = 11040)
    const int rbs = ceil_div(g.m, YROWS);
    int nks = min(min(ceil_div(g.m, KSPLIT_COLS), 12), max(1, 768 / rbs));     // <= 12 partials for trailing_yk to add up
    const int krange = ceil_div(ceil_div(g.m, nks), KSPLIT_COLS) * KSPLIT_COLS;
    nks = ceil_div(g.m, krange);
    hipLaunchKernelGGL(panel_gram_kernel, dim3(g.nchunk), dim3(256), 0, st, g, Gp);
    after("panel_gram_kernel", j0);
    hipLaunchKernelGGL(panel_q_kernel, dim3(g.nchunk), dim3(256), 0, st, g, Gp, Q, Gp2);
    after("panel_q_kernel", j0);
    hipLaunchKernelGGL(panel_v_kernel, dim3(g.nchunk), dim3(256), 0, st, g, Gp2, Q, Vtop, V, VtVp, VtPp);
    after("panel_v_kernel", j0);
    const int off = g.lo & 63;                           // the trailing block starts `off` into its first tile of the absolute grid
    if (join_pending) {                                  // the rest of the previous update
      if (hipError_t e2 = hipStreamWaitEvent(st, la->join, 0); e2 != hipSuccess) { set_error("sbr: hipStreamWaitEvent: %s", hipGetErrorString(e2)); return (int)e2; }
      join_pending = false;
    }
    if ((int64_t)D * D * 8 < 0xFFFFFFFFll)
      hipLaunchKernelGGL(trailing_y_kernel<true>, dim3(rbs, nks), dim3(256), 0, st, g, V, Ypart, krange, full ? -1 : off);
    else
      hipLaunchKernelGGL(trailing_y_kernel<false>, dim3(rbs, nks), dim3(256), 0, st, g, V, Ypart, krange, full ? -1 : off);
    after("trailing_y_kernel", j0);
    hipLaunchKernelGGL(trailing_yk_kernel, dim3(g.nchunk), dim3(256), 0, st, g, Ypart, nks, V, Y, Kp);
    after("trailing_yk_kernel", j0);
    hipLaunchKernelGGL(trailing_tz_kernel, dim3(g.nchunk), dim3(256), 0, st, g, VtVp, VtPp, Vtop, Kp, Y, V, Z, scal);
    after("trailing_tz_kernel", j0);
    const int tiles = ceil_div(g.m, 64);
    if (full) {
      hipLaunchKernelGGL(trailing_update_kernel, dim3(tiles, tiles), dim3(256), 0, st, g, V, Z);
    } else {
      const int nt = ceil_div(g.m + off, 64);
      if (la && g.m >= LOOKAHEAD_MIN_M) {
        hipLaunchKernelGGL(trailing_update_lower_kernel, dim3(nt), dim3(256), 0, st, g, off, V, Z, 1, 1, nt);
        hipError_t e2 = hipEventRecord(la->fork, st);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(la->side, la->fork, 0);
        if (e2 != hipSuccess) { set_error("sbr: look-ahead fork: %s", hipGetErrorString(e2)); return (int)e2; }
        launch_update(la->side, g, off, V, Z, 2, nt);
        e2 = hipEventRecord(la->join, la->side);
        if (e2 != hipSuccess) { set_error("sbr: look-ahead join: %s", hipGetErrorString(e2)); return (int)e2; }
        join_pending = true;
      } else {
        launch_update(st, g, off, V, Z, 0, nt);
      }
    }
    after("trailing_update_kernel", j0);
    j0 += BW;
  }
  if (join_pending) {
    if (hipError_t e2 = hipStreamWaitEvent(st, la->join, 0); e2 != hipSuccess) { set_error("sbr: hipStreamWaitEvent: %s", hipGetErrorString(e2)); return (int)e2; }
  }
  {
    const int n = D - j0;
    const size_t lds = ((size_t)n * (n + 1) + 2 * n + 8) * sizeof(double);
    static AttrGuard guard;
    const void *fn = reinterpret_cast<const void *>(corner_kernel);
    if (int rc = set_dynamic_lds_once(guard, &fn, 1, (int)(((size_t)CORNER * (CORNER + 1) + 2 * CORNER + 8) * sizeof(double)), "sbr corner"))
      return rc;
    if (n > BW + 1) hipLaunchKernelGGL(corner_kernel, dim3(1), dim3(256), lds, st, G, D, j0);
    after("corner_kernel", j0);
  }
  hipLaunchKernelGGL(extract_band_kernel, dim3((unsigned)ceil_div64((int64_t)(D + PAD) * LDB, 256)), dim3(256), 0, st, G, D, AB);
  return launch_status("sbr_to_band");
}

// The systolic chase is a persistent kernel whose ceil((D + PAD - 2) / BW) workgroups wait on each other, so ALL of them
// must be resident at once.  How many the device holds is asked of the runtime (99 VGPRs at 256 threads: 4 workgroups
// per CU on gfx950, 1024 on 256 CUs) per device, and only HALF of it is used: the kernel runs on a side stream beside the
// convolutions, a second chase may run in another stream or process, and a partitioned / CU-masked device exposes fewer
// CUs than the part number suggests.  Beyond the limit the chase runs wavefront by wavefront (chase_kernel), which needs
// no co-residency.  IDIFF_FAKE_CU_COUNT replaces the CU count (tests of this selection).
int systolic_node_limit() {
  static int per_cu[64], cus[64];                 // cached per device ordinal (0 = not asked yet)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  const int slot = dev & 63;
  int n = __atomic_load_n(&per_cu[slot], __ATOMIC_ACQUIRE), c = __atomic_load_n(&cus[slot], __ATOMIC_ACQUIRE);
  if (n == 0 || c == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, chase_systolic_kernel, 256, 0) != hipSuccess || n <= 0) return 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) return 0;
    __atomic_store_n(&cus[slot], c, __ATOMIC_RELEASE);
    __atomic_store_n(&per_cu[slot], n, __ATOMIC_RELEASE);
  }
  const int fake = option_value(OPT_FAKE_CU_COUNT);
  if (fake > 0) c = fake;
  return n * c / 2;
}

// 1: the single-launch systolic chase takes a D x D matrix on the current device, 0: the wavefront chase does
int sbr_chase_is_systolic(int D) {
  return !option(OPT_CHASE_WAVEFRONT) && ceil_div(D + PAD - 2, BW) <= systolic_node_limit();
}

// Stage 2: the band left in scratch by sbr_to_band -> diag / offd.
int sbr_chase(int D, double *diag, double *offd, double *scratch, hipStream_t st) {
  double *AB = scratch;
  double *vs = AB + (int64_t)(D + PAD) * LDB;
  double *scal = scratch + sbr_scratch_doubles(D) - 16;     // [0] residual^2, [1] ||G||_F^2, [2] abort word
  if (!option(OPT_CHASE_WAVEFRONT) && ceil_div(D + PAD - 2, BW) <= systolic_node_limit()) {
    SysArgs a;
    a.AB = AB; a.mbox = reinterpret_cast<unsigned long long *>(vs); a.diag = diag; a.offd = offd;
    a.abort_flag = reinterpret_cast<int *>(scal + 2); a.D = D; a.Dp = D + PAD;
    const int spin_opt = option_value(OPT_CHASE_SPIN_LIMIT);
    a.spin_limit = spin_opt > 0 ? spin_opt : SPIN_LIMIT;
    const int T = ceil_div(a.Dp - 2, BW);                    // nodes with at least one sweep
    hipError_t e = hipMemsetAsync(vs, 0, (size_t)T * 512 * sizeof(unsigned long long), st);
    if (e == hipSuccess) e = hipMemsetAsync(scal + 2, 0, sizeof(double), st);
    if (e != hipSuccess) { set_error("sbr: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(chase_systolic_kernel, dim3(T), dim3(256), 0, st, a);
    hipLaunchKernelGGL(finish_de_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, st, D, diag, offd, scal, scal + 1, 1e-22,
                       reinterpret_cast<const int *>(scal + 2));
    return launch_status("sbr_chase_systolic");
  }
  // launch k = tasks with 2 s + t = k; task (s, t) exists iff D - (s + 1 + t BW) >= 2
  ChaseArgs c;
  c.AB = AB; c.vs = vs; c.D = D;
  for (int k = 0; k <= 2 * (D - 3); ++k) {
    const int s_hi = min(k / 2, D - 3);
    const int64_t num = (int64_t)k * BW + 3 - D;
    int s_lo = num > 0 ? (int)((num + 2 * BW - 2) / (2 * BW - 1)) : 0;
    if (s_lo > s_hi) continue;
    c.k = k; c.s_hi = s_hi;
    hipLaunchKernelGGL(chase_kernel, dim3(s_hi - s_lo + 1), dim3(256), 0, st, c);
  }
  // tolerance^2 on residual^2 / ||G||_F^2: an exact annihilation leaves ~1e-30
  hipLaunchKernelGGL(band_to_de_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, st, AB, D, diag, offd, scal, scal + 1, 1e-22);
  return launch_status("sbr_chase");
}

int sbr_tridiagonalize(double *G, int D, double *diag, double *offd, double *scratch, hipStream_t st) {
  if (int rc = sbr_to_band(G, D, scratch, st)) return rc;
  return sbr_chase(D, diag, offd, scratch, st);
}

int sbr_band_ld() { return LDB; }

}  // namespace idiff
