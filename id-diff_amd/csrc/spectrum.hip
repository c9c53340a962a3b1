// Singular values of the centred score matrix on gfx950 (replaces the CPU torch.linalg.svd of
// dim_reduction.py:193-198, whose U and V are discarded).
//
//   mean_j   = (1/M) sum_m S[m, j]                                    fp64, two-stage deterministic sum
//   G        = sum_m (S[m, :] - mean)^T (S[m, :] - mean)              fp64 on v_mfma_f64_16x16x4_f64
//   T        = Q^T G Q  tridiagonal (Householder reflections)         fp64
//   lambda_i = eigenvalues of T by Sturm-sequence bisection           fp64, one thread per eigenvalue
//   sigma_i  = sqrt(max(lambda_i, 0)), descending                     rounded once to fp32
//
// The fp32 entries of S are exact in fp64 and so are their pairwise products (24+24 < 53 bits), so G is
// the exact Gram matrix up to fp64 summation error: squaring the condition number costs ~1e-16 * cond^2
// relative, far inside the 1e-4 tolerance for the cliffs the ID rule looks for (cond ~ 1e2..1e4).
#include "common.h"

namespace {

typedef double doublex4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ means
constexpr int MEAN_SPLITS = 32;

__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float *__restrict__ S, int M, int D, double *__restrict__ part /* [P][SPLITS][D] */) {
  const int p = blockIdx.z, sp = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= D) return;
  const int rows_per = (M + MEAN_SPLITS - 1) / MEAN_SPLITS;
  const int lo = sp * rows_per, hi = min(M, lo + rows_per);
  const float *s = S + (int64_t)p * M * D + col;
  double acc = 0.0;
  for (int m = lo; m < hi; ++m) acc += (double)s[(int64_t)m * D];
  part[((int64_t)p * MEAN_SPLITS + sp) * D + col] = acc;
}

__global__ void __launch_bounds__(256)
colsum_final_kernel(const double *__restrict__ part, int M, int D, double *__restrict__ mean) {
  const int p = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= D) return;
  double acc = 0.0;
  for (int sp = 0; sp < MEAN_SPLITS; ++sp) acc += part[((int64_t)p * MEAN_SPLITS + sp) * D + col];
  mean[(int64_t)p * D + col] = acc / (double)M;
}

// ------------------------------------------------------------------------------------------------ Gram
// Workgroup = 4 waves -> one 64 x 64 tile of G (only tiles with tj >= ti are computed, then mirrored).
// Wave w owns the 32 x 32 quadrant (w>>1, w&1) as 2 x 2 MFMA tiles of 16 x 16.  A k-step is 4 rows of S:
// lane l feeds A[i = l&15][k = l>>4] = S[m0 + k][i0 + i] - mean[i0 + i] and likewise B from the j panel
// (v_mfma_f64_16x16x4_f64: C/D row = (l>>4) + 4*reg, col = l&15).
constexpr int GT = 64;       // tile edge
constexpr int GROWS = 32;    // rows of S staged per iteration
constexpr int GPITCH = GT + 16;  // row offset = 16 banks: the 4 rows of a k-step never collide

__global__ void __launch_bounds__(256)
gram_kernel(const float *__restrict__ S, const double *__restrict__ mean, int M, int D, int tiles,
            double *__restrict__ G, int ti_first, int mirror) {
  __shared__ float Si[GROWS * GPITCH];
  __shared__ float Sj[GROWS * GPITCH];
  // decode upper-triangular tile index
  int t = blockIdx.x, ti = ti_first;       // upper-triangular tiles of the tile rows ti_first, ti_first + 1, ...
  while (t >= tiles - ti) { t -= tiles - ti; ++ti; }
  const int tj = ti + t;
  const int p = blockIdx.y;
  const float *Sp = S + (int64_t)p * M * D;
  const double *mu = mean + (int64_t)p * D;
  const int i0 = ti * GT, j0 = tj * GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const int fl = lane & 15, fk = lane >> 4;

  double mi[2], mj[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ci = i0 + wi + a * 16 + fl, cj = j0 + wj + a * 16 + fl;
    mi[a] = ci < D ? mu[ci] : 0.0;
    mj[a] = cj < D ? mu[cj] : 0.0;
  }
  doublex4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (doublex4){0.0, 0.0, 0.0, 0.0};

  // staging map: 256 threads x 8 elements = 32 rows x 64 cols, consecutive threads -> consecutive columns
  const int sc = tid & 63, sr = tid >> 6;  // rows sr, sr+4, ...
  for (int m0 = 0; m0 < M; m0 += GROWS) {
#pragma unroll
    for (int r = 0; r < GROWS / 4; ++r) {
      const int row = sr + r * 4, m = m0 + row;
      const bool rok = m < M;
      // out-of-range rows/cols are staged as 0 and masked again when the fragments are formed
      const int ci = i0 + sc, cj = j0 + sc;
      Si[row * GPITCH + sc] = (rok && ci < D) ? Sp[(int64_t)m * D + ci] : 0.f;
      Sj[row * GPITCH + sc] = (rok && cj < D) ? Sp[(int64_t)m * D + cj] : 0.f;
    }
    __syncthreads();
    const int rows_here = min(GROWS, M - m0);
#pragma unroll
    for (int ks = 0; ks < GROWS / 4; ++ks) {
      const int row = ks * 4 + fk;
      double a[2], b[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float va = Si[row * GPITCH + wi + q * 16 + fl];
        const float vb = Sj[row * GPITCH + wj + q * 16 + fl];
        const bool oka = row < rows_here && (i0 + wi + q * 16 + fl) < D;
        const bool okb = row < rows_here && (j0 + wj + q * 16 + fl) < D;
        a[q] = oka ? (double)va - mi[q] : 0.0;
        b[q] = okb ? (double)vb - mj[q] : 0.0;
      }
#pragma unroll
      for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          acc[qa][qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qa], b[qb], acc[qa][qb], 0, 0, 0);
    }
    __syncthreads();
  }
  double *Gp = G + (int64_t)p * D * D;
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + qa * 16 + fk + 4 * r;
        const int gj = j0 + wj + qb * 16 + fl;
        if (gi < D && gj < D) {
          const double v = acc[qa][qb][r];
          Gp[(int64_t)gi * D + gj] = v;
          if (ti != tj && mirror) Gp[(int64_t)gj * D + gi] = v;
        }
      }
}

// The same product for matrices that give the chip enough tiles of 128 x 128 (D >= 1024, D % 4 == 0, one matrix): the
// 64 x 64 kernel above converts and centres every operand element in every wave that reads it (12 VALU instructions
// beside 4 MFMAs per k-step) and waits on unprefetched loads: 25 TFLOP/s, 0.31 of the fp64 MFMA peak.  Here a wave owns
// 64 x 64 outputs as 4 x 4 MFMA tiles (16 MFMAs per 8 eight-byte LDS reads), the operands are converted to fp64 and
// centred ONCE, by the thread that stages them (16-byte global loads, two stages of [16 rows][128 cols] fp64 per
// operand = 64 KB of LDS, one barrier per stage), and the loads of stage t+2 are in flight while stage t multiplies.
// Only tiles with tj >= ti are computed; diagonal tiles come out whole (and exactly symmetric: the products commute and
// are summed in the same order), idiff_symmetrize_upper_f64's kernel mirrors the rest.
// The reduction over rows runs in the same order and the same groups of four as in gram_kernel, so the two agree bit for
// bit (checked by tests/test_hip_spectrum.py).
constexpr int GBT = 128;      // tile edge
constexpr int BROWS = 16;    // rows of S per stage
constexpr size_t GRAM_BIG_LDS = (size_t)2 * 2 * BROWS * GBT * sizeof(double);   // 64 KB

__global__ void __launch_bounds__(256, 2)
gram_big_kernel(const float *__restrict__ S, const double *__restrict__ mean, int M, int D, int tiles,
                double *__restrict__ G, int ti_first) {
  extern __shared__ __attribute__((aligned(16))) double glds[];   // [stage][operand][BROWS][GBT]
  int t = blockIdx.x, ti = ti_first;
  while (t >= tiles - ti) { t -= tiles - ti; ++ti; }
  const int tj = ti + t;
  const int i0 = ti * GBT, j0 = tj * GBT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  const int fl = lane & 15, fk = lane >> 4;

  // staging map: thread -> 4 consecutive columns of rows sr and sr + 8 of both operands
  const int sc = (tid & 31) * 4, sr = tid >> 5;
  const bool ci_ok = i0 + sc < D, cj_ok = j0 + sc < D;          // D % 4 == 0: a run of four is inside or outside as a whole
  double mi4[4] = {0.0, 0.0, 0.0, 0.0}, mj4[4] = {0.0, 0.0, 0.0, 0.0};
  if (ci_ok) { for (int e = 0; e < 4; ++e) mi4[e] = mean[i0 + sc + e]; }
  if (cj_ok) { for (int e = 0; e < 4; ++e) mj4[e] = mean[j0 + sc + e]; }
  const float *Si = S + i0 + sc, *Sj = S + j0 + sc;
  float4 pi[2], pj[2];
  int f_m0 = 0;                     // first row of the stage held in pi / pj
  auto fetch = [&](int m0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = m0 + sr + 8 * q;
      const bool ok = m < M;
      pi[q] = (ok && ci_ok) ? *reinterpret_cast<const float4 *>(Si + (int64_t)m * D) : make_float4(0.f, 0.f, 0.f, 0.f);
      pj[q] = (ok && cj_ok) ? *reinterpret_cast<const float4 *>(Sj + (int64_t)m * D) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f_m0 = m0;
  };
  auto stage = [&](int buf) {
    double *base = glds + (size_t)buf * 2 * BROWS * GBT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = sr + 8 * q;
      const bool ok = f_m0 + row < M;          // rows past the end contribute 0, not -mean
      double *di = base + row * GBT + sc, *dj = base + BROWS * GBT + row * GBT + sc;
      const double vi[4] = {(double)pi[q].x - mi4[0], (double)pi[q].y - mi4[1], (double)pi[q].z - mi4[2], (double)pi[q].w - mi4[3]};
      const double vj[4] = {(double)pj[q].x - mj4[0], (double)pj[q].y - mj4[1], (double)pj[q].z - mj4[2], (double)pj[q].w - mj4[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) { di[e] = ok ? vi[e] : 0.0; dj[e] = ok ? vj[e] : 0.0; }
    }
  };

  doublex4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (doublex4){0.0, 0.0, 0.0, 0.0};

  const int nst = (M + BROWS - 1) / BROWS;
  fetch(0);
  stage(0);
  if (nst > 1) fetch(BROWS);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) stage(buf ^ 1);
    if (st + 2 < nst) fetch((st + 2) * BROWS);
    const double *Ai = glds + (size_t)buf * 2 * BROWS * GBT + wi + fl;
    const double *Bj = glds + (size_t)buf * 2 * BROWS * GBT + BROWS * GBT + wj + fl;
#pragma unroll
    for (int ks = 0; ks < BROWS / 4; ++ks) {
      const int row = ks * 4 + fk;
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = Ai[row * GBT + q * 16]; b[q] = Bj[row * GBT + q * 16]; }
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
          acc[qa][qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qa], b[qb], acc[qa][qb], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qa = 0; qa < 4; ++qa)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + qa * 16 + fk + 4 * r;
        const int gj = j0 + wj + qb * 16 + fl;
        if (gi < D && gj < D) G[(int64_t)gi * D + gj] = acc[qa][qb][r];
      }
}

// Matrices the large-tile kernel takes (IDIFF_GRAM_SMALL_TILES=1 forces the 64 x 64 kernel, for A/B runs)
bool gram_big_ok(const float *S, int D) {
  return D >= 1024 && D % 4 == 0 && ((uintptr_t)S & 15) == 0 && !idiff::option(idiff::OPT_GRAM_SMALL_TILES);
}

// diagonal tiles are computed in full by gram_kernel (both triangles inside the tile come from the
// same products in a different order); force exact symmetry afterwards so the Householder sweep may read
// rows where the textbook reads columns.
// The batched small case (BASELINE config 2: P matrices of 1501 x 100): ONE workgroup per matrix.  gram_kernel gives such a
// matrix three 64 x 64 tiles whose lanes are a quarter to a half empty (100 = 64 + 36), re-reads S once per tile and converts /
// centres every operand in every wave that touches it: 13 TFLOP/s, 0.17 of the fp64 matrix peak.  Here D <= 112 = 7 MFMA column
// blocks; a stage of 32 rows is converted to fp64 and centred ONCE by the thread that stages it ([32][112] doubles, two
// stages: 56 KB, two workgroups per CU), the 28 blocks (bi <= bj) of the upper triangle are dealt 7 to a wave so that a wave
// needs at most 7 fragments per k-step (one 8-byte LDS read each) for its 7 MFMAs, the next stage's 16-byte global loads are in
// flight while the current one multiplies, and S is read from HBM exactly once.  Diagonal blocks come out whole and exactly
// symmetric (same products, same order); the rest is mirrored by the epilogue.  Row order of the reduction: k-steps of four
// consecutive rows in ascending order, as gram_kernel -- the two agree bit for bit.
constexpr int SB_COLS = 112;      // 7 column blocks of 16
constexpr int SB_NB = SB_COLS / 16;
constexpr int SB_ROWS = 32;       // rows of S per stage
constexpr int SB_PITCH = SB_COLS; // 224 dwords = 32 mod 64 banks: rows k, k + 1 of a k-step fall on disjoint bank halves
constexpr size_t GRAM_SMALL_LDS = (size_t)(2 * SB_ROWS * SB_PITCH + SB_COLS) * sizeof(double);

// wave W's blocks: row W from the diagonal to the right, plus the short rows that fill it up to 7 blocks
template <int W> struct SbBlocks;
template <> struct SbBlocks<0> { static constexpr int n = 7, bi[7] = {0, 0, 0, 0, 0, 0, 0}, bj[7] = {0, 1, 2, 3, 4, 5, 6}; };
template <> struct SbBlocks<1> { static constexpr int n = 7, bi[7] = {1, 1, 1, 1, 1, 1, 6}, bj[7] = {1, 2, 3, 4, 5, 6, 6}; };
template <> struct SbBlocks<2> { static constexpr int n = 7, bi[7] = {2, 2, 2, 2, 2, 5, 5}, bj[7] = {2, 3, 4, 5, 6, 5, 6}; };
template <> struct SbBlocks<3> { static constexpr int n = 7, bi[7] = {3, 3, 3, 3, 4, 4, 4}, bj[7] = {3, 4, 5, 6, 4, 5, 6}; };

template <int W>
__device__ __forceinline__ void sb_multiply(const double *__restrict__ X /* [SB_ROWS][SB_PITCH] */, doublex4 (&acc)[7], int fl, int fk) {
  using B = SbBlocks<W>;
#pragma unroll
  for (int ks = 0; ks < SB_ROWS / 4; ++ks) {
    const double *row = X + (ks * 4 + fk) * SB_PITCH + fl;
    double f[SB_NB];
#pragma unroll
    for (int b = W; b < SB_NB; ++b) f[b] = row[16 * b];          // wave W touches column blocks W .. 6 only
#pragma unroll
    for (int q = 0; q < B::n; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[B::bi[q]], f[B::bj[q]], acc[q], 0, 0, 0);
  }
}

template <int W>
__device__ __forceinline__ void sb_store(double *__restrict__ Gp, int D, const doublex4 (&acc)[7], int fl, int fk) {
  using B = SbBlocks<W>;
#pragma unroll
  for (int q = 0; q < B::n; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * B::bi[q] + fk + 4 * r, gj = 16 * B::bj[q] + fl;
      if (gi < D && gj < D) {
        Gp[(int64_t)gi * D + gj] = acc[q][r];
        if (B::bi[q] != B::bj[q]) Gp[(int64_t)gj * D + gi] = acc[q][r];
      }
    }
}

__global__ void __launch_bounds__(256, 2)
gram_small_batched_kernel(const float *__restrict__ S, const double *__restrict__ mean, int M, int D, double *__restrict__ G) {
  extern __shared__ __attribute__((aligned(16))) double sbm[];
  double *X0 = sbm, *X1 = sbm + SB_ROWS * SB_PITCH, *mu = sbm + 2 * SB_ROWS * SB_PITCH;
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, fk = lane >> 4;
  const float *Sp = S + (int64_t)p * M * D;
  for (int c = tid; c < SB_COLS; c += 256) mu[c] = c < D ? mean[(int64_t)p * D + c] : 0.0;
  // zero both stages once: the columns D .. 111 and the rows past M are never written again
  for (int e = tid; e < 2 * SB_ROWS * SB_PITCH; e += 256) sbm[e] = 0.0;
  __syncthreads();
  // staging map: a stage is SB_ROWS rows x D/4 float4 (D % 4 == 0 and a 16-byte aligned S are checked by the launcher)
  const int c4n = D >> 2, per_stage = SB_ROWS * c4n;
  constexpr int SLOTS = (SB_ROWS * (SB_COLS / 4) + 255) / 256;     // 4
  float4 v[SLOTS];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int idx = tid + 256 * q, r = idx / c4n, c4 = idx - r * c4n;
      v[q] = (idx < per_stage && m0 + r < M) ? *reinterpret_cast<const float4 *>(Sp + (int64_t)(m0 + r) * D + 4 * c4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](double *X, int m0) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int idx = tid + 256 * q, r = idx / c4n, c4 = idx - r * c4n;
      if (idx < per_stage) {
        const bool ok = m0 + r < M;                              // a row past the end is zero AFTER centring
        double2 lo, hi;
        lo.x = ok ? (double)v[q].x - mu[4 * c4] : 0.0;
        lo.y = ok ? (double)v[q].y - mu[4 * c4 + 1] : 0.0;
        hi.x = ok ? (double)v[q].z - mu[4 * c4 + 2] : 0.0;
        hi.y = ok ? (double)v[q].w - mu[4 * c4 + 3] : 0.0;
        double2 *dst = reinterpret_cast<double2 *>(X + r * SB_PITCH + 4 * c4);
        dst[0] = lo; dst[1] = hi;
      }
    }
  };
  doublex4 acc[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) acc[q] = (doublex4){0.0, 0.0, 0.0, 0.0};
  fetch(0);
  stage(X0, 0);
  __syncthreads();
  const int nst = (M + SB_ROWS - 1) / SB_ROWS;
  for (int t = 0; t < nst; ++t) {
    double *cur = (t & 1) ? X1 : X0, *nxt = (t & 1) ? X0 : X1;
    const bool more = t + 1 < nst;
    if (more) fetch((t + 1) * SB_ROWS);                          // in flight under this stage's MFMAs
    switch (wave) {
      case 0: sb_multiply<0>(cur, acc, fl, fk); break;
      case 1: sb_multiply<1>(cur, acc, fl, fk); break;
      case 2: sb_multiply<2>(cur, acc, fl, fk); break;
      default: sb_multiply<3>(cur, acc, fl, fk); break;
    }
    if (more) stage(nxt, (t + 1) * SB_ROWS);
    __syncthreads();
  }
  double *Gp = G + (int64_t)p * D * D;
  switch (wave) {
    case 0: sb_store<0>(Gp, D, acc, fl, fk); break;
    case 1: sb_store<1>(Gp, D, acc, fl, fk); break;
    case 2: sb_store<2>(Gp, D, acc, fl, fk); break;
    default: sb_store<3>(Gp, D, acc, fl, fk); break;
  }
}

__global__ void __launch_bounds__(256)
symmetrize_diag_tiles_kernel(double *__restrict__ G, int D) {
  const int p = blockIdx.y, tile = blockIdx.x;
  double *Gp = G + (int64_t)p * D * D;
  for (int e = threadIdx.x; e < GT * GT; e += 256) {
    const int a = e / GT, b = e % GT;
    const int i = tile * GT + a, j = tile * GT + b;
    if (a < b && i < D && j < D) Gp[(int64_t)j * D + i] = Gp[(int64_t)i * D + j];
  }
}

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread workgroup; every thread gets the result.  `buf` holds >= 4 doubles.
__device__ __forceinline__ double block_sum_d(double v, double *buf) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
  __syncthreads();
  return buf[0] + buf[1] + buf[2] + buf[3];
}

// Householder reflector for x (length n, x[0] is the sub-diagonal entry): H = I - tau v v^T,
// H x = alpha e_0, v = x - alpha e_0.  Returns tau = 0 when x[1:] is already zero.
struct Reflector { double alpha, v0, tau; };
__device__ __forceinline__ Reflector make_reflector(double x0, double tail_sq) {
  Reflector h;
  if (tail_sq == 0.0) { h.alpha = x0; h.v0 = 0.0; h.tau = 0.0; return h; }
  const double norm = sqrt(x0 * x0 + tail_sq);
  h.alpha = x0 > 0.0 ? -norm : norm;
  h.v0 = x0 - h.alpha;
  h.tau = 2.0 / (h.v0 * h.v0 + tail_sq);
  return h;
}

// ------------------------------------------------------------------------------------------------ tridiag, large D
// Step k works on the trailing block A22 = A[k+1:, k+1:] (n = D-k-1) with x = A[k, k+1:] (row == column).
//   symv  : every workgroup rebuilds the reflector from x (one pass over <= D doubles), then
//           p_i = tau * sum_j A22[i][j] v_j for its rows (one wave per row, coalesced along j).
//           Workgroup 0 also emits v, tau, diag[k], offdiag[k].
//   rank2 : every workgroup rebuilds K = (tau/2) p.v, w = p - K v on the fly and applies
//           A22 -= v w^T + w v^T to its tile (both triangles kept, so rows stay readable as columns).
__global__ void __launch_bounds__(256)
tridiag_symv_kernel(const double *__restrict__ A, int D, int k, double *__restrict__ v, double *__restrict__ pvec,
                    double *__restrict__ diag, double *__restrict__ offd, double *__restrict__ tau_out) {
  __shared__ double red[4];
  extern __shared__ double vs[];  // v for this step, length n
  const int n = D - k - 1;
  const double *x = A + (int64_t)k * D + (k + 1);
  double part = 0.0;
  for (int j = 1 + threadIdx.x; j < n; j += 256) { const double t = x[j]; part += t * t; }
  const double tail_sq = block_sum_d(part, red);
  const Reflector h = make_reflector(x[0], tail_sq);
  for (int j = threadIdx.x; j < n; j += 256) vs[j] = j == 0 ? h.v0 : x[j];
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int j = threadIdx.x; j < n; j += 256) v[j] = vs[j];
    if (threadIdx.x == 0) { diag[k] = A[(int64_t)k * D + k]; offd[k] = h.alpha; *tau_out = h.tau; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const double *row = A + (int64_t)(k + 1 + i) * D + (k + 1);
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += row[j] * vs[j];
    acc = wave_sum_d(acc);
    if (lane == 0) pvec[i] = h.tau * acc;
  }
}

__global__ void __launch_bounds__(256)
tridiag_rank2_kernel(double *__restrict__ A, int D, int k, const double *__restrict__ v,
                     const double *__restrict__ pvec, const double *__restrict__ tau_in) {
  __shared__ double red[4];
  const int n = D - k - 1;
  const double tau = *tau_in;
  if (tau == 0.0) return;
  double part = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) part += pvec[j] * v[j];
  const double K = 0.5 * tau * block_sum_d(part, red);
  // tile: 16 rows x all columns per workgroup iteration, threads along columns
  for (int i0 = blockIdx.x * 16; i0 < n; i0 += gridDim.x * 16) {
    const int rows = min(16, n - i0);
    for (int j = threadIdx.x; j < n; j += 256) {
      const double vj = v[j], wj = pvec[j] - K * vj;
      for (int r = 0; r < rows; ++r) {
        const int i = i0 + r;
        const double vi = v[i], wi = pvec[i] - K * vi;
        A[(int64_t)(k + 1 + i) * D + (k + 1 + j)] -= vi * wj + wi * vj;
      }
    }
  }
}

__global__ void tridiag_tail_kernel(const double *__restrict__ A, int D, double *__restrict__ diag,
                                    double *__restrict__ offd) {
  // last 2 x 2 block: nothing left to reflect
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (D >= 2) {
    diag[D - 2] = A[(int64_t)(D - 2) * D + (D - 2)];
    offd[D - 2] = A[(int64_t)(D - 2) * D + (D - 1)];
  }
  diag[D - 1] = A[(int64_t)(D - 1) * D + (D - 1)];
}


// ------------------------------------------------------------------------------------------------ tridiag, fused step
// One launch per Householder step (D <= FUSED_D_MAX): with (v, p = tau*A22*v, tau) of step k in hand,
//   1. K = (tau/2) p.v, w = p - K v                                   (every workgroup, redundantly, in LDS)
//   2. first row of the updated block -> next reflector (v', tau'), diag[k+1], offd[k+1]
//   3. for its rows: a = A[i][j] - v_i w_j - w_i v_j, store, and accumulate p'_i = tau' * sum_j a * v'_j
// so the trailing matrix is read and written ONCE per step (16 B/element) instead of read twice and
// written once by the symv + rank2 pair.  v/p/tau are ping-ponged between steps.
constexpr int FUSED_D_MAX = 6000;   // 3 vectors of D doubles in LDS

template <int R>
__global__ void __launch_bounds__(256)
tridiag_fused_kernel(double *__restrict__ A, int D, int k, const double *__restrict__ v_in,
                     const double *__restrict__ p_in, const double *__restrict__ tau_in, double *__restrict__ v_out,
                     double *__restrict__ p_out, double *__restrict__ tau_out, double *__restrict__ diag,
                     double *__restrict__ offd) {
  __shared__ double red[4 * 16];
  extern __shared__ double sm[];
  const int base = k + 1, n = D - base;
  double *vs = sm, *ws = sm + n, *vn = sm + 2 * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double tau = *tau_in;
  double part = 0.0;
  for (int j = tid; j < n; j += 256) { const double a = v_in[j], b = p_in[j]; vs[j] = a; ws[j] = b; part += a * b; }
  const double K = 0.5 * tau * block_sum_d(part, red);
  for (int j = tid; j < n; j += 256) ws[j] -= K * vs[j];
  __syncthreads();
  // updated first row of the block (local row 0); entries 1.. form the next x
  const double *row0 = A + (int64_t)base * D + base;
  const double v0 = vs[0], w0 = ws[0];
  part = 0.0;
  for (int j = 1 + tid; j < n; j += 256) {
    const double r = row0[j] - v0 * ws[j] - w0 * vs[j];
    vn[j] = r;
    if (j >= 2) part += r * r;
  }
  const double tail_sq = block_sum_d(part, red);   // contains a barrier: vn[] is visible afterwards
  const Reflector h = make_reflector(vn[1], tail_sq);
  __syncthreads();
  if (tid == 0) vn[1] = h.v0;
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int j = 1 + tid; j < n; j += 256) v_out[j - 1] = vn[j];
    if (tid == 0) {
      *tau_out = h.tau;
      diag[k + 1] = row0[0] - 2.0 * v0 * w0;
      offd[k + 1] = h.alpha;
    }
  }
  for (int i0 = 1 + blockIdx.x * R; i0 < n; i0 += gridDim.x * R) {
    double vi[R], wi[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = min(i0 + r, n - 1);
      vi[r] = vs[i]; wi[r] = ws[i]; acc[r] = 0.0;
    }
    for (int j = 1 + tid; j < n; j += 256) {
      const double vj = vs[j], wj = ws[j], xj = vn[j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (i0 + r < n) {
          double *ap = A + (int64_t)(base + i0 + r) * D + base + j;
          const double a = *ap - vi[r] * wj - wi[r] * vj;
          *ap = a;
          acc[r] += a * xj;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum_d(acc[r]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[wave * 16 + r] = acc[r];
    }
    __syncthreads();
    if (tid < R && i0 + tid < n)
      p_out[i0 + tid - 1] = h.tau * (red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid]);
  }
}

__global__ void tridiag_fused_tail_kernel(const double *__restrict__ A, int D, double *__restrict__ diag,
                                          double *__restrict__ offd) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { diag[D - 1] = A[(int64_t)(D - 1) * D + (D - 1)]; offd[D - 1] = 0.0; }
}


// ------------------------------------------------------------------------------------------------ tridiag, prep + stream
// Two launches per Householder step, vectors indexed by GLOBAL row/column g (step k works on g >= base = k+1):
//   prep   (1 workgroup, 1024 threads): p = tau * (sum of the column-segment partials of the previous stream pass),
//          K = (tau/2) p.v, w = p - K v, then the updated first row of the block -> next reflector v', tau',
//          diag[k+1], offd[k+1].
//   stream (grid = row blocks x 512-column segments): a = A[i][j] - v_i w_j - w_i v_j, store, and the partial
//          dot products part[seg][i] = sum_{j in seg} a * v'_j for the next step.
// The stream pass touches every element once with 16-byte accesses and needs no LDS for the vectors (a thread
// keeps the 3 x 2 vector entries of its two columns in registers), so it also serves D = 12288 (config 5), where
// the Gram matrix (1.2 GB) no longer fits the Infinity Cache and three LDS-resident vectors would not fit a CU.
constexpr int SEG_CHUNKS = 8;                    // 512-column chunks per workgroup of the stream pass
constexpr int SEG_COLS = 512 * SEG_CHUNKS;
constexpr int PREP_THREADS = 512;
constexpr int STREAM_D_MAX = 12288;              // config 5; nothing in the two kernels depends on it any more

__device__ __forceinline__ double block_sum_prep(double v, double *buf /* >= PREP_THREADS / 64 */) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < PREP_THREADS / 64; ++i) s += buf[i];
  return s;
}

// One 512-thread workgroup (two waves per SIMD at 34 registers: inside the 80 the convolutions leave free), two strided passes over the live indices, every per-index value re-read from memory instead
// of being held in per-thread arrays: ~40 registers and 100 bytes of LDS.  That matters more than its own speed: the
// spectrum runs on a side stream beside the score evaluations, whose convolution workgroups leave 80 registers per SIMD
// and 32 KB of LDS free on every CU.  The former 1024-thread / 95-register form needed an EMPTY CU and got one only
// when a convolution kernel drained: each of the 3072 steps then waited ~100 us and the spectrum of one point took as
// long as all the score evaluations of the next (550 ms instead of 68), i.e. it set the pace of the whole pipeline.
__global__ void __launch_bounds__(PREP_THREADS)
tridiag_prep_kernel(const double *__restrict__ A, int D, int k, const double *__restrict__ v, const double *__restrict__ part,
                    int nseg, int direct_p, const double *__restrict__ tau_in, double *__restrict__ w,
                    double *__restrict__ v_next, double *__restrict__ tau_next, double *__restrict__ diag,
                    double *__restrict__ offd) {
  __shared__ double red[PREP_THREADS / 64];
  __shared__ double bc[3];
  const int base = k + 1, tid = threadIdx.x;
  const double tau = *tau_in;
  // pass 1: p = tau * (sum of the column-segment partials), parked in w[]; dot = p.v
  double dot = 0.0;
  for (int g = base + tid; g < D; g += PREP_THREADS) {
    double sum = 0.0;
    if (direct_p) sum = part[g];
    else { for (int sg = 0; sg < nseg; ++sg) sum += part[(int64_t)sg * D + g]; sum *= tau; }
    const double vg = v[g];
    w[g] = sum;
    dot += sum * vg;
    if (g == base) { bc[0] = vg; bc[1] = sum; }
  }
  const double K = 0.5 * tau * block_sum_prep(dot, red);     // (its barriers also publish bc[0..1])
  const double v0 = bc[0], w0 = bc[1] - K * v0;               // v[base], w[base]
  // pass 2: w = p - K v; the updated first row of the block rv = A[base][g] - v0 w_g - w0 v_g becomes the next
  // reflector's tail (parked in v_next[]); its squared norm beyond the leading element
  const double *row0 = A + (int64_t)base * D;
  double tail = 0.0;
  for (int g = base + tid; g < D; g += PREP_THREADS) {
    const double vg = v[g];
    const double wg = w[g] - K * vg;                          // the same thread wrote w[g] in pass 1
    w[g] = wg;
    if (g > base) {
      const double r = row0[g] - v0 * wg - w0 * vg;
      v_next[g] = r;
      if (g > base + 1) tail += r * r;
      else bc[2] = r;                                         // g == base + 1: the leading element
    }
  }
  const double tail_sq = block_sum_prep(tail, red);
  if (tid == 0) {
    const Reflector h = make_reflector(base + 1 < D ? bc[2] : 0.0, tail_sq);
    if (base + 1 < D) v_next[base + 1] = h.v0;
    *tau_next = h.tau;
    diag[k + 1] = row0[base] - 2.0 * v0 * w0;
    offd[k + 1] = h.alpha;
  }
}

template <int R>
__global__ void __launch_bounds__(256)
tridiag_stream_kernel(double *__restrict__ A, int D, int k, const double *__restrict__ v, const double *__restrict__ w,
                      const double *__restrict__ x, double *__restrict__ part) {
  __shared__ double red[4 * R];
  const int lo = k + 2;                       // first live row / column of the shrunken block
  const int galign = lo & ~1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = lo + blockIdx.x * R;
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  // a workgroup sweeps SEG_CHUNKS chunks of 512 columns for its R rows: one partial per (segment, row) instead of one
  // per 512 columns, i.e. four times fewer values for the single-workgroup prep pass to gather
#pragma unroll 1
  for (int c = 0; c < SEG_CHUNKS; ++c) {
    const int g0 = galign + (blockIdx.y * SEG_CHUNKS + c) * 512 + 2 * tid;   // this thread's column pair (g0, g0+1), g0 even
    if (g0 >= D) break;
    const bool in0 = g0 >= lo, in1 = g0 + 1 >= lo && g0 + 1 < D;
    double vj0 = 0, vj1 = 0, wj0 = 0, wj1 = 0, xj0 = 0, xj1 = 0;
    if (in0) { vj0 = v[g0]; wj0 = w[g0]; xj0 = x[g0]; }
    if (in1) { vj1 = v[g0 + 1]; wj1 = w[g0 + 1]; xj1 = x[g0 + 1]; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gi = i0 + r;
      if (gi < D && (in0 || in1)) {
        const double vi = v[gi], wi = w[gi];
        double2 *ap = reinterpret_cast<double2 *>(A + (int64_t)gi * D + g0);
        double2 a = *ap;
        a.x -= vi * wj0 + wi * vj0;             // masked columns carry zeros: the element is rewritten unchanged
        a.y -= vi * wj1 + wi * vj1;
        *ap = a;
        acc[r] += a.x * xj0 + a.y * xj1;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = wave_sum_d(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[wave * R + r] = acc[r];
  }
  __syncthreads();
  if (tid < R && i0 + tid < D)
    part[(int64_t)blockIdx.y * D + i0 + tid] = red[tid] + red[R + tid] + red[2 * R + tid] + red[3 * R + tid];
}

// ------------------------------------------------------------------------------------------------ tridiag, D <= 128 in LDS
// One workgroup per matrix; the whole matrix lives in LDS (pitch D+1 doubles).  Used for the batched
// k-sphere workload (P matrices of 100 x 100).
__global__ void __launch_bounds__(256)
tridiag_small_kernel(const double *__restrict__ G, int D, double *__restrict__ diag, double *__restrict__ offd) {
  extern __shared__ double sm[];
  const int pitch = D + 1;
  double *A = sm;                 // [D][pitch]
  double *v = sm + D * pitch;     // [D]
  double *w = v + D;              // [D]
  double *red = w + D;            // [4]
  const int p = blockIdx.x, tid = threadIdx.x;
  const double *Gp = G + (int64_t)p * D * D;
  for (int e = tid; e < D * D; e += 256) A[(e / D) * pitch + (e % D)] = Gp[e];
  __syncthreads();
  double *dg = diag + (int64_t)p * D, *od = offd + (int64_t)p * D;
  for (int k = 0; k + 2 < D; ++k) {
    const int n = D - k - 1;
    const double *x = A + k * pitch + (k + 1);
    double part = 0.0;
    for (int j = 1 + tid; j < n; j += 256) part += x[j] * x[j];
    const double tail_sq = block_sum_d(part, red);
    const Reflector h = make_reflector(x[0], tail_sq);
    if (tid == 0) { dg[k] = A[k * pitch + k]; od[k] = h.alpha; }
    if (h.tau == 0.0) continue;  // uniform across the workgroup
    for (int j = tid; j < n; j += 256) v[j] = j == 0 ? h.v0 : x[j];
    __syncthreads();
    // p = tau * A22 v : two threads per row
    {
      const int row = tid >> 1, half = tid & 1;
      double acc = 0.0;
      if (row < n) {
        const double *ar = A + (k + 1 + row) * pitch + (k + 1);
        for (int j = half; j < n; j += 2) acc += ar[j] * v[j];
      }
      acc += __shfl_xor(acc, 1, 64);
      if (row < n && half == 0) w[row] = h.tau * acc;
    }
    __syncthreads();
    double pv = 0.0;
    for (int j = tid; j < n; j += 256) pv += w[j] * v[j];
    const double K = 0.5 * h.tau * block_sum_d(pv, red);
    for (int j = tid; j < n; j += 256) w[j] -= K * v[j];
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e - i * n;
      A[(k + 1 + i) * pitch + (k + 1 + j)] -= v[i] * w[j] + w[i] * v[j];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (D >= 2) { dg[D - 2] = A[(D - 2) * pitch + (D - 2)]; od[D - 2] = A[(D - 2) * pitch + (D - 1)]; }
    dg[D - 1] = A[(D - 1) * pitch + (D - 1)];
    if (D >= 1) od[D - 1] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ tridiag, D <= 128 in registers
// The batched k-sphere case again (4096 matrices of 100 x 100): tridiag_small_kernel keeps the matrix in LDS and walks it with
// ~8 workgroup barriers per Householder step -- 4 us per step, 0.034 of the LDS bandwidth, one matrix per CU at a time.  Here
// the matrix lives in REGISTERS: 256 threads per matrix, the lane pair (2 i, 2 i + 1) holds row i -- lane h of the pair the
// columns j = 2 t + h as 64 doubles with compile-time indices only (the step loop is unrolled in blocks of 16 columns: block b
// runs steps k = 16 b .. 16 b + 15 on the columns >= 16 b).  The symmetric matrix-vector product p_i = sum_j A[i][j] v_j and
// the rank-2 update A[i][j] -= v_i p_j + u_i v_j (u = p - 2 c v, c = tau v.p / 2) are fp64 FMA streams over a lane's own
// registers; the only cross-lane traffic is one DPP swap inside the pair per sum (the "wavefront shuffle" of a step) and the
// exchange of the vectors x (column k: row j holds A[j][k] = A[k][j]), v and p through 4 KB of LDS, read back as broadcast
// 16-byte loads; the scalars of a step (|x_tail|^2, v.p) are recomputed by every lane pair from those broadcasts, so a step
// has TWO workgroup barriers and no reduction tree.  Finished rows / columns are masked by ZEROS in the exchanged vectors
// (x_j = 0 for j <= k, p_i = 0 for i <= k), never by predicates in the inner loops; a wave whose 32 rows are all finished
// skips the loops (a uniform branch) and only keeps the barriers.  252 registers: two waves per SIMD, two matrices per CU at once.
constexpr int RT_NP = 128;        // padded dimension
constexpr int RT_NH = RT_NP / 2;  // columns per lane of a pair (largest form)
constexpr int RT_THREADS = 256;
constexpr int RT_BLK = 16;        // steps (= columns) per unrolled block
constexpr int RT_CHUNK = 8;       // register slots per scheduling group of the unrolled inner loops
constexpr int RT_PL = RT_NH + 2;  // plane pitch in LDS: the two parities of a 16-byte broadcast read fall on different banks

// exchange a double with the other lane of the pair (DPP quad_perm [1, 0, 3, 2]: no LDS, no ds_bpermute)
__device__ __forceinline__ double rt_partner(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// `cond ? a[i] : r` chains over register-array elements are rewritten by the optimiser into ONE load through a selected
// POINTER -- a dynamic index, which sends the whole array to scratch.  Passing the candidate through an empty asm keeps it a
// value (no instruction is emitted).
__device__ __forceinline__ double rt_val(double v) {
  asm volatile("" : "+v"(v));
  return v;
}

// v_rsq_f64 / v_rcp_f64 seeds + two Newton steps: full double accuracy in ~12 instructions; the IEEE sqrt and division of
// make_reflector are ~200 instructions on the critical path of EVERY step here (as in sbr.hip's factorisations)
__device__ __forceinline__ double rt_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double rt_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
__device__ __forceinline__ Reflector rt_reflector(double x0, double tail_sq) {
  Reflector h;
  if (tail_sq == 0.0) { h.alpha = x0; h.v0 = 0.0; h.tau = 0.0; return h; }
  const double n2 = fma(x0, x0, tail_sq);
  const double norm = n2 * rt_rsqrt(n2);
  h.alpha = x0 > 0.0 ? -norm : norm;
  h.v0 = x0 - h.alpha;
  h.tau = 2.0 * rt_rcp(fma(h.v0, h.v0, tail_sq));
  return h;
}

struct RtShared {
  double x[2][2][RT_PL];          // [step parity][column parity][column >> 1]
  double p[2][RT_PL], v[2][RT_PL];
};

// Inner loops: RT_CHUNK register slots per group, the NEXT group's broadcast reads issued before the current group's FMAs
// (two register sets), groups fenced for the scheduler -- unfenced it hoists every read of the unrolled loop to the top and
// spills; fenced without the look-ahead every group pays the LDS latency in full (5800 cycles per step measured).
template <int NH, int JLO>
__device__ __forceinline__ void rt_block(double (&a)[NH], int D, int tid, double *__restrict__ dg, double *__restrict__ od,
                                         RtShared &sh) {
  constexpr int T0 = JLO / 2;
  constexpr int TP = T0 + RT_BLK / 2 + 1 < NH ? T0 + RT_BLK / 2 + 1 : NH;   // slots that can hold column k + 1 in this block
  constexpr int NG = (NH - T0 + RT_CHUNK - 1) / RT_CHUNK;                   // groups of the inner loops
  const int row = tid >> 1, h = tid & 1;
  const int wave_rows_end = ((tid >> 6) + 1) * 32;       // first row beyond this wave's
  for (int k = JLO; k < JLO + RT_BLK && k + 2 < D; ++k) {
    double (*X)[RT_PL] = sh.x[k & 1];
    // my element of column k (slot k >> 1 of the lanes with h == (k & 1)): a select over the block's 8 candidate slots
    double xk = rt_val(a[T0]);
#pragma unroll
    for (int t = 1; t < RT_BLK / 2; ++t)
      if (T0 + t < NH) xk = ((k >> 1) == T0 + t) ? rt_val(a[T0 + t]) : xk;
    if (h == (k & 1)) {
      X[row & 1][row >> 1] = row > k ? xk : 0.0;        // x_row = A[row][k]
      if (row == k) dg[k] = xk;
    }
    __syncthreads();
    const bool live = wave_rows_end > k + 1;             // some row of this wave is still below the pivot (uniform per wave)
    const int kp = k + 1, kpt = kp >> 1;
    const bool mine = h == (kp & 1);                     // this lane holds column k + 1
    double q = 0.0, ss = 0.0, akp1 = 0.0;
    if (live) {
      double q0 = 0.0, q1 = 0.0, s0 = 0.0, s1 = 0.0;
      double2 xb[2][RT_CHUNK / 2];
#pragma unroll
      for (int u = 0; u < RT_CHUNK / 2; ++u)
        if (T0 + 2 * u < NH) xb[0][u] = *reinterpret_cast<const double2 *>(&X[h][T0 + 2 * u]);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < RT_CHUNK / 2; ++u)
            if (T0 + (g + 1) * RT_CHUNK + 2 * u < NH)
              xb[(g + 1) & 1][u] = *reinterpret_cast<const double2 *>(&X[h][T0 + (g + 1) * RT_CHUNK + 2 * u]);
        }
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 2; ++u) {
          const int t = T0 + g * RT_CHUNK + 2 * u;
          if (t < NH) {
            const double2 xx = xb[g & 1][u];
            q0 = fma(a[t], xx.x, q0);
            q1 = fma(a[t + 1], xx.y, q1);
            if (t < TP) {                        // compile-time: only these slots can be column k + 1
              const bool hit = mine && t == kpt;
              s0 = fma(hit ? 0.0 : xx.x, xx.x, s0);
              akp1 = hit ? rt_val(a[t]) : akp1;
            } else {
              s0 = fma(xx.x, xx.x, s0);
            }
            if (t + 1 < TP) {
              const bool hit = mine && t + 1 == kpt;
              s1 = fma(hit ? 0.0 : xx.y, xx.y, s1);
              akp1 = hit ? rt_val(a[t + 1]) : akp1;
            } else {
              s1 = fma(xx.y, xx.y, s1);
            }
          }
        }
      }
      q = q0 + q1; ss = s0 + s1;
      q += rt_partner(q);
      ss += rt_partner(ss);                  // both lanes of a pair add the same two numbers: identical in every lane
      akp1 += rt_partner(akp1);              // one of the two is zero
    }
    const double x0 = X[kp & 1][kpt];
    // a finished wave needs no scalars: it writes p = 0 and v = x = 0 for its rows (already there from its last live step)
    const Reflector hh = live ? rt_reflector(x0, ss) : Reflector{0.0, 0.0, 0.0};
    if (row == kp && h == 0) od[k] = hh.alpha;           // row k + 1's wave is live at step k (row k's is not when k % 32 == 31)
    const double vi = row == kp ? hh.v0 : X[row & 1][row >> 1];
    // v = x - alpha e_{k+1}  ->  A v = A x - alpha A[:, k + 1];   tau = 0 (nothing to annihilate): p = 0, the step is a no-op
    const double pi = (live && row > k) ? hh.tau * (q - hh.alpha * akp1) : 0.0;
    if (h == 0) { sh.p[row & 1][row >> 1] = pi; sh.v[row & 1][row >> 1] = vi; }
    __syncthreads();
    if (live) {
      double c0 = 0.0, c1 = 0.0;
      double2 pb[2][RT_CHUNK / 2], vb[2][RT_CHUNK / 2];
#pragma unroll
      for (int u = 0; u < RT_CHUNK / 2; ++u)
        if (T0 + 2 * u < NH) {
          pb[0][u] = *reinterpret_cast<const double2 *>(&sh.p[h][T0 + 2 * u]);
          vb[0][u] = *reinterpret_cast<const double2 *>(&sh.v[h][T0 + 2 * u]);
        }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < RT_CHUNK / 2; ++u)
            if (T0 + (g + 1) * RT_CHUNK + 2 * u < NH) {
              pb[(g + 1) & 1][u] = *reinterpret_cast<const double2 *>(&sh.p[h][T0 + (g + 1) * RT_CHUNK + 2 * u]);
              vb[(g + 1) & 1][u] = *reinterpret_cast<const double2 *>(&sh.v[h][T0 + (g + 1) * RT_CHUNK + 2 * u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 2; ++u)
          if (T0 + g * RT_CHUNK + 2 * u < NH) {
            c0 = fma(pb[g & 1][u].x, vb[g & 1][u].x, c0);
            c1 = fma(pb[g & 1][u].y, vb[g & 1][u].y, c1);
          }
      }
      double c = c0 + c1;
      c += rt_partner(c);
      c *= 0.5 * hh.tau;
      const double ui = pi - 2.0 * c * vi, nvi = -vi, nui = -ui;
#pragma unroll
      for (int u = 0; u < RT_CHUNK / 2; ++u)
        if (T0 + 2 * u < NH) {
          pb[0][u] = *reinterpret_cast<const double2 *>(&sh.p[h][T0 + 2 * u]);
          vb[0][u] = *reinterpret_cast<const double2 *>(&sh.v[h][T0 + 2 * u]);
        }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < RT_CHUNK / 2; ++u)
            if (T0 + (g + 1) * RT_CHUNK + 2 * u < NH) {
              pb[(g + 1) & 1][u] = *reinterpret_cast<const double2 *>(&sh.p[h][T0 + (g + 1) * RT_CHUNK + 2 * u]);
              vb[(g + 1) & 1][u] = *reinterpret_cast<const double2 *>(&sh.v[h][T0 + (g + 1) * RT_CHUNK + 2 * u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 2; ++u) {
          const int t = T0 + g * RT_CHUNK + 2 * u;
          if (t < NH) {
            a[t] = fma(nui, vb[g & 1][u].x, fma(nvi, pb[g & 1][u].x, a[t]));
            a[t + 1] = fma(nui, vb[g & 1][u].y, fma(nvi, pb[g & 1][u].y, a[t + 1]));
          }
        }
      }
    }
    // (p / v are rewritten only after the next step's first barrier, x alternates between two buffers: no third barrier)
  }
}

// NH = register slots per lane: 64 covers D <= 128, 52 covers D <= 104 (the config-2 width 100 with 19 % fewer FMAs and LDS
// reads per step and fewer registers), 32 covers D <= 64.
template <int NH, int JLO>
__device__ __forceinline__ void rt_blocks(double (&a)[NH], int D, int tid, double *__restrict__ dg, double *__restrict__ od,
                                          RtShared &sh) {
  if constexpr (JLO < 2 * NH) {
    rt_block<NH, JLO>(a, D, tid, dg, od, sh);
    rt_blocks<NH, JLO + RT_BLK>(a, D, tid, dg, od, sh);
  }
}

template <int NH>
__global__ void __launch_bounds__(RT_THREADS, 2)
tridiag_reg_kernel(const double *__restrict__ G, int D, double *__restrict__ diag, double *__restrict__ offd) {
  __shared__ __attribute__((aligned(16))) RtShared sh;
  const int p = blockIdx.x, tid = threadIdx.x, row = tid >> 1, h = tid & 1;
  const double *Gp = G + (int64_t)p * D * D;
  double *dg = diag + (int64_t)p * D, *od = offd + (int64_t)p * D;
  double a[NH];
#pragma unroll
  for (int t = 0; t < NH; ++t) a[t] = (row < D && 2 * t + h < D) ? Gp[(int64_t)row * D + 2 * t + h] : 0.0;
  for (int e = tid; e < (int)(sizeof(RtShared) / sizeof(double)); e += RT_THREADS) reinterpret_cast<double *>(&sh)[e] = 0.0;
  __syncthreads();
  rt_blocks<NH, 0>(a, D, tid, dg, od, sh);
  // the last 2 x 2 block: A[D-2][D-2], A[D-2][D-1] (row D - 2) and A[D-1][D-1] (row D - 1)
  double e2 = 0.0, e1 = 0.0;
#pragma unroll
  for (int t = 0; t < NH; ++t) {
    e2 = (2 * t + h == D - 2) ? rt_val(a[t]) : e2;
    e1 = (2 * t + h == D - 1) ? rt_val(a[t]) : e1;
  }
  e2 += rt_partner(e2);           // the other lane of the pair contributes 0
  e1 += rt_partner(e1);
  if (h == 0) {
    if (D >= 2 && row == D - 2) { dg[D - 2] = e2; od[D - 2] = e1; }
    if (row == D - 1) { dg[D - 1] = e1; od[D - 1] = 0.0; }
  }
}

// ------------------------------------------------------------------------------------------------ bisection
// Thread j of matrix p brackets the j-th smallest eigenvalue of the symmetric tridiagonal (d, e) with the
// Sturm sequence of leading principal minors, division-free:
//   p_0 = 1, p_1 = d_0 - x, p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1};   #{eigenvalues < x} = #sign changes.
// (d_i, e_{i-1}^2) pairs sit in LDS as double2 (one broadcast ds_read_b128 per step); the loop is unrolled by 8
// with the next 8 pairs loaded before the current 8 are consumed, so one dependent FMA paces a step.
// The matrix is normalised by 1/span first (|d_i - x| <= 2, e^2 <= 1), so p grows by < 3x per step and can only
// shrink by ~2^-53 per step: renormalising the pair to exponent 0 every 8 steps (v_frexp_exp + v_ldexp) keeps it
// far from overflow/underflow.  A sign change is the xor of two sign bits (integer ops); an exact zero counts as
// positive, which yields the same total as the textbook convention because p_{i+1} = -e^2 p_{i-1} after a zero.
struct SturmState { double pm, pc; int count; };

__device__ __forceinline__ void sturm_step(SturmState &s, double di, double e2, double x) {
  const double pn = fma(di - x, s.pc, -(e2 * s.pm));
  s.count += (unsigned)(__double2hiint(pn) ^ __double2hiint(s.pc)) >> 31;
  s.pm = s.pc; s.pc = pn;
}

__device__ __forceinline__ void sturm_renorm(SturmState &s) {
  const double mag = fabs(s.pc);
  if (mag != 0.0) {
    const int ex = -ilogb(mag);
    s.pc = ldexp(s.pc, ex); s.pm = ldexp(s.pm, ex);
  }
}

template <bool LDS, int G>
__global__ void __launch_bounds__(256)
bisect_kernel(const double *__restrict__ diag, const double *__restrict__ offd, int D, double *__restrict__ eig,
              float *__restrict__ sv) {
  extern __shared__ double2 sh2[];  // [D] when LDS: (d_i, e_{i-1}^2) / span, e_{-1} = 0
  __shared__ double red[8];
  const int p = blockIdx.y;
  const double *d = diag + (int64_t)p * D, *e = offd + (int64_t)p * D;
  const int tid = threadIdx.x;
  // Gershgorin interval
  double lo = INFINITY, hi = -INFINITY;
  bool poisoned = false;                 // fmin / fmax DROP a NaN operand: a poisoned tridiagonal must be seen explicitly
  for (int i = tid; i < D; i += 256) {
    const double el = i > 0 ? e[i - 1] : 0.0, er = i + 1 < D ? e[i] : 0.0;
    const double r = fabs(el) + fabs(er);
    poisoned |= !(fabs(d[i]) + r < INFINITY);
    lo = fmin(lo, d[i] - r);
    hi = fmax(hi, d[i] + r);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
  if ((tid & 63) == 0) { red[tid >> 6] = lo; red[4 + (tid >> 6)] = hi; }
  if (__syncthreads_or(poisoned)) {
    // the tridiagonalisation reported a failure (NaN / inf diag or offdiag: band-reduction residual, stalled chase, non-finite
    // scores): every output of this matrix is NaN -- never a silently wrong spectrum (sqrt(fmax(NaN, 0)) would be 0)
    const int jj = blockIdx.x * 256 + tid;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    if (jj < D) {
      if (eig) eig[(int64_t)p * D + jj] = nanv;
      if (sv) sv[(int64_t)p * D + jj] = (float)nanv;
    }
    return;
  }
  lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  const double span = fmax(fmax(fabs(lo), fabs(hi)), 1e-300);
  const double inv = 1.0 / span;
  if (LDS) {
    for (int i = tid; i < D; i += 256) {
      const double el = i > 0 ? e[i - 1] * inv : 0.0;
      sh2[i] = make_double2(d[i] * inv, el * el);
    }
    __syncthreads();
  }
  // G = 1: one thread per eigenvalue, plain bisection (the batched k-sphere case already fills the chip).
  // G = 8: eight adjacent lanes share an eigenvalue and probe the bracket at a + (b - a) * (1..8) / 8 in one sweep: three
  //        bits per sweep instead of one.  A sweep is ISSUE bound (3 fp64 + 4 integer instructions per step), so when one
  //        matrix leaves most SIMDs idle, 8x the threads cost nothing and the 45 dependent sweeps become ~16.
  const int j = (blockIdx.x * 256 + tid) / G, sub = tid % G;
  if (j >= D) return;
  // normalised bracket, widened by the rounding of the Gershgorin sums
  double a = lo * inv - 2.3e-16 * (double)D, b = hi * inv + 2.3e-16 * (double)D;
  auto count_below = [&](double x) -> int {
    SturmState s;
    s.pm = 1.0; s.pc = (LDS ? sh2[0].x : d[0] * inv) - x; s.count = (unsigned)__double2hiint(s.pc) >> 31;
    int i = 1;
    if (LDS) {
      double2 cur[8], nxt[8];
      if (i + 8 <= D) {
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = sh2[i + u];
      }
      for (; i + 8 <= D; i += 8) {
        const bool more = i + 16 <= D;
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) nxt[u] = sh2[i + 8 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sturm_step(s, cur[u].x, cur[u].y, x);
        sturm_renorm(s);
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
      }
      for (; i < D; ++i) sturm_step(s, sh2[i].x, sh2[i].y, x);
    } else {
      for (; i < D; ++i) {
        const double ev = e[i - 1] * inv;
        sturm_step(s, d[i] * inv, ev * ev, x);
        if ((i & 7) == 0) sturm_renorm(s);
      }
    }
    return s.count;
  };
  if (G == 1) {
    for (int it = 0; it < 120; ++it) {
      const double mid = 0.5 * (a + b);
      if (mid <= a || mid >= b) break;
      if (count_below(mid) > j) b = mid; else a = mid;
      if (b - a <= 1e-13 * fmax(fabs(a), fabs(b)) + 1e-22) break;
    }
  } else {
    const int lane = tid & 63;
    bool done = false;
    for (int it = 0; it < 48; ++it) {
      const double w8 = (b - a) * (1.0 / G);
      const double x = sub == G - 1 ? b : a + w8 * (double)(sub + 1);
      const bool above = count_below(x) > j;                     // lambda_j < x
      const unsigned long long m = __ballot(above);
      const unsigned bits = (unsigned)(m >> (lane & ~(G - 1))) & ((1u << G) - 1u);
      const int f = bits ? __ffs(bits) - 1 : G - 1;              // counts are monotone in x: first probe above lambda_j
      const double na = f == 0 ? a : a + w8 * (double)f;
      const double nb = f == G - 1 ? b : a + w8 * (double)(f + 1);
      if (!(nb - na < b - a)) done = true;                       // no representable progress left
      a = na; b = nb;
      if (b - a <= 1e-13 * fmax(fabs(a), fabs(b)) + 1e-22) done = true;
      if (__all(done)) break;                                    // the groups of a wave leave together (ballot needs them)
    }
    if (sub != 0) return;
  }
  const double lam = 0.5 * (a + b) * span;
  if (eig) eig[(int64_t)p * D + j] = lam;
  if (sv) sv[(int64_t)p * D + (D - 1 - j)] = (float)sqrt(fmax(lam, 0.0));
}

// D > 8192: the (d_i, e_{i-1}^2) pairs no longer fit the LDS of one workgroup.  All threads of a workgroup walk the same
// index range in every Sturm sweep, so the pairs are staged tile by tile (BT pairs = 32 KB) and each thread advances its
// own recurrence through the tile: the one-thread-streams-from-L2 form it replaces took 80 ms at D = 12288.
constexpr int BT = 2048;
__global__ void __launch_bounds__(256)
bisect_tiled_kernel(const double *__restrict__ diag, const double *__restrict__ offd, int D, double *__restrict__ eig,
                    float *__restrict__ sv) {
  __shared__ double2 tile[BT];
  __shared__ double red[8];
  const int p = blockIdx.y;
  const double *d = diag + (int64_t)p * D, *e = offd + (int64_t)p * D;
  const int tid = threadIdx.x;
  double lo = INFINITY, hi = -INFINITY;
  bool poisoned = false;                 // fmin / fmax DROP a NaN operand: a poisoned tridiagonal must be seen explicitly
  for (int i = tid; i < D; i += 256) {
    const double el = i > 0 ? e[i - 1] : 0.0, er = i + 1 < D ? e[i] : 0.0;
    const double r = fabs(el) + fabs(er);
    poisoned |= !(fabs(d[i]) + r < INFINITY);
    lo = fmin(lo, d[i] - r);
    hi = fmax(hi, d[i] + r);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
  if ((tid & 63) == 0) { red[tid >> 6] = lo; red[4 + (tid >> 6)] = hi; }
  if (__syncthreads_or(poisoned)) {
    // the tridiagonalisation reported a failure (NaN / inf diag or offdiag: band-reduction residual, stalled chase, non-finite
    // scores): every output of this matrix is NaN -- never a silently wrong spectrum (sqrt(fmax(NaN, 0)) would be 0)
    const int jj = blockIdx.x * 256 + tid;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    if (jj < D) {
      if (eig) eig[(int64_t)p * D + jj] = nanv;
      if (sv) sv[(int64_t)p * D + jj] = (float)nanv;
    }
    return;
  }
  lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  const double span = fmax(fmax(fabs(lo), fabs(hi)), 1e-300);
  const double inv = 1.0 / span;
  const int j = min(blockIdx.x * 256 + tid, D - 1);           // surplus threads shadow the last eigenvalue
  double a = lo * inv - 2.3e-16 * (double)D, b = hi * inv + 2.3e-16 * (double)D;
  for (int it = 0; it < 120; ++it) {
    const double mid = 0.5 * (a + b);
    const bool live = mid > a && mid < b && (b - a > 1e-13 * fmax(fabs(a), fabs(b)) + 1e-22);
    if (!__syncthreads_or(live)) break;
    SturmState st;
    st.pm = 1.0; st.pc = d[0] * inv - mid; st.count = (unsigned)__double2hiint(st.pc) >> 31;
    for (int base = 0; base < D; base += BT) {
      __syncthreads();
      for (int i = tid; i < BT; i += 256) {
        const int gi = base + i;
        double2 v = make_double2(0.0, 0.0);
        if (gi < D) { const double el = gi > 0 ? e[gi - 1] * inv : 0.0; v = make_double2(d[gi] * inv, el * el); }
        tile[i] = v;
      }
      __syncthreads();
      const int i0 = base == 0 ? 1 : 0, n = min(BT, D - base);
      int i = i0;
      for (; i + 8 <= n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sturm_step(st, tile[i + u].x, tile[i + u].y, mid);
        sturm_renorm(st);
      }
      for (; i < n; ++i) sturm_step(st, tile[i].x, tile[i].y, mid);
      sturm_renorm(st);
    }
    if (live) { if (st.count > j) b = mid; else a = mid; }
  }
  if (blockIdx.x * 256 + tid >= D) return;
  const double lam = 0.5 * (a + b) * span;
  if (eig) eig[(int64_t)p * D + j] = lam;
  if (sv) sv[(int64_t)p * D + (D - 1 - j)] = (float)sqrt(fmax(lam, 0.0));
}

constexpr int SMALL_D_MAX = 128;

size_t small_lds_bytes(int D) { return ((size_t)D * (D + 1) + 2 * D + 8) * sizeof(double); }

int bisect_launch(const double *diag, const double *offd, int P, int D, double *eig, float *sv, hipStream_t st) {
  const bool use_lds = D <= 8192;
  const size_t lds = use_lds ? (size_t)D * sizeof(double2) : 0;
  if (lds > 64 * 1024) {
    static idiff::AttrGuard guard;
    const void *fns[3] = {reinterpret_cast<const void *>(bisect_kernel<true, 1>), reinterpret_cast<const void *>(bisect_kernel<true, 8>),
                          reinterpret_cast<const void *>(bisect_kernel<true, 16>)};
    if (int rc = idiff::set_dynamic_lds_once(guard, fns, 3, 8192 * (int)sizeof(double2), "bisect")) return rc;
  }
  // 64 eigenvalues per workgroup would spread the work over more CUs, but the chain length (D steps per
  // bisection) is the latency; 256 threads keep the broadcast LDS traffic low.
  const bool few = (int64_t)P * idiff::ceil_div(D, 256) <= 64;     // most SIMDs would idle: 8 lanes per eigenvalue
  // 16 lanes per eigenvalue (four bits per sweep: ~12 dependent sweeps instead of ~16) while that still leaves one wave per SIMD
  // (P D / 4 waves on 1024 SIMDs); beyond it the extra lanes would queue behind each other
  if (use_lds && few && (int64_t)P * D * 16 / 64 <= 1024)
    hipLaunchKernelGGL((bisect_kernel<true, 16>), dim3(idiff::ceil_div(D * 16, 256), P), dim3(256), lds, st, diag, offd, D, eig, sv);
  else if (use_lds && few)
    hipLaunchKernelGGL((bisect_kernel<true, 8>), dim3(idiff::ceil_div(D * 8, 256), P), dim3(256), lds, st, diag, offd, D, eig, sv);
  else if (use_lds)
    hipLaunchKernelGGL((bisect_kernel<true, 1>), dim3(idiff::ceil_div(D, 256), P), dim3(256), lds, st, diag, offd, D, eig, sv);
  else   // D > 8192: pairs staged through LDS tile by tile
    hipLaunchKernelGGL(bisect_tiled_kernel, dim3(idiff::ceil_div(D, 256), P), dim3(256), 0, st, diag, offd, D, eig, sv);
  return idiff::launch_status("bisect");
}

}  // namespace

namespace idiff {
int64_t sbr_scratch_doubles(int D);                                                                     // sbr.hip
int sbr_tridiagonalize(double *G, int D, double *diag, double *offd, double *scratch, hipStream_t st);  // sbr.hip
int sbr_to_band(double *G, int D, double *scratch, hipStream_t st);
int sbr_band_ld();
int sbr_chase_is_systolic(int D);
}

using namespace idiff;

// doubles of scratch idiff_symtridiag_f64 needs for a D x D matrix (0 for the LDS-resident path)
IDIFF_API int64_t idiff_symtridiag_scratch_doubles(int D) {
  if (D <= SMALL_D_MAX) return 0;
  const int64_t onestage = 4 * (int64_t)D + 16 + (int64_t)((D + 511) / 512 + 1) * D;
  const int64_t twostage = sbr_scratch_doubles(D);
  return onestage > twostage ? onestage : twostage;
}

// Stage 1 of the two-stage path alone (parity tests, profiler): G -> lower band of half-width 32, column-major with
// leading dimension idiff_symband_ld(): band[j * ld + k] = B[j + k][j]; the band is the first D * ld doubles of scratch.
IDIFF_API int idiff_symband_ld(void) { return sbr_band_ld(); }
IDIFF_API int idiff_symband_f64(double *G, int D, double *scratch, void *stream) {
  if (!G || !scratch || D <= SMALL_D_MAX) return fail("symband: needs D > %d, G and scratch", SMALL_D_MAX);
  return sbr_to_band(G, D, scratch, (hipStream_t)stream);
}

IDIFF_API int idiff_colmean_f64(const float *S, int P, int M, int D, double *mean, double *scratch, void *stream) {
  if (!S || !mean || !scratch || P <= 0 || M <= 0 || D <= 0) return fail("colmean: bad arguments");
  if (P > 65535) return fail("colmean: P too large");
  double *part = scratch;  // [P][MEAN_SPLITS][D]
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(D, 256), MEAN_SPLITS, P), dim3(256), 0, st, S, M, D, part);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(D, 256), P), dim3(256), 0, st, part, M, D, mean);
  return launch_status("colmean");
}

IDIFF_API int idiff_centered_gram_f64(const float *S, const double *mean, int P, int M, int D, double *G, void *stream) {
  if (!S || !mean || !G || P <= 0 || M <= 0 || D <= 0) return fail("centered_gram: bad arguments");
  if (P > 65535) return fail("centered_gram: P too large");
  hipStream_t st = (hipStream_t)stream;
  if (P == 1 && gram_big_ok(S, D)) {
    const int tb = ceil_div(D, GBT);
    hipLaunchKernelGGL(gram_big_kernel, dim3(tb * (tb + 1) / 2), dim3(256), GRAM_BIG_LDS, st, S, mean, M, D, tb, G, 0);
    if (int rc = launch_status("centered_gram")) return rc;
    return idiff_symmetrize_upper_f64(G, D, stream);
  }
  if (D > 48 && D <= SB_COLS && D % 4 == 0 && ((uintptr_t)S & 15) == 0 && !option(OPT_GRAM_SMALL_TILES)) {
    // batched small matrices (config 2): one workgroup per matrix, S read once, operands centred once
    static AttrGuard guard;
    const void *fn = reinterpret_cast<const void *>(gram_small_batched_kernel);
    if (int rc = set_dynamic_lds_once(guard, &fn, 1, (int)GRAM_SMALL_LDS, "centered_gram")) return rc;
    hipLaunchKernelGGL(gram_small_batched_kernel, dim3(P), dim3(256), GRAM_SMALL_LDS, st, S, mean, M, D, G);
    return launch_status("centered_gram");
  }
  const int tiles = ceil_div(D, GT);
  hipLaunchKernelGGL(gram_kernel, dim3(tiles * (tiles + 1) / 2, P), dim3(256), 0, st, S, mean, M, D, tiles, G, 0, 1);
  hipLaunchKernelGGL(symmetrize_diag_tiles_kernel, dim3(tiles, P), dim3(256), 0, st, G, D);
  return launch_status("centered_gram");
}

// Rows [row0, row1) of the UPPER triangle of the centred Gram matrix (row0, row1 multiples of 64 or D): G[i][j] for
// row0 <= i < row1, j >= T * (i / T) with T = 64, or 128 when the rows are multiples of 128 and the large-matrix kernel
// takes them; nothing else is written.  The row-sharded pipeline computes G one block of rows at
// a time so that the all-reduce of block b runs while block b + 1 is being computed; idiff_symmetrize_upper_f64
// completes the matrix afterwards.
IDIFF_API int idiff_centered_gram_rows_f64(const float *S, const double *mean, int M, int D, int row0, int row1, double *G,
                                           void *stream) {
  if (!S || !mean || !G || M <= 0 || D <= 0) return fail("centered_gram_rows: bad arguments");
  if (row0 < 0 || row1 > D || row0 >= row1 || row0 % GT || (row1 % GT && row1 != D))
    return fail("centered_gram_rows: rows [%d, %d) must be tile-aligned (%d) inside [0, %d)", row0, row1, GT, D);
  if (gram_big_ok(S, D) && row0 % GBT == 0 && (row1 % GBT == 0 || row1 == D)) {
    const int tb = ceil_div(D, GBT), b0 = row0 / GBT, b1 = ceil_div(row1, GBT);
    int count = 0;
    for (int ti = b0; ti < b1; ++ti) count += tb - ti;
    hipLaunchKernelGGL(gram_big_kernel, dim3(count), dim3(256), GRAM_BIG_LDS, (hipStream_t)stream, S, mean, M, D, tb, G, b0);
    return launch_status("centered_gram_rows");
  }
  const int tiles = ceil_div(D, GT), t0 = row0 / GT, t1 = ceil_div(row1, GT);
  int count = 0;
  for (int ti = t0; ti < t1; ++ti) count += tiles - ti;
  hipLaunchKernelGGL(gram_kernel, dim3(count, 1), dim3(256), 0, (hipStream_t)stream, S, mean, M, D, tiles, G, t0, 0);
  return launch_status("centered_gram_rows");
}

namespace {
__global__ void __launch_bounds__(256) symmetrize_upper_kernel(double *__restrict__ G, int D) {
  // lower <- upper, 32 x 32 tiles through LDS so that both sides are coalesced
  __shared__ double tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    tile[r][tx] = (i < D && j < D) ? G[(int64_t)i * D + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = bj * 32 + r, i = bi * 32 + tx;      // element (j, i) of the lower triangle <- (i, j)
    if (i < D && j < D && j > i) G[(int64_t)j * D + i] = tile[tx][r];
  }
}
}  // namespace

IDIFF_API int idiff_symmetrize_upper_f64(double *G, int D, void *stream) {
  if (!G || D <= 0) return fail("symmetrize_upper: bad arguments");
  const int nb = ceil_div(D, 32);
  hipLaunchKernelGGL(symmetrize_upper_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, G, D);
  return launch_status("symmetrize_upper");
}

// Which tridiagonalisation idiff_symtridiag_f64 would run for a D x D matrix on the current device, given the switches:
// 0 LDS-resident (D <= 128), 1 two-stage with the single-launch systolic chase, 2 two-stage with the wavefront chase
// (too many nodes to be co-resident on this device, or IDIFF_CHASE_WAVEFRONT), 3 one-stage sweep (IDIFF_TRIDIAG_ONESTAGE).
IDIFF_API int idiff_symtridiag_plan(int D) {
  if (D <= SMALL_D_MAX) return 0;
  if (option(OPT_TRIDIAG_ONESTAGE)) return 3;
  return sbr_chase_is_systolic(D) ? 1 : 2;
}

IDIFF_API int idiff_symtridiag_f64(double *G, int P, int D, double *diag, double *offdiag, double *scratch, void *stream) {
  if (!G || !diag || !offdiag || P <= 0 || D <= 0) return fail("symtridiag: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (D <= RT_NP && !option(OPT_TRIDIAG_ONESTAGE)) {
    // register-resident form (a lane pair per row); IDIFF_TRIDIAG_ONESTAGE keeps the LDS-resident kernel for A/B and as the
    // fail-soft alternative
    if (D <= 64) hipLaunchKernelGGL(tridiag_reg_kernel<32>, dim3(P), dim3(RT_THREADS), 0, st, G, D, diag, offdiag);
    else if (D <= 104) hipLaunchKernelGGL(tridiag_reg_kernel<52>, dim3(P), dim3(RT_THREADS), 0, st, G, D, diag, offdiag);
    else hipLaunchKernelGGL(tridiag_reg_kernel<64>, dim3(P), dim3(RT_THREADS), 0, st, G, D, diag, offdiag);
    return launch_status("tridiag_reg");
  }
  if (D <= SMALL_D_MAX) {
    const size_t lds = small_lds_bytes(D);
    if (lds > 64 * 1024) {
      static AttrGuard guard;
      const void *fn = reinterpret_cast<const void *>(tridiag_small_kernel);
      if (int rc = set_dynamic_lds_once(guard, &fn, 1, (int)small_lds_bytes(SMALL_D_MAX), "symtridiag")) return rc;
    }
    hipLaunchKernelGGL(tridiag_small_kernel, dim3(P), dim3(256), lds, st, G, D, diag, offdiag);
    return launch_status("tridiag_small");
  }
  if (!scratch) return fail("symtridiag: scratch (idiff_symtridiag_scratch_doubles(D) doubles) required for D > %d", SMALL_D_MAX);
  if (!option(OPT_TRIDIAG_ONESTAGE)) {
    // two-stage: blocked band reduction on the fp64 matrix cores, then bulge chasing on the compact band (sbr.hip)
    for (int p = 0; p < P; ++p) {
      const int rc = sbr_tridiagonalize(G + (int64_t)p * D * D, D, diag + (int64_t)p * D, offdiag + (int64_t)p * D, scratch, st);
      if (rc) return rc;
    }
    return 0;
  }
  if ((size_t)D * sizeof(double) > 60 * 1024) {
    static AttrGuard guard2;
    const void *fn = reinterpret_cast<const void *>(tridiag_symv_kernel);
    if (int rc = set_dynamic_lds_once(guard2, &fn, 1, 160 * 1024 - 256, "symtridiag")) return rc;
    if ((size_t)D * sizeof(double) > 160 * 1024 - 256) return fail("symtridiag: D=%d exceeds the LDS-resident reflector", D);
  }
  double *v = scratch, *pvec = scratch + D, *tau = scratch + 2 * D;
  if (D % 2 == 0 && D <= STREAM_D_MAX && true /* the per-step fused variant below is kept for odd D */) {
    // scratch: v0[D] | v1[D] | w[D] | p0[D] | tau[2] | part[nseg_max][D]
    const int nseg_max = ceil_div(D, SEG_COLS) + 1;
    double *vb[2] = {scratch, scratch + (int64_t)D};
    double *wv = scratch + 2 * (int64_t)D, *p0 = scratch + 3 * (int64_t)D, *tb = scratch + 4 * (int64_t)D;
    double *part = scratch + 4 * (int64_t)D + 16;
    (void)nseg_max;
    for (int p = 0; p < P; ++p) {
      double *A = G + (int64_t)p * D * D;
      double *dg = diag + (int64_t)p * D, *od = offdiag + (int64_t)p * D;
      const int n0 = D - 1;
      // step 0: reflector + p_0 by the symv kernel (v and p are local-indexed there: shift to global index 1)
      hipLaunchKernelGGL(tridiag_symv_kernel, dim3(max(1, min(ceil_div(n0, 4), 1024))), dim3(256),
                         (size_t)n0 * sizeof(double), st, A, D, 0, vb[0] + 1, p0 + 1, dg, od, tb);
      int nseg_prev = 1;
      for (int k = 0; k + 2 < D; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        hipLaunchKernelGGL(tridiag_prep_kernel, dim3(1), dim3(PREP_THREADS), 0, st, A, D, k, vb[cur], k == 0 ? p0 : part,
                           nseg_prev, k == 0 ? 1 : 0, tb + cur, wv, vb[nxt], tb + nxt, dg, od);
        const int lo = k + 2, galign = lo & ~1;
        const int rows = D - lo, nseg = ceil_div(D - galign, SEG_COLS);
        // rows per workgroup: 4 while the block is tall, 2 near the end (8 rows per workgroup left too few workgroups in
        // flight for a bandwidth-bound pass: 63.8 -> 57.4 ms at D = 3072, 2.62 -> 2.48 s at D = 12288)
        if (rows >= 768)
          hipLaunchKernelGGL(tridiag_stream_kernel<4>, dim3(ceil_div(rows, 4), nseg), dim3(256), 0, st, A, D, k, vb[cur], wv, vb[nxt], part);
        else
          hipLaunchKernelGGL(tridiag_stream_kernel<2>, dim3(ceil_div(rows, 2), nseg), dim3(256), 0, st, A, D, k, vb[cur], wv, vb[nxt], part);
        nseg_prev = nseg;
      }
      hipLaunchKernelGGL(tridiag_fused_tail_kernel, dim3(1), dim3(64), 0, st, A, D, dg, od);
    }
    return launch_status("symtridiag_stream");
  }
  if (D <= FUSED_D_MAX) {
    {
      static AttrGuard guard_fused;
      const void *fns[3] = {reinterpret_cast<const void *>(tridiag_fused_kernel<16>),
                            reinterpret_cast<const void *>(tridiag_fused_kernel<8>),
                            reinterpret_cast<const void *>(tridiag_fused_kernel<4>)};
      if (int rc = set_dynamic_lds_once(guard_fused, fns, 3, 3 * FUSED_D_MAX * 8, "symtridiag")) return rc;
    }
    // ping-pong buffers: [v0 | p0 | v1 | p1 | tau0 tau1]
    double *vb[2] = {scratch, scratch + 2 * (int64_t)D}, *pb[2] = {scratch + D, scratch + 3 * (int64_t)D};
    double *tb = scratch + 4 * (int64_t)D;
    for (int p = 0; p < P; ++p) {
      double *A = G + (int64_t)p * D * D;
      double *dg = diag + (int64_t)p * D, *od = offdiag + (int64_t)p * D;
      const int n0 = D - 1;
      hipLaunchKernelGGL(tridiag_symv_kernel, dim3(max(1, min(ceil_div(n0, 4), 1024))), dim3(256),
                         (size_t)n0 * sizeof(double), st, A, D, 0, vb[0], pb[0], dg, od, tb);
      for (int k = 0; k + 2 < D; ++k) {
        const int n = D - k - 1, cur = k & 1, nxt = cur ^ 1;
        const size_t lds = (size_t)3 * n * sizeof(double);
        const int rows = n - 1;
        if (rows >= 16 * 100000) {  // measured: 16 rows per workgroup leave CUs idle (n/16 < 256 workgroups)
          hipLaunchKernelGGL(tridiag_fused_kernel<16>, dim3(max(1, ceil_div(rows, 16))), dim3(256), lds, st, A, D, k, vb[cur],
                             pb[cur], tb + cur, vb[nxt], pb[nxt], tb + nxt, dg, od);
        } else if (rows >= 8 * 256) {
          hipLaunchKernelGGL(tridiag_fused_kernel<8>, dim3(max(1, ceil_div(rows, 8))), dim3(256), lds, st, A, D, k, vb[cur],
                             pb[cur], tb + cur, vb[nxt], pb[nxt], tb + nxt, dg, od);
        } else {
          hipLaunchKernelGGL(tridiag_fused_kernel<4>, dim3(max(1, ceil_div(rows, 4))), dim3(256), lds, st, A, D, k, vb[cur],
                             pb[cur], tb + cur, vb[nxt], pb[nxt], tb + nxt, dg, od);
        }
      }
      hipLaunchKernelGGL(tridiag_fused_tail_kernel, dim3(1), dim3(64), 0, st, A, D, dg, od);
    }
    return launch_status("symtridiag_fused");
  }
  for (int p = 0; p < P; ++p) {
    double *A = G + (int64_t)p * D * D;
    double *dg = diag + (int64_t)p * D, *od = offdiag + (int64_t)p * D;
    for (int k = 0; k + 2 < D; ++k) {
      const int n = D - k - 1;
      const int g1 = max(1, min(ceil_div(n, 4), 1024));
      hipLaunchKernelGGL(tridiag_symv_kernel, dim3(g1), dim3(256), (size_t)n * sizeof(double), st, A, D, k, v, pvec, dg, od, tau);
      const int g2 = max(1, min(ceil_div(n, 16), 1024));
      hipLaunchKernelGGL(tridiag_rank2_kernel, dim3(g2), dim3(256), 0, st, A, D, k, v, pvec, tau);
    }
    hipLaunchKernelGGL(tridiag_tail_kernel, dim3(1), dim3(64), 0, st, A, D, dg, od);
  }
  return launch_status("symtridiag");
}

IDIFF_API int idiff_tridiag_eigvals_f64(const double *diag, const double *offdiag, int P, int D, double *eig, void *stream) {
  if (!diag || !offdiag || !eig || P <= 0 || D <= 0) return fail("tridiag_eigvals: bad arguments");
  if (P > 65535) return fail("tridiag_eigvals: P too large");
  return bisect_launch(diag, offdiag, P, D, eig, nullptr, (hipStream_t)stream);
}

// workspace layout (doubles): mean[P*D] | colsum partials[P*32*D] | G[P*D*D] | diag[P*D] | offd[P*D] |
//                             scratch[idiff_symtridiag_scratch_doubles(D)]
IDIFF_API int64_t idiff_spectrum_workspace_bytes(int P, int M, int D) {
  (void)M;
  if (P <= 0 || D <= 0) return 0;
  const int64_t n = (int64_t)P * D * (1 + MEAN_SPLITS) + (int64_t)P * D * D + 2 * (int64_t)P * D + idiff_symtridiag_scratch_doubles(D);
  return n * (int64_t)sizeof(double);
}

IDIFF_API int idiff_spectrum_f32(const float *S, int P, int M, int D, void *workspace, int64_t workspace_bytes, float *sv,
                                 double *eig_out, void *stream) {
  if (!S || !workspace || !sv || P <= 0 || M <= 0 || D <= 0) return fail("spectrum: bad arguments");
  // M < D is fine: the Gram matrix is D x D either way and has D - min(M - 1, D) zero eigenvalues; the caller keeps the
  // leading min(M, D) singular values, which is what torch.linalg.svd returns (dim_reduction.py:197)
  if (P > 65535) return fail("spectrum: P too large");
  if (workspace_bytes < idiff_spectrum_workspace_bytes(P, M, D)) return fail("spectrum: workspace too small");
  if (((uintptr_t)workspace & 7) != 0) return fail("spectrum: workspace must be 8-byte aligned");
  double *mean = (double *)workspace;
  double *G = mean + (int64_t)P * D * (1 + MEAN_SPLITS);
  double *diag = G + (int64_t)P * D * D;
  double *offd = diag + (int64_t)P * D;
  double *scratch = offd + (int64_t)P * D;
  int rc;
  if ((rc = idiff_colmean_f64(S, P, M, D, mean, mean + (int64_t)P * D, stream))) return rc;
  if ((rc = idiff_centered_gram_f64(S, mean, P, M, D, G, stream))) return rc;
  if ((rc = idiff_symtridiag_f64(G, P, D, diag, offd, scratch, stream))) return rc;
  return bisect_launch(diag, offd, P, D, eig_out, sv, (hipStream_t)stream);
}
