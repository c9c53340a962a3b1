// Dense contractions of the score networks on the fp32 matrix cores of gfx950.
//
//   C[m, n] = epilogue( sum_k A[m, k] * Bt[n, k] )
//
// One kernel template serves every contraction on the hot path:
//   LINEAR  A is a row-major [M, K] matrix (Linear layers, NIN / 1x1 convs on NHWC activations, the
//           attention products Q K^T and P V with V held transposed), optionally batched by strides.
//   CONV    A is gathered on the fly from an NHWC activation tensor: m = (b, oy, ox),
//           k = (ky, kx, cin), A[m, k] = x[b, oy*stride + ky - pad, ox*stride + kx - pad, cin] or 0
//           outside the image (implicit GEMM; im2col is never written to HBM).
// Both operands are "K-contiguous panels": weights are packed once at load time as Bt[N][K]
// (= [Cout][KH][KW][Cin] for convs, the native [out][in] layout for Linear).
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak; there is no
// TF32/xf32 on gfx950 and the parity target is the reference's fp32 CPU path).  Lane l feeds
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; since the order of the k-reduction is free, each
// lane half takes FOUR consecutive k from one ds_read_b128 (half h supplies k = 8g + 4h + j to the j-th
// MFMA of k-group g), so a wave issues (TM + TN) 16-byte LDS reads per 4*TM*TN MFMAs.
//
// Tile: BM x BN x 32 per workgroup, waves arranged WARPS_M x WARPS_N, each wave TM x TN MFMA tiles.
// LDS panels are [rows][32 + 4] floats: the 16-byte pad makes the ds_read_b128 of 16 consecutive rows hit
// 16 distinct 4-bank slots (conflict-free) and keeps every row 16-byte aligned for ds_write_b128.
// Pipeline: global -> registers for k-tile t+1 is issued before the MFMAs of tile t (latency hides under
// 64-cycle MFMAs), registers -> the other LDS buffer after them, one barrier per k-tile.
// Workgroup ids are remapped so that consecutive M-tiles (which share conv halo rows and the weight
// panel) land on the same XCD's L2.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_PITCH = BK + 4;

struct IgemmParams {
  const float *A;   // LINEAR: matrix; CONV: NHWC activations
  const float *Bt;
  float *C;
  int64_t lda, ldb, ldc;
  int64_t strideA, strideB, strideC;
  int M, N, K;
  int tiles_m, tiles_n;
  // conv geometry
  int H, W, Cin, OH, OW, KW, stride, pad;
  idiff_epilogue ep;
  int has_ep;
  uint32_t a_bytes, b_bytes;  // extent of one batch slice of A / Bt for the buffer-addressed kernel
  int vec_ep;                 // epilogue operands allow 16-byte accesses (set by the launcher)
  const float *A2;            // LINEAR, optional: columns K1.. of A live in a second matrix with the same row pitch
  int K1;                     // (the concatenation [A | A2] is never materialised); 0 = single source
  uint32_t a2_bytes;
  uint32_t c_bytes, res_bytes;  // extents of one batch slice of C and of the residual; buf_ep = they fit a buffer descriptor
  int buf_ep;
  const float *scale_a, *scale_b;   // fp16-pair form: device pointers to {s, 1 / s}, the power of two A / Bt is multiplied by before
                                    // its cut (NULL: 1); the sums are multiplied by the product of the inverses
};

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int BM, int BN, int WARPS_M, int WARPS_N, bool CONV, bool VEC>
__global__ void __launch_bounds__(WARPS_M *WARPS_N * 64)
igemm_kernel(const IgemmParams p) {
  constexpr int T = WARPS_M * WARPS_N * 64;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_PER_T = BM * (BK / 4) / T;  // float4 loads per thread per k-tile
  constexpr int B_PER_T = BN * (BK / 4) / T;
  static_assert(A_PER_T >= 1 && B_PER_T >= 1, "tile too small for the thread count");
  static_assert(T % (BK / 4) == 0, "a thread keeps one k-column of float4s");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *As = lds;                           // [2][BM][LDS_PITCH]
  float *Bs = lds + 2 * BM * LDS_PITCH;      // [2][BN][LDS_PITCH]

  // ---- XCD-aware, bijective block -> tile map (blocks b and b+8 share an XCD)
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int batch = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WARPS_N) * WTM, wn0 = (wave % WARPS_N) * WTN;

  const float *Ab = p.A + (int64_t)batch * p.strideA;
  const float *Bb = p.Bt + (int64_t)batch * p.strideB;

  // ---- per-thread load coordinates: float4 column kc (fixed), rows tid/8 + i*(T/8)
  const int kc = (tid & 7) * 4;
  const int row_base = tid >> 3;
  constexpr int ROW_STEP = T / 8;

  // A rows
  int64_t a_off[A_PER_T];  // LINEAR: row offset; CONV: pixel-plane base (b*H*W)
  int a_iy[A_PER_T], a_ix[A_PER_T];
  bool a_ok[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int m = m0 + row_base + i * ROW_STEP;
    a_ok[i] = m < p.M;
    if (CONV) {
      const int mm = a_ok[i] ? m : 0;
      const int ox = mm % p.OW;
      const int t = mm / p.OW;
      const int oy = t % p.OH;
      const int b = t / p.OH;
      a_off[i] = (int64_t)b * p.H * p.W;
      a_iy[i] = oy * p.stride - p.pad;
      a_ix[i] = ox * p.stride - p.pad;
    } else {
      a_off[i] = (int64_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  int64_t b_off[B_PER_T];
  bool b_ok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int n = n0 + row_base + i * ROW_STEP;
    b_ok[i] = n < p.N;
    b_off[i] = (int64_t)n * p.ldb;
  }

  float4 a_reg[A_PER_T], b_reg[B_PER_T];

  auto load_scalar4 = [&](const float *base, int k, bool ok) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      if (k + 0 < p.K) v.x = base[0];
      if (k + 1 < p.K) v.y = base[1];
      if (k + 2 < p.K) v.z = base[2];
      if (k + 3 < p.K) v.w = base[3];
    }
    return v;
  };

  auto fetch = [&](int kt) {
    const int k = kt * BK + kc;
    if (CONV) {
      // k -> (tap, cin); Cin % 4 == 0 so a float4 never straddles a tap
      const int tap = k / p.Cin, c = k - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const bool kok = k < p.K;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        const bool ok = kok && a_ok[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        a_reg[i] = ok ? ldg4(Ab + (a_off[i] + (int64_t)iy * p.W + ix) * p.Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        if (VEC) a_reg[i] = (a_ok[i] && k < p.K) ? ldg4(Ab + a_off[i] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        else a_reg[i] = load_scalar4(Ab + a_off[i] + k, k, a_ok[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      if (VEC) b_reg[i] = (b_ok[i] && k < p.K) ? ldg4(Bb + b_off[i] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      else b_reg[i] = load_scalar4(Bb + b_off[i] + k, k, b_ok[i]);
    }
  };

  auto stage = [&](int buf) {
    float *a_dst = As + buf * BM * LDS_PITCH;
    float *b_dst = Bs + buf * BN * LDS_PITCH;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
      *reinterpret_cast<float4 *>(a_dst + (row_base + i * ROW_STEP) * LDS_PITCH + kc) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      *reinterpret_cast<float4 *>(b_dst + (row_base + i * ROW_STEP) * LDS_PITCH + kc) = b_reg[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  const int nkt = (p.K + BK - 1) / BK;

  fetch(0);
  stage(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    const float *a_src = As + buf * BM * LDS_PITCH + (wm0 + frag_row) * LDS_PITCH + frag_k;
    const float *b_src = Bs + buf * BN * LDS_PITCH + (wn0 + frag_row) * LDS_PITCH + frag_k;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a_src + i * 32 * LDS_PITCH + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b_src + j * 32 * LDS_PITCH + g * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nkt) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float *Cb = p.C + (int64_t)batch * p.strideC;
  const idiff_epilogue &ep = p.ep;
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + col_l;
    if (n >= p.N) continue;
    const float bias = (p.has_ep && ep.bias) ? ep.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bias;
        if (p.has_ep) {
          if (ep.rowbias) v += ep.rowbias[(int64_t)(m / ep.rows_per_group) * ep.ld_rowbias + n];
          v = idiff::act_apply(v, ep.act);
          if (ep.residual) v += ep.residual[(int64_t)m * ep.ld_residual + n];
          v *= ep.out_scale;
          if (ep.rowscale) v *= ep.rowscale[m / ep.rows_per_group];
        }
        Cb[(int64_t)m * p.ldc + n] = v;
      }
    }
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool CONV, bool VEC>
int launch_cfg(IgemmParams &p, int batch, hipStream_t st) {
  p.tiles_m = idiff::ceil_div(p.M, BM);
  p.tiles_n = idiff::ceil_div(p.N, BN);
  constexpr size_t lds_bytes = (size_t)2 * (BM + BN) * LDS_PITCH * sizeof(float);
  auto kern = igemm_kernel<BM, BN, WARPS_M, WARPS_N, CONV, VEC>;
  if (lds_bytes > 64 * 1024) {
    static idiff::AttrGuard guard;                 // one per template instantiation, one bit per device
    const void *fn = reinterpret_cast<const void *>(kern);
    if (int rc = idiff::set_dynamic_lds_once(guard, &fn, 1, (int)lds_bytes, "igemm")) return rc;
  }
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WARPS_M * WARPS_N * 64), lds_bytes, st, p);
  return idiff::launch_status(CONV ? "igemm_conv" : "igemm_linear");
}


// ------------------------------------------------------------------------------------------------------------
// Pipelined variant (the one the score networks run on): same tiling and MFMA core, but
//   * operands are addressed through raw buffer descriptors with 32-bit byte offsets; an offset of
//     0xFFFFFFFF is out of range and returns zeros, so image borders, M/N tails and the K tail need no
//     branches and no data selects (the compiled loop above spends ~250 VALU + 8 exec-masked branches per
//     k-tile on that);
//   * the conv loader keeps (tap, channel) as uniform counters advanced per k-tile (Cin % 32 == 0: a k-tile
//     never straddles a tap) and a per-row 9-bit tap-validity mask computed once, instead of two integer
//     divisions per thread per k-tile;
//   * global loads run TWO k-tiles ahead: in iteration t the registers holding tile t+1 are written to
//     LDS, then reloaded with tile t+2, while the 64 MFMAs of tile t issue.  The non-MFMA part of an iteration
//     shrinks to ~60 instructions, which matters because the waves that share a SIMD run the same program and
//     drift into lockstep (all in their non-MFMA phase at once = idle matrix pipe);
//   * two residency forms: DBUF (two LDS buffers, one barrier per k-tile, 2 workgroups per CU) for grids that
//     cannot fill the chip four times over, and the single-buffer form (two barriers per k-tile, 36.9 KB LDS,
//     128 registers -> FOUR workgroups per CU) whose extra resident waves cover each other's barrier and
//     staging phases: 135-139 TFLOP/s on the conv shapes of NCSN++ (0.86-0.88 of the fp32 MFMA peak).
// Requires 16-byte aligned K-contiguous operands, slices < 4 GiB and (conv) Cin % 32 == 0; everything else
// takes the general kernel above.
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t OOB = 0xFFFFFFFFu;

// ---- split-precision products (SPLIT): every fp32 operand element is cut EXACTLY into three bf16 pieces of 8 mantissa bits
// each, x = x1 + x2 + x3 (truncation: x1 = the top 16 bits of x, r = x - x1 is exact, x2 = the top 16 bits of r, x3 = r - x2 is
// exact and has at most 8 significant bits), and a product a b is formed as the six partial products of weight >= 2^-16,
//     a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1        (smallest first),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  What is left out (a2 b3 + a3 b2 + a3 b3) is below 2^-23 |a b|: the
// size of the rounding of ONE fp32 product.  Measured on [2.3 M x 128] x [128 x 128] against an fp64 contraction of the same
// fp32 inputs (scripts/bf16x6_probe.hip): 1.7e-7 relative for this form, 2.0e-7 for the k-ordered fp32 fma chain of
// v_mfma_f32_32x32x2_f32 -- the split form is not a reduced-precision mode.  The bf16 matrix core retires a 32x32x16 block in
// 32 cycles where the fp32 one needs 8 x 64 for the same k extent: six of them are 2.7x cheaper, and unlike the fp32 MFMA they
// run beside the vector ALU, which is where the splitting (about 5 instructions per element) goes.
// bf16 elements per LDS row of a split plane: 64 bytes, NO padding -- the four 16-byte chunks of a row are XOR-swizzled by
// (row ^ (row >> 2)) & 3, which makes the b128 fragment reads of 16 consecutive rows hit every bank once, and the three
// planes of a 128 x 32 + 128 x 32 tile pair take 48 KB instead of 60: THREE workgroups per CU instead of two (the k-tile
// stamps of scripts/igemm_phases.py: 2300 cycles of LDS reads + MFMAs against 2170 of barrier / cutting / staging / fetch
// per k-tile and wave, 13 k-cycles of prologue + epilogue per 8 k-tiles -- work that only other resident waves can cover).
constexpr int SP = BK;
__device__ __forceinline__ int split_swz(int row) { return (row ^ (row >> 2)) & 3; }

__device__ __forceinline__ void split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3) {
#ifdef IDIFF_IGEMM_DIAG_NO_CUT   // timing-only build (scripts/igemm_ab.py): no cutting arithmetic, results wrong by construction
  p1 = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
  p2 = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
  p3 = p1;
  return;
#endif
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t t1[4], t2[4], t3[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    t1[e] = __float_as_uint(x[e]) & 0xffff0000u;
    const float r1 = x[e] - __uint_as_float(t1[e]);
    t2[e] = __float_as_uint(r1) & 0xffff0000u;
    t3[e] = __float_as_uint(r1 - __uint_as_float(t2[e]));
  }
  p1 = make_uint2((t1[0] >> 16) | t1[1], (t1[2] >> 16) | t1[3]);
  p2 = make_uint2((t2[0] >> 16) | t2[1], (t2[2] >> 16) | t2[3]);
  p3 = make_uint2((t3[0] >> 16) | (t3[1] & 0xffff0000u), (t3[2] >> 16) | (t3[3] & 0xffff0000u));
}
// ---- fp16-pair products (SPLIT == 2): an fp32 operand as hi = fp16(v), lo = fp16(v - hi) (22 significand bits), a product as
// lo hi + hi lo + hi hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- THREE matrix instructions per 32 x 32 x 16 block where
// the bf16 cut needs six, and 3.5 vector instructions per element to cut where that needs 6.5.  The matrix pipes are what the
// six-product form runs out of (profiles/HISTORY_r04.md: 180 us of pipe time at 2.4 GHz per [573440 x 256] x [256 x 256]^T launch,
// ~270 at the clock the chip holds under that load), so this is the form that is faster, not another tiling.  fp16 has 5 exponent
// bits: the caller vouches for the ranges (include/idiff_hip.h, idiff_gemm_pairs_f32) -- A as it is, |a| < 65504 and of order one
// (below 0.25 the low half goes subnormal: absolute error 2^-25 per element instead of relative 2^-22), Bt multiplied by the power
// of two that brings its largest element into [2^11, 2^12) and the sums divided by it again, exactly.
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void pair4(const float4 v, uint2 &hi, uint2 &lo) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const f2 v01 = {v.x, v.y}, v23 = {v.z, v.w};
  const h2 h01 = __builtin_convertvector(v01, h2), h23 = __builtin_convertvector(v23, h2);
  const f2 r01 = v01 - __builtin_convertvector(h01, f2), r23 = v23 - __builtin_convertvector(h23, f2);   // exact
  const h2 l01 = __builtin_convertvector(r01, h2), l23 = __builtin_convertvector(r23, h2);
  hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
  lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
}
bool aligned16(const void *ptr) { return ((uintptr_t)ptr & 15) == 0; }

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(float4, v);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool CONV, bool DBUF, int SPLIT>   // SPLIT: 0 fp32 MFMA, 1 three bf16, 2 fp16 pairs
__global__ void __launch_bounds__(WARPS_M *WARPS_N * 64, (DBUF || SPLIT) ? 2 : 4)
igemm_pipe_kernel(const IgemmParams p) {
#ifdef IDIFF_IGEMM_PHASES
  const uint64_t ph_entry = __builtin_amdgcn_s_memtime();
#endif
  static_assert(!(SPLIT && DBUF), "the split form uses one LDS buffer");
  constexpr int T = WARPS_M * WARPS_N * 64;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_PER_T = BM * (BK / 4) / T;
  constexpr int B_PER_T = BN * (BK / 4) / T;
  constexpr int ROW_STEP = T / 8;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *As = lds;
  float *Bs = lds + (DBUF ? 2 : 1) * BM * LDS_PITCH;
  // SPLIT: three bf16 (two fp16) planes per operand and buffer, [buffer][plane][row][SP]
  constexpr int PLANES = SPLIT == 2 ? 2 : 3;
  unsigned short *As16 = reinterpret_cast<unsigned short *>(lds);
  unsigned short *Bs16 = As16 + (DBUF ? 2 : 1) * PLANES * BM * SP;

  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int batch = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WARPS_N) * WTM, wn0 = (wave % WARPS_N) * WTN;

  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + (int64_t)batch * p.strideA), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB =
      __builtin_amdgcn_make_buffer_rsrc((void *)(p.Bt + (int64_t)batch * p.strideB), 0, (int)p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 =
      __builtin_amdgcn_make_buffer_rsrc((void *)(p.K1 > 0 ? p.A2 : p.A), 0, (int)(p.K1 > 0 ? p.a2_bytes : p.a_bytes), 0x00020000);

  const int kc = (tid & 7) * 4;
  const int row_base = tid >> 3;

  uint32_t a_base[A_PER_T], a_mask[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int m = m0 + row_base + i * ROW_STEP;
    const bool ok = m < p.M;
    if (CONV) {
      const int mm = ok ? m : 0;
      const int ox = mm % p.OW;
      const int t = mm / p.OW;
      const int oy = t % p.OH;
      const int b = t / p.OH;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      // modular 32-bit arithmetic: the pixel offset may be "negative" until the tap offset is added
      a_base[i] = (uint32_t)(((b * p.H + iy0) * p.W + ix0) * p.Cin + kc) * 4u;
      uint32_t mask = 0;
      const int KH = p.K / (p.KW * p.Cin);
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx) {
          const int iy = iy0 + ky, ix = ix0 + kx;
          if (ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << (ky * p.KW + kx);
        }
      a_mask[i] = mask;
    } else {
      a_base[i] = (uint32_t)((int64_t)m * p.lda + kc) * 4u;
      a_mask[i] = ok ? 1u : 0u;
    }
  }
  uint32_t b_base[B_PER_T];
  bool b_ok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int n = n0 + row_base + i * ROW_STEP;
    b_ok[i] = n < p.N;
    b_base[i] = (uint32_t)((int64_t)n * p.ldb + kc) * 4u;
  }

  float4 a_reg[A_PER_T], b_reg[B_PER_T];
  // uniform k-tile state of the NEXT fetch
  int f_kt = 0, f_c0 = 0, f_tap = 0, f_ky = 0, f_kx = 0;

  const int ntaps = CONV ? p.K / p.Cin : 1;
  auto fetch = [&]() {
    const int k = f_kt * BK + kc;
    const bool kok = k < p.K;
    uint32_t koffb = (uint32_t)f_kt * (BK * 4u);
    if (CONV && p.Cin == 4) {
      // the 4-channel stems (image + padding channel): a 16-byte load is one whole tap, so the eight float4 columns of a
      // k-tile are eight taps and K = KH*KW*4 takes ceil(KH*KW / 8) k-tiles (two for 3x3); the tap is lane-dependent
      // but fixed per thread and k-tile.  a_base carries + kc * 4 for the channel offset of the wide layers: undone here.
      const int tap = f_kt * 8 + (tid & 7);
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const uint32_t tapoff = (uint32_t)((ky * p.W + kx) * 4 - kc) * 4u;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const bool ok = tap < ntaps && ((a_mask[i] >> tap) & 1u);
        a_reg[i] = buf_load4(rA, ok ? a_base[i] + tapoff : OOB);
      }
    } else if (CONV) {
      // k-tiles walk the taps of one 32-channel slice before moving to the next slice (the reduction order is
      // free): the same input lines are then re-read one (kx) or KW (ky) k-tiles later instead of Cin/32 and
      // KW*Cin/32 tiles later, i.e. while they are still in the XCD's L2
      const uint32_t tapoff = (uint32_t)((f_ky * p.W + f_kx) * p.Cin + f_c0) * 4u;
      koffb = (uint32_t)(f_tap * p.Cin + f_c0) * 4u;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const bool ok = (a_mask[i] >> f_tap) & 1u;
        a_reg[i] = buf_load4(rA, ok ? a_base[i] + tapoff : OOB);
      }
      ++f_tap;
      if (++f_kx == p.KW) { f_kx = 0; ++f_ky; }
      if (f_tap == ntaps) { f_tap = 0; f_kx = 0; f_ky = 0; f_c0 += BK; }
    } else {
      // wave-uniform choice of the source (columns K1.. live in the second matrix, same row offsets): a select of the
      // descriptor, not a branch -- the k-tile body stays one basic block, which the interleaving below needs
      const bool second = p.K1 > 0 && f_kt * BK >= p.K1;
      const __amdgpu_buffer_rsrc_t rS = second ? rA2 : rA;
      const uint32_t koff = (uint32_t)(f_kt * BK - (second ? p.K1 : 0)) * 4u;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) a_reg[i] = buf_load4(rS, (a_mask[i] && kok) ? a_base[i] + koff : OOB);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) b_reg[i] = buf_load4(rB, (b_ok[i] && kok) ? b_base[i] + koffb : OOB);
    ++f_kt;
  };

  // ROW_STEP is a multiple of 16, so the swizzle of a thread's rows is one constant
  static_assert(ROW_STEP % 16 == 0 && BM % 32 == 0 && BN % 32 == 0, "split-plane swizzle");
  const int kc_sw = (((kc >> 3) ^ split_swz(row_base)) << 3) + (kc & 7);
  const float pair_sa = (SPLIT == 2 && p.scale_a) ? p.scale_a[0] : 1.f, pair_s = (SPLIT == 2 && p.scale_b) ? p.scale_b[0] : 1.f;
  auto stage = [&](int buf) {
    if (SPLIT == 2) {
      unsigned short *a_dst = As16, *b_dst = Bs16;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        uint2 hi, lo;
        const float4 av = a_reg[i];
        pair4(make_float4(av.x * pair_sa, av.y * pair_sa, av.z * pair_sa, av.w * pair_sa), hi, lo);
        unsigned short *d = a_dst + (row_base + i * ROW_STEP) * SP + kc_sw;
        *reinterpret_cast<uint2 *>(d) = hi;
        *reinterpret_cast<uint2 *>(d + BM * SP) = lo;
      }
#pragma unroll
      for (int i = 0; i < B_PER_T; ++i) {
        uint2 hi, lo;
        const float4 w = b_reg[i];
        pair4(make_float4(w.x * pair_s, w.y * pair_s, w.z * pair_s, w.w * pair_s), hi, lo);
        unsigned short *d = b_dst + (row_base + i * ROW_STEP) * SP + kc_sw;
        *reinterpret_cast<uint2 *>(d) = hi;
        *reinterpret_cast<uint2 *>(d + BN * SP) = lo;
      }
      return;
    }
    if (SPLIT) {
      unsigned short *a_dst = As16 + buf * 3 * BM * SP, *b_dst = Bs16 + buf * 3 * BN * SP;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        uint2 q1, q2, q3;
        split4(a_reg[i], q1, q2, q3);
        unsigned short *d = a_dst + (row_base + i * ROW_STEP) * SP + kc_sw;
        *reinterpret_cast<uint2 *>(d) = q1;
        *reinterpret_cast<uint2 *>(d + BM * SP) = q2;
        *reinterpret_cast<uint2 *>(d + 2 * BM * SP) = q3;
      }
#pragma unroll
      for (int i = 0; i < B_PER_T; ++i) {
        uint2 q1, q2, q3;
        split4(b_reg[i], q1, q2, q3);
        unsigned short *d = b_dst + (row_base + i * ROW_STEP) * SP + kc_sw;
        *reinterpret_cast<uint2 *>(d) = q1;
        *reinterpret_cast<uint2 *>(d + BN * SP) = q2;
        *reinterpret_cast<uint2 *>(d + 2 * BN * SP) = q3;
      }
      return;
    }
    float *a_dst = As + buf * BM * LDS_PITCH;
    float *b_dst = Bs + buf * BN * LDS_PITCH;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
      *reinterpret_cast<float4 *>(a_dst + (row_base + i * ROW_STEP) * LDS_PITCH + kc) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      *reinterpret_cast<float4 *>(b_dst + (row_base + i * ROW_STEP) * LDS_PITCH + kc) = b_reg[i];
  };

  floatx16 acc[TM][TN];   // first written by the MFMAs of k-tile 0, which take C = 0 as an inline constant: no TM*TN*16 v_mov_b32

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  const int nkt = (p.K + BK - 1) / BK;

  fetch();
  stage(0);
  if (nkt > 1) fetch();
  __syncthreads();

#ifdef IDIFF_IGEMM_PHASES   // diagnostic build only (scripts/igemm_phases.py): shader-clock ticks per phase of a k-tile
  uint32_t ph[6] = {0, 0, 0, 0, 0, 0};
  uint64_t ph_last = __builtin_amdgcn_s_memtime();
  const uint64_t ph_first = ph_last;
#define IDIFF_PH(k) { __builtin_amdgcn_sched_barrier(0); const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[k] += (uint32_t)(t_ - ph_last); ph_last = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define IDIFF_PH(k)
#endif
  auto ktile = [&](int kt, auto first) {
    const int buf = DBUF ? (kt & 1) : 0;
    if (DBUF && !SPLIT) {
      if (kt + 1 < nkt) stage(buf ^ 1);   // tile kt+1: loaded one iteration ago
      if (kt + 2 < nkt) fetch();          // tile kt+2: lands while this tile's 64 MFMAs run
    }
    if (SPLIT == 2) {
      const int c0 = (lane >> 5) ^ split_swz(frag_row);
      const unsigned short *a16 = As16 + (wm0 + frag_row) * SP, *b16 = Bs16 + (wn0 + frag_row) * SP;
      halfx8 af[BK / 16][TM][2], bf[BK / 16][TN][2];
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) af[s][i][q] = *reinterpret_cast<const halfx8 *>(a16 + (q * BM + i * 32) * SP + ((c0 ^ (2 * s)) << 3));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q) bf[s][j][q] = *reinterpret_cast<const halfx8 *>(b16 + (q * BN + j * 32) * SP + ((c0 ^ (2 * s)) << 3));
      }
      constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0};     // lo hi, hi lo, hi hi
#pragma unroll
      for (int s = 0; s < BK / 16; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (decltype(first)::value && s == 0 && t == 0) {
                const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s][i][QA[t]], bf[s][j][QB[t]], zero, 0, 0, 0);
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s][i][QA[t]], bf[s][j][QB[t]], acc[i][j], 0, 0, 0);
              }
            }
    } else if (SPLIT) {
      // lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8 h + e] and B[k = 8 h + e][column r], e = 0 .. 7
      // chunk (lane >> 5) + 2 s of row frag_row (+ multiples of 32: same swizzle): c0 for s = 0, c0 ^ 2 for s = 1
      const int c0 = (lane >> 5) ^ split_swz(frag_row);
      const unsigned short *a16 = As16 + buf * 3 * BM * SP + (wm0 + frag_row) * SP;
      const unsigned short *b16 = Bs16 + buf * 3 * BN * SP + (wn0 + frag_row) * SP;
      bf16x8 af[BK / 16][TM][3], bf[BK / 16][TN][3];
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) af[s][i][q] = *reinterpret_cast<const bf16x8 *>(a16 + (q * BM + i * 32) * SP + ((c0 ^ (2 * s)) << 3));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) bf[s][j][q] = *reinterpret_cast<const bf16x8 *>(b16 + (q * BN + j * 32) * SP + ((c0 ^ (2 * s)) << 3));
      }
      // the six partial products, smallest first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int s = 0; s < BK / 16; ++s)
#pragma unroll
#ifdef IDIFF_IGEMM_DIAG_ONE_PRODUCT   // timing-only build: one of the six partial products (and a third of the fragment reads)
        for (int t = 0; t < 1; ++t)
#else
        for (int t = 0; t < 6; ++t)
#endif
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (decltype(first)::value && s == 0 && t == 0) {
                const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][PA[t]], bf[s][j][PB[t]], zero, 0, 0, 0);
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][PA[t]], bf[s][j][PB[t]], acc[i][j], 0, 0, 0);
              }
            }
    }
    const float *a_src = As + buf * BM * LDS_PITCH + (wm0 + frag_row) * LDS_PITCH + frag_k;
    const float *b_src = Bs + buf * BN * LDS_PITCH + (wn0 + frag_row) * LDS_PITCH + frag_k;
#pragma unroll
    for (int g = 0; g < (SPLIT ? 0 : BK / 8); ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a_src + i * 32 * LDS_PITCH + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b_src + j * 32 * LDS_PITCH + g * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (decltype(first)::value && g == 0) {
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, zero, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    IDIFF_PH(0)
    __syncthreads();
    IDIFF_PH(1)
    if (!DBUF) {
      // single LDS buffer (half the LDS -> three workgroups per CU): restage between two barriers; the other
      // resident workgroups keep the matrix pipe busy meanwhile
      if (kt + 1 < nkt) stage(0);
      IDIFF_PH(2)
#ifdef IDIFF_IGEMM_DIAG_NO_FETCH   // timing-only build: operands fetched for the first two k-tiles only
      if (kt + 2 < nkt) ++f_kt;
#else
      if (kt + 2 < nkt) fetch();
#endif
      IDIFF_PH(3)
      __syncthreads();
      IDIFF_PH(4)
    }
  };
#ifdef IDIFF_IGEMM_PHASES
  const uint64_t ph_loop0 = __builtin_amdgcn_s_memtime();
  ph_last = ph_loop0;
#endif
  ktile(0, std::true_type());
  for (int kt = 1; kt < nkt; ++kt) ktile(kt, std::false_type());
#ifdef IDIFF_IGEMM_PHASES
  const uint64_t ph_loop1 = __builtin_amdgcn_s_memtime();
#endif

  if (SPLIT == 2) {
    const float inv = (p.scale_a ? p.scale_a[1] : 1.f) * (p.scale_b ? p.scale_b[1] : 1.f);    // powers of two: exact
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  float *Cb = p.C + (int64_t)batch * p.strideC;
  const idiff_epilogue &ep = p.ep;
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
  // optional per-tile column statistics (sum, sum of squares in fp64) of the values being stored: the GroupNorm that
  // consumes this tensor then needs no pass of its own over HBM (idiff_epilogue.colstats)
#ifdef IDIFF_IGEMM_PHASES
  const bool want_stats = false;                   // epilogue.colstats carries the stamp buffer in this build
#else
  const bool want_stats = p.has_ep && ep.colstats != nullptr;
#endif
  double *red = reinterpret_cast<double *>(lds);   // [WARPS_M][BN][2]; operand staging is finished
  if (p.vec_ep && p.buf_ep) {
    // The form every contraction of the score networks takes.  As the vector form below (each wave turns its 32-row
    // blocks through a private LDS patch and finishes runs of four columns with 16-byte accesses), with the address
    // and predicate arithmetic taken out of the per-row work -- at K = 256 the epilogue's ~40 VALU instructions per row
    // pass were a fifth of a wave's time next to 512 MFMAs: C and the residual are addressed through buffer descriptors
    // (one 32-bit offset per thread + a wave-uniform row step; rows >= M fall outside the descriptor and columns >= N
    // start from an out-of-range offset, so nothing is predicated), bias + per-group bias and the two scales are folded
    // once per 32-row block when a block lies inside one row group (rows_per_group % 32 == 0).
    constexpr int C4 = WTN / 4;
    constexpr int RPP = 64 / C4;
    constexpr int PASSES = 32 / RPP;
    constexpr uint32_t OOB_BASE = 0xF0000000u;       // launch_pipe: extents < 0xE0000000, BM rows < 0x0FFFFFF0 bytes
    float *tr = lds + (tid >> 6) * (32 * WTN);
    const int c4 = lane % C4, r0 = lane / C4;
    const int n = n0 + wn0 + c4 * 4;
    const bool n_ok = n < p.N;
    const bool has_ep = p.has_ep != 0;
    const bool has_res = has_ep && ep.residual != nullptr;
    const bool has_rb = has_ep && ep.rowbias != nullptr, has_rs = has_ep && ep.rowscale != nullptr;
    const bool scaled = has_ep && (ep.out_scale != 1.f || has_rs);
    const int act = has_ep ? ep.act : (int)IDIFF_ACT_NONE;
    const bool block_groups = ep.rows_per_group % 32 == 0;   // a 32-row block never straddles a row group
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void *)Cb, 0, (int)p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)ep.residual, 0, (int)p.res_bytes, 0x00020000);
    const int mrow0 = m0 + wm0 + r0;
    const uint32_t ldc_b = (uint32_t)p.ldc * 4u, ldr_b = (uint32_t)ep.ld_residual * 4u;
    const uint32_t c_off = n_ok ? (uint32_t)mrow0 * ldc_b + (uint32_t)n * 4u : OOB_BASE;
    const uint32_t r_off = n_ok ? (uint32_t)mrow0 * ldr_b + (uint32_t)n * 4u : OOB_BASE;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok && has_ep && ep.bias) bias4 = *reinterpret_cast<const float4 *>(ep.bias + n);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // the block's residual rows are requested before its accumulators go through the LDS patch: a read issued at its
      // point of use is a dependent HBM round trip per row pass (the residual-fed K = 256 contractions ran at 105
      // TFLOP/s against 123 for the same shape without one)
      float4 resv[PASSES];
      if (has_res) {
#pragma unroll
        for (int t = 0; t < PASSES; ++t)
          resv[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)(r_off + (uint32_t)(i * 32 + RPP * t) * ldr_b), 0, 0));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + row_l) * WTN + j * 32 + col_l] = acc[i][j][r];
      float4 badd = bias4;
      float sc = has_ep ? ep.out_scale : 1.f;
      const int mblk = m0 + wm0 + i * 32;             // wave-uniform
      if (block_groups && mblk < p.M && (has_rb || has_rs)) {
        const int grp = mblk / ep.rows_per_group;
        if (has_rb && n_ok) {
          const float4 rb = *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)grp * ep.ld_rowbias + n);
          badd = make_float4(bias4.x + rb.x, bias4.y + rb.y, bias4.z + rb.z, bias4.w + rb.w);
        }
        if (has_rs) sc *= ep.rowscale[grp];
      }
#pragma unroll
      for (int t = 0; t < PASSES; ++t) {
        const int row = r0 + RPP * t;
        const float4 a4 = *reinterpret_cast<const float4 *>(tr + row * WTN + c4 * 4);
        const uint32_t step = (uint32_t)(i * 32 + RPP * t);
        float v[4] = {a4.x + badd.x, a4.y + badd.y, a4.z + badd.z, a4.w + badd.w};
        float rsv = 1.f;
        if (!block_groups && (has_rb || has_rs)) {
          const int m = mrow0 + (int)step;
          if (m < p.M && n_ok) {
            const int grp = m / ep.rows_per_group;
            if (has_rb) {
              const float4 rb = *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)grp * ep.ld_rowbias + n);
              v[0] += rb.x; v[1] += rb.y; v[2] += rb.z; v[3] += rb.w;
            }
            if (has_rs) rsv = ep.rowscale[grp];
          }
        }
        if (act != IDIFF_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = idiff::act_apply(v[e], act);
        }
        if (has_res) {
          const float4 rs = resv[t];
          v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
        }
        if (scaled) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= sc;
          if (!block_groups && has_rs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= rsv;
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, make_float4(v[0], v[1], v[2], v[3])), rC,
                                               (int)(c_off + step * ldc_b), 0, 0);
        if (want_stats) {      // whole tiles only (idiff_gemm_colstats_split): every row is a row of C
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += (double)v[e]; s2[e] += (double)v[e] * (double)v[e]; }
        }
      }
    }
    if (want_stats) {
#pragma unroll
      for (int off = C4; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
      __syncthreads();                               // every wave is done with its patch
      if (lane < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int slot = ((wave / WARPS_N) * BN + wn0 + c4 * 4 + e) * 2;
          red[slot] = s1[e]; red[slot + 1] = s2[e];
        }
      }
    }
  } else if (p.vec_ep) {
    // The same with 64-bit addresses, for C or residual extents beyond one buffer descriptor.
    // One dword per lane per store makes short-K contractions store-issue bound (a 128x128 tile is 256 wave-level
    // stores, ~100 cycles each through the CU's one address unit: 0.8 of the 1.24 ms of the K = 128 shortcut GEMMs).
    // Each wave turns its 32-row blocks through a private 8 KB LDS patch instead and finishes runs of four columns:
    // 16-byte loads of bias / residual, 16-byte stores, a quarter of the store instructions.
    constexpr int C4 = WTN / 4;                      // float4 per row of the wave tile
    constexpr int RPP = 64 / C4;                     // rows per pass of 64 lanes
    constexpr int PASSES = 32 / RPP;
    float *tr = lds + (tid >> 6) * (32 * WTN);
    const int c4 = lane % C4, r0 = lane / C4;
    const int n = n0 + wn0 + c4 * 4;
    const bool n_ok = n < p.N;                       // N % 4 == 0: a run of four is inside or outside as a whole
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok && p.has_ep && ep.bias) bias4 = *reinterpret_cast<const float4 *>(ep.bias + n);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + row_l) * WTN + j * 32 + col_l] = acc[i][j][r];
#pragma unroll
      for (int t = 0; t < PASSES; ++t) {
        const int row = r0 + RPP * t;
        const float4 a4 = *reinterpret_cast<const float4 *>(tr + row * WTN + c4 * 4);
        const int m = m0 + wm0 + i * 32 + row;
        if (m >= p.M || !n_ok) continue;
        float v[4] = {a4.x + bias4.x, a4.y + bias4.y, a4.z + bias4.z, a4.w + bias4.w};
        if (p.has_ep) {
          const int grp = (ep.rowbias || ep.rowscale) ? m / ep.rows_per_group : 0;
          if (ep.rowbias) {
            const float4 rb = *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)grp * ep.ld_rowbias + n);
            v[0] += rb.x; v[1] += rb.y; v[2] += rb.z; v[3] += rb.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = idiff::act_apply(v[e], ep.act);
          if (ep.residual) {
            const float4 rs = *reinterpret_cast<const float4 *>(ep.residual + (int64_t)m * ep.ld_residual + n);
            v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
          }
          const float rsc = ep.rowscale ? ep.rowscale[grp] : 1.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= ep.out_scale; if (ep.rowscale) v[e] *= rsc; }
        }
        *reinterpret_cast<float4 *>(Cb + (int64_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += (double)v[e]; s2[e] += (double)v[e] * (double)v[e]; }
        }
      }
    }
    if (want_stats) {
      // lanes sharing c4 hold the same four columns
#pragma unroll
      for (int off = C4; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
      __syncthreads();                               // every wave is done with its patch
      if (lane < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int slot = ((wave / WARPS_N) * BN + wn0 + c4 * 4 + e) * 2;
          red[slot] = s1[e]; red[slot + 1] = s2[e];
        }
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + col_l;
    const bool n_ok = n < p.N;
    const float bias = (n_ok && p.has_ep && ep.bias) ? ep.bias[n] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
        if (m >= p.M || !n_ok) continue;
        float v = acc[i][j][r] + bias;
        if (p.has_ep) {
          if (ep.rowbias) v += ep.rowbias[(int64_t)(m / ep.rows_per_group) * ep.ld_rowbias + n];
          v = idiff::act_apply(v, ep.act);
          if (ep.residual) v += ep.residual[(int64_t)m * ep.ld_residual + n];
          v *= ep.out_scale;
          if (ep.rowscale) v *= ep.rowscale[m / ep.rows_per_group];
        }
        Cb[(int64_t)m * p.ldc + n] = v;
        if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
      }
    }
    if (want_stats) {
      // lanes l and l+32 hold the same column: fold, then one slot per (wave row, column)
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32) {
        const int slot = ((wave / WARPS_N) * BN + wn0 + j * 32 + col_l) * 2;
        red[slot] = s1; red[slot + 1] = s2;
      }
    }
  }
  }
  if (want_stats) {
    __syncthreads();
    for (int c = tid; c < BN; c += T) {
      const int n = n0 + c;
      if (n >= p.N) continue;
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < WARPS_M; ++w) { a += red[(w * BN + c) * 2]; b += red[(w * BN + c) * 2 + 1]; }
      double *dst = ep.colstats + ((int64_t)tile_m * p.N + n) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
#ifdef IDIFF_IGEMM_PHASES
  if (p.has_ep && ep.colstats && lane == 0) {
    // a buffer nothing else reads: [workgroup][wave][8]
    const int64_t wg = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    uint32_t *st = reinterpret_cast<uint32_t *>(ep.colstats) + (wg * (T / 64) + (tid >> 6)) * 8;
    for (int k = 0; k < 5; ++k) st[k] = ph[k];
    st[5] = (uint32_t)(ph_loop0 - ph_entry);                       // prologue: kernel entry to the first k-tile
    st[6] = (uint32_t)(__builtin_amdgcn_s_memtime() - ph_loop1);   // epilogue
    st[7] = (uint32_t)nkt;
  }
#endif
}
#undef IDIFF_PH

template <int BM, int BN, int WARPS_M, int WARPS_N, bool CONV, bool DBUF = true, int SPLIT = 0>
int launch_pipe(IgemmParams &p, int batch, hipStream_t st) {
  p.tiles_m = idiff::ceil_div(p.M, BM);
  p.tiles_n = idiff::ceil_div(p.N, BN);
  constexpr size_t stage_bytes = SPLIT ? (size_t)(DBUF ? 2 : 1) * (BM + BN) * (SPLIT == 2 ? 2 : 3) * SP * sizeof(unsigned short)
                                       : (size_t)(DBUF ? 2 : 1) * (BM + BN) * LDS_PITCH * sizeof(float);
  // (the epilogue turns the accumulators through one 32 x WTN fp32 patch per wave in the same memory)
  constexpr size_t patch_bytes = (size_t)WARPS_M * WARPS_N * 32 * (BN / WARPS_N) * sizeof(float);
  constexpr size_t lds_bytes = stage_bytes > patch_bytes ? stage_bytes : patch_bytes;
  auto kern = igemm_pipe_kernel<BM, BN, WARPS_M, WARPS_N, CONV, DBUF, SPLIT>;
  if (lds_bytes > 64 * 1024) {
    static idiff::AttrGuard guard;                 // one per template instantiation, one bit per device
    const void *fn = reinterpret_cast<const void *>(kern);
    if (int rc = idiff::set_dynamic_lds_once(guard, &fn, 1, (int)lds_bytes, "igemm")) return rc;
  }
  {
    const idiff_epilogue &e = p.ep;
    const bool al = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.strideC % 4 == 0) && aligned16(p.C) &&
                    (!p.has_ep || ((!e.bias || aligned16(e.bias)) && (!e.rowbias || (aligned16(e.rowbias) && e.ld_rowbias % 4 == 0)) &&
                                   (!e.residual || (aligned16(e.residual) && e.ld_residual % 4 == 0))));
    p.vec_ep = al && !idiff::option(idiff::OPT_SCALAR_EPILOGUE);
    // 32-bit addressing of C and the residual (igemm_pipe_kernel's first epilogue form): extents below 0xE0000000 so that
    // an out-of-range start stays out of range after the row steps of one tile are added
    const int64_t c_bytes = ((int64_t)(p.M - 1) * p.ldc + p.N) * 4;
    const int64_t res_bytes = (p.has_ep && e.residual) ? ((int64_t)(p.M - 1) * e.ld_residual + p.N) * 4 : 0;
    const int64_t pitch = p.ldc > e.ld_residual ? p.ldc : e.ld_residual;
    p.buf_ep = p.vec_ep && c_bytes < 0xE0000000ll && res_bytes < 0xE0000000ll && (int64_t)BM * pitch * 4 < 0x0FFFFFF0ll;
    p.c_bytes = p.buf_ep ? (uint32_t)c_bytes : 0; p.res_bytes = p.buf_ep ? (uint32_t)res_bytes : 0;
  }
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WARPS_M * WARPS_N * 64), lds_bytes, st, p);
  return idiff::launch_status(CONV ? "igemm_pipe_conv" : "igemm_pipe_linear");
}

// Rows per workgroup tile the pipelined dispatcher picks for (M, N, batch); keep in step with dispatch_pipe.
int pipe_tile_rows(int M, int N, int batch) {
  const int64_t wg_big = (int64_t)idiff::ceil_div(M, 128) * idiff::ceil_div(N, 128) * batch;
  if (N > 64 && wg_big >= 256) return 128;
  const int64_t wg_mid = (int64_t)idiff::ceil_div(M, 128) * idiff::ceil_div(N, 64) * batch;
  if (wg_mid >= 256 || M >= 4096) return 128;
  return 64;
}

template <bool CONV>
int dispatch_pipe(IgemmParams &p, int batch, hipStream_t st) {
  const int64_t wg_big = (int64_t)idiff::ceil_div(p.M, 128) * idiff::ceil_div(p.N, 128) * batch;
  if (!idiff::option(idiff::OPT_NO_SPLIT)) {
    // split-precision products on the bf16 matrix cores (see split4): single LDS buffer (60 KB at 128 x 128: two workgroups
    // per CU), the same tile choice as below.  (A double-buffered form with the splitting of tile t + 1 interleaved between
    // the matrix instructions of tile t by sched_group_barrier was built and measured: 118-128 TFLOP/s against 140-180 for
    // this one -- at 32 k per tile a 128 x 128 workgroup asks the L2 for 32 KB per 1600 matrix-pipe cycles, which is what
    // bounds it, not the instruction mix.)
    if (p.N > 64 && wg_big >= 256) return launch_pipe<128, 128, 2, 2, CONV, false, 1>(p, batch, st);
    if (p.N <= 32 && p.M >= 4096) return launch_pipe<128, 32, 4, 1, CONV, false, 1>(p, batch, st);
    const int64_t wg_mid_s = (int64_t)idiff::ceil_div(p.M, 128) * idiff::ceil_div(p.N, 64) * batch;
    if (wg_mid_s >= 256 || p.M >= 4096) return launch_pipe<128, 64, 2, 2, CONV, false, 1>(p, batch, st);
    return launch_pipe<64, 64, 2, 2, CONV, false, 1>(p, batch, st);
  }
  // >= 4 workgroups per CU available: single LDS buffer, 128 registers, four resident workgroups per CU
  // (measured 135-142 TFLOP/s vs 124-135 for the double-buffered two-workgroup form)
  if (p.N > 64 && wg_big >= 1024 && !idiff::option(idiff::OPT_DBUF_ONLY)) return launch_pipe<128, 128, 2, 2, CONV, false>(p, batch, st);
  if (p.N > 64 && wg_big >= 256) return launch_pipe<128, 128, 2, 2, CONV>(p, batch, st);
  // narrow outputs (the 3-channel image conv at the end of the U-Nets): one 32-wide MFMA column instead of two
  if (p.N <= 32 && p.M >= 4096) return launch_pipe<128, 32, 4, 1, CONV>(p, batch, st);
  const int64_t wg_mid = (int64_t)idiff::ceil_div(p.M, 128) * idiff::ceil_div(p.N, 64) * batch;
  if (wg_mid >= 256 || p.M >= 4096) return launch_pipe<128, 64, 2, 2, CONV>(p, batch, st);
  return launch_pipe<64, 64, 2, 2, CONV>(p, batch, st);
}

template <bool CONV, bool VEC>
int dispatch(IgemmParams &p, int batch, hipStream_t st) {
  // Pick the largest tile that still yields >= ~2 workgroups per CU pair; small problems get small tiles.
  const int64_t wg_big = (int64_t)idiff::ceil_div(p.M, 128) * idiff::ceil_div(p.N, 128) * batch;
  if (p.N > 64 && wg_big >= 256) return launch_cfg<128, 128, 2, 2, CONV, VEC>(p, batch, st);
  const int64_t wg_mid = (int64_t)idiff::ceil_div(p.M, 128) * idiff::ceil_div(p.N, 64) * batch;
  if (wg_mid >= 256 || p.M >= 4096) return launch_cfg<128, 64, 2, 2, CONV, VEC>(p, batch, st);
  return launch_cfg<64, 64, 2, 2, CONV, VEC>(p, batch, st);
}

void fill_epilogue(IgemmParams &p, const idiff_epilogue *ep) {
  if (ep) {
    p.ep = *ep;
    p.has_ep = 1;
    if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1;
  } else {
    p.has_ep = 0;
    p.ep.bias = nullptr; p.ep.rowbias = nullptr; p.ep.residual = nullptr;
    p.ep.ld_rowbias = 0; p.ep.ld_residual = 0; p.ep.rows_per_group = 1; p.ep.act = 0; p.ep.out_scale = 1.f;
    p.ep.rowscale = nullptr;
  }
}

// The fast kernel addresses an operand through one 32-bit-offset buffer descriptor (< 4 GiB).  Larger problems
// are cut into row ranges on the host (rows are independent): this returns the epilogue of the range that
// starts at row m0, which must be a multiple of rows_per_group.
idiff_epilogue shift_epilogue(const idiff_epilogue &ep, int64_t m0) {
  idiff_epilogue e = ep;
  const int64_t g0 = m0 / (ep.rows_per_group > 0 ? ep.rows_per_group : 1);
  if (e.rowbias) e.rowbias += g0 * ep.ld_rowbias;
  if (e.residual) e.residual += m0 * ep.ld_residual;
  if (e.rowscale) e.rowscale += g0;
  return e;
}
constexpr int64_t BUF_LIMIT = 0xFFFFFFF0ll;

}  // namespace

IDIFF_API int idiff_gemm_f32(const float *A, int64_t lda, int64_t strideA, const float *Bt, int64_t ldb,
                             int64_t strideB, float *C, int64_t ldc, int64_t strideC, int M, int N, int K,
                             int batch, const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (M < 0 || N < 0 || K <= 0 || batch < 0) return fail("gemm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  if (M == 0 || N == 0 || batch == 0) return 0;
  if (!A || !Bt || !C) return fail("gemm: null pointer");
  if (lda < K || ldb < K || ldc < N) return fail("gemm: leading dimension smaller than the row length");
  if (batch > 65535) return fail("gemm: batch %d exceeds grid.y", batch);
  IgemmParams p = {};
  p.A = A; p.Bt = Bt; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.strideA = strideA; p.strideB = strideB; p.strideC = strideC; p.M = M; p.N = N; p.K = K;
  fill_epilogue(p, ep);
  const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (strideA % 4 == 0) && (strideB % 4 == 0) &&
                   aligned16(A) && aligned16(Bt);
  hipStream_t st = (hipStream_t)stream;
  const int64_t a_bytes = ((int64_t)(M - 1) * lda + K) * 4, b_bytes = ((int64_t)(N - 1) * ldb + K) * 4;
  if (vec && batch == 1 && a_bytes >= BUF_LIMIT && b_bytes < BUF_LIMIT && !idiff::option(idiff::OPT_NO_PIPE)) {
    if (ep && ep->colstats) return fail("gemm: colstats is not available for operands beyond 4 GiB");
    const int rpg = (ep && ep->rows_per_group > 0) ? ep->rows_per_group : 1;
    const int mid = (M / 2 / rpg) * rpg;
    if (mid > 0) {
      idiff_epilogue lo, hi;
      if (ep) { lo = *ep; hi = shift_epilogue(*ep, mid); }
      int rc = idiff_gemm_f32(A, lda, 0, Bt, ldb, 0, C, ldc, 0, mid, N, K, 1, ep ? &lo : nullptr, stream);
      if (rc) return rc;
      return idiff_gemm_f32(A + (int64_t)mid * lda, lda, 0, Bt, ldb, 0, C + (int64_t)mid * ldc, ldc, 0, M - mid, N, K, 1,
                            ep ? &hi : nullptr, stream);
    }
  }
  if (vec && a_bytes < BUF_LIMIT && b_bytes < BUF_LIMIT && !idiff::option(idiff::OPT_NO_PIPE)) {
    p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
    return dispatch_pipe<false>(p, batch, st);
  }
  if (ep && ep->colstats) return fail("gemm: colstats requested for a problem the pipelined kernel does not take "
                                      "(ask idiff_gemm_colstats_split first)");
  return vec ? dispatch<false, true>(p, batch, st) : dispatch<false, false>(p, batch, st);
}

namespace {
// {s, 1 / s} with s the power of two that brings max |Bt| into [2^11, 2^12) (1 for an all-zero matrix); one workgroup
__global__ void __launch_bounds__(256) pairs_scale_kernel(const float *__restrict__ bt, int64_t ldb, int N, int K, float *__restrict__ out) {
  __shared__ float red[256];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < (int64_t)N * K; i += 256) m = fmaxf(m, fabsf(bt[(i / K) * ldb + (i % K)]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mx = red[0];
    int e = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
    const int sh = mx > 0.f && mx < INFINITY ? 12 - e : 0;        // mx * 2^sh in [2^11, 2^12)
    out[0] = ldexpf(1.f, sh);
    out[1] = ldexpf(1.f, -sh);
  }
}
bool pairs_geometry_ok(int M, int N, int K, int batch) {
  if (M <= 0 || N <= 64 || K <= 0 || K % 4 || batch <= 0 || batch > 65535) return false;
  const int want = idiff::option_value(idiff::OPT_PAIRS_MIN_TILES);                  // tests: serve small problems too
  return (int64_t)idiff::ceil_div(M, 128) * idiff::ceil_div(N, 128) * batch >= (want > 0 ? want : 256);   // the 128 x 128 tiles fill the chip
}
}  // namespace

IDIFF_API int idiff_gemm_pairs_ok(int M, int N, int K, int batch) {
  if (idiff::option(idiff::OPT_NO_PIPE) || idiff::option(idiff::OPT_NO_SPLIT) || idiff::option(idiff::OPT_NO_PAIRS)) return 0;
  return pairs_geometry_ok(M, N, K, batch) ? 1 : 0;
}

IDIFF_API int idiff_gemm_pairs_scale_f32(const float *Bt, int64_t ldb, int N, int K, float *scale, void *stream) {
  using namespace idiff;
  if (!Bt || !scale) return fail("gemm_pairs_scale: null pointer");
  if (N <= 0 || K <= 0 || ldb < K) return fail("gemm_pairs_scale: bad shape N=%d K=%d ldb=%lld", N, K, (long long)ldb);
  hipLaunchKernelGGL(pairs_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, Bt, ldb, N, K, scale);
  return launch_status("gemm_pairs_scale");
}

IDIFF_API int idiff_gemm_pairs_f32(const float *A, int64_t lda, int64_t strideA, const float *Bt, int64_t ldb, int64_t strideB,
                                   const float *w_scale, int weight_is_a, const float *act_scale, float *C, int64_t ldc, int64_t strideC,
                                   int M, int N, int K, int batch, const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (M < 0 || N < 0 || K <= 0 || batch < 0) return fail("gemm_pairs: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  if (M == 0 || N == 0 || batch == 0) return 0;
  if (!A || !Bt || !C || !w_scale) return fail("gemm_pairs: null pointer");
  if (lda < K || ldb < K || ldc < N) return fail("gemm_pairs: leading dimension smaller than the row length");
  if (!pairs_geometry_ok(M, N, K, batch)) return fail("gemm_pairs: M=%d N=%d K=%d batch=%d not served (ask idiff_gemm_pairs_ok)", M, N, K, batch);
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (strideA % 4 == 0) && (strideB % 4 == 0) && aligned16(A) && aligned16(Bt);
  const int64_t a_bytes = ((int64_t)(M - 1) * lda + K) * 4, b_bytes = ((int64_t)(N - 1) * ldb + K) * 4;
  if (!vec || a_bytes >= BUF_LIMIT || b_bytes >= BUF_LIMIT)
    return fail("gemm_pairs: operands must be 16-byte aligned with row pitches and batch strides that are multiples of 4, one "
                "batch slice inside 4 GiB");
  if (ep && ep->colstats && batch != 1) return fail("gemm_pairs: colstats only for unbatched problems");
  IgemmParams p = {};
  p.A = A; p.Bt = Bt; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.strideA = strideA; p.strideB = strideB; p.strideC = strideC;
  p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.scale_a = weight_is_a ? w_scale : act_scale; p.scale_b = weight_is_a ? act_scale : w_scale;
  fill_epilogue(p, ep);
  return launch_pipe<128, 128, 2, 2, false, false, 2>(p, batch, (hipStream_t)stream);
}

IDIFF_API int idiff_gemm_pairs_2src_f32(const float *A1, const float *A2, int64_t lda, int K1, const float *act_scale, const float *Bt,
                                        int64_t ldb, const float *w_scale, float *C, int64_t ldc, int M, int N, int K,
                                        const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (M < 0 || N < 0 || K <= 0 || K1 <= 0 || K1 >= K) return fail("gemm_pairs_2src: bad sizes M=%d N=%d K=%d K1=%d", M, N, K, K1);
  if (M == 0 || N == 0) return 0;
  if (!A1 || !A2 || !Bt || !C || !w_scale) return fail("gemm_pairs_2src: null pointer");
  if (K1 % BK) return fail("gemm_pairs_2src: the split column K1 = %d must be a multiple of %d", K1, BK);
  if (lda < K1 || lda < K - K1 || ldb < K || ldc < N) return fail("gemm_pairs_2src: leading dimension smaller than the row length");
  if (!pairs_geometry_ok(M, N, K, 1)) return fail("gemm_pairs_2src: M=%d N=%d K=%d not served (ask idiff_gemm_pairs_ok)", M, N, K);
  const int64_t a1_bytes = ((int64_t)(M - 1) * lda + K1) * 4, a2_bytes = ((int64_t)(M - 1) * lda + (K - K1)) * 4;
  const int64_t b_bytes = ((int64_t)(N - 1) * ldb + K) * 4;
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A1) && aligned16(A2) && aligned16(Bt);
  if (!vec || a1_bytes >= BUF_LIMIT || a2_bytes >= BUF_LIMIT || b_bytes >= BUF_LIMIT)
    return fail("gemm_pairs_2src: operands must be 16-byte aligned with row pitches that are multiples of 4 and lie inside 4 GiB");
  IgemmParams p = {};
  p.A = A1; p.A2 = A2; p.K1 = K1; p.Bt = Bt; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.a_bytes = (uint32_t)a1_bytes; p.a2_bytes = (uint32_t)a2_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.scale_a = act_scale; p.scale_b = w_scale;
  fill_epilogue(p, ep);
  return launch_pipe<128, 128, 2, 2, false, false, 2>(p, 1, (hipStream_t)stream);
}

namespace {
// {s, 1 / s} with s the power of two nearest to 1 / rms of the tensor(s) whose per-tile column sums (sum, sum of squares; fp64)
// the producing contractions wrote: out = [s, 1 / s, <fp64 accumulator>, <block counter>] (8 floats, zeroed by the launcher)
__global__ void __launch_bounds__(256) pairs_act_scale_kernel(const double *__restrict__ ws1, int64_t n1, const double *__restrict__ ws2,
                                                              int64_t n2, double count, float *__restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += stride) acc += ws1[2 * i + 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) acc += ws2[2 * i + 1];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double *total = reinterpret_cast<double *>(out + 2);
    unsigned int *done = reinterpret_cast<unsigned int *>(out + 4);
    atomicAdd(total, red[0]);
    __threadfence();
    if (atomicAdd(done, 1u) == gridDim.x - 1) {
      __threadfence();
      const double ms = *reinterpret_cast<volatile double *>(total) / count;      // mean square
      int sh = 0;
      if (ms > 0.0 && ms < 1e300) {
        int e = 0;
        (void)frexp(sqrt(ms), &e);                  // rms = f * 2^e, f in [0.5, 1): rms * 2^-e in [0.5, 1), * 2^(1-e) in [1, 2)
        sh = -e + ((sqrt(ms) * ldexp(1.0, -e) < 0.70710678118654752) ? 1 : 0);   // rms * 2^sh in [0.707, 1.414)
      }
      out[0] = ldexpf(1.f, sh);
      out[1] = ldexpf(1.f, -sh);
    }
  }
}
}  // namespace

IDIFF_API int idiff_pairs_act_scale_f32(const double *ws1, int nsplit1, int C1, const double *ws2, int nsplit2, int C2, int B, int HW,
                                        float *out, void *stream) {
  using namespace idiff;
  if (!ws1 || !out || nsplit1 <= 0 || C1 <= 0 || B <= 0 || HW <= 0 || (ws2 && (nsplit2 <= 0 || C2 <= 0)))
    return fail("pairs_act_scale: bad arguments");
  const int64_t n1 = (int64_t)B * nsplit1 * C1, n2 = ws2 ? (int64_t)B * nsplit2 * C2 : 0;
  hipError_t e = hipMemsetAsync(out, 0, 32, (hipStream_t)stream);
  if (e != hipSuccess) return fail("pairs_act_scale: hipMemsetAsync: %s", hipGetErrorString(e));
  const int blocks = (int)std::min<int64_t>(512, (n1 + n2 + 255) / 256);
  hipLaunchKernelGGL(pairs_act_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws1, n1, ws2, n2,
                     (double)B * HW * (C1 + (ws2 ? C2 : 0)), out);
  return launch_status("pairs_act_scale");
}

IDIFF_API int idiff_gemm_2src_f32(const float *A1, const float *A2, int64_t lda, int K1, const float *Bt, int64_t ldb,
                                  float *C, int64_t ldc, int M, int N, int K, const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (M < 0 || N < 0 || K <= 0 || K1 <= 0 || K1 >= K) return fail("gemm_2src: bad sizes M=%d N=%d K=%d K1=%d", M, N, K, K1);
  if (M == 0 || N == 0) return 0;
  if (!A1 || !A2 || !Bt || !C) return fail("gemm_2src: null pointer");
  if (K1 % BK) return fail("gemm_2src: the split column K1 = %d must be a multiple of %d", K1, BK);
  if (lda < K1 || lda < K - K1 || ldb < K || ldc < N) return fail("gemm_2src: leading dimension smaller than the row length");
  const int64_t a1_bytes = ((int64_t)(M - 1) * lda + K1) * 4, a2_bytes = ((int64_t)(M - 1) * lda + (K - K1)) * 4;
  const int64_t b_bytes = ((int64_t)(N - 1) * ldb + K) * 4;
  const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A1) && aligned16(A2) && aligned16(Bt);
  if (vec && (a1_bytes >= BUF_LIMIT || a2_bytes >= BUF_LIMIT) && b_bytes < BUF_LIMIT && M > 1 && !idiff::option(idiff::OPT_NO_PIPE)) {
    // rows are independent: cut them until each half fits one buffer descriptor (as idiff_gemm_f32 does)
    if (ep && ep->colstats) return fail("gemm_2src: colstats is not available for operands beyond 4 GiB");
    const int rpg = (ep && ep->rows_per_group > 0) ? ep->rows_per_group : 1;
    int mid = (M / 2 / rpg) * rpg;
    if (mid <= 0) mid = M / 2;
    if (mid % rpg) return fail("gemm_2src: cannot split %d rows inside an epilogue row group of %d", M, rpg);
    idiff_epilogue lo, hi;
    if (ep) { lo = *ep; hi = shift_epilogue(*ep, mid); }
    int rc = idiff_gemm_2src_f32(A1, A2, lda, K1, Bt, ldb, C, ldc, mid, N, K, ep ? &lo : nullptr, stream);
    if (rc) return rc;
    return idiff_gemm_2src_f32(A1 + (int64_t)mid * lda, A2 + (int64_t)mid * lda, lda, K1, Bt, ldb, C + (int64_t)mid * ldc, ldc,
                               M - mid, N, K, ep ? &hi : nullptr, stream);
  }
  if (!vec || a1_bytes >= BUF_LIMIT || a2_bytes >= BUF_LIMIT || b_bytes >= BUF_LIMIT || idiff::option(idiff::OPT_NO_PIPE))
    return fail("gemm_2src: operands must be 16-byte aligned with K %% 4 == 0 and lda %% 4 == 0 (use two idiff_gemm_f32 calls)");
  IgemmParams p = {};
  p.A = A1; p.A2 = A2; p.K1 = K1; p.Bt = Bt; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.a_bytes = (uint32_t)a1_bytes; p.a2_bytes = (uint32_t)a2_bytes; p.b_bytes = (uint32_t)b_bytes;
  fill_epilogue(p, ep);
  return dispatch_pipe<false>(p, 1, (hipStream_t)stream);
}

IDIFF_API int idiff_conv2d_nhwc_f32(const float *x, const float *wt, float *out, int B, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad_lo, int pad_hi,
                                    const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_lo < 0 || pad_hi < 0)
    return fail("conv2d: bad geometry");
  if (Cin % 4 != 0) return fail("conv2d: Cin must be a multiple of 4 (pad the channels), got %d", Cin);
  if (B == 0) return 0;
  if (!x || !wt || !out) return fail("conv2d: null pointer");
  if (!aligned16(x) || !aligned16(wt)) return fail("conv2d: x and wt must be 16-byte aligned");
  const int OH = (H + pad_lo + pad_hi - KH) / stride + 1, OW = (W + pad_lo + pad_hi - KW) / stride + 1;
  const int pad = pad_lo;
  if (H + pad_lo + pad_hi < KH || W + pad_lo + pad_hi < KW || OH <= 0 || OW <= 0) return fail("conv2d: empty output");
  const int64_t M64 = (int64_t)B * OH * OW;
  if (M64 > 0x7fffffff) return fail("conv2d: B*OH*OW overflows int32");
  // the 128 -> 3 image heads: ten times fewer multiplications on the vector ALUs than padded to an MFMA column
  if (conv3x3_narrow_ok(B, H, W, Cin, Cout, KH, KW, stride, pad_lo, pad_hi, ep))
    return conv3x3_narrow(x, wt, out, B, H, W, Cin, Cout, ep, (hipStream_t)stream);
  IgemmParams p = {};
  p.A = x; p.Bt = wt; p.C = out;
  p.M = (int)M64; p.N = Cout; p.K = KH * KW * Cin;
  p.lda = 0; p.ldb = p.K; p.ldc = Cout;
  p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KW = KW; p.stride = stride; p.pad = pad;
  fill_epilogue(p, ep);
  const int64_t a_bytes = (int64_t)B * H * W * Cin * 4, b_bytes = (int64_t)Cout * p.K * 4;
  const bool fast_ok = (Cin % BK == 0 || Cin == 4) && KH * KW <= 32 && b_bytes < BUF_LIMIT && !idiff::option(idiff::OPT_NO_PIPE);
  if (ep && ep->colstats && !(fast_ok && a_bytes < BUF_LIMIT))
    return fail("conv2d: colstats requested for a problem the pipelined kernel does not take "
                "(ask idiff_conv2d_colstats_split first)");
  if (fast_ok && a_bytes >= BUF_LIMIT && B > 1) {
    // split the batch (images are independent) so that each half fits one buffer descriptor
    const int rpg = (ep && ep->rows_per_group > 0) ? ep->rows_per_group : 1;
    const int b_lo = B / 2;
    const int64_t m_lo = (int64_t)b_lo * OH * OW;
    if (m_lo % rpg == 0) {
      idiff_epilogue lo, hi;
      if (ep) { lo = *ep; hi = shift_epilogue(*ep, m_lo); }
      int rc = idiff_conv2d_nhwc_f32(x, wt, out, b_lo, H, W, Cin, Cout, KH, KW, stride, pad_lo, pad_hi, ep ? &lo : nullptr, stream);
      if (rc) return rc;
      return idiff_conv2d_nhwc_f32(x + (int64_t)b_lo * H * W * Cin, wt, out + m_lo * Cout, B - b_lo, H, W, Cin, Cout, KH, KW,
                                   stride, pad_lo, pad_hi, ep ? &hi : nullptr, stream);
    }
  }
  if (fast_ok && a_bytes < BUF_LIMIT) {
    p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
    return dispatch_pipe<true>(p, 1, (hipStream_t)stream);
  }
  return dispatch<true, true>(p, 1, (hipStream_t)stream);
}


// Number of workgroup row-tiles per sample when `rows_per_sample` consecutive output rows form one sample, i.e. the
// `nsplit` of the [samples, nsplit, N, 2] fp64 layout idiff_epilogue.colstats is written in -- or 0 when the fused
// statistics are not available for this problem (then the consumer runs idiff_groupnorm_stats_f32 as usual).
IDIFF_API int idiff_gemm_colstats_split(int M, int N, int K, int64_t lda, int64_t ldb, int rows_per_sample) {
  if (M <= 0 || N <= 0 || K <= 0 || rows_per_sample <= 0 || M % rows_per_sample) return 0;
  if (idiff::option(idiff::OPT_NO_PIPE) || idiff::option(idiff::OPT_NO_COLSTATS)) return 0;
  if (K % 4 || lda % 4 || ldb % 4) return 0;
  if (((int64_t)(M - 1) * lda + K) * 4 >= BUF_LIMIT || ((int64_t)(N - 1) * ldb + K) * 4 >= BUF_LIMIT) return 0;
  const int bm = pipe_tile_rows(M, N, 1);
  return rows_per_sample % bm == 0 ? rows_per_sample / bm : 0;
}

IDIFF_API int idiff_conv2d_colstats_split(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_lo,
                                          int pad_hi) {
  if (B <= 0 || (Cin % BK && Cin != 4) || KH * KW > 32 || idiff::option(idiff::OPT_NO_PIPE) || idiff::option(idiff::OPT_NO_COLSTATS)) return 0;
  const int OH = (H + pad_lo + pad_hi - KH) / stride + 1, OW = (W + pad_lo + pad_hi - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return 0;
  if ((int64_t)B * H * W * Cin * 4 >= BUF_LIMIT || (int64_t)Cout * KH * KW * Cin * 4 >= BUF_LIMIT) return 0;
  const int64_t M = (int64_t)B * OH * OW;
  if (M > 0x7fffffff) return 0;
  const int bm = pipe_tile_rows((int)M, Cout, 1), rps = OH * OW;
  return rps % bm == 0 ? rps / bm : 0;
}
