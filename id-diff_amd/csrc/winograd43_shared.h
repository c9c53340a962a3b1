// Shared by winograd43.hip (fp32 contraction) and winograd43h.hip (contraction on fp16 pairs): the F(4x4, 3x3) geometry, the
// kernel arguments, the 6-point transforms' constants, the workgroup tail and the fp64 filter transform.  See winograd43.hip's header
// for the algorithm.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

// Interpolation points 0, +-a, +-b, infinity with a b = 1 (reciprocal pairs keep the transforms balanced): a = 2/3, b = 3/2.
// (scripts/f43_emulation.py on the whole network: rel_err(S) 3.3e-6 for this set, 4.9e-6 for 1/2, 2 -- whose constants are all
// dyadic -- and 7.0e-6 for Lavin's 1, 2.)  The transforms use these fp32 constants; G is evaluated in fp64 from the same a, b.
#ifdef IDIFF_W43_DYADIC_POINTS   // A/B builds only (scripts/wino43_ab.py): the dyadic set 1/2, 2
constexpr double F4_A = 0.5, F4_B = 2.0;
#else
constexpr double F4_A = 2.0 / 3.0, F4_B = 1.5;
#endif
constexpr float F4_a = (float)F4_A, F4_b = (float)F4_B, F4_a2 = (float)(F4_A * F4_A), F4_b2 = (float)(F4_B * F4_B),
                F4_a3 = (float)(F4_A * F4_A * F4_A), F4_b3 = (float)(F4_B * F4_B * F4_B), F4_ab2 = (float)(F4_A * F4_A + F4_B * F4_B);
constexpr int F4_TILES = 32;
constexpr int F4_THREADS = 512;
constexpr int F4_COUT = 64;
constexpr int F4_KC = 8;
constexpr int F4_NPOS = 36;
constexpr int F4_Z_FLOATS = 6 * F4_TILES * 2 * F4_COUT;  // the tail's exchange: [6 rows][tiles][2 columns][64 cout] = 24576 floats
constexpr int64_t F4_X_LIMIT = 0xFFFF0000ll;
constexpr uint32_t F4_INVALID = 0xFFFF8000u;             // beyond any valid extent (the scalar step offset is not range-checked)

struct Wino43Params {
  const float *x;
  const float *u;
  float *out;
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_y, tiles_per_img, total_tiles;
  int tx_shift, tpi_shift;        // log2 of tiles_x / tiles_per_img when both are powers of two, else -1 (division)
  int tiles_m, tiles_n, ngroup;
  uint32_t x_bytes, u_bytes, out_bytes, res_bytes;
  idiff_epilogue ep;
  int has_ep;
  // the input transform's constants as kernel arguments: they then live in SGPRs and the twelve operations are plain VOP3 fmas
  // with a scalar operand; as compile-time literals they became v_fmamk_f32 (a 32-bit literal per instruction), measured 3.7 %
  // slower over a forward than the dyadic point set whose constants are inline operands (scripts/wino43_ab.py)
  float c_nb2, c_na2, c_nab2, c_a, c_b;
#ifdef IDIFF_W43H_STAMP
  uint64_t *stamps;               // diagnostic build: 8 ticks per workgroup (the launcher takes the address from IDIFF_W43H_STAMP_PTR)
#endif
};

struct F4Consts { float nb2, na2, nab2, a, b; };

__device__ __forceinline__ void f4_split_tile(const Wino43Params &p, int T, int &img, int &ty, int &tx) {
  if (p.tx_shift >= 0) {
    img = T >> p.tpi_shift;
    const int rem = T & (p.tiles_per_img - 1);
    ty = rem >> p.tx_shift; tx = rem & (p.tiles_x - 1);
  } else {
    img = T / p.tiles_per_img;
    const int rem = T - img * p.tiles_per_img;
    ty = rem / p.tiles_x; tx = rem - ty * p.tiles_x;
  }
}

// the 6-point input transform t = B^T d on one component: 12 operations (the odd parts are formed unscaled, d3 - b^2 d1 and
// d3 - a^2 d1, and their factors a and b ride in the fused multiply-adds that combine them with the even parts)
__device__ __forceinline__ void f4_bt(const F4Consts &k, const float d0, const float d1, const float d2, const float d3, const float d4,
                                      const float d5, float &t0, float &t1, float &t2, float &t3, float &t4, float &t5) {
  const float pe = fmaf(k.nb2, d2, d4), po = fmaf(k.nb2, d1, d3);          // t1, t2 = pe +- a po
  const float re = fmaf(k.na2, d2, d4), ro = fmaf(k.na2, d1, d3);          // t3, t4 = re +- b ro
  t0 = fmaf(k.nab2, d2, d0 + d4);
  t1 = fmaf(k.a, po, pe); t2 = fmaf(-k.a, po, pe);
  t3 = fmaf(k.b, ro, re); t4 = fmaf(-k.b, ro, re);
  t5 = fmaf(k.nab2, d3, d1 + d5);
}

// The tail of a workgroup (see the file header): acc -> row mixing through LDS in two passes -> column mixing, epilogue, stores.
// DESCALE: the accumulators carry U's power-of-two scaling, undone on the finished sums before the epilogue.
template <bool DESCALE>
__device__ __forceinline__ void f4_tail(const Wino43Params &p, float *lds, floatx16 (&acc)[9], const int tile0, const int tile_m,
                                        const int n0, const int wh, const int wa, const int wb, const float descale, uint64_t *stamp_out = nullptr) {
  const idiff_epilogue &ep = p.ep;
  const bool has_ep = p.has_ep != 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int cq = tid & 15, tl = tid >> 4;            // this thread finishes channels n .. n + 3 of tile tl
  const int n = n0 + 4 * cq;
  const bool has_res = has_ep && ep.residual != nullptr;
  const bool scaled = has_ep && (ep.out_scale != 1.f || ep.rowscale != nullptr);
  const int act = has_ep ? ep.act : (int)IDIFF_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)ep.residual, 0, (int)p.res_bytes, 0x00020000);
  const int ld_res = (int)ep.ld_residual;
  uint32_t ooff, roff;
  float4 badd;
  float sc;
  auto prep = [&]() {
    badd = make_float4(0.f, 0.f, 0.f, 0.f);
    sc = has_ep ? ep.out_scale : 1.f;
    const int T = tile0 + tl;
    const bool ok = T < p.total_tiles;
    int img, ty, tx;
    f4_split_tile(p, ok ? T : 0, img, ty, tx);
    const int m00 = (img * p.H + 4 * ty) * p.W + 4 * tx;
    ooff = ok ? ((uint32_t)m00 * (uint32_t)p.Cout + (uint32_t)n) * 4u : F4_INVALID;
    roff = ok ? ((uint32_t)m00 * (uint32_t)ld_res + (uint32_t)n) * 4u : F4_INVALID;
    if (has_ep && ep.bias) badd = *reinterpret_cast<const float4 *>(ep.bias + n);
    if (has_ep && ok) {
      if (ep.rowbias) {
        const float4 rb = *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)img * ep.ld_rowbias + n);
        badd.x += rb.x; badd.y += rb.y; badd.z += rb.z; badd.w += rb.w;
      }
      if (ep.rowscale) sc *= ep.rowscale[img];
    }
  };
  const bool want_stats = has_ep && ep.colstats != nullptr;
#ifdef IDIFF_W43H_STAMP
  uint64_t st_tail[5] = {0, 0, 0, 0, 0};
#define IDIFF_TAIL_STAMP(k) st_tail[k] = __builtin_amdgcn_s_memrealtime()
#else
#define IDIFF_TAIL_STAMP(k)
#endif
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  // accumulator register `reg` of lane l is tile row (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), cout wh * 32 + (l & 31)
  float *zbase = lds + (size_t)(4 * (lane >> 5)) * 2 * F4_COUT + wh * 32 + (lane & 31);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
    float4 res[2][4];
    // the output addresses and per-tile epilogue operands (bias, per-sample bias, scale: loads from L2) are formed HERE, in front of
    // the exchange that covers their latency; formed behind it they stood in the way of every output phase
    prep();
    if (has_res) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          res[bb][a] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)roff, (a * p.W + 2 * pass + bb) * ld_res * 4, 0));
    }
    // Row mixing z_{i,b} = sum_j A^T[b][j] m_{i,j}: a wave has three of a row's six columns, its partner (other column block, same
    // rows and channels) the rest.  Each wave PARKS its part for one half of the tile rows (block 1: accumulator registers 0-7 = tile
    // rows < 16, block 0: registers 8-15) and after the barrier ADDS its part for the other half onto what the partner parked: every
    // wave works in both phases and moves half of what a park-all / add-all split (block 1 parks, block 0 adds) would make it move.
    auto exchange = [&](auto blk, auto lo, auto park) __attribute__((always_inline)) {
      constexpr int BLK = decltype(blk)::value, LO = decltype(lo)::value;
      constexpr bool PARK = decltype(park)::value;
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int reg = LO + r8;
          const float ma = acc[ii * 3][reg], mb = acc[ii * 3 + 1][reg], mc = acc[ii * 3 + 2][reg];
          float p0, p1;
          if (BLK == 1) {                               // columns 3, 4, 5: m3, m4, m5
            if (pass == 0) { p0 = ma + mb; p1 = F4_b * (ma - mb); }
            else { p0 = F4_b2 * (ma + mb); p1 = fmaf(F4_b3, ma - mb, mc); }
          } else {                                      // columns 0, 1, 2: m0, m1, m2
            if (pass == 0) { p0 = ma + (mb + mc); p1 = F4_a * (mb - mc); }
            else { p0 = F4_a2 * (mb + mc); p1 = F4_a3 * (mb - mc); }
          }
          float *zp = zbase + (((3 * wa + ii) * F4_TILES + (reg & 3) + 8 * (reg >> 2)) * 2) * F4_COUT;
          if (PARK) { zp[0] = p0; zp[F4_COUT] = p1; }
          // (an LDS float add, ds_add_f32, instead of this read - add - write was measured TWICE as slow for the whole kernel: 146.3 against
          //  78.6 ms per forward, profiles/r05_tail_ab.txt -- LDS float atomics do not run at the store rate)
          else { zp[0] += p0; zp[F4_COUT] += p1; }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I8 = std::integral_constant<int, 8>;
    if (wb == 1) exchange(I1(), I0(), std::true_type()); else exchange(I0(), I8(), std::true_type());
    __syncthreads();
    if (wb == 1) exchange(I1(), I8(), std::false_type()); else exchange(I0(), I0(), std::false_type());
    __syncthreads();
    IDIFF_TAIL_STAMP(2 * pass);                       // z of this pass exchanged
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const float *zr = lds + (tl * 2 + bb) * F4_COUT + 4 * cq;
      float4 z[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) z[i] = *reinterpret_cast<const float4 *>(zr + i * (F4_TILES * 2 * F4_COUT));
      const int b = 2 * pass + bb;
      float y[4][4];
#define IDIFF_F4_AT(cmp, e)                                                                                  \
      {                                                                                                      \
        const float s12 = z[1].cmp + z[2].cmp, d12 = z[1].cmp - z[2].cmp, s34 = z[3].cmp + z[4].cmp, d34 = z[3].cmp - z[4].cmp; \
        y[0][e] = z[0].cmp + (s12 + s34);                                                                     \
        y[1][e] = fmaf(F4_b, d34, F4_a * d12);                                                                \
        y[2][e] = fmaf(F4_b2, s34, F4_a2 * s12);                                                              \
        y[3][e] = fmaf(F4_b3, d34, fmaf(F4_a3, d12, z[5].cmp));                                               \
      }
      IDIFF_F4_AT(x, 0) IDIFF_F4_AT(y, 1) IDIFF_F4_AT(z, 2) IDIFF_F4_AT(w, 3)
#undef IDIFF_F4_AT
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (DESCALE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[a][e] *= descale;
        }
        y[a][0] += badd.x; y[a][1] += badd.y; y[a][2] += badd.z; y[a][3] += badd.w;
        if (act != IDIFF_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[a][e] = idiff::act_apply(y[a][e], act);
        }
        if (has_res) { y[a][0] += res[bb][a].x; y[a][1] += res[bb][a].y; y[a][2] += res[bb][a].z; y[a][3] += res[bb][a].w; }
        if (scaled) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[a][e] *= sc;
        }
      }
      if (want_stats) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += (double)y[a][e]; s2[e] += (double)y[a][e] * (double)y[a][e]; }
      }
      // the four rows are stored together from registers nothing writes again before the next column (see winograd43_kernel)
      const int so = b * p.Cout * 4, rp = p.W * p.Cout * 4;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, make_float4(y[a][0], y[a][1], y[a][2], y[a][3])), rO, (int)ooff,
                                               so + a * rp, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    IDIFF_TAIL_STAMP(2 * pass + 1);                   // outputs of this pass stored (issued)
  }
  if (want_stats) {
    __syncthreads();
    double *red = reinterpret_cast<double *>(lds);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[((tl * F4_COUT) + 4 * cq + e) * 2] = s1[e];
      red[((tl * F4_COUT) + 4 * cq + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    const int per = p.tiles_per_img < F4_TILES ? p.tiles_per_img : F4_TILES;      // tiles added up per slot
    const int slots = F4_TILES / per;
    if (per >= 4) {
      // two levels, so that all 512 threads add: thread (channel, q) the four tiles 4q .. 4q + 3 (never across a sample: per is a
      // multiple of 4), then one thread per (slot, channel) the per / 4 partial sums -- a chain of 4 + 8 additions, not 32
      double *part = red + F4_TILES * F4_COUT * 2;                                // [8][64][2] behind the per-tile sums
      {
        const int ch = tid & 63, q = tid >> 6;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a += red[((4 * q + k) * F4_COUT + ch) * 2]; b += red[((4 * q + k) * F4_COUT + ch) * 2 + 1]; }
        part[(q * F4_COUT + ch) * 2] = a; part[(q * F4_COUT + ch) * 2 + 1] = b;
      }
      __syncthreads();
      const int quads = per / 4;
      for (int o = tid; o < slots * F4_COUT; o += F4_THREADS) {
        const int smp = o / F4_COUT, ch = o - smp * F4_COUT;
        const int64_t slot = (int64_t)tile_m * slots + smp;                         // sample, or (sample, split) = workgroup row
        if (slots > 1 && slot >= p.B) continue;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < quads; ++k) { a += part[((smp * quads + k) * F4_COUT + ch) * 2]; b += part[((smp * quads + k) * F4_COUT + ch) * 2 + 1]; }
        double *dst = ep.colstats + (slot * p.Cout + n0 + ch) * 2;
        dst[0] = a; dst[1] = b;
      }
    } else {
      for (int o = tid; o < slots * F4_COUT; o += F4_THREADS) {
        const int smp = o / F4_COUT, ch = o - smp * F4_COUT;
        const int64_t slot = (int64_t)tile_m * slots + smp;
        if (slots > 1 && slot >= p.B) continue;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < per; ++k) { a += red[((smp * per + k) * F4_COUT + ch) * 2]; b += red[((smp * per + k) * F4_COUT + ch) * 2 + 1]; }
        double *dst = ep.colstats + (slot * p.Cout + n0 + ch) * 2;
        dst[0] = a; dst[1] = b;
      }
    }
  }
#ifdef IDIFF_W43H_STAMP
  IDIFF_TAIL_STAMP(4);
  if (threadIdx.x == 0 && stamp_out) { for (int k = 0; k < 5; ++k) stamp_out[3 + k] = st_tail[k]; }
#endif
#undef IDIFF_TAIL_STAMP
}


// U = G g G^T in fp64 for one (cin, cout) pair: the 36 values, and their largest magnitude
__device__ __forceinline__ void f4_u_of_pair(const float *wt, int Cin, int cin, int cout, double (&U)[36]) {
  const double a = F4_A, b = F4_B, na = 1.0 / (2.0 * a * a * (a * a - b * b)), nb = 1.0 / (2.0 * b * b * (b * b - a * a)), n0 = 1.0 / (a * a * b * b);
  const double G[6][3] = {{n0, 0.0, 0.0}, {na, a * na, a * a * na}, {na, -a * na, a * a * na}, {nb, b * nb, b * b * nb}, {nb, -b * nb, b * b * nb},
                          {0.0, 0.0, 1.0}};
  double g[3][3];
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)wt[((int64_t)cout * 9 + ky * 3 + kx) * Cin + cin];
  double gg[6][3];
  for (int i = 0; i < 6; ++i)
    for (int kx = 0; kx < 3; ++kx) gg[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) U[6 * i + j] = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
}


bool f4_geometry_ok(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return false;
  if (H % 4 || W % 4 || Cin % F4_KC || Cout % F4_COUT) return false;
  if ((int64_t)Cin * 4 > 0x7000) return false;                               // channel offset must stay below the invalid-pixel bias
  if ((int64_t)36 * Cin * Cout * 4 >= F4_X_LIMIT) return false;
  if ((int64_t)B * (H / 4) * (W / 4) > 0x7fffffff / 4) return false;
  if ((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) * 4 >= F4_X_LIMIT) return false;   // one buffer descriptor per tensor
  return true;
}


}  // namespace
