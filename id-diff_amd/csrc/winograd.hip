// 3x3 / stride 1 / pad 1 convolutions of the score networks by Winograd's minimal filtering F(2x2, 3x3) on the
// fp32 matrix cores of gfx950: 16 multiplications per 2x2 output tile and (cin, cout) pair instead of 36, i.e.
// 2.25x fewer MFMA flops than the implicit GEMM of igemm.hip, all of it still exact-fp32 arithmetic
// (v_mfma_f32_32x32x2_f32; the transforms only add, subtract and halve).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch at (2ty-1, 2tx-1), Y: 2x2 outputs at (2ty, 2tx)
//
//   U = G g G^T is packed once per layer (idiff_winograd_pack_f32) as [Cin/8][Cout/64][16 positions][64 cout][8 cin].
//   V = B^T d B is formed by the loader on its way from HBM/L2 to LDS and never written to memory.
//   M_p = V_p U_p^T for the 16 positions p are 16 independent [tiles x Cin] x [Cin x Cout] contractions; a workgroup
//   owns 64 tiles x 64 output channels of ALL 16 of them, so the inverse transform A^T M A is local to the
//   workgroup and M never leaves registers either.
//
// Workgroup: 512 threads = 8 waves = 2 (position halves: rows {0,1} / {2,3} of the 4x4 transform domain)
//   x 2 (32-tile halves) x 2 (32-channel halves); a wave holds 8 positions x one 32x32 MFMA tile = 128 accumulators.
// K loop: 8 input channels per step, two LDS stages of [16][64][8] floats for V and for U (128 KB together), one
//   barrier per step; global loads run two steps ahead in registers, the transform + LDS writes of step s+1 sit
//   between the two MFMA halves of step s.
// LDS rows are 32 bytes (8 channels); the two 16-byte halves of row r are swapped when bit 3 of r is set, which
//   makes the ds_read_b128 of a 32-row MFMA operand conflict-free (lane groups of ds_read_b128:
//   MI355X_MICROARCH.md section LDS).  As in igemm.hip each lane half feeds four consecutive channels to four
//   successive MFMAs (the order of the k-reduction is free), one 16-byte read per operand per 4 MFMAs.
// Loader roles are per wave: waves 0-3 gather and transform V (a lane pair 8 lanes apart shares one (tile,
//   4-channel) unit: each loads two rows of the 4x4 patch, mixes columns locally and swaps one row through DPP
//   row_ror:8 for the row mixing), waves 4-7 copy the pre-swizzled U slab (fully coalesced, LDS image = HBM image).
// Epilogue: partial A^T M A per position half, halves exchanged through LDS, then the same fused epilogue as
//   igemm.hip (bias, per-sample bias, activation, residual, scales, optional per-tile column statistics).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int WG_TILES = 64;
constexpr int WG_COUT = 64;
constexpr int KC = 8;
constexpr int NPOS = 16;
constexpr int OPER_FLOATS = NPOS * 64 * KC;       // one operand of one stage in HBM order: 8192 floats = 32 KB
constexpr int V_SLOT = 64 * KC + 8;               // LDS floats per position slot of V: 32 bytes of padding rotate the
                                                  // banks so that the four patch rows of a unit store conflict-free
constexpr int U_SLOT = 64 * KC;
constexpr int V_FLOATS = NPOS * V_SLOT;           // 8320
constexpr int STAGE_FLOATS = V_FLOATS + NPOS * U_SLOT;   // V then U: 16512 floats
constexpr size_t LDS_BYTES = (size_t)2 * STAGE_FLOATS * sizeof(float);   // 129 KB
constexpr int64_t X_LIMIT = 0xFFFF0000ll;          // one buffer descriptor, with room for the invalid-pixel bias
constexpr uint32_t INVALID_PIXEL = 0xFFFF8000u;    // + channel offset (< 32 KB) stays beyond any valid extent

struct WinoParams {
  const float *x;
  const float *u;
  float *out;
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_per_img, total_tiles;
  int tiles_m, tiles_n;
  uint32_t x_bytes, u_bytes;
  idiff_epilogue ep;
  int has_ep;
};

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float qperm(float v) {   // DPP quad_perm:[2,2,1,1]: lanes 0,1 of a quad read lane 2, lanes 2,3 read lane 1
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, true));
}

struct TailCtx { int tile0, tile_m, n0, tb, cb, lane, tid; };

// Inverse transform + fused epilogue for the wave of position half PH (rows {2PH, 2PH+1} of the transform domain):
//   t0j = m0j + m1j + m2j, t1j = m1j - m2j - m3j;  Y[a][0] = ta0 + ta1 + ta2, Y[a][1] = ta1 - ta2 - ta3
// acc[j] holds row 2PH, acc[4+j] row 2PH+1.  The wave finishes accumulator registers 8PH..8PH+7 (two runs of four
// consecutive tiles) and hands its partial sums for the other eight to the partner wave through LDS.  PH is a
// template parameter so that every accumulator index is a compile-time constant.
template <int PH>
__device__ __forceinline__ void winograd_tail(const WinoParams &p, floatx16 (&acc)[8], float *lds, const TailCtx &c) {
  auto partial = [&](int reg) -> float4 {
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ma = acc[j][reg], mb = acc[4 + j][reg];
      if (PH == 0) { t0[j] = ma + mb; t1[j] = mb; }
      else { t0[j] = ma; t1[j] = -ma - mb; }
    }
    return make_float4(t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3], t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]);
  };
  float4 *xch = reinterpret_cast<float4 *>(lds);          // [pair 4][dst half 2][8][64 lanes] float4 = 64 KB
  const int pair = c.tb + 2 * c.cb;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    xch[((pair * 2 + (1 - PH)) * 8 + k) * 64 + c.lane] = partial(8 * (1 - PH) + k);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

  const idiff_epilogue &ep = p.ep;
  const bool want_stats = p.has_ep && ep.colstats != nullptr;
  const int n = c.n0 + c.cb * 32 + (c.lane & 31);
  const float bias = (p.has_ep && ep.bias) ? ep.bias[n] : 0.f;
  const bool per_image = p.ep.rows_per_group == p.H * p.W;   // the usual per-sample bias / scale: group = image
  const int tiles_y = p.tiles_per_img / p.tiles_x;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int run = 0; run < 2; ++run) {
    // tiles of registers 8PH + 4run + (0..3): consecutive, starting at
    int T = c.tile0 + c.tb * 32 + 16 * PH + 8 * run + 4 * (c.lane >> 5);
    int img = T / p.tiles_per_img;
    int rem = T - img * p.tiles_per_img;
    int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * run + kk;
      float4 y = partial(8 * PH + k);
      const float4 z = xch[((pair * 2 + PH) * 8 + k) * 64 + c.lane];
      const float yv[4] = {y.x + z.x, y.y + z.y, y.z + z.z, y.w + z.w};
      if (T < p.total_tiles) {
        const int64_t m00 = ((int64_t)img * p.H + 2 * ty) * p.W + 2 * tx;
        float rb = 0.f, rs = 1.f;
        if (p.has_ep && per_image) {
          if (ep.rowbias) rb = ep.rowbias[(int64_t)img * ep.ld_rowbias + n];
          if (ep.rowscale) rs = ep.rowscale[img];
        }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
          const int64_t m = m00 + (ab >> 1) * p.W + (ab & 1);
          float v = yv[ab] + bias;
          if (p.has_ep) {
            if (!per_image) {
              const int64_t g = m / ep.rows_per_group;
              rb = ep.rowbias ? ep.rowbias[g * ep.ld_rowbias + n] : 0.f;
              rs = ep.rowscale ? ep.rowscale[g] : 1.f;
            }
            v = idiff::act_apply(v + rb, ep.act);
            if (ep.residual) v += ep.residual[m * ep.ld_residual + n];
            v *= ep.out_scale;
            v *= rs;
          }
          p.out[m * p.Cout + n] = v;
          if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ++T;
      if (++tx == p.tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++img; } }
    }
  }
  if (want_stats) {
    double *red = reinterpret_cast<double *>(lds + 16384);   // behind the 64 KB exchange area: [4][64][2]
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (c.lane < 32) {
      const int slot = ((c.tb * 2 + PH) * WG_COUT + c.cb * 32 + c.lane) * 2;
      red[slot] = s1; red[slot + 1] = s2;
    }
    __syncthreads();
    if (c.tid < WG_COUT) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * WG_COUT + c.tid) * 2]; b += red[(w * WG_COUT + c.tid) * 2 + 1]; }
      double *dst = ep.colstats + ((int64_t)c.tile_m * p.Cout + c.n0 + c.tid) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
}

__global__ void __launch_bounds__(512)
winograd_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int tile0 = tile_m * WG_TILES, n0 = tile_n * WG_COUT;
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index in an SGPR: roles, operand halves and buffer descriptors stay provably wave-uniform (no waterfall
  // loops around the buffer loads, scalar branches for the role split)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave & 1, tb = (wave >> 1) & 1, cb = wave >> 2;

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);

  // ---------------------------------------------------------------- loader state
  // Every thread owns one row r of the 4x4 patch of one (tile, 4-channel quad) unit -- 64 tiles x 2 quads x 4 rows = 512
  // threads -- plus four 16-byte pieces of the U slab.  The four rows of a unit sit in the four lanes of a DPP quad:
  // the column mixing of B^T d B is local, the row mixing needs one other row (rows 0,1 <- row 2; rows 2,3 <- row 1),
  // fetched with quad_perm:[2,2,1,1] and folded in with one fma by the lane's sign (-1, +1, -1, -1).  That yields row 3
  // negated; the packed U carries the same sign on its row 3, so the products are unchanged.
  // Transform-domain position (i, j) lives in LDS slot 4j + i (and U is packed in that order).
  const bool early = wave < 4;    // waves w and w+4 share a SIMD: one transforms while the other feeds the matrix pipe
  uint32_t v_src[4];
  int v_dst;
  float sgn;
  {
    const int r = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
    const int T = tile0 + tl;
    const bool tv = T < p.total_tiles;
    const int TT = tv ? T : 0;
    const int img = TT / p.tiles_per_img, rem = TT - img * p.tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y = 2 * ty - 1 + r, x0 = 2 * tx - 1;
    const bool yok = tv && y >= 0 && y < p.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x0 + j;
      v_src[j] = (yok && xx >= 0 && xx < p.W) ? (uint32_t)(((img * p.H + y) * p.W + xx) * p.Cin + q * 4) * 4u : INVALID_PIXEL;
    }
    v_dst = r * V_SLOT + tl * KC + 4 * (q ^ ((tl >> 3) & 1));
    sgn = (r == 1) ? 1.f : -1.f;
  }

  float4 ldv[4], ldu[4];
  const int nsteps = p.Cin / KC;
  int f_step = 0;
  uint32_t u_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) u_src[i] = (uint32_t)tid * 16u + (uint32_t)i * 8192u;
  // the per-step offsets are wave-uniform: they ride in the buffer instruction's scalar offset (not part of the range
  // check of a raw buffer, so an out-of-range pixel stays out of range) instead of costing a VALU add per load
  auto fetch = [&]() {
    const int choff = f_step * (KC * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) ldv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)v_src[j], choff, 0));
    const int slab = (f_step * p.tiles_n + tile_n) * (OPER_FLOATS * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) ldu[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_src[i], slab, 0));
    ++f_step;
  };

  auto stage = [&](int buf) {
    float *Vd = lds + buf * STAGE_FLOATS + v_dst;
    float4 c[4];
    c[0] = f4sub(ldv[0], ldv[2]); c[1] = f4add(ldv[1], ldv[2]); c[2] = f4sub(ldv[2], ldv[1]); c[3] = f4sub(ldv[1], ldv[3]);
    // c += sgn * quad_perm[2,2,1,1](c), the permute folded into the fma's DPP operand (hipcc keeps v_mov_b32_dpp + v_fmac
    // apart).  s_nop 1: a VALU write of a VGPR needs two wait states before a DPP read of it, and the hazard
    // recogniser does not look inside inline asm.
#define IDIFF_QFMA(x) "v_fmac_f32_dpp " x ", " x ", %16 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm volatile("s_nop 1\n" IDIFF_QFMA("%0") IDIFF_QFMA("%1") IDIFF_QFMA("%2") IDIFF_QFMA("%3") IDIFF_QFMA("%4") IDIFF_QFMA("%5")
                 IDIFF_QFMA("%6") IDIFF_QFMA("%7") IDIFF_QFMA("%8") IDIFF_QFMA("%9") IDIFF_QFMA("%10") IDIFF_QFMA("%11")
                 IDIFF_QFMA("%12") IDIFF_QFMA("%13") IDIFF_QFMA("%14") IDIFF_QFMA("%15")
                 : "+v"(c[0].x), "+v"(c[0].y), "+v"(c[0].z), "+v"(c[0].w), "+v"(c[1].x), "+v"(c[1].y), "+v"(c[1].z), "+v"(c[1].w),
                   "+v"(c[2].x), "+v"(c[2].y), "+v"(c[2].z), "+v"(c[2].w), "+v"(c[3].x), "+v"(c[3].y), "+v"(c[3].z), "+v"(c[3].w)
                 : "v"(sgn));
#undef IDIFF_QFMA
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(Vd + j * 4 * V_SLOT) = c[j];
    float *Ud = lds + buf * STAGE_FLOATS + V_FLOATS + tid * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(Ud + i * 2048) = ldu[i];
  };

  floatx16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int frag = fr * KC + 4 * (fh ^ ((fr >> 3) & 1));
  const int a_frag = 2 * ph * V_SLOT + tb * 32 * KC + frag;             // slot of accumulator pp: 4 (pp & 3) + 2 ph + (pp >> 2)
  const int b_frag = V_FLOATS + 2 * ph * U_SLOT + cb * 32 * KC + frag;

  auto compute = [&](int buf, int pp0) {
    const float *S = lds + buf * STAGE_FLOATS;
#pragma unroll
    for (int pp = pp0; pp < pp0 + 4; ++pp) {
      const int slot = 4 * (pp & 3) + (pp >> 2);
      const float4 a = *reinterpret_cast<const float4 *>(S + a_frag + slot * V_SLOT);
      const float4 b = *reinterpret_cast<const float4 *>(S + b_frag + slot * U_SLOT);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[pp], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[pp], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[pp], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[pp], 0, 0, 0);
    }
  };

  fetch();
  stage(0);
  if (nsteps > 1) fetch();
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    // fp32 MFMAs and VALU work share the SIMD's fp32 lanes, and a wave's transform instructions trickle out slowly
    // beside another wave's MFMA stream; so per SIMD one wave stages step s+1 at the START of step s under its
    // partner's 32 MFMAs, the partner at the END under the first wave's MFMAs -- the pipe never waits for a transform
    if (early) {
      if (s + 1 < nsteps) stage(buf ^ 1);   // loaded one step ago
      if (s + 2 < nsteps) fetch();
    }
    compute(buf, 0);
    __builtin_amdgcn_sched_barrier(0);      // keep the operand reads of the two halves from being hoisted together
    compute(buf, 4);
    if (!early) {
      if (s + 1 < nsteps) stage(buf ^ 1);
      if (s + 2 < nsteps) fetch();
    }
    __syncthreads();
  }

  TailCtx c;
  c.tile0 = tile0; c.tile_m = tile_m; c.n0 = n0; c.tb = tb; c.cb = cb; c.lane = lane; c.tid = tid;
  if (ph == 0) winograd_tail<0>(p, acc, lds, c);
  else winograd_tail<1>(p, acc, lds, c);
}

// U = G g G^T in fp64, rounded once; g[ky][kx] = wt[cout][ky][kx][cin] (the K-contiguous panel of the direct kernel).
__global__ void winograd_pack_kernel(const float *wt, float *u, int Cin, int Cout) {
  const int64_t total = (int64_t)Cin * Cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)wt[((int64_t)cout * 9 + ky * 3 + kx) * Cin + cin];
    double gg[4][3];   // G g
    for (int kx = 0; kx < 3; ++kx) {
      gg[0][kx] = g[0][kx];
      gg[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
      gg[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
      gg[3][kx] = g[2][kx];
    }
    const int s = cin / KC, c8 = cin % KC, nt = cout / WG_COUT, co = cout % WG_COUT;
    const int slot = 4 * ((c8 >> 2) ^ ((co >> 3) & 1)) + (c8 & 3);
    float *dst = u + ((int64_t)(s * (Cout / WG_COUT) + nt) * NPOS * 64 + co) * KC + slot;
    for (int i = 0; i < 4; ++i) {
      const double r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
      const double v[4] = {r0, 0.5 * (r0 + r1 + r2), 0.5 * (r0 - r1 + r2), r2};
      const double sign = i == 3 ? -1.0 : 1.0;     // the kernel's input transform produces row 3 negated
      for (int j = 0; j < 4; ++j) dst[(int64_t)(j * 4 + i) * 64 * KC] = (float)(sign * v[j]);
    }
  }
}

bool geometry_ok(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return false;
  if (H % 2 || W % 2 || Cin % KC || Cout % WG_COUT) return false;
  if ((int64_t)Cin * 4 > 0x7000) return false;                               // channel offset must stay below the invalid-pixel bias
  if ((int64_t)16 * Cin * Cout * 4 >= X_LIMIT) return false;
  if ((int64_t)B * (H / 2) * (W / 2) > 0x7fffffff / 4) return false;
  return true;
}

}  // namespace

IDIFF_API int idiff_conv2d_winograd_ok(int B, int H, int W, int Cin, int Cout) {
  if (getenv("IDIFF_NO_WINOGRAD")) return 0;
  return geometry_ok(B, H, W, Cin, Cout) ? 1 : 0;
}

IDIFF_API int idiff_conv2d_winograd_colstats_split(int B, int H, int W, int Cin, int Cout) {
  if (!idiff_conv2d_winograd_ok(B, H, W, Cin, Cout) || getenv("IDIFF_NO_COLSTATS")) return 0;
  if ((int64_t)B * H * W * Cin * 4 >= X_LIMIT) return 0;
  const int tpi = (H / 2) * (W / 2);
  return tpi % WG_TILES == 0 ? tpi / WG_TILES : 0;
}

IDIFF_API int64_t idiff_winograd_weight_floats(int Cin, int Cout) { return (int64_t)16 * Cin * Cout; }

IDIFF_API int idiff_winograd_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % KC || Cout % WG_COUT)
    return fail("winograd_pack: Cin must be a multiple of %d and Cout of %d (got %d, %d)", KC, WG_COUT, Cin, Cout);
  if (!wt || !u) return fail("winograd_pack: null pointer");
  const int64_t total = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(winograd_pack_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt, u, Cin, Cout);
  return launch_status("winograd_pack");
}

IDIFF_API int idiff_conv2d_winograd_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                        const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (B == 0) return 0;
  if (!geometry_ok(B, H, W, Cin, Cout))
    return fail("conv2d_winograd: geometry B=%d H=%d W=%d Cin=%d Cout=%d not supported (ask idiff_conv2d_winograd_ok)", B, H, W, Cin, Cout);
  if (!x || !u || !out) return fail("conv2d_winograd: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)u & 15)) return fail("conv2d_winograd: x and u must be 16-byte aligned");
  const int64_t x_bytes = (int64_t)B * H * W * Cin * 4;
  if (x_bytes >= X_LIMIT) {
    if (ep && ep->colstats) return fail("conv2d_winograd: colstats is not available for inputs beyond one buffer descriptor");
    if (B < 2) return fail("conv2d_winograd: a single image exceeds one buffer descriptor");
    const int rpg = (ep && ep->rows_per_group > 0) ? ep->rows_per_group : 1;
    const int b_lo = B / 2;
    const int64_t m_lo = (int64_t)b_lo * H * W;
    if (m_lo % rpg) return fail("conv2d_winograd: cannot split the batch inside an epilogue row group");
    idiff_epilogue lo, hi;
    if (ep) {
      lo = *ep; hi = *ep;
      const int64_t g0 = m_lo / rpg;
      if (hi.rowbias) hi.rowbias += g0 * ep->ld_rowbias;
      if (hi.residual) hi.residual += m_lo * ep->ld_residual;
      if (hi.rowscale) hi.rowscale += g0;
    }
    int rc = idiff_conv2d_winograd_f32(x, u, out, b_lo, H, W, Cin, Cout, ep ? &lo : nullptr, stream);
    if (rc) return rc;
    return idiff_conv2d_winograd_f32(x + m_lo * Cin, u, out + m_lo * Cout, B - b_lo, H, W, Cin, Cout, ep ? &hi : nullptr, stream);
  }
  WinoParams p = {};
  p.x = x; p.u = u; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tiles_x = W / 2; p.tiles_per_img = (H / 2) * (W / 2); p.total_tiles = B * p.tiles_per_img;
  p.tiles_m = ceil_div(p.total_tiles, WG_TILES); p.tiles_n = Cout / WG_COUT;
  p.x_bytes = (uint32_t)x_bytes; p.u_bytes = (uint32_t)((int64_t)16 * Cin * Cout * 4);
  if (ep) {
    p.ep = *ep; p.has_ep = 1;
    if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1;
    if (ep->colstats && p.tiles_per_img % WG_TILES)
      return fail("conv2d_winograd: colstats needs whole workgroups per sample (ask idiff_conv2d_winograd_colstats_split)");
  } else {
    p.has_ep = 0; p.ep.rows_per_group = 1; p.ep.out_scale = 1.f;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(winograd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_BYTES);
    if (e != hipSuccess) { set_error("conv2d_winograd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL(winograd_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
  return launch_status("conv2d_winograd");
}
