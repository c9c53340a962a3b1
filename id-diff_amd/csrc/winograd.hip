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
constexpr int OPER_FLOATS = NPOS * 64 * KC;       // one operand of one stage: 8192 floats = 32 KB
constexpr int STAGE_FLOATS = 2 * OPER_FLOATS;     // V then U
constexpr size_t LDS_BYTES = (size_t)2 * STAGE_FLOATS * sizeof(float);   // 128 KB
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
__device__ __forceinline__ float ror8(float v) {   // value of the lane 8 away inside the 16-lane DPP row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
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

  // ---------------------------------------------------------------- loader state (role is uniform per wave)
  const bool v_role = wave < 4;
  uint32_t src_off[8];      // V: byte offsets of 2 patch rows x 4 columns (outer row first); U: float4 slots
  int v_dst = 0, pos_outer = 0, pos_inner = 0;
  float sgn = 1.f;
  if (v_role) {
    const int tt = tid & 3, q = (tid >> 2) & 1, half = (tid >> 3) & 1, g = tid >> 4;
    const int tl = g * 4 + tt;
    const int T = tile0 + tl;
    const bool tv = T < p.total_tiles;
    const int TT = tv ? T : 0;
    const int img = TT / p.tiles_per_img, rem = TT - img * p.tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    const int r_outer = half ? 3 : 0, r_inner = half ? 2 : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x0 + j;
      const bool xok = tv && xx >= 0 && xx < p.W;
      const int yo = y0 + r_outer, yi = y0 + r_inner;
      src_off[j] = (xok && yo >= 0 && yo < p.H) ? (uint32_t)(((img * p.H + yo) * p.W + xx) * p.Cin + q * 4) * 4u : INVALID_PIXEL;
      src_off[4 + j] = (xok && yi >= 0 && yi < p.H) ? (uint32_t)(((img * p.H + yi) * p.W + xx) * p.Cin + q * 4) * 4u : INVALID_PIXEL;
    }
    v_dst = tl * KC + 4 * (q ^ ((tl >> 3) & 1));
    pos_outer = half ? 12 : 0;   // transform-domain row 3 / 0
    pos_inner = half ? 8 : 4;    // transform-domain row 2 / 1
    sgn = half ? -1.f : 1.f;
  } else {
    const int u_idx = tid - 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) src_off[i] = (uint32_t)(u_idx + 256 * i) * 16u;
  }

  float4 ld[8];
  const int nsteps = p.Cin / KC;
  int f_step = 0;
  auto fetch = [&]() {
    if (v_role) {
      const uint32_t choff = (uint32_t)f_step * (KC * 4u);
#pragma unroll
      for (int i = 0; i < 8; ++i) ld[i] = buf_load4(rX, src_off[i] + choff);
    } else {
      const uint32_t slab = (uint32_t)(f_step * p.tiles_n + tile_n) * (uint32_t)(OPER_FLOATS * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) ld[i] = buf_load4(rU, slab + src_off[i]);
    }
    ++f_step;
  };

  auto stage = [&](int buf) {
    float *Vd = lds + buf * STAGE_FLOATS;
    if (v_role) {
      // column mixing of B^T d B inside each of this lane's two patch rows
      float4 o[4], n[4];
      o[0] = f4sub(ld[0], ld[2]); o[1] = f4add(ld[1], ld[2]); o[2] = f4sub(ld[2], ld[1]); o[3] = f4sub(ld[1], ld[3]);
      n[0] = f4sub(ld[4], ld[6]); n[1] = f4add(ld[5], ld[6]); n[2] = f4sub(ld[6], ld[5]); n[3] = f4sub(ld[5], ld[7]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // row mixing: rows (0,1) live in one lane, rows (3,2) in its partner; each needs the partner's inner row
        const float4 r = make_float4(ror8(n[j].x), ror8(n[j].y), ror8(n[j].z), ror8(n[j].w));
        const float4 a = make_float4(sgn * (o[j].x - r.x), sgn * (o[j].y - r.y), sgn * (o[j].z - r.z), sgn * (o[j].w - r.w));
        const float4 b = make_float4(n[j].x + sgn * r.x, n[j].y + sgn * r.y, n[j].z + sgn * r.z, n[j].w + sgn * r.w);
        *reinterpret_cast<float4 *>(Vd + (pos_outer + j) * (64 * KC) + v_dst) = a;
        *reinterpret_cast<float4 *>(Vd + (pos_inner + j) * (64 * KC) + v_dst) = b;
      }
    } else {
      float *Ud = Vd + OPER_FLOATS + (tid - 256) * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(Ud + i * 1024) = ld[i];
    }
  };

  floatx16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int frag = fr * KC + 4 * (fh ^ ((fr >> 3) & 1));
  const int a_frag = (ph * 8 * 64 + tb * 32) * KC + frag;
  const int b_frag = OPER_FLOATS + (ph * 8 * 64 + cb * 32) * KC + frag;

  auto compute = [&](int buf, int pp0) {
    const float *S = lds + buf * STAGE_FLOATS;
#pragma unroll
    for (int pp = pp0; pp < pp0 + 4; ++pp) {
      const float4 a = *reinterpret_cast<const float4 *>(S + a_frag + pp * (64 * KC));
      const float4 b = *reinterpret_cast<const float4 *>(S + b_frag + pp * (64 * KC));
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
    compute(buf, 0);
    if (s + 1 < nsteps) stage(buf ^ 1);   // step s+1: loaded one step ago
    if (s + 2 < nsteps) fetch();          // step s+2: lands during the rest of this step and the next one's first half
    compute(buf, 4);
    __syncthreads();
  }

  // ---------------------------------------------------------------- inverse transform, halves exchanged through LDS
  // this wave holds m[i][j] for i in {2ph, 2ph+1}: acc[j] (first row), acc[4+j] (second row).
  //   t0j = m0j + m1j + m2j, t1j = m1j - m2j - m3j;  Y[a][0] = ta0 + ta1 + ta2, Y[a][1] = ta1 - ta2 - ta3
  // Register indices must be compile-time constants: both 8-register halves are formed with static indices and
  // the wave-uniform `ph` picks which one this wave finishes (`mine`) and which one it hands over (`other`).
  auto partial = [&](const float (&ma)[4], const float (&mb)[4]) -> float4 {
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ph == 0) { t0[j] = ma[j] + mb[j]; t1[j] = mb[j]; }
      else { t0[j] = ma[j]; t1[j] = -ma[j] - mb[j]; }
    }
    return make_float4(t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3], t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]);
  };
  auto gather = [&](int k, bool own, float (&ma)[4], float (&mb)[4]) {
    const bool hi = own ? (ph == 1) : (ph == 0);   // registers 8..15 belong to the ph = 1 wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ma[j] = hi ? acc[j][8 + k] : acc[j][k];
      mb[j] = hi ? acc[4 + j][8 + k] : acc[4 + j][k];
    }
  };
  float4 *xch = reinterpret_cast<float4 *>(lds);          // [pair 4][dst half 2][8][64 lanes] float4 = 64 KB
  const int pair = tb + 2 * cb;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float ma[4], mb[4];
    gather(k, false, ma, mb);                              // registers the partner wave finishes
    xch[((pair * 2 + (1 - ph)) * 8 + k) * 64 + lane] = partial(ma, mb);
  }
  __syncthreads();

  const idiff_epilogue &ep = p.ep;
  const bool want_stats = p.has_ep && ep.colstats != nullptr;
  const int n = n0 + cb * 32 + (lane & 31);
  const float bias = (p.has_ep && ep.bias) ? ep.bias[n] : 0.f;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int reg = 8 * ph + k;
    float ma[4], mb[4];
    gather(k, true, ma, mb);
    float4 y = partial(ma, mb);
    const float4 z = xch[((pair * 2 + ph) * 8 + k) * 64 + lane];
    y = f4add(y, z);
    const int tl = tb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int T = tile0 + tl;
    if (T >= p.total_tiles) continue;
    const int img = T / p.tiles_per_img, rem = T - img * p.tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int64_t m = ((int64_t)img * p.H + 2 * ty + (ab >> 1)) * p.W + 2 * tx + (ab & 1);
      float v = yv[ab] + bias;
      if (p.has_ep) {
        if (ep.rowbias) v += ep.rowbias[(m / ep.rows_per_group) * ep.ld_rowbias + n];
        v = idiff::act_apply(v, ep.act);
        if (ep.residual) v += ep.residual[m * ep.ld_residual + n];
        v *= ep.out_scale;
        if (ep.rowscale) v *= ep.rowscale[m / ep.rows_per_group];
      }
      p.out[m * p.Cout + n] = v;
      if (want_stats) { s1 += (double)v; s2 += (double)v * (double)v; }
    }
  }
  if (want_stats) {
    double *red = reinterpret_cast<double *>(lds + 16384);   // behind the 64 KB exchange area: [4][64][2]
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 32) {
      const int slot = ((tb * 2 + ph) * WG_COUT + cb * 32 + lane) * 2;
      red[slot] = s1; red[slot + 1] = s2;
    }
    __syncthreads();
    if (tid < WG_COUT) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * WG_COUT + tid) * 2]; b += red[(w * WG_COUT + tid) * 2 + 1]; }
      double *dst = ep.colstats + ((int64_t)tile_m * p.Cout + n0 + tid) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
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
      for (int j = 0; j < 4; ++j) dst[(int64_t)(i * 4 + j) * 64 * KC] = (float)v[j];
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
