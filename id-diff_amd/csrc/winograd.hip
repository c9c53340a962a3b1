// 3x3 / stride 1 / pad 1 convolutions of the score networks by Winograd's minimal filtering F(2x2, 3x3) on the
// fp32 matrix cores of gfx950: 16 multiplications per 2x2 output tile and (cin, cout) pair instead of 36, i.e.
// 2.25x fewer MFMA flops than the implicit GEMM of igemm.hip, all of it still exact-fp32 arithmetic
// (v_mfma_f32_32x32x2_f32; the transforms only add, subtract and halve).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch at (2ty-1, 2tx-1), Y: 2x2 outputs at (2ty, 2tx)
//
//   U = G g G^T is packed once per layer (idiff_winograd_pack_f32) as [Cin/8][Cout/64][16 slots][64 cout][8 cin].
//   V = B^T d B is formed by the loader on its way from HBM/L2 to LDS and never written to memory.
//   M_p = V_p U_p^T for the 16 positions p = (i, j) are 16 independent [tiles x Cin] x [Cin x Cout] contractions; a
//   workgroup owns 64 tiles x 64 output channels of ALL 16 of them, so the inverse transform A^T M A is local to the
//   workgroup and M never leaves the CU either.  Position (i, j) lives in slot 4j + i (LDS and packed U alike).
//
// Cost model this kernel is built on (measured with in-kernel stamps and PMC): a SIMD's time is
//   64 cycles x (fp32 MFMAs) + ~6.5 cycles x (every other instruction its waves issue) -- fp32 MFMAs run on the
//   same fp32 lanes as the VALU and nothing overlaps them -- so the design minimises non-MFMA instructions per MFMA.
//
// Workgroup (WG_TILES = 32): 256 threads = 4 waves = the 4 rows i of the 4x4 transform domain, 32 tiles x 64 channels,
//   64 KB of LDS, 216 registers -> TWO workgroups per CU, each one's prologue and tail running under the other's K loop
//   (WG_TILES = 64, 8 waves with 2 tile halves and one workgroup per CU, also builds: 2-4 % slower, 10 % on 4x4 maps, but
//   half as many workgroups stream each filter slab).  A wave holds
//   4 positions (i, 0..3) x [32 tiles x 64 channels] = 8 MFMA tiles = 128 accumulators, and per 8-channel step issues
//   32 MFMAs from 4 sixteen-byte LDS reads of V and 8 sixteen-byte global loads of U (each lane half feeds four
//   consecutive channels to four successive MFMAs, the order of the k-reduction being free; one V fragment serves both
//   32-channel halves).
// K loop: 8 input channels per step, two LDS stages of V (16 slots x [64 tiles][8] + 32 bytes of padding per slot), one
//   barrier per step, V loads two steps ahead in registers.  U never touches LDS: the packed slab is laid out so that the
//   32 channels x 32 bytes a half wave needs for one MFMA operand are contiguous, and each lane loads its own fragments
//   (8 x 16 bytes per step) straight into the registers the MFMAs read, re-requesting a position's pair for the next
//   step as soon as that position's MFMAs have issued (4 + 4 LDS writes and 8 LDS reads per wave-step fewer than the
//   staged form: +4 %).  The allocation is what the tail's exchange needs (64 KB at 32 tiles).
// Loader: every thread owns row r of the 4x4 patch of one (tile, 4-channel quad) unit -- tiles x 2 x 4 threads.  The four
//   rows of a unit are the four lanes of a DPP quad: column mixing is local (16 VALU), row mixing is one v_fmac_f32_dpp per value (quad_perm:[2,2,1,1], signs
//   -1,+1,-1,-1; row 3 comes out negated and the packed U carries the same sign).  Per-step address offsets ride in the
//   buffer instructions' scalar offset.  Half of the waves transform at the start of a step, the other half at the end,
//   so that the matrix pipe of a SIMD always has some wave's MFMAs.
// LDS rows are 32 bytes (8 channels); the two 16-byte halves of row r are swapped when bit 3 of r is set (in LDS for V,
//   in the packed slab for U), which makes the ds_read_b128 of a 32-row MFMA operand conflict-free, and the V slot pitch
//   of 520 floats makes the four rows' ds_write_b128 conflict-free (lane groups per instruction: MI355X_MICROARCH.md section LDS).
// Tail: each wave mixes its row over j (z_ib), the four rows meet in LDS ([4 rows][tiles][2][64 cout], over the dead
//   stage buffers), and every thread finishes float4 runs of 4 channels: Y[0][b] = z0b + z1b + z2b,
//   Y[1][b] = z1b - z2b - z3b, fused epilogue (bias, per-sample bias, activation, residual, scales, optional per-tile
//   column statistics), 16-byte stores -- the one-dword-per-lane store tail of the first version was store-issue bound
//   (29.8k of 131k cycles per workgroup at Cin = 128).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int WG_TILES = 32;                      // 32: two workgroups per CU (see the header); 64 also builds
constexpr int THREADS = WG_TILES * 8;
constexpr int NWAVES = THREADS / 64;
constexpr int WG_COUT = 64;
constexpr int KC = 8;
constexpr int NPOS = 16;
constexpr int OPER_FLOATS = NPOS * 64 * KC;       // one operand of one stage in HBM order: 8192 floats = 32 KB
constexpr int V_SLOT = WG_TILES * KC + 8;         // LDS floats per position slot of V (see above); = 8 mod 32
constexpr int V_FLOATS = NPOS * V_SLOT;           // 8320
constexpr int STAGE_FLOATS = V_FLOATS;            // U never touches LDS
constexpr int TAIL_FLOATS = 4 * WG_TILES * 2 * 64;     // the tail's exchange: [4 rows][tiles][2][64 cout]
constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(2 * STAGE_FLOATS > TAIL_FLOATS ? 2 * STAGE_FLOATS : TAIL_FLOATS);   // 64 KB
constexpr int64_t X_LIMIT = 0xFFFF0000ll;          // one buffer descriptor, with room for the invalid-pixel bias
constexpr int PEEL_MIN_WORKGROUPS = 4 * 512;      // four rounds of 2 workgroups on each of 256 CUs
constexpr uint32_t INVALID_PIXEL = 0xFFFF8000u;    // beyond any valid extent (the scalar step offset is not range-checked)

struct WinoParams {
  const float *x;
  const float *u;
  float *out;
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_y, tiles_per_img, total_tiles;
  int tx_shift, tpi_shift;        // log2 of tiles_x / tiles_per_img when both are powers of two, else -1 (division)
  int tiles_m, tiles_n, ngroup;   // ngroup: output-channel tiles scheduled together (tile_n innermost inside a group)
  uint32_t x_bytes, u_bytes, out_bytes, res_bytes;
  idiff_epilogue ep;
  int has_ep;
};

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// tile index -> (image, tile row, tile column); the maps of the score networks are powers of two, where this is two
// shifts and two masks instead of two ~20-instruction integer divisions
__device__ __forceinline__ void split_tile(const WinoParams &p, int T, int &img, int &ty, int &tx) {
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

// ---------------------------------------------------------------- tail (shared by the fp32 and the split-precision kernel)
// MODE 0 (fp32 kernel, 4 waves): acc[2 j + h] = M at position (wi, j) for this wave's 32 tiles x cout half h, in the 32x32
//   accumulator layout.
// MODE 1 (split kernel, 8 waves): wave (wi, jh) holds only positions j = 2 jh, 2 jh + 1: acc[2 jj + h].  The waves of
//   jh = 1 park their share of z_ib in LDS, the waves of jh = 0 add theirs on top, and waves 0 .. 3 (tid < THREADS) finish as
//   in mode 0; the other four only keep the barriers company.
template <int MODE>
__device__ __forceinline__ void wino_tail(const WinoParams &p, const floatx16 *acc, float *lds, const int tile0, const int n0,
                                          const int tile_m, const int tid, const int lane, const int wave, const int wi, const int tb) {
  const bool active = MODE == 0 || tid < THREADS;     // wave-uniform
  // This thread finishes 4 channels (n .. n+3) of tiles 2g and 2g+1: output pixels (2ty + a, 2tx + b).  Everything here
  // is VALU work that no MFMA hides (a wave's non-MFMA VALU instructions were split about evenly between the K loop and
  // prologue + tail at Cin = 256), so the tail is written to issue as few of them as it can: 32-bit byte offsets into
  // buffer descriptors of `out` and `residual` (one VGPR per tile; the four pixels of a tile differ by wave-uniform
  // amounts that ride in the scalar offset), bias + per-sample bias and out_scale x per-sample scale folded once per
  // tile, an out-of-range offset instead of a predicate for tiles beyond the end.  Every epilogue operand that comes
  // from memory is requested BEFORE the transform-domain exchange, so its latency hides under it (the residual read
  // issued at its point of use cost 0.3 ms of a 3.1 ms launch).
  const idiff_epilogue &ep = p.ep;
  const bool has_ep = p.has_ep != 0;
#if defined(IDIFF_WINO_STAMP) || defined(IDIFF_SPLIT_PHASES)
  const bool want_stats = false;          // epilogue.colstats carries the stamp buffer in this build
#else
  const bool want_stats = has_ep && ep.colstats != nullptr;
#endif
  const int cq = tid & 15, g = tid >> 4;
  const int n = n0 + 4 * cq;
  const bool per_image = ep.rows_per_group == p.H * p.W;   // the usual per-sample bias / scale: group = image
  const bool grouped = has_ep && !per_image && (ep.rowbias || ep.rowscale);   // any other row group: cold path below
  const bool has_res = has_ep && ep.residual != nullptr;
  const bool scaled = has_ep && (ep.out_scale != 1.f || ep.rowscale != nullptr);
  const int act = has_ep ? ep.act : (int)IDIFF_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)ep.residual, 0, (int)p.res_bytes, 0x00020000);
  const int ld_res = (int)ep.ld_residual;
  int m00[2];                  // row of `out` of pixel (0, 0) of the tile
  uint32_t ooff[2];            // its byte offset (+ this thread's channels), INVALID_PIXEL beyond the last tile
  float4 badd[2], res[2][4];
  float sc[2];
  if (active) {
    int T = tile0 + 2 * g, img, ty, tx;
    split_tile(p, T, img, ty, tx);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_ep && ep.bias) bias = *reinterpret_cast<const float4 *>(ep.bias + n);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bool ok = T < p.total_tiles;
      m00[t] = (img * p.H + 2 * ty) * p.W + 2 * tx;
      ooff[t] = ok ? ((uint32_t)m00[t] * (uint32_t)p.Cout + (uint32_t)n) * 4u : INVALID_PIXEL;
      badd[t] = bias;
      sc[t] = has_ep ? ep.out_scale : 1.f;
      if (has_ep && per_image && ok) {
        if (ep.rowbias) badd[t] = f4add(bias, *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)img * ep.ld_rowbias + n));
        if (ep.rowscale) sc[t] *= ep.rowscale[img];
      }
      if (has_res) {
        const uint32_t roff = ok ? ((uint32_t)m00[t] * (uint32_t)ld_res + (uint32_t)n) * 4u : INVALID_PIXEL;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
          res[t][ab] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)roff, ((ab >> 1) * p.W + (ab & 1)) * ld_res * 4, 0));
      }
      ++T;
      if (++tx == p.tiles_x) { tx = 0; if (++ty == p.tiles_y) { ty = 0; ++img; } }
    }
  }

  // z_ib for this wave's row i: z_i0 = m_i0 + m_i1 + m_i2, z_i1 = m_i1 - m_i2 - m_i3 -> LDS [i][tile][b][cout]
  {
    float *zp = lds + ((wi * WG_TILES + tb * 32 + 4 * (lane >> 5)) * 2) * 64 + (lane & 31);
    if (MODE == 0) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float m0 = acc[ch][reg], m1 = acc[2 + ch][reg], m2 = acc[4 + ch][reg], m3 = acc[6 + ch][reg];
          const int trow = (reg & 3) + 8 * (reg >> 2);        // + 4 * (lane >> 5): row of the 32x32 accumulator tile
          zp[(trow * 2) * 64 + ch * 32] = m0 + m1 + m2;
          zp[(trow * 2 + 1) * 64 + ch * 32] = m1 - m2 - m3;
        }
    } else {
      const bool second = wave >= 4;                           // jh = 1: positions (wi, 2), (wi, 3)
      if (second) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const float m2 = acc[ch][reg], m3 = acc[2 + ch][reg];
            const int trow = (reg & 3) + 8 * (reg >> 2);
            zp[(trow * 2) * 64 + ch * 32] = m2;
            zp[(trow * 2 + 1) * 64 + ch * 32] = -m2 - m3;
          }
      }
      __syncthreads();
      if (!second) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const float m0 = acc[ch][reg], m1 = acc[2 + ch][reg];
            const int trow = (reg & 3) + 8 * (reg >> 2);
            zp[(trow * 2) * 64 + ch * 32] = (m0 + m1) + zp[(trow * 2) * 64 + ch * 32];
            zp[(trow * 2 + 1) * 64 + ch * 32] = m1 + zp[(trow * 2 + 1) * 64 + ch * 32];
          }
      }
    }
  }
  __syncthreads();

  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (active) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float *zr = lds + ((2 * g + t) * 2 + b) * 64 + 4 * cq;
      const float4 z0 = *reinterpret_cast<const float4 *>(zr);
      const float4 z1 = *reinterpret_cast<const float4 *>(zr + 1 * WG_TILES * 2 * 64);
      const float4 z2 = *reinterpret_cast<const float4 *>(zr + 2 * WG_TILES * 2 * 64);
      const float4 z3 = *reinterpret_cast<const float4 *>(zr + 3 * WG_TILES * 2 * 64);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float v[4];
        if (a == 0) { v[0] = z0.x + z1.x + z2.x; v[1] = z0.y + z1.y + z2.y; v[2] = z0.z + z1.z + z2.z; v[3] = z0.w + z1.w + z2.w; }
        else { v[0] = z1.x - z2.x - z3.x; v[1] = z1.y - z2.y - z3.y; v[2] = z1.z - z2.z - z3.z; v[3] = z1.w - z2.w - z3.w; }
        v[0] += badd[t].x; v[1] += badd[t].y; v[2] += badd[t].z; v[3] += badd[t].w;
        float rsv = 1.f;
        if (grouped) {
          const int64_t grp = ((int64_t)m00[t] + a * p.W + b) / ep.rows_per_group;
          if (ep.rowbias) {
            const float4 rbv = *reinterpret_cast<const float4 *>(ep.rowbias + grp * ep.ld_rowbias + n);
            v[0] += rbv.x; v[1] += rbv.y; v[2] += rbv.z; v[3] += rbv.w;
          }
          if (ep.rowscale) rsv = ep.rowscale[grp];
        }
        if (act != IDIFF_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = idiff::act_apply(v[e], act);
        }
        if (has_res) {
          const float4 r4 = res[t][2 * a + b];
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (scaled) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= sc[t];
          if (grouped) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= rsv;
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, make_float4(v[0], v[1], v[2], v[3])), rO, (int)ooff[t],
                                               (a * p.W + b) * p.Cout * 4, 0);
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += (double)v[e]; s2[e] += (double)v[e] * (double)v[e]; }
        }
      }
    }
  }
  }
  if (want_stats && p.tiles_per_img < WG_TILES) {
    // Maps smaller than a workgroup (8x8: 16 tiles, 4x4: 4 tiles per sample): the workgroup holds WG_TILES / tiles_per_img
    // whole samples and writes one slot per (sample, channel), layout [B][1][Cout][2].  A thread's two tiles lie in one
    // sample (tiles_per_img is even); the 16 (tile pair, 4-channel) partial sums of the workgroup meet in LDS and
    // 64 x samples threads add up the tile pairs of their sample in a fixed order.
    __syncthreads();                                          // every z has been read: the area is free again
    double *red = reinterpret_cast<double *>(lds);           // [16 tile pairs][64 channels][2]
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[((g * WG_COUT) + 4 * cq + e) * 2] = s1[e];
        red[((g * WG_COUT) + 4 * cq + e) * 2 + 1] = s2[e];
      }
    }
    __syncthreads();
    const int pairs = p.tiles_per_img >> 1, samples = WG_TILES / p.tiles_per_img;
    for (int o = active ? tid : samples * WG_COUT; o < samples * WG_COUT; o += THREADS) {
      const int smp = o / WG_COUT, ch = o - smp * WG_COUT;
      const int64_t img = (int64_t)tile_m * samples + smp;
      if (img >= p.B) continue;
      double a = 0.0, b = 0.0;
      for (int k = 0; k < pairs; ++k) { a += red[((smp * pairs + k) * WG_COUT + ch) * 2]; b += red[((smp * pairs + k) * WG_COUT + ch) * 2 + 1]; }
      double *dst = ep.colstats + (img * p.Cout + n0 + ch) * 2;
      dst[0] = a; dst[1] = b;
    }
  } else if (want_stats) {
    // lanes l, l+16, l+32, l+48 hold the same four channels; then one slot per (wave, channel)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += __shfl_xor(s1[e], 16, 64); s1[e] += __shfl_xor(s1[e], 32, 64);
      s2[e] += __shfl_xor(s2[e], 16, 64); s2[e] += __shfl_xor(s2[e], 32, 64);
    }
    __syncthreads();                                          // every z has been read: the area is free again
    double *red = reinterpret_cast<double *>(lds);           // [NWAVES][64][2]
    if (lane < 16 && active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[((wave * WG_COUT) + 4 * cq + e) * 2] = s1[e];
        red[((wave * WG_COUT) + 4 * cq + e) * 2 + 1] = s2[e];
      }
    }
    __syncthreads();
    if (tid < WG_COUT) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) { a += red[(w * WG_COUT + tid) * 2]; b += red[(w * WG_COUT + tid) * 2 + 1]; }
      double *dst = ep.colstats + ((int64_t)tile_m * p.Cout + n0 + tid) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
}

// PEEL: step 0 is peeled and its MFMAs take C = 0 as an inline constant, so no accumulator is initialised (1-3 % on the
// large maps); launches of only a few rounds of workgroups (the 4x4 maps) measured 7 % faster with the plain loop.
template <bool PEEL>
__global__ void __launch_bounds__(THREADS, 2)   // two waves per SIMD (256 VGPRs): two 256-thread workgroups per CU
winograd_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef IDIFF_WINO_STAMP   // diagnostic build only (scripts/wino_clock.py): the clock the chip holds inside this kernel
  const uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  // groups of `ngroup` output-channel tiles: inside a group tile_n is innermost (the workgroups of an XCD share the input
  // patches of a run of tile_m and stream only ngroup filter slabs), the groups follow one another
  const int per_group = p.tiles_m * p.ngroup;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int tile_n = grp * p.ngroup + in_grp % p.ngroup, tile_m = in_grp / p.ngroup;
  const int tile0 = tile_m * WG_TILES, n0 = tile_n * WG_COUT;
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index in an SGPR: roles, operand halves and buffer descriptors stay provably wave-uniform (no waterfall
  // loops around the buffer loads, scalar branches)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, tb = wave >> 2;   // transform-domain row, 32-tile half

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);

  // ---------------------------------------------------------------- loader state
  const bool early = wave < NWAVES / 2;    // waves w and w+4 share a SIMD: one transforms while the other feeds the matrix pipe
  uint32_t v_src[4], u_src[4];
  int v_dst;
  float sgn;
  {
    const int r = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
    const int T = tile0 + tl;
    const bool tv = T < p.total_tiles;
    const int TT = tv ? T : 0;
    int img, ty, tx;
    split_tile(p, TT, img, ty, tx);
    const int y = 2 * ty - 1 + r, x0 = 2 * tx - 1;
    const bool yok = tv && y >= 0 && y < p.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x0 + j;
      v_src[j] = (yok && xx >= 0 && xx < p.W) ? (uint32_t)(((img * p.H + y) * p.W + xx) * p.Cin + q * 4) * 4u : INVALID_PIXEL;
    }
    v_dst = r * V_SLOT + tl * KC + 4 * (q ^ ((tl >> 3) & 1));
    sgn = (r == 1) ? 1.f : -1.f;
    // U fragments go straight from the slab to the MFMA operand registers of the lane that uses them (32 channels x 32
    // bytes contiguous per half wave): no LDS round trip for the filter bank
    const int fr_ = lane & 31, fh_ = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) u_src[j] = (uint32_t)(((4 * j + wi) * 64 + fr_) * 32 + 16 * (fh_ ^ ((fr_ >> 3) & 1)));
  }

  float4 ldv[4], bfr[4][2];
  const int nsteps = p.Cin / KC;
  int f_step = 0;
  // the per-step offsets are wave-uniform: they ride in the buffer instruction's scalar offset (not part of the range
  // check of a raw buffer, so an out-of-range pixel stays out of range) instead of costing a VALU add per load
  auto fetch = [&]() {
    const int choff = min(f_step, nsteps - 1) * (KC * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) ldv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)v_src[j], choff, 0));
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
  };

  floatx16 acc[8];   // [j][channel half]; PEEL: first written by the MFMAs of step 0 (the 2 x 128 v_mov_b32 of an explicit
                     // initialisation were 11 % of a wave's non-MFMA VALU work)

  const int fr = lane & 31, fh = lane >> 5;
  const int frag = fr * KC + 4 * (fh ^ ((fr >> 3) & 1));
  const int a_frag = wi * V_SLOT + tb * 32 * KC + frag;       // slot of position (wi, j): 4j + wi

  auto load_b = [&](int j, int step) {
    const int slab = (step * p.tiles_n + tile_n) * (OPER_FLOATS * 4);
    bfr[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_src[j], slab, 0));
    bfr[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_src[j] + 32 * 32, slab, 0));
  };
  // The four positions of a step one after the other, their V fragments through TWO registers sets in rotation: the fragment
  // of position j + 2 is read from LDS right after the eight MFMAs of position j have been issued, so the eight MFMAs of
  // position j + 1 (512 cycles) cover its latency.  (Before, the compiler interleaved positions 0 / 1 and then read the
  // fragments of positions 2 and 3 each immediately in front of its use: two exposed LDS round trips per step.)
  auto mfma8 = [&](int j, const float4 a, auto first) {
    const float4 b0 = bfr[j][0], b1 = bfr[j][1];
    if constexpr (decltype(first)::value) {
      const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[2 * j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, zero, 0, 0, 0);
      acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, zero, 0, 0, 0);
    } else {
      acc[2 * j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[2 * j], 0, 0, 0);
      acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[2 * j + 1], 0, 0, 0);
    }
    acc[2 * j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[2 * j], 0, 0, 0);
    acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[2 * j + 1], 0, 0, 0);
    acc[2 * j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[2 * j], 0, 0, 0);
    acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[2 * j + 1], 0, 0, 0);
    acc[2 * j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[2 * j], 0, 0, 0);
    acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[2 * j + 1], 0, 0, 0);
  };
  auto compute = [&](int buf, int s, auto first, auto before_last) {
    const float *S = lds + buf * STAGE_FLOATS + a_frag;
    const int snext = min(s + 1, nsteps - 1);
    float4 f0 = *reinterpret_cast<const float4 *>(S), f1 = *reinterpret_cast<const float4 *>(S + 4 * V_SLOT);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(0, f0, first);
    // the registers are free once these MFMAs have read them; the last step re-requests its own slab (clamped, not
    // skipped: no branch per position, and the scalar offset of a raw buffer is not range-checked, so it must stay valid)
    load_b(0, snext);
    f0 = *reinterpret_cast<const float4 *>(S + 8 * V_SLOT);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1, f1, first);
    load_b(1, snext);
    f1 = *reinterpret_cast<const float4 *>(S + 12 * V_SLOT);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(2, f0, first);
    load_b(2, snext);
    __builtin_amdgcn_sched_barrier(0);
    before_last();                      // the late role writes its stage HERE: its LDS writes then land under eight MFMAs
    __builtin_amdgcn_sched_barrier(0);  // instead of in front of the barrier
    mfma8(3, f1, first);
    load_b(3, snext);
  };
  // The role is a compile-time constant of the loop a wave runs, and writing the next stage / requesting the one after are
  // unconditional (the last two steps re-request the last stage and write a stage nobody reads): the compiler's wait counts
  // are static, and with `if (early)` / `if (s + 2 < nsteps)` inside one shared loop the paths merged into `s_waitcnt
  // vmcnt(1) / vmcnt(0)` in front of the input transform -- an early wave waited for the filter loads of the CURRENT step
  // (issued at the end of the previous one) before starting the work that is meant to cover them.
  auto step = [&](int s, auto first, auto is_early) {
    const int buf = s & 1;
    if constexpr (decltype(is_early)::value) {
      stage(buf ^ 1); fetch();
      compute(buf, s, first, [] {});
    } else {
      compute(buf, s, first, [&] { stage(buf ^ 1); fetch(); });
    }
    __syncthreads();
  };

#pragma unroll
  for (int j = 0; j < 4; ++j) load_b(j, 0);
  fetch();
  stage(0);
  fetch();
  __syncthreads();

  auto run = [&](auto is_early) {
    if constexpr (PEEL) {
      step(0, std::true_type(), is_early);
      for (int s = 1; s < nsteps; ++s) step(s, std::false_type(), is_early);
    } else {
      for (int s = 0; s < nsteps; ++s) step(s, std::false_type(), is_early);
    }
  };
  if constexpr (!PEEL) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  if (early) run(std::true_type()); else run(std::false_type());

  wino_tail<0>(p, acc, lds, tile0, n0, tile_m, tid, lane, wave, wi, tb);
#ifdef IDIFF_WINO_STAMP
  const idiff_epilogue &ep = p.ep;
  const bool has_ep = p.has_ep != 0;
  if (has_ep && ep.colstats && tid == 0) {
    // shader-clock ticks and 100 MHz ticks of this workgroup's lifetime, in a buffer nothing else reads
    uint64_t *st = reinterpret_cast<uint64_t *>(ep.colstats) + 2 * (int64_t)blockIdx.x;
    st[0] = __builtin_amdgcn_s_memtime() - stamp_t0;
    st[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Split-precision form: the same algorithm with the 16 position-wise contractions M_p = V_p U_p^T on the bf16 matrix cores.
// Every fp32 operand element is cut EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits, x = x1 + x2 + x3) and a product
// is the six partial products of weight >= 2^-16, fp32 accumulation: what is left out is < 2^-23 |a b| per product, one fp32
// rounding (igemm.hip, split4; measured 1.7e-7 against 2.0e-7 for the fp32 MFMA chain).  v_mfma_f32_32x32x16_bf16 retires
// 16 channels in 32 cycles where the fp32 form needs 8 x 64: six of them are 2.7x cheaper, and they leave the vector ALU
// free for the transforms and the splitting.
//   * U is split once, in idiff_winograd_pack_split_f32: per 16-channel step, output tile, position and piece a block of
//     [64 cout][16 cin] bf16 (32 bytes per output channel: the fragment of lane half 0, then that of lane half 1).
//   * V stays fp32 in LDS, in the layout of the fp32 kernel: a 16-channel step is two of its 8-channel stages.  Lane
//     (tile fr, half fh) reads the eight channels of stage fh for its tile (two 16-byte reads, the conflict-free one first:
//     lane half 1 therefore gets its two quads in the opposite order, and the packed U matches) and splits them in
//     registers: each V element is consumed by exactly one wave, so splitting on the reading side costs the same vector
//     work as on the writing side and keeps the LDS traffic at 4 bytes per element.
//   * 8 waves = 4 rows of the transform domain x 2 pairs of positions, 32 tiles x 64 output channels per workgroup, one
//     workgroup per CU: a wave holds 2 positions x 2 channel halves (64 accumulators) and its 12 U fragments a step ahead
//     (48 registers), which fits TWO waves per SIMD (256 registers each) -- with one wave per SIMD (the first version: 4
//     waves x 4 positions, 128 accumulators + 96 registers of U) nothing covered a wait and the kernel ran at 0.8x of the
//     fp32 one, its matrix pipe busy 24 % of the time.  One LDS stage of V: fragments go to registers, a barrier, then
//     the stage is rewritten under this step's matrix instructions.
//   * What bounds it (round 3, scripts/wino_split_phases.py + PMC): a 16-channel step moves 96 KB of U and 32 KB of x from
//     L2 into the CU, 128 KB in ~3.9k cycles = 33 B/clk/CU = 13.6 TB/s chip-wide -- the rate the vector-memory path of a
//     CU sustains from L2 (MI355X_MICROARCH.md, 'Indexed rows': 66-73 GB/s per CU), against 1536 cycles of matrix work per
//     SIMD.  Re-ordering the step (M1 held back across the barrier, roles in separate loops) only moved the waiting from
//     one phase to another.  Fewer bytes per product need 64 tiles per workgroup (U read once for twice the tiles), i.e.
//     128 accumulators per wave next to ~130 other registers at two waves per SIMD: does not fit.  Hence opt-in, not default.
constexpr int SPLIT_KC = 16;
constexpr size_t SPLIT_LDS_BYTES = sizeof(float) * (size_t)(4 * STAGE_FLOATS > TAIL_FLOATS ? 4 * STAGE_FLOATS : TAIL_FLOATS);   // 133 KB: two stages of 2 x 8 channels
constexpr int SPLIT_POS_BYTES = 3 * 64 * SPLIT_KC * 2;          // one position of one slab: 3 pieces x [64 cout][16 cin] bf16
constexpr int SPLIT_SLAB_BYTES = NPOS * SPLIT_POS_BYTES;        // 98304

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4 lo, const float4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uintx4 q1, q2, q3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t t1[2], t2[2], t3[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      t1[e] = __float_as_uint(x[2 * j + e]) & 0xffff0000u;
      const float r1 = x[2 * j + e] - __uint_as_float(t1[e]);
      t2[e] = __float_as_uint(r1) & 0xffff0000u;
      t3[e] = __float_as_uint(r1 - __uint_as_float(t2[e]));
    }
    q1[j] = (t1[0] >> 16) | t1[1];
    q2[j] = (t2[0] >> 16) | t2[1];
    q3[j] = (t3[0] >> 16) | (t3[1] & 0xffff0000u);
  }
  p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2); p3 = __builtin_bit_cast(bf16x8, q3);
}

constexpr int SPLIT_THREADS = 512;

__global__ void __launch_bounds__(SPLIT_THREADS, 1)
winograd_split_kernel(const WinoParams p) {
  static_assert(WG_TILES == 32 && THREADS == 256, "the split kernel is written for 32 tiles: 8 waves = 4 rows x 2 position pairs");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int per_group = p.tiles_m * p.ngroup;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int tile_n = grp * p.ngroup + in_grp % p.ngroup, tile_m = in_grp / p.ngroup;
  const int tile0 = tile_m * WG_TILES, n0 = tile_n * WG_COUT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, jh = wave >> 2;          // transform-domain row; positions j = 2 jh, 2 jh + 1

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);

  // ---------------------------------------------------------------- loader state: thread = row r of the 4x4 patch of one
  // (tile, 8-channel stage h, 4-channel quad q) unit; the four rows of a unit are the four lanes of a DPP quad
  uint32_t v_src[4], u_src[2];
  int v_dst;
  float sgn;
  {
    const int r = tid & 3, q = (tid >> 2) & 1, h = (tid >> 3) & 1, tl = tid >> 4;
    const int T = tile0 + tl;
    const bool tv = T < p.total_tiles;
    const int TT = tv ? T : 0;
    int img, ty, tx;
    split_tile(p, TT, img, ty, tx);
    const int y = 2 * ty - 1 + r, x0 = 2 * tx - 1;
    const bool yok = tv && y >= 0 && y < p.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x0 + j;
      v_src[j] = (yok && xx >= 0 && xx < p.W) ? (uint32_t)(((img * p.H + y) * p.W + xx) * p.Cin + h * KC + q * 4) * 4u : INVALID_PIXEL;
    }
    v_dst = h * STAGE_FLOATS + r * V_SLOT + tl * KC + 4 * (q ^ ((tl >> 3) & 1));
    sgn = (r == 1) ? 1.f : -1.f;
    const int fr_ = lane & 31, fh_ = lane >> 5;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) u_src[jj] = (uint32_t)((4 * (2 * jh + jj) + wi) * SPLIT_POS_BYTES + fr_ * (SPLIT_KC * 2) + 16 * fh_);
  }

  float4 ldv[2][4];                 // the inputs of the next two steps (fetched two steps ahead: HBM latency under load)
  uintx4 bfr[2][2][3];              // [position jj][cout half][piece]
  const int nsteps = p.Cin / SPLIT_KC;
  int f_step = 0;
  auto fetch = [&](float4 (&lv)[4]) {
    // beyond the last step the LAST step is fetched again (never consumed): the scalar offset of a raw buffer load is not
    // range-checked, so it must not run past the tensor
    const int choff = min(f_step, nsteps - 1) * (SPLIT_KC * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) lv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)v_src[j], choff, 0));
    ++f_step;
  };
  auto stage = [&](int buf, const float4 (&lv)[4]) {
    float *Vd = lds + buf * 2 * STAGE_FLOATS + v_dst;
    float4 c[4];
    c[0] = f4sub(lv[0], lv[2]); c[1] = f4add(lv[1], lv[2]); c[2] = f4sub(lv[2], lv[1]); c[3] = f4sub(lv[1], lv[3]);
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
  };

  floatx16 acc[4];                  // [position jj][cout half]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5, sw = (fr >> 3) & 1;
  // stage fh, row fr; first the half the fp32 kernel's lane would read (conflict-free), then the other one
  const int a_first = fh * STAGE_FLOATS + (4 * 2 * jh + wi) * V_SLOT + fr * KC + 4 * (fh ^ sw);
  const int a_second = fh * STAGE_FLOATS + (4 * 2 * jh + wi) * V_SLOT + fr * KC + 4 * (fh ^ sw ^ 1);

  auto load_b = [&](int jj, int step) {
    const int slab = (step * p.tiles_n + tile_n) * SPLIT_SLAB_BYTES;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        bfr[jj][hb][q] = __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_src[jj] + q * (64 * SPLIT_KC * 2) + hb * (32 * SPLIT_KC * 2), slab, 0);
  };

  load_b(0, 0);
  load_b(1, 0);
  fetch(ldv[0]);
  stage(0, ldv[0]);
  fetch(ldv[0]);                    // step 1
  fetch(ldv[1]);                    // step 2
  __syncthreads();

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // the six partial products, smallest first
  const bool early = jh == 0;
#ifdef IDIFF_SPLIT_PHASES   // diagnostic build only (scripts/wino_split_phases.py): shader-clock ticks per phase of a step
  uint32_t ph[6] = {0, 0, 0, 0, 0, 0};
  uint64_t ph_last = __builtin_amdgcn_s_memtime();
  const uint64_t ph_first = ph_last;
#define IDIFF_PH(k) { __builtin_amdgcn_sched_barrier(0); const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[k] += (uint32_t)(t_ - ph_last); ph_last = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define IDIFF_PH(k)
#endif
  // step s: the V of step s + 1 is in ldv[s & 1] (requested two steps ago); `lv` is that set
  auto step = [&](int s, float4 (&lv)[4]) {
    const int buf = s & 1;
    const float *S = lds + buf * 2 * STAGE_FLOATS;
    float4 alo[2], ahi[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      alo[jj] = *reinterpret_cast<const float4 *>(S + a_first + 4 * jj * V_SLOT);
      ahi[jj] = *reinterpret_cast<const float4 *>(S + a_second + 4 * jj * V_SLOT);
    }
    // waves w and w + 4 share a SIMD: one writes the other stage (vector work) while its partner's matrix instructions run
    if (early) { stage(buf ^ 1, lv); fetch(lv); }
    IDIFF_PH(0)
    const int snext = min(s + 1, nsteps - 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bf16x8 a[3];
      split8(alo[jj], ahi[jj], a[0], a[1], a[2]);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          acc[2 * jj + hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]], __builtin_bit_cast(bf16x8, bfr[jj][hb][PB[t]]), acc[2 * jj + hb], 0, 0, 0);
      load_b(jj, snext);
      IDIFF_PH(1 + jj)
    }
    if (!early) { stage(buf ^ 1, lv); fetch(lv); }
    IDIFF_PH(3)
    __syncthreads();
    IDIFF_PH(4)
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(s, ldv[0]);
    if (s + 1 < nsteps) step(s + 1, ldv[1]);
  }
#ifdef IDIFF_SPLIT_PHASES
  const uint64_t ph_loop = __builtin_amdgcn_s_memtime();
#endif
  wino_tail<1>(p, acc, lds, tile0, n0, tile_m, tid, lane, wave, wi, 0);
#ifdef IDIFF_SPLIT_PHASES
  if (p.has_ep && p.ep.colstats && lane == 0) {
    // a buffer nothing else reads: [workgroup][wave][8]
    uint32_t *st = reinterpret_cast<uint32_t *>(p.ep.colstats) + ((int64_t)blockIdx.x * 8 + wave) * 8;
    for (int k = 0; k < 5; ++k) st[k] = ph[k];
    st[5] = (uint32_t)(ph_loop - ph_first);
    st[6] = (uint32_t)(__builtin_amdgcn_s_memtime() - ph_loop);
    st[7] = (uint32_t)nsteps;
  }
#endif
}
#undef IDIFF_PH

// U = G g G^T (fp64, rounded once to fp32 exactly as idiff_winograd_pack_f32 does), then cut into the three bf16 pieces and
// laid out for winograd_split_kernel: [Cin / 16][Cout / 64][16 positions][3 pieces][64 cout][16 cin], where the 16 channels
// of a row are the eight of lane half 0 (channels 0 .. 7 of the step) followed by the eight of lane half 1 in ITS reading
// order (channels 12 .. 15, then 8 .. 11).
__global__ void winograd_pack_split_kernel(const float *wt, unsigned short *u, int Cin, int Cout) {
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
    const int s = cin / SPLIT_KC, c16 = cin % SPLIT_KC, nt = cout / WG_COUT, co = cout % WG_COUT;
    const int half = c16 >> 3, e = c16 & 7;
    const int pos_in_row = half * 8 + (half ? ((e >> 2) ^ 1) * 4 + (e & 3) : e);
    unsigned short *dst = u + ((int64_t)(s * (Cout / WG_COUT) + nt) * NPOS * 3 * 64 + co) * SPLIT_KC + pos_in_row;
    for (int i = 0; i < 4; ++i) {
      const double r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
      const double v[4] = {r0, 0.5 * (r0 + r1 + r2), 0.5 * (r0 - r1 + r2), r2};
      const double sign = i == 3 ? -1.0 : 1.0;     // the kernel's input transform produces row 3 negated
      for (int j = 0; j < 4; ++j) {
        const float x = (float)(sign * v[j]);
        const uint32_t t1 = __float_as_uint(x) & 0xffff0000u;
        const float r1f = x - __uint_as_float(t1);
        const uint32_t t2 = __float_as_uint(r1f) & 0xffff0000u;
        const uint32_t t3 = __float_as_uint(r1f - __uint_as_float(t2));
        unsigned short *d = dst + (int64_t)(j * 4 + i) * 3 * 64 * SPLIT_KC;
        d[0] = (unsigned short)(t1 >> 16);
        d[64 * SPLIT_KC] = (unsigned short)(t2 >> 16);
        d[2 * 64 * SPLIT_KC] = (unsigned short)(t3 >> 16);
      }
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
  if (idiff::option(idiff::OPT_NO_WINOGRAD)) return 0;
  return geometry_ok(B, H, W, Cin, Cout) ? 1 : 0;
}

IDIFF_API int idiff_conv2d_winograd_colstats_split(int B, int H, int W, int Cin, int Cout) {
  if (!idiff_conv2d_winograd_ok(B, H, W, Cin, Cout) || idiff::option(idiff::OPT_NO_COLSTATS)) return 0;
  if ((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) * 4 >= X_LIMIT) return 0;
  const int tpi = (H / 2) * (W / 2);
  if (tpi % WG_TILES == 0) return tpi / WG_TILES;
  // maps smaller than a workgroup's 32 tiles (8x8, 4x4): whole samples per workgroup, one slot per sample
  return (tpi >= 2 && tpi % 2 == 0 && WG_TILES % tpi == 0) ? 1 : 0;
}

IDIFF_API int64_t idiff_winograd_weight_floats(int Cin, int Cout) { return (int64_t)16 * Cin * Cout; }

// The split-precision form (winograd_split_kernel): Cin % 16 == 0 on top of the fp32 kernel's conditions.  OPT-IN
// (IDIFF_WINO_SPLIT): correct to the same bars, but measured 0.8x the fp32 kernel on the NCSN++ layers.
IDIFF_API int idiff_conv2d_winograd_split_ok(int B, int H, int W, int Cin, int Cout) {
  if (!idiff::option(idiff::OPT_WINO_SPLIT) || idiff::option(idiff::OPT_NO_SPLIT) || idiff::option(idiff::OPT_NO_WINOGRAD)) return 0;
  if (!geometry_ok(B, H, W, Cin, Cout) || Cin % SPLIT_KC) return 0;
  if ((int64_t)24 * Cin * Cout * 4 >= X_LIMIT) return 0;
  return 1;
}

IDIFF_API int64_t idiff_winograd_split_weight_floats(int Cin, int Cout) { return (int64_t)24 * Cin * Cout; }   // 3 bf16 per weight

IDIFF_API int idiff_winograd_pack_split_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % SPLIT_KC || Cout % WG_COUT)
    return fail("winograd_pack_split: Cin must be a multiple of %d and Cout of %d (got %d, %d)", SPLIT_KC, WG_COUT, Cin, Cout);
  if (!wt || !u) return fail("winograd_pack_split: null pointer");
  const int64_t total = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(winograd_pack_split_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<unsigned short *>(u), Cin, Cout);
  return launch_status("winograd_pack_split");
}

IDIFF_API int idiff_winograd_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % KC || Cout % WG_COUT)
    return fail("winograd_pack: Cin must be a multiple of %d and Cout of %d (got %d, %d)", KC, WG_COUT, Cin, Cout);
  if (!wt || !u) return fail("winograd_pack: null pointer");
  const int64_t total = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(winograd_pack_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt, u, Cin, Cout);
  return launch_status("winograd_pack");
}

namespace {
int conv2d_winograd_impl(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                         const idiff_epilogue *ep, void *stream, bool split);
}
IDIFF_API int idiff_conv2d_winograd_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                        const idiff_epilogue *ep, void *stream) {
  return conv2d_winograd_impl(x, u, out, B, H, W, Cin, Cout, ep, stream, false);
}
// u: the bank of idiff_winograd_pack_split_f32
IDIFF_API int idiff_conv2d_winograd_split_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                              const idiff_epilogue *ep, void *stream) {
  if (Cin % SPLIT_KC) return idiff::fail("conv2d_winograd_split: Cin must be a multiple of %d (got %d)", SPLIT_KC, Cin);
  return conv2d_winograd_impl(x, u, out, B, H, W, Cin, Cout, ep, stream, true);
}
namespace {
int conv2d_winograd_impl(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                         const idiff_epilogue *ep, void *stream, bool split) {
  using namespace idiff;
  if (B == 0) return 0;
  if (!geometry_ok(B, H, W, Cin, Cout))
    return fail("conv2d_winograd: geometry B=%d H=%d W=%d Cin=%d Cout=%d not supported (ask idiff_conv2d_winograd_ok)", B, H, W, Cin, Cout);
  if (!x || !u || !out) return fail("conv2d_winograd: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)out & 15)) return fail("conv2d_winograd: x, u and out must be 16-byte aligned");
  if (ep && ep->residual && (((uintptr_t)ep->residual & 15) || ep->ld_residual % 4))
    return fail("conv2d_winograd: residual must be 16-byte aligned with a row pitch that is a multiple of 4");
  const int64_t x_bytes = (int64_t)B * H * W * Cin * 4, out_bytes = (int64_t)B * H * W * Cout * 4;
  if (ep && ep->residual && (ep->ld_residual < Cout || ep->ld_residual > 0x7fffffff / 4))
    return fail("conv2d_winograd: ld_residual %lld is not a row pitch for %d channels", (long long)ep->ld_residual, Cout);
  const int64_t res_bytes = (ep && ep->residual) ? (int64_t)B * H * W * ep->ld_residual * 4 : 0;
  // the kernel addresses x, out and residual through one buffer descriptor each (32-bit byte offsets)
  if (x_bytes >= X_LIMIT || out_bytes >= X_LIMIT || res_bytes >= X_LIMIT) {
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
    int rc = conv2d_winograd_impl(x, u, out, b_lo, H, W, Cin, Cout, ep ? &lo : nullptr, stream, split);
    if (rc) return rc;
    return conv2d_winograd_impl(x + m_lo * Cin, u, out + m_lo * Cout, B - b_lo, H, W, Cin, Cout, ep ? &hi : nullptr, stream, split);
  }
  WinoParams p = {};
  p.x = x; p.u = u; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tiles_x = W / 2; p.tiles_y = H / 2; p.tiles_per_img = (H / 2) * (W / 2); p.total_tiles = B * p.tiles_per_img;
  p.tx_shift = p.tpi_shift = -1;
  if ((p.tiles_x & (p.tiles_x - 1)) == 0 && (p.tiles_per_img & (p.tiles_per_img - 1)) == 0) {
    p.tx_shift = __builtin_ctz((unsigned)p.tiles_x); p.tpi_shift = __builtin_ctz((unsigned)p.tiles_per_img);
  }
  p.tiles_m = ceil_div(p.total_tiles, WG_TILES); p.tiles_n = Cout / WG_COUT;
  {
    // two output-channel tiles per scheduling group: the workgroups of an XCD then stream two filter slabs instead of
    // four (1.5-3 % on the Cout = 256 layers; the slabs of four tiles plus the live input lines overflow the 4 MB L2)
    const int want = option_value(OPT_WINO_NGROUP);
    p.ngroup = (want > 0 && p.tiles_n % want == 0) ? want : ((p.tiles_n > 2 && p.tiles_n % 2 == 0) ? 2 : p.tiles_n);
  }
  p.x_bytes = (uint32_t)x_bytes; p.u_bytes = (uint32_t)((int64_t)(split ? 24 : 16) * Cin * Cout * 4);
  p.out_bytes = (uint32_t)out_bytes; p.res_bytes = (uint32_t)res_bytes;
  if (ep) {
    p.ep = *ep; p.has_ep = 1;
    if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1;
    if (ep->colstats && p.tiles_per_img % WG_TILES && !(p.tiles_per_img >= 2 && p.tiles_per_img % 2 == 0 && WG_TILES % p.tiles_per_img == 0))
      return fail("conv2d_winograd: colstats needs whole workgroups per sample or whole samples per workgroup "
                  "(ask idiff_conv2d_winograd_colstats_split)");
  } else {
    p.has_ep = 0; p.ep.rows_per_group = 1; p.ep.out_scale = 1.f;
  }
  const void *fns[2] = {reinterpret_cast<const void *>(winograd_kernel<false>), reinterpret_cast<const void *>(winograd_kernel<true>)};
  {
    static AttrGuard guard;
    if (int rc = set_dynamic_lds_once(guard, fns, 2, (int)LDS_BYTES, "conv2d_winograd")) return rc;
  }
  const int nwg = p.tiles_m * p.tiles_n;
  if (split) {
    static AttrGuard sguard;
    const void *fn = reinterpret_cast<const void *>(winograd_split_kernel);
    if (int rc = set_dynamic_lds_once(sguard, &fn, 1, (int)SPLIT_LDS_BYTES, "conv2d_winograd_split")) return rc;
    hipLaunchKernelGGL(winograd_split_kernel, dim3(nwg), dim3(SPLIT_THREADS), SPLIT_LDS_BYTES, (hipStream_t)stream, p);
    return launch_status("conv2d_winograd_split");
  }
  if (nwg >= PEEL_MIN_WORKGROUPS)
    hipLaunchKernelGGL(winograd_kernel<true>, dim3(nwg), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(winograd_kernel<false>, dim3(nwg), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, p);
  return launch_status("conv2d_winograd");
}
}  // namespace
