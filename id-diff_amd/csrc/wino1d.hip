// 3x3 / stride 1 / pad 1 convolutions by Winograd's F(4, 3) ALONG THE IMAGE ROWS ONLY -- the three filter rows stay a direct sum -- with
// the 18 contractions per 1 x 4 output tile on the fp16 matrix cores of gfx950 (fp16 pairs, as winograd43h.hip).
//
// Why a second pair kernel.  winograd43h_kernel (F(4x4, 3x3): 2.25 multiplications per output) is not bound by its matrix instructions
// (22 % of the SIMD cycles) but by what its workgroup has to bring in per K step: 36 positions x 64 channels of U (147 KB) for 512 output
// pixels, because 36 accumulators per tile and channel leave room for no more pixels in the register file, plus 6 x 6 patches that repeat
// their neighbours' pixels -- 2,040 - 2,230 128-byte lines from L2 per workgroup and step, at the 16 TB/s the L2 -> L1 path delivers chip-wide
// (profiles/r05_wino43h_l1_requests.txt: the time follows the line count).  Transforming along x only costs twice the matrix work
// (4.5 multiplications per output) and HALF the lines: six positions x three filter rows of U are 74 KB per step, shared by all 512 pixels
// of the workgroup, and the input is read as whole image rows -- every pixel once per step, no horizontal halo at all, one halo row above or
// below the block.  The matrix cores have the room: 108 instructions per wave and step are 3,456 cycles of a step that took 9,300.
//
//   out[r][4t + a] = sum_i A^T[a][i] sum_ky ( V[r + ky - 1][t][i] . U[i][ky] ),   V[r][t][.] = B^T x[r][4t - 1 .. 4t + 4],   U[.][ky] = G g[ky][.]
//   (same points 0, +-2/3, +-3/2, inf and the same B^T, G, A^T as winograd43_shared.h, used in one dimension: the transform's gain is
//   5.4 where the 2-D form has 29.3, so |x| may reach 12,000 before a pair overflows, and the rounding of the transform enters once)
//
// Workgroup: 256 threads = 4 waves, ONE PER SIMD (up to 512 registers each), 512 output pixels x 64 output channels:
//   RB = 512 / W whole image rows (W = 64: eight rows, 32: half a 32 x 32 image, 16: two images, 8: eight, 4: thirty-two), 128 row-tiles of 1 x 4 pixels, six positions each.
//   Wave (h, ph): output channels 32 h .. 32 h + 31, positions 3 ph .. 3 ph + 2, all four groups of 32 row-tiles: 12 accumulator tiles of
//   32 x 32 = 192 registers.  Every U fragment (position, filter row, channel half: 2 KB of pairs) is loaded by exactly one wave, straight from
//   L2 into registers, three (position, filter row) combinations ahead.
// K step: 16 input channels.  Per wave and step nine combinations (position, filter row) x four groups x three matrix instructions
//   (hi hi, hi lo, lo hi); the V operand of (group, filter row ky) is the stage's rows 4 g + ky - 1 .. 4 g + ky + 2: the same LDS image read at a
//   row's distance, no copy per filter row.  Rows that lie outside the image of an output row (a block of small maps holds several images) are
//   read from an all-zero entry instead.  The step is written as 108 GAPS -- one matrix instruction, a hand-dealt slice of the other work, a
//   scheduling fence (see step()): one wave per SIMD issues in order, and nothing else hides behind the matrix instructions.
// Loader: lane lm = tid % 8 holds the channel PAIR 2 lm, 2 lm + 1 (8-byte loads; eight consecutive lanes read the 64 contiguous bytes of a
//   pixel), segment set tid / 8: four row-tiles = 16 pixels + the neighbour left and right of one image row (W = 32: half a row, W = 16: a row,
//   W = 8: two rows) -- 36 registers that hold the NEXT step's pixels while this step is contracted; threads tid / 8 < 2 W / 4 additionally one
//   row-tile (six pixels) of the halo row above or below the block.  A row-tile's staging is 13 groups of <= 5 vector instructions: t = B^T d
//   for both channels (12 plain fmas each), then per position one dword of hi parts (v_cvt_pk_f16_f32 of the two channels: no lane exchange)
//   and one of lo parts (v - hi by v_fma_mix_f32, exact), two positions of a plane per ds_write2st64_b32, then the requests for the step after next.
// Stage (LDS): [6 positions][(RB + 2) W / 4 + 1 entries, rounded up to eight][64 B], entry = (row slot, tile), the last one all zero; entry e lives
//   in slot r1_slot(e) (pairs of every second group of four swapped: the four loader threads of a write pass then fall on four bank quarters),
//   its four 16-byte pieces (plane, channel half) at piece ^ ((slot >> 2) & 3): a ds_read_b128 of 32 consecutive entries is conflict-free from
//   any even first entry (checked by enumeration over the instruction's lane groups).  Two stages, 64 KB apart (the other stage: one XOR).
// U (global): [Cin/16][Cout/64][18 slots = position * 3 + filter row][2 planes][64 cout][16 cin] fp16 + 4 floats of header (the factor that
//   undoes the power-of-two scaling first).
// Tail: two rounds (groups {0, 1}, {2, 3}); every wave parks its three positions' accumulators as they are ([64 row-tiles][6 positions][64
//   channels] fp32), and thread (4 channels, 16 consecutive pixels) forms A^T m from six 16-byte reads per row-tile, applies the epilogue of
//   idiff_conv2d_nhwc_f32 -- its switches as compile-time constants -- and stores 16 bytes per pixel.
// Measured (profiles/r05_wino1d_*.txt, HISTORY_r05.md 5): 1.03 - 1.23x winograd43h_kernel by shape, matrix cores 52 % busy, 1,120 lines from L2 per
// workgroup and step.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// LDS by its 32-bit address: the K loop keeps complete addresses in registers (stage, entry, piece; positions are immediate offsets), so no
// instruction re-adds the allocation's base per access
typedef __attribute__((address_space(3))) char r1_lds_char;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"           // (the host pass sees 64-bit pointers; LDS pointers of the device pass are 32 bits wide)
typedef __attribute__((address_space(3))) halfx8 r1_lds_halfx8;
typedef __attribute__((address_space(3))) uint32_t r1_lds_u32;
__device__ __forceinline__ halfx8 r1_lds_read16(uint32_t at) { return *(const r1_lds_halfx8 *)at; }
__device__ __forceinline__ void r1_lds_write4(uint32_t at, uint32_t v) { *(r1_lds_u32 *)at = v; }
#pragma clang diagnostic pop

constexpr double R1_A = 2.0 / 3.0, R1_B = 1.5;                     // the interpolation points of winograd43_shared.h
constexpr float R1_a = (float)R1_A, R1_b = (float)R1_B, R1_a2 = (float)(R1_A * R1_A), R1_b2 = (float)(R1_B * R1_B),
                R1_a3 = (float)(R1_A * R1_A * R1_A), R1_b3 = (float)(R1_B * R1_B * R1_B), R1_ab2 = (float)(R1_A * R1_A + R1_B * R1_B);
constexpr int R1_THREADS = 256;
constexpr int R1_COUT = 64;
constexpr int R1_KC = 16;
constexpr int R1_PIXELS = 512;                                     // output pixels of a workgroup
constexpr int R1_RT = R1_PIXELS / 4;                               // its row-tiles
constexpr int R1_NSLOT = 18;                                       // (position, filter row)
constexpr int R1_SLOT_BYTES = 2 * 64 * R1_KC * 2;                  // 4096
constexpr int R1_PLANE_BYTES = 64 * R1_KC * 2;                     // 2048
constexpr int R1_Z_BYTES = R1_RT * 4 * R1_COUT * 4;                // the tail's exchange: 131,072
constexpr size_t R1_LDS_BYTES = R1_Z_BYTES;
constexpr uint32_t R1_STAGE_STRIDE = 0x10000;                      // the two stages 64 KB apart: the other stage is one XOR away
constexpr uint32_t R1_INVALID = 0xFFFF8000u;                       // beyond any valid extent (the scalar offset is not range-checked)
constexpr int64_t R1_X_LIMIT = 0xFFFF0000ll;
#ifndef IDIFF_W1D_BRING
#define IDIFF_W1D_BRING 3
#endif
#ifndef IDIFF_W1D_STORE_AUX
#define IDIFF_W1D_STORE_AUX 0      // cache policy bits of the output stores (1 = sc0, 2 = nt, 16 = sc1)
#endif
constexpr int R1_BRING = IDIFF_W1D_BRING;                          // combinations of U requested ahead (register sets of 8)
static_assert(9 % R1_BRING == 0, "the ring of U registers must divide the nine combinations of a step");

// Where entry e of a stage lives: the entries of every second group of four trade places in pairs.  The 32 lanes of one pass of a ds_write_b32
// are four loader threads' dwords of the entries e, e + 4, e + 8, e + 12 (32 bytes each); at e * 64 they would fall on two of the four
// 32-byte quarters of the 128 bytes the banks span (a two-way conflict on every stage write, SQ_LDS_BANK_CONFLICT = 24 % of the LDS cycles);
// the trade alternates the entries' parity and the four land on four quarters.  A fragment read covers both members of every pair, so its
// conflict-free pattern is unchanged.
__device__ __forceinline__ int r1_slot(int e) { return e ^ ((e >> 2) & 1); }

template <int W> struct R1Geo {
  static constexpr int TPR = W / 4;                                // row-tiles per image row
  static constexpr int RB = R1_PIXELS / W;                         // image rows per workgroup
  static constexpr int EZ = (RB + 2) * TPR;                        // stage entries: row slot 0 = the row above the block, RB + 1 = the row below, then the zero entry
  static constexpr int E = (EZ + 1 + 7) & ~7;                      // (whole groups of eight: r1_slot() stays inside)
  static constexpr int POS_BYTES = E * 64;
  static constexpr int STAGE_BYTES = 6 * POS_BYTES;
  static_assert(RB * TPR == R1_RT, "128 row-tiles per workgroup");
  static_assert(STAGE_BYTES <= R1_STAGE_STRIDE, "a stage inside its half of the LDS allocation");
  static_assert(5 * POS_BYTES < 65536, "positions are reached by the LDS instructions' immediate offset");
  static_assert(POS_BYTES % 256 == 0, "two positions of one entry: one ds_write2st64_b32");
};

struct Wino1dParams {
  const float *x;
  const float *u;
  float *out;
  int B, H, W, Cin, Cout;
  int rows_total, blocks_m, tiles_n, ngroup;
  uint32_t x_bytes, u_bytes, out_bytes, res_bytes;
  idiff_epilogue ep;
  int has_ep;
  float c_nb2, c_na2, c_nab2, c_a, c_b;            // the transform's constants as kernel arguments (SGPR operands of plain fmas)
#ifdef IDIFF_W1D_STAMP
  uint64_t *stamps;                                // diagnostic build (scripts/wino1d_stamps.py): 8 ticks of 100 MHz per workgroup
#endif
};

template <int W>
__global__ void __launch_bounds__(R1_THREADS)
wino1d_kernel(const Wino1dParams p) {
  using G = R1Geo<W>;
  constexpr int TPR = G::TPR, RB = G::RB, POS = G::POS_BYTES, BRING = R1_BRING;
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef IDIFF_W1D_STAMP
  uint64_t st[8];
#define IDIFF_W1D_T(k) { __builtin_amdgcn_sched_barrier(0); st[k] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define IDIFF_W1D_T(k)
#endif
  IDIFF_W1D_T(0)
  char *const ldsb = reinterpret_cast<char *>(lds);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(r1_lds_char *)ldsb;   // the allocation's LDS address
  const int nwg = p.blocks_m * p.tiles_n;
  int bid = blockIdx.x;
  {   // consecutive workgroups of the launch order land on different XCDs: give every XCD a contiguous range of the (row block, cout tile) order
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int per_group = p.blocks_m * p.ngroup;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int tile_n = grp * p.ngroup + in_grp % p.ngroup, tile_m = in_grp / p.ngroup;
  const int row0 = tile_m * RB, n0 = tile_n * R1_COUT;            // first global image row (sample * H + y) of the block
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wh = wave & 1, ph = wave >> 1;

  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);
  const uint32_t cin4 = (uint32_t)p.Cin * 4u;
  const int roww = W * (int)cin4;                                  // bytes per image row
  const int nsteps = p.Cin / R1_KC;

  // ---------------------------------------------------------------- loader: thread = (channel pair, row segment)
  // Lane lm holds channels 2 lm and 2 lm + 1 (8-byte loads: eight consecutive lanes read the 64 contiguous bytes of a pixel), so the two halves of
  // every stage dword are in ONE lane and a value pair costs 2 conversions + 2 mixed fmas -- no exchange with a neighbouring lane, no selects.
  // Segment set lk: W = 64 / 32: a quarter / half of a row (16 pixels and their two neighbours), W = 16: row lk, W = 8 / 4: two / four whole rows -- four
  // row-tiles and 16 pixels (+ the neighbours left and right) per thread and step; lk < 2 W / 4 additionally one row-tile of a halo row.
  constexpr int TS = TPR < 4 ? TPR : 4;                            // row-tiles per segment
  constexpr int NSEG = 4 / TS;                                     // row segments of a thread
  constexpr int NPX = 4 * TS + 2;                                  // its pixels, the one to the left and the one to the right included
  const int lm = tid & 7, lk = tid >> 3;
  // base one pixel to the LEFT: offset v + j * cin4 is the segment's pixel j - 1 ... (the range check covers the vector offset only)
  const __amdgpu_buffer_rsrc_t rXm = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x - p.Cin), 0, (int)(p.x_bytes + cin4), 0x00020000);
  uint32_t v_main, v_halo;
  bool m_l, m_r, h_c0, h_c5;
  int e_halo;
  {
#ifdef IDIFF_W1D_DIAG_SAME_X       // timing-only build: every workgroup reads the first block's pixels (L2 hits instead of HBM misses)
    const int rho = (W >= 32 ? lk / (W / 16) : NSEG * lk);
#else
    const int rho = row0 + (W >= 32 ? lk / (W / 16) : NSEG * lk);
#endif
    const int xf = W >= 32 ? 16 * (lk % (W / 16)) : 0;            // first pixel of my first tile
    v_main = rho < p.rows_total ? (uint32_t)rho * (uint32_t)roww + (uint32_t)xf * cin4 + (uint32_t)lm * 8u : R1_INVALID;
    m_l = xf > 0; m_r = xf + 4 * TS < W;                           // the neighbours are pixels of the row (else the zero padding)
    const bool part = lk < 2 * TPR, top = lk < TPR;
    const int tx = top ? lk : lk - TPR;
    const int hr = top ? row0 - 1 : row0 + RB;                     // the halo row: inside the same image as the block's first / last row?
    const bool valid = part && (top ? (row0 % p.H != 0) : ((row0 + RB) % p.H != 0 && row0 + RB < p.rows_total));
    v_halo = valid ? (uint32_t)hr * (uint32_t)roww + (uint32_t)(4 * tx) * cin4 + (uint32_t)lm * 8u : R1_INVALID;
    h_c0 = tx > 0; h_c5 = tx + 1 < TPR;
    e_halo = part ? (top ? tx : (RB + 1) * TPR + tx) : G::EZ;      // threads without a halo tile transform zeros into the zero entry
  }
  // does any lane of this wave hold a halo tile that is inside the image?  (W = 32: wave 0 the row above, wave 1 the row below; else wave 0 both)
  const bool halo_wave = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_ballot_w64(v_halo != R1_INVALID)) != 0
                         || __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(v_halo != R1_INVALID) >> 32)) != 0;
  const int e_main0 = TPR + 4 * lk;                                // my four row-tiles' entries: consecutive
  f2 px[NSEG][NPX];                                                // (channel 2 lm, channel 2 lm + 1) per pixel
  f2 hp[6];
#pragma unroll
  for (int g = 0; g < NSEG; ++g)
#pragma unroll
    for (int j = 0; j < NPX; ++j) px[g][j] = f2{0.f, 0.f};
  uint32_t invalid_s = R1_INVALID;
  // pixels [j0, j1] of segment g for K step `step` (clamped: the last stages request their own step again, nobody reads it)
  auto fetch_px = [&](int g, int j0, int j1, int step) __attribute__((always_inline)) {
    const int choff = min(step, nsteps - 1) * (R1_KC * 4);
    asm volatile("" : "+s"(invalid_s));
#pragma unroll
    for (int j = j0; j <= j1; ++j) {
      if (W < 32 && (j == 0 || j == NPX - 1)) continue;            // always the padding
      const uint32_t vo = j == 0 ? (m_l ? v_main : invalid_s) : (j == NPX - 1 ? (m_r ? v_main : invalid_s) : v_main);
      px[g][j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rXm, (int)vo, choff + g * roww + j * (int)cin4, 0));
    }
  };
  auto fetch_hp = [&](int j0, int j1, int step) __attribute__((always_inline)) {
    const int choff = min(step, nsteps - 1) * (R1_KC * 4);
    asm volatile("" : "+s"(invalid_s));
#pragma unroll
    for (int j = j0; j <= j1; ++j) {
      const uint32_t vo = j == 0 ? (h_c0 ? v_halo : invalid_s) : (j == 5 ? (h_c5 ? v_halo : invalid_s) : v_halo);
      hp[j] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rXm, (int)vo, choff + j * (int)cin4, 0));
    }
  };
  // where my hi dword of row-tile i (4: the halo tile) goes in stage 0, position 0 (LDS address); the lo dword: bit 5 flipped; stage 1: bit 16
  uint32_t wtab[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int e = r1_slot(i < 4 ? e_main0 + i : e_halo);
    wtab[i] = lds0 + (uint32_t)(e * 64 + (((lm >> 2) ^ ((e >> 2) & 3)) << 4) + (lm & 3) * 4);
  }
  const float k_nb2 = p.c_nb2, k_na2 = p.c_na2, k_nab2 = p.c_nab2, k_a = p.c_a, k_b = p.c_b;
#ifdef IDIFF_W1D_DIAG_NO_VWRITE
  uint32_t diag_sink = 0;
#endif
  // The staging of ONE row-tile (two channels) in 13 groups of at most five vector instructions, the units the K loop deals over its gaps:
  //   0 - 4   t = B^T d for both channels (12 operations each, the arithmetic of winograd43_shared.h's f4_bt)
  //   5 - 10  position grp - 5: the two channels' values -> one dword of hi parts, one of lo parts (v - hi in one mixed-precision fma each), stored
  //   11, 12  the pixels no later tile reads are requested for K step `next`
  float t_pe[2], t_po[2], t_re[2], t_ro[2], t_s04[2], t_s15[2], tv[2][6];
  uint32_t w_hi = 0, w_lo = 0, p_hi = 0, p_lo = 0;
  auto tile_group = [&](auto ii, auto gg, uint32_t wbase, int next) __attribute__((always_inline)) {
    constexpr int i = decltype(ii)::value, grp = decltype(gg)::value;
    constexpr int sg = i < 4 ? i / TS : 0, t = i < 4 ? i % TS : 0;
    auto d = [&](int c, int j) __attribute__((always_inline)) { return i < 4 ? px[sg][4 * t + j][c] : hp[j][c]; };
    auto head = [&](int c, int part) __attribute__((always_inline)) {          // the twelve operations of channel c in five parts
      if (part == 0) { t_pe[c] = fmaf(k_nb2, d(c, 2), d(c, 4)); t_po[c] = fmaf(k_nb2, d(c, 1), d(c, 3)); t_re[c] = fmaf(k_na2, d(c, 2), d(c, 4)); }
      if (part == 1) { t_ro[c] = fmaf(k_na2, d(c, 1), d(c, 3)); t_s04[c] = d(c, 0) + d(c, 4); t_s15[c] = d(c, 1) + d(c, 5); }
      if (part == 2) { tv[c][0] = fmaf(k_nab2, d(c, 2), t_s04[c]); tv[c][5] = fmaf(k_nab2, d(c, 3), t_s15[c]); }
      if (part == 3) { tv[c][1] = fmaf(k_a, t_po[c], t_pe[c]); tv[c][2] = fmaf(-k_a, t_po[c], t_pe[c]); }
      if (part == 4) { tv[c][3] = fmaf(k_b, t_ro[c], t_re[c]); tv[c][4] = fmaf(-k_b, t_ro[c], t_re[c]); }
    };
    if (grp == 0) { w_hi = wtab[i] ^ wbase; w_lo = w_hi ^ 32u; head(0, 0); head(0, 1); }     // (wbase: 0 or the stage stride)
    if (grp == 1) { head(0, 2); head(0, 3); }
    if (grp == 2) { head(0, 4); head(1, 0); }
    if (grp == 3) { head(1, 1); head(1, 2); }
    if (grp == 4) { head(1, 3); head(1, 4); }
    if (grp >= 5 && grp <= 10) {
      // positions in pairs (0, 1), (2, 3), (4, 5): the first group of a pair cuts its position, the second cuts the other and stores both -- two
      // dwords of one plane at one base and a distance that is a multiple of 256 bytes go out as ONE ds_write2st64_b32
      constexpr int pos = grp >= 5 && grp <= 10 ? grp - 5 : 0;
      const f2 vv = {tv[0][pos], tv[1][pos]};
      const uint32_t xh = __builtin_bit_cast(uint32_t, __builtin_convertvector(vv, h2));
      f2 rest;                                                      // v - hi in one mixed-precision instruction per channel (exact)
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest.x) : "v"(xh), "v"(vv.x));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest.y) : "v"(xh), "v"(vv.y));
      const uint32_t xl = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, h2));
      if ((pos & 1) == 0) { p_hi = xh; p_lo = xl; }
      else {
#ifdef IDIFF_W1D_DIAG_NO_VWRITE    // timing-only build: the pairs are summed into one register instead of written
        diag_sink += xh + xl + p_hi + p_lo;
#else
        r1_lds_write4(w_hi + (pos - 1) * POS, p_hi); r1_lds_write4(w_hi + pos * POS, xh);
        r1_lds_write4(w_lo + (pos - 1) * POS, p_lo); r1_lds_write4(w_lo + pos * POS, xl);
#endif
      }
    }
#ifdef IDIFF_W1D_DIAG_NO_XLOAD     // timing-only build: the input is loaded for the first two steps only
    if (next <= 1)
#endif
    if (grp >= 11) {
      if (i < 4) {
        constexpr int j0 = 4 * t, j1 = t == TS - 1 ? 4 * t + 5 : 4 * t + 3, mid = (j0 + j1) / 2;
        if (grp == 11) fetch_px(sg, j0, mid, next); else fetch_px(sg, mid + 1, j1, next);
      } else {
        if (grp == 11) fetch_hp(0, 2, next); else fetch_hp(3, 5, next);
      }
    }
  };
  // all of one tile at once (the first step's stage, before the loop)
  auto tile_all = [&](auto ii, uint32_t wbase, int next) __attribute__((always_inline)) {
    tile_group(ii, std::integral_constant<int, 0>(), wbase, next); tile_group(ii, std::integral_constant<int, 1>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 2>(), wbase, next); tile_group(ii, std::integral_constant<int, 3>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 4>(), wbase, next); tile_group(ii, std::integral_constant<int, 5>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 6>(), wbase, next); tile_group(ii, std::integral_constant<int, 7>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 8>(), wbase, next); tile_group(ii, std::integral_constant<int, 9>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 10>(), wbase, next); tile_group(ii, std::integral_constant<int, 11>(), wbase, next);
    tile_group(ii, std::integral_constant<int, 12>(), wbase, next);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---------------------------------------------------------------- contraction
  floatx16 acc[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][i][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  // V fragment of (group g, filter row ky): the entry of row-tile 32 g + fr one row up / at / one row down, or the zero entry where that row
  // is outside the output row's image; piece fh of plane 0, plane 1 lies at the address with bit 5 flipped
  uint32_t ra[4][3];                                  // LDS addresses (plane 0; plane 1: bit 5 flipped) in the stage being read: toggled to the other stage after every step
  {
    const int y0 = row0 % p.H;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = 32 * g + fr, r = m / TPR, tx = m % TPR;
      const int y = (y0 + r) % p.H;
      const bool inside = row0 + r < p.rows_total;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const bool ok = inside && (ky == 1 || (ky == 0 ? y > 0 : y + 1 < p.H));
        const int e = r1_slot(ok ? (r + ky) * TPR + tx : G::EZ);
        ra[g][ky] = lds0 + (uint32_t)(ph * 3 * POS + e * 64 + ((fh ^ ((e >> 2) & 3)) << 4));
      }
    }
  }
  const uint32_t u_lane = (uint32_t)((wh * 32 + fr) * (R1_KC * 2) + fh * 16);
  halfx8 bh[BRING], bl[BRING];
  auto load_b = [&](int c, int step) __attribute__((always_inline)) {
    const int soff = ((step * p.tiles_n + tile_n) * R1_NSLOT + 9 * ph + c) * R1_SLOT_BYTES;     // slot = (3 ph + c / 3) * 3 + c % 3
    bh[c % BRING] = __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_lane, soff, 0));
    bl[c % BRING] = __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_lane + R1_PLANE_BYTES, soff, 0));
  };
  // A step: nine combinations c = (position 3 ph + c / 3, filter row c % 3), each twelve matrix instructions on four V fragments read one
  // combination ahead and a U fragment requested BRING ahead.  One wave per SIMD issues in order: vector work placed behind the matrix
  // instructions would start when the last of them has been issued.  So the step is written as 108 GAPS -- one matrix instruction, then at
  // most ~five other instructions that fit in the 32 cycles it runs, then a scheduling fence -- and everything else is dealt over them by hand:
  //   gaps 1, 2, 6, 9 of a combination: two LDS reads each of the NEXT combination's V fragments (plane 0 first: its products come first)
  //   gap 4: the two U loads
  //   gaps 0, 3, 5, 7, 8, 10, 11 (and 1, 2, 6, 9 of the last combination): the next staging group -- 4 x 13 for my row-tiles, then 13 for the halo tile.
  auto step = [&](int s, auto last) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last)::value;
    const uint32_t wbase = (s & 1) ? 0u : R1_STAGE_STRIDE;         // the stage being written: the one not read
    halfx8 ah[2][4], al[2][4];
    auto rd_hi = [&](int c, int g) __attribute__((always_inline)) { ah[c & 1][g] = r1_lds_read16(ra[g][c % 3] + (c / 3) * POS); };
    auto rd_lo = [&](int c, int g) __attribute__((always_inline)) { al[c & 1][g] = r1_lds_read16((ra[g][c % 3] ^ 32u) + (c / 3) * POS); };
#pragma unroll
    for (int g = 0; g < 4; ++g) { rd_hi(0, g); rd_lo(0, g); }
    auto slice = [&](auto cc, auto kk) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value, k = decltype(kk)::value;
      if (c + 1 < 9) {
        if (k == 1) { rd_hi(c + 1, 0); rd_hi(c + 1, 1); }
        if (k == 2) { rd_hi(c + 1, 2); rd_hi(c + 1, 3); }
        if (k == 6) { rd_lo(c + 1, 0); rd_lo(c + 1, 1); }
        if (k == 9) { rd_lo(c + 1, 2); rd_lo(c + 1, 3); }
      }
      if (k == 4) {
#ifdef IDIFF_W1D_DIAG_NO_BLOAD     // timing-only build: U is loaded for the first step only
        if (s == 0) { if (c + BRING < 9) load_b(c + BRING, s); }
#else
        if (c + BRING < 9) load_b(c + BRING, s); else if (!LAST) load_b(c + BRING - 9, s + 1);
#endif
      }
#ifndef IDIFF_W1D_DIAG_NO_STAGE    // timing-only build: no transform, no stage writes, no input loads after the prologue
      if (!LAST) {
        // the staging slot of this gap: 7 per combination, 11 in the last one
        constexpr int idx7 = k == 0 ? 0 : k == 3 ? 1 : k == 5 ? 2 : k == 7 ? 3 : k == 8 ? 4 : k == 10 ? 5 : k == 11 ? 6 : -1;
        constexpr int idx11 = k < 4 ? k : (k == 4 ? -1 : k - 1);
        constexpr int slot = c < 8 ? (idx7 < 0 ? -1 : 7 * c + idx7) : (idx11 < 0 ? -1 : 56 + idx11);
        if (slot >= 0 && slot < 52)
          tile_group(std::integral_constant<int, (slot >= 0 && slot < 52 ? slot / 13 : 0)>(), std::integral_constant<int, (slot >= 0 ? slot % 13 : 0)>(), wbase, s + 2);
        // the halo tile: only in the waves that hold one (a wave-uniform branch; the others' gaps stay empty)
        if (slot >= 52 && slot < 65 && halo_wave)
          tile_group(std::integral_constant<int, 4>(), std::integral_constant<int, (slot >= 52 ? slot - 52 : 0)>(), wbase, s + 2);
      }
#endif
    };
    auto combo = [&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      constexpr int pi = c / 3;
      const halfx8 yh = bh[c % BRING], yl = bl[c % BRING];
      auto gap = [&](auto kk) __attribute__((always_inline)) {
        constexpr int k = decltype(kk)::value, g = k & 3, pr = k >> 2;       // products: plane 0 x plane 0, plane 0 x plane 1, plane 1 x plane 0
#ifdef IDIFF_W1D_DIAG_NO_MFMA      // timing-only build: the operands are consumed by one vector instruction instead
        acc[g][pi][0] += (float)(pr == 2 ? al[c & 1][g] : ah[c & 1][g])[0] + (float)(pr == 1 ? yl : yh)[0];
#else
        acc[g][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[c & 1][g] : ah[c & 1][g], pr == 1 ? yl : yh, acc[g][pi], 0, 0, 0);
#endif
        slice(cc, kk);
        __builtin_amdgcn_sched_barrier(0);
      };
      gap(std::integral_constant<int, 0>()); gap(std::integral_constant<int, 1>()); gap(std::integral_constant<int, 2>());
      gap(std::integral_constant<int, 3>()); gap(std::integral_constant<int, 4>()); gap(std::integral_constant<int, 5>());
      gap(std::integral_constant<int, 6>()); gap(std::integral_constant<int, 7>()); gap(std::integral_constant<int, 8>());
      gap(std::integral_constant<int, 9>()); gap(std::integral_constant<int, 10>()); gap(std::integral_constant<int, 11>());
    };
    __builtin_amdgcn_sched_barrier(0);
    combo(std::integral_constant<int, 0>()); combo(std::integral_constant<int, 1>()); combo(std::integral_constant<int, 2>());
    combo(std::integral_constant<int, 3>()); combo(std::integral_constant<int, 4>()); combo(std::integral_constant<int, 5>());
    combo(std::integral_constant<int, 6>()); combo(std::integral_constant<int, 7>()); combo(std::integral_constant<int, 8>());
#ifndef IDIFF_W1D_DIAG_NO_BARRIER  // timing-only build (a race by construction): what the step barrier and the waves' skew at it cost
    __syncthreads();
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) ra[g][ky] ^= R1_STAGE_STRIDE;
  };

  // the zero entry of both stages
  if (tid < 192) {
    const int b = tid / 96, rem = tid - 96 * b;
    *reinterpret_cast<uint32_t *>(ldsb + b * R1_STAGE_STRIDE + (rem >> 4) * POS + r1_slot(G::EZ) * 64 + (rem & 15) * 4) = 0u;
  }
#pragma unroll
  for (int c = 0; c < BRING; ++c) load_b(c, 0);
#pragma unroll
  for (int g = 0; g < NSEG; ++g) fetch_px(g, 0, NPX - 1, 0);
  fetch_hp(0, 5, 0);
#ifdef IDIFF_W1D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  IDIFF_W1D_T(1)                                     // first operands arrived
  tile_all(std::integral_constant<int, 0>(), 0u, 1); tile_all(std::integral_constant<int, 1>(), 0u, 1); tile_all(std::integral_constant<int, 2>(), 0u, 1);
  tile_all(std::integral_constant<int, 3>(), 0u, 1);
  if (halo_wave) tile_all(std::integral_constant<int, 4>(), 0u, 1);
  __syncthreads();
  IDIFF_W1D_T(2)
  {
    int s = 0;                                        // at least two steps (Cin >= 32, checked by the launcher)
    do step(s, std::false_type()); while (++s + 1 < nsteps);
  }
  step(nsteps - 1, std::true_type());
  IDIFF_W1D_T(3)
#ifdef IDIFF_W1D_DIAG_NO_VWRITE
  if (diag_sink == 12345u) ldsb[tid] = 1;
#endif

  // ---------------------------------------------------------------- tail
  // Two rounds, groups {0, 1} then {2, 3} (64 row-tiles each).  A wave PARKS its three positions' accumulators as they stand --
  // zr[row-tile][position][cout], 98,304 B -- and after a barrier thread (4 channels, 4 row-tiles = 16 consecutive pixels) reads the six
  // positions of a row-tile (six 16-byte reads), forms A^T m, applies the epilogue and stores 16 bytes per pixel.  (The first form had each
  // wave mix its positions into four partial outputs, park half and read - add - write the other half on top of the partner's: a third more
  // LDS writes, at 64 B per clock, and 128 dependent LDS round trips per lane: 72.4 -> 69.9 ms per forward with this one.)
  const float descale = p.u[(int64_t)R1_NSLOT * p.Cin * p.Cout];
  const idiff_epilogue &ep = p.ep;
  const bool has_ep = p.has_ep != 0;
  // accumulator register `reg` of lane l is row-tile (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) of its group, cout wh * 32 + (l & 31)
  float *zbase = lds + (size_t)(4 * (lane >> 5)) * 6 * R1_COUT + (size_t)(3 * ph) * R1_COUT + wh * 32 + (lane & 31);
  auto park = [&](auto grp) __attribute__((always_inline)) {
    constexpr int GI = decltype(grp)::value;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      float *zp = zbase + (size_t)(32 * (GI & 1) + (reg & 3) + 8 * (reg >> 2)) * 6 * R1_COUT;
      zp[0] = acc[GI][0][reg]; zp[R1_COUT] = acc[GI][1][reg]; zp[2 * R1_COUT] = acc[GI][2][reg];
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // the finishing thread: channels n .. n + 3 of row-tiles 64 round + 4 tl .. + 3: 16 consecutive pixels of one sample
  const int cq = tid & 15, tl = tid >> 4;
  const int n = n0 + 4 * cq;
#ifdef IDIFF_W1D_DIAG_PLAIN_EP      // timing-only build: the epilogue's switches as compile-time constants (bias only)
  const bool has_res = false, scaled = false, want_stats = false;
  const int act = (int)IDIFF_ACT_NONE;
#else
  const bool has_res = has_ep && ep.residual != nullptr;
  const bool scaled = has_ep && (ep.out_scale != 1.f || ep.rowscale != nullptr);
  const int act = has_ep ? ep.act : (int)IDIFF_ACT_NONE;
  const bool want_stats = has_ep && ep.colstats != nullptr;
#endif
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)ep.residual, 0, (int)p.res_bytes, 0x00020000);
  const int ld_res = (int)ep.ld_residual;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_ep && ep.bias) bias4 = *reinterpret_cast<const float4 *>(ep.bias + n);
  // the residual's 16 pixels of a round are requested one phase ahead (round 0: before the first park, round 1: behind round 0's outputs) and
  // arrive behind the parks and barriers
  float4 res[16];
  auto request_res = [&](int rnd) __attribute__((always_inline)) {
    const uint32_t pxl = (uint32_t)(4 * (64 * rnd + 4 * tl));
    const bool ook = row0 + (int)(pxl / W) < p.rows_total;
    const uint32_t roff = ook ? (((uint32_t)row0 * (uint32_t)W + pxl) * (uint32_t)ld_res + (uint32_t)n) * 4u : R1_INVALID;
#pragma unroll
    for (int i = 0; i < 16; ++i) res[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)roff, i * ld_res * 4, 0));
  };
  double s1[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, s2[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  // One round's 16 pixels with the epilogue's switches as COMPILE-TIME constants: tested at run time inside the loop every pixel became a chain
  // of scalar branches between its LDS reads and its store and nothing of one pixel overlapped the next -- 9.6 us per workgroup where the
  // constant form took 1.7 (profiles/r05_wino1d_stamps.txt).  ACT < 0: the activation by its run-time code.
  auto finish = [&](auto round_c, auto act_c, auto res_c, auto scaled_c, auto stats_c) __attribute__((always_inline)) {
    constexpr int RND = decltype(round_c)::value, ACT = decltype(act_c)::value;
    constexpr bool RES = decltype(res_c)::value, SCALED = decltype(scaled_c)::value, STATS = decltype(stats_c)::value;
    // row-tile m of the block is its pixels 4 m .. 4 m + 3 in row-major order
    const uint32_t pxl = (uint32_t)(4 * (64 * RND + 4 * tl));                  // my first pixel, counted from the block's first
    const int orow = row0 + (int)(pxl / W);
    const bool ook = orow < p.rows_total;
    const uint32_t px0 = (uint32_t)row0 * (uint32_t)W + pxl;
    const uint32_t ooff = ook ? (px0 * (uint32_t)p.Cout + (uint32_t)n) * 4u : R1_INVALID;
    float4 badd = bias4;
    float sc = has_ep ? ep.out_scale : 1.f;
    if (has_ep && ook) {
      const int img = orow / p.H;
      if (ep.rowbias) {
        const float4 rb = *reinterpret_cast<const float4 *>(ep.rowbias + (int64_t)img * ep.ld_rowbias + n);
        badd.x += rb.x; badd.y += rb.y; badd.z += rb.z; badd.w += rb.w;
      }
      if (ep.rowscale) sc *= ep.rowscale[img];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // row-tile 4 tl + j of the round
      const float *zr = lds + (size_t)(4 * tl + j) * 6 * R1_COUT + 4 * cq;
      float4 z[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) z[i] = *reinterpret_cast<const float4 *>(zr + i * R1_COUT);
      float y[4][4];
#define IDIFF_R1_AT(cmp, e)                                                                                  \
      {                                                                                                      \
        const float s12 = z[1].cmp + z[2].cmp, d12 = z[1].cmp - z[2].cmp, s34 = z[3].cmp + z[4].cmp, d34 = z[3].cmp - z[4].cmp; \
        y[0][e] = z[0].cmp + (s12 + s34);                                                                     \
        y[1][e] = fmaf(R1_b, d34, R1_a * d12);                                                                \
        y[2][e] = fmaf(R1_b2, s34, R1_a2 * s12);                                                              \
        y[3][e] = fmaf(R1_b3, d34, fmaf(R1_a3, d12, z[5].cmp));                                               \
      }
      IDIFF_R1_AT(x, 0) IDIFF_R1_AT(y, 1) IDIFF_R1_AT(z, 2) IDIFF_R1_AT(w, 3)
#undef IDIFF_R1_AT
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        y[a][0] = fmaf(y[a][0], descale, badd.x); y[a][1] = fmaf(y[a][1], descale, badd.y);
        y[a][2] = fmaf(y[a][2], descale, badd.z); y[a][3] = fmaf(y[a][3], descale, badd.w);
        if (ACT != (int)IDIFF_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[a][e] = idiff::act_apply(y[a][e], ACT < 0 ? act : ACT);
        }
        if (RES) { const float4 r = res[4 * j + a]; y[a][0] += r.x; y[a][1] += r.y; y[a][2] += r.z; y[a][3] += r.w; }
        if (SCALED) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[a][e] *= sc;
        }
        if (STATS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[RND][e] += (double)y[a][e]; s2[RND][e] += (double)y[a][e] * (double)y[a][e]; }
        }
#ifdef IDIFF_W1D_DIAG_NO_STORE     // timing-only build: one store per thread and round instead of 16
        if (4 * j + a == 15)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, make_float4(y[a][0], y[a][1], y[a][2], y[a][3])), rO, (int)ooff,
                                               (4 * j + a) * p.Cout * 4, IDIFF_W1D_STORE_AUX);
      }
    }
  };
  auto finish_round = [&](auto round_c) __attribute__((always_inline)) {
    using T = std::true_type; using F = std::false_type;
    using ANone = std::integral_constant<int, (int)IDIFF_ACT_NONE>; using AAny = std::integral_constant<int, -1>;
    auto by_stats = [&](auto a, auto r, auto sc_) __attribute__((always_inline)) { if (want_stats) finish(round_c, a, r, sc_, T()); else finish(round_c, a, r, sc_, F()); };
    auto by_scale = [&](auto a, auto r) __attribute__((always_inline)) { if (scaled) by_stats(a, r, T()); else by_stats(a, r, F()); };
    auto by_res = [&](auto a) __attribute__((always_inline)) { if (has_res) by_scale(a, T()); else by_scale(a, F()); };
    if (act == (int)IDIFF_ACT_NONE) by_res(ANone()); else by_res(AAny());
  };
  // the last step ended with a barrier: nobody reads the stages any more
  if (has_res) request_res(0);
  park(I0()); park(I1());
  __syncthreads();
  IDIFF_W1D_T(4)
  finish_round(I0());
  if (has_res) request_res(1);
  __syncthreads();
  park(I2()); park(I3());
  __syncthreads();
  finish_round(I1());
  IDIFF_W1D_T(5)
  if (want_stats) {
    // per-sample (or per-block, where a sample spans several blocks) column sums for the GroupNorm that reads this output: thread sums ->
    // [32 thread rows = (round, tl)][64 channels] in LDS -> one thread per (slot, channel)
    __syncthreads();
    double *red = reinterpret_cast<double *>(lds);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(((16 * r + tl) * R1_COUT) + 4 * cq + e) * 2] = s1[r][e];
        red[(((16 * r + tl) * R1_COUT) + 4 * cq + e) * 2 + 1] = s2[r][e];
      }
    __syncthreads();
    const int slots = RB > p.H ? RB / p.H : 1;                     // samples per block
    const int per = 32 / slots;                                    // thread rows per sample
    for (int o = tid; o < slots * R1_COUT; o += R1_THREADS) {
      const int smp = o / R1_COUT, ch = o - smp * R1_COUT;
      const int64_t slot = (int64_t)tile_m * slots + smp;           // sample, or (sample, split) = block
      if (slots > 1 && slot >= p.B) continue;
      double a = 0.0, b = 0.0;
      for (int k = 0; k < per; ++k) { a += red[((smp * per + k) * R1_COUT + ch) * 2]; b += red[((smp * per + k) * R1_COUT + ch) * 2 + 1]; }
      double *dst = ep.colstats + (slot * p.Cout + n0 + ch) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
#ifdef IDIFF_W1D_STAMP
  IDIFF_W1D_T(6)
  if (tid == 0 && p.stamps) {
    uint64_t *q = p.stamps + 8 * (int64_t)blockIdx.x;
    for (int k = 0; k < 7; ++k) q[k] = st[k];
    q[7] = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 4) ;   // HW_ID low byte: wave / SIMD / pipe ... (bits 8..11: CU)
  }
#endif
#undef IDIFF_W1D_T
}

// U[i][ky] = sum_kx G[i][kx] g[ky][kx] in fp64 for one (cin, cout) pair
__device__ __forceinline__ void r1_u_of_pair(const float *wt, int Cin, int cin, int cout, double (&U)[18]) {
  const double a = R1_A, b = R1_B, na = 1.0 / (2.0 * a * a * (a * a - b * b)), nb = 1.0 / (2.0 * b * b * (b * b - a * a)), n0 = 1.0 / (a * a * b * b);
  const double Gm[6][3] = {{n0, 0.0, 0.0}, {na, a * na, a * a * na}, {na, -a * na, a * a * na}, {nb, b * nb, b * b * nb}, {nb, -b * nb, b * b * nb},
                           {0.0, 0.0, 1.0}};
  for (int ky = 0; ky < 3; ++ky) {
    double g[3];
    for (int kx = 0; kx < 3; ++kx) g[kx] = (double)wt[((int64_t)cout * 9 + ky * 3 + kx) * Cin + cin];
    for (int i = 0; i < 6; ++i) U[3 * i + ky] = Gm[i][0] * g[0] + Gm[i][1] * g[1] + Gm[i][2] * g[2];
  }
}

// pass 1: max |U| over the layer (bits of a non-negative float order like unsigned integers; the word was zeroed by the launcher)
__global__ void wino1d_absmax_kernel(const float *wt, unsigned int *absmax_bits, int Cin, int Cout) {
  const int64_t total = (int64_t)Cin * Cout;
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    double U[18];
    r1_u_of_pair(wt, Cin, (int)(idx % Cin), (int)(idx / Cin), U);
    for (int k = 0; k < 18; ++k) m = fmaxf(m, fabsf((float)U[k]));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(absmax_bits, __float_as_uint(m));
}

// pass 2: the pairs.  The scale 2^k brings max |U| into [2^11, 2^12); header[0] receives 2^-k.
__global__ void wino1d_pack_kernel(const float *wt, _Float16 *u, float *header, int Cin, int Cout) {
  const float amax = __uint_as_float(*reinterpret_cast<const unsigned int *>(header + 1));
  int e = 0;
  if (amax > 0.f && isfinite(amax)) { (void)frexpf(amax, &e); }          // amax = f 2^e, f in [0.5, 1)
  const int k = (amax > 0.f && isfinite(amax)) ? 12 - e : 0;
  const double scale = ldexp(1.0, k);
  const int64_t total = (int64_t)Cin * Cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
    double U[18];
    r1_u_of_pair(wt, Cin, cin, cout, U);
    const int s = cin / R1_KC, c16 = cin % R1_KC, nt = cout / R1_COUT, co = cout % R1_COUT;
    _Float16 *dst = u + ((int64_t)(s * (Cout / R1_COUT) + nt) * R1_NSLOT) * (R1_SLOT_BYTES / 2) + co * R1_KC + c16;
    for (int q = 0; q < 18; ++q) {
      const float v = (float)(U[q] * scale);                                   // rounded once to fp32, then cut
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      dst[(int64_t)q * (R1_SLOT_BYTES / 2)] = hi;
      dst[(int64_t)q * (R1_SLOT_BYTES / 2) + R1_PLANE_BYTES / 2] = lo;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) header[0] = (float)ldexp(1.0, -k);
}

bool r1_geometry_ok(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || Cin <= 0 || Cout <= 0) return false;
  if (W != 4 && W != 8 && W != 16 && W != 32 && W != 64) return false;
  const int RB = R1_PIXELS / W;
  if (H % 4 || (RB % H != 0 && H % RB != 0)) return false;                     // a block is whole images or a whole part of one
  if (Cin % R1_KC || Cin < 2 * R1_KC || Cin > 1024 || Cout % R1_COUT) return false;
  if ((int64_t)R1_NSLOT * Cin * Cout * 4 >= R1_X_LIMIT) return false;
  if ((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) * 4 >= R1_X_LIMIT - 0x4000) return false;   // one buffer descriptor per tensor
  return true;
}

template <int W>
int r1_launch(const Wino1dParams &p, hipStream_t stream) {
  static idiff::AttrGuard guard;
  const void *fn = reinterpret_cast<const void *>(wino1d_kernel<W>);
  if (int rc = idiff::set_dynamic_lds_once(guard, &fn, 1, (int)R1_LDS_BYTES, "conv2d_wino1d")) return rc;
  hipLaunchKernelGGL(wino1d_kernel<W>, dim3(p.blocks_m * p.tiles_n), dim3(R1_THREADS), R1_LDS_BYTES, stream, p);
  return idiff::launch_status("conv2d_wino1d");
}
}  // namespace

IDIFF_API int idiff_conv2d_wino1d_ok(int B, int H, int W, int Cin, int Cout) {
  using namespace idiff;
  if (option(OPT_NO_WINOGRAD) || option(OPT_NO_WINO43H) || option(OPT_NO_WINO1D)) return 0;
  return r1_geometry_ok(B, H, W, Cin, Cout) ? 1 : 0;
}

// nsplit of epilogue.colstats ([samples, nsplit, Cout, 2]) or 0 when the statistics cannot be produced
IDIFF_API int idiff_conv2d_wino1d_colstats_split(int B, int H, int W, int Cin, int Cout) {
  if (!idiff_conv2d_wino1d_ok(B, H, W, Cin, Cout) || idiff::option(idiff::OPT_NO_COLSTATS)) return 0;
  const int RB = R1_PIXELS / W;
  return H > RB ? H / RB : 1;
}

IDIFF_API int64_t idiff_wino1d_weight_floats(int Cin, int Cout) { return (int64_t)R1_NSLOT * Cin * Cout + 4; }

IDIFF_API int idiff_wino1d_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % R1_KC || Cout % R1_COUT)
    return fail("wino1d_pack: Cin must be a multiple of %d and Cout of %d (got %d, %d)", R1_KC, R1_COUT, Cin, Cout);
  if (!wt || !u) return fail("wino1d_pack: null pointer");
  if ((uintptr_t)u & 15) return fail("wino1d_pack: u must be 16-byte aligned");
  const int64_t total = (int64_t)Cin * Cout;
  float *header = u + (int64_t)R1_NSLOT * Cin * Cout;
  hipError_t e = hipMemsetAsync(header, 0, 16, (hipStream_t)stream);
  if (e != hipSuccess) return fail("wino1d_pack: hipMemsetAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(wino1d_absmax_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<unsigned int *>(header + 1), Cin, Cout);
  hipLaunchKernelGGL(wino1d_pack_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<_Float16 *>(u), header, Cin, Cout);
  return launch_status("wino1d_pack");
}

IDIFF_API int idiff_conv2d_wino1d_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                      const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (B == 0) return 0;
  if (!r1_geometry_ok(B, H, W, Cin, Cout))
    return fail("conv2d_wino1d: geometry B=%d H=%d W=%d Cin=%d Cout=%d not supported (ask idiff_conv2d_wino1d_ok)", B, H, W, Cin, Cout);
  if (!x || !u || !out) return fail("conv2d_wino1d: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)out & 15)) return fail("conv2d_wino1d: x, u and out must be 16-byte aligned");
  if (ep && ep->colstats && idiff_conv2d_wino1d_colstats_split(B, H, W, Cin, Cout) <= 0)
    return fail("conv2d_wino1d: colstats are switched off (IDIFF_NO_COLSTATS): ask idiff_conv2d_wino1d_colstats_split");
  if (ep && (ep->rowbias || ep->rowscale) && ep->rows_per_group != H * W)
    return fail("conv2d_wino1d: per-row-group bias / scale only per image (rows_per_group = H * W = %d, got %d)", H * W, ep->rows_per_group);
  if (ep && ep->residual && (((uintptr_t)ep->residual & 15) || ep->ld_residual % 4 || ep->ld_residual < Cout || ep->ld_residual > 0x7fffffff / 4))
    return fail("conv2d_wino1d: residual must be 16-byte aligned with a row pitch >= Cout that is a multiple of 4");
  const int64_t res_bytes = (ep && ep->residual) ? (int64_t)B * H * W * ep->ld_residual * 4 : 0;
  if (res_bytes >= R1_X_LIMIT) return fail("conv2d_wino1d: residual beyond one buffer descriptor");
  Wino1dParams p = {};
  p.x = x; p.u = u; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.rows_total = B * H;
  p.blocks_m = ceil_div(p.rows_total, R1_PIXELS / W); p.tiles_n = Cout / R1_COUT;
  {
    // all cout tiles of a row block next to each other in the launch order (same XCD, same K step at about the same time: the block's pixels come
    // from HBM once): 1.49x -> 1.39x algorithmic bytes over a forward at equal time (pairs of tiles: 73.0 ms, all four: 72.9, one: 74.1)
    const int want = option_value(OPT_WINO_NGROUP);
    p.ngroup = (want > 0 && p.tiles_n % want == 0) ? want : p.tiles_n;
  }
  p.x_bytes = (uint32_t)((int64_t)B * H * W * Cin * 4); p.u_bytes = (uint32_t)((int64_t)R1_NSLOT * Cin * Cout * 4);
  p.out_bytes = (uint32_t)((int64_t)B * H * W * Cout * 4); p.res_bytes = (uint32_t)res_bytes;
  if (ep) {
    p.ep = *ep; p.has_ep = 1;
    if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1;
  } else {
    p.has_ep = 0; p.ep.rows_per_group = 1; p.ep.out_scale = 1.f;
  }
  p.c_nb2 = -R1_b2; p.c_na2 = -R1_a2; p.c_nab2 = -R1_ab2; p.c_a = R1_a; p.c_b = R1_b;
#ifdef IDIFF_W1D_STAMP
  { const char *e = getenv("IDIFF_W1D_STAMP_PTR"); p.stamps = e ? reinterpret_cast<uint64_t *>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  switch (W) {
    case 4: return r1_launch<4>(p, (hipStream_t)stream);
    case 8: return r1_launch<8>(p, (hipStream_t)stream);
    case 16: return r1_launch<16>(p, (hipStream_t)stream);
    case 64: return r1_launch<64>(p, (hipStream_t)stream);
    default: return r1_launch<32>(p, (hipStream_t)stream);
  }
}
