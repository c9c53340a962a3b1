// 3x3 / stride 1 / pad 1 convolutions by Winograd's F(4x4, 3x3) with the 36 contractions on the fp16 matrix cores of gfx950.
// Algorithm, points, transforms, tail: winograd43.hip / winograd43_shared.h.
#include "winograd43_shared.h"

namespace {

// The same convolution with the 36 contractions on the fp16 matrix cores: v_mfma_f32_32x32x16_f16 on PAIRS of fp16 values.
//   v = hi + lo,  hi = fp16(v),  lo = fp16(v - hi)      (22 significand bits; the matrix core keeps fp16 subnormals, measured by
//   scripts/mfma_f16_probe.hip, so the pair is as good for small v as for large)
//   V U^T  ~  Vhi Uhi^T + Vhi Ulo^T + Vlo Uhi^T         (lo x lo is 2^-22 of the product: dropped), fp32 accumulation
// U is scaled by a power of two at pack time so that max |U| = 2^11 .. 2^12 (its typical element then has a NORMAL low part; the
// outputs are scaled back exactly); V is used as is: |V| must stay below 65504, i.e. activations below ~2000 -- beyond that the
// high part is +-inf and the outputs are NaN, loudly.  Measured: scripts/f43_emulation.py (whole nf = 128 network, CPU) per layer
// 8.0e-7 (fp32 contraction 7.8e-7), rel_err(S) 3.33e-6 (3.27e-6); scripts/mfma_f16_probe.hip 9.0e-8 per K = 16 contraction
// against 7.1e-8 of an fp32 fma chain.
// Why: v_mfma_f32_32x32x2_f32 runs at the vector ALU's own rate (157 TFLOP/s, 64 cycles for 4096 flops) and holds the SIMD
// while it does, so every transform instruction above comes out of the contraction's time; the fp16 instruction does 32768 flops in
// 32 cycles and holds the SIMD's issue for 8 of them: three of them per position and 16 channels = 96 cycles where the fp32 form
// spends 512, and the transforms run beside them.  What bounds the kernel then is memory latency: a wave has one in-order counter for
// its loads, the input patch misses to HBM and the U loads queue behind it (profiles/r04_wino43h_loop_experiments.txt).
//
// Workgroup, waves, accumulators, tail: as winograd43_kernel.  K step: 16 input channels.
// Loader: thread (tile t = tid / 16, channel c = tid % 16) holds the 6 x 6 patch of its tile and channel in 36 registers (4-byte
//   loads, 64 contiguous bytes per tile and pixel), transforms it in place -- rows, then columns, no transposition --, splits each
//   V(i, j) into its pair, trades halves with the neighbouring lane (channels 2m, 2m + 1) and stores whole dwords to the stage; as soon as a
//   column of the patch is consumed the same registers receive the next step's column.  The work is cut into seven parts, one behind
//   each of the first seven positions' matrix instructions of the current step (step()).
// Stage (LDS): [36 positions][32 tiles][64 B]; the 64 bytes of a tile are four 16-byte chunks (plane, channel half), chunk c
//   stored at c ^ ((tile >> 2) & 3) so that the 16 lanes of a ds_read_b128 group (tiles r .. r + 15, one chunk each) cover all
//   64 banks.  Two stages = 147,456 B.
// U (global): [Cin/16][Cout/64][36 slots][2 planes][64 cout][16 cin] fp16 + one float (the factor that undoes the scaling) at the
//   end; a wave's 16-byte loads of one plane are 1 KB contiguous.
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
constexpr int H4_KC = 16;
constexpr int H4_TILE_BYTES = 64;
constexpr int H4_POS_BYTES = F4_TILES * H4_TILE_BYTES;                 // 2048
constexpr int H4_STAGE_BYTES = F4_NPOS * H4_POS_BYTES;                 // 73,728
constexpr int H4_SLOT_BYTES = 2 * 64 * H4_KC * 2;                      // 4096: one position of one (step, cout tile)
constexpr int H4_PLANE_BYTES = 64 * H4_KC * 2;                         // 2048
constexpr int H4_TOUCH_BYTES = 1024;                                   // landing zone of the look-ahead touches (see touch() in the kernel)
constexpr size_t H4_LDS_BYTES = (2 * H4_STAGE_BYTES > (int)sizeof(float) * F4_Z_FLOATS ? 2 * H4_STAGE_BYTES : sizeof(float) * F4_Z_FLOATS) + H4_TOUCH_BYTES;
#ifndef IDIFF_W43H_TOUCH_AHEAD
#define IDIFF_W43H_TOUCH_AHEAD 4
#endif
#ifndef IDIFF_W43H_TOUCH_AT
#define IDIFF_W43H_TOUCH_AT 8
#endif
#ifndef IDIFF_W43H_DMA_AT
#define IDIFF_W43H_DMA_AT 0
#endif
#ifndef IDIFF_W43H_BRING
#define IDIFF_W43H_BRING 3
#endif
constexpr int H4_BRING = IDIFF_W43H_BRING;     // positions of U requested ahead (register sets of 8)

#ifndef IDIFF_W43H_STAGE_AT
#define IDIFF_W43H_STAGE_AT 0
#endif
#ifndef IDIFF_W43H_U_AUX
#define IDIFF_W43H_U_AUX 0          // cache policy bits of the U loads (1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef IDIFF_W43H_X_AUX
#define IDIFF_W43H_X_AUX 0          // ... of the patch loads
#endif
#ifndef IDIFF_W43H_COLORDER
#define IDIFF_W43H_COLORDER 0
#endif
// the order in which a step's six patch columns are transformed, written and re-requested
#if IDIFF_W43H_COLORDER == 1
__device__ constexpr int H4_COL[6] = {0, 4, 1, 5, 2, 3};     // columns that neighbouring tiles share (4 = the right neighbour's 0, 5 = its 1) back to back
#else
__device__ constexpr int H4_COL[6] = {0, 1, 2, 3, 4, 5};
#endif
#ifndef IDIFF_W43H_COLS_PER_PART
#define IDIFF_W43H_COLS_PER_PART 1
#endif
constexpr int H4_COLS_PER_PART = IDIFF_W43H_COLS_PER_PART;   // columns staged behind one position (1, 2, 3 or 6)
constexpr int H4_STAGE_AT = IDIFF_W43H_STAGE_AT;   // the next step's staging starts behind this position (0 .. 2): seven parts, one per position

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// the transforms' constants as the packed instructions take them: (-b^2, -a^2), (a, -a), (b, -b) in scalar register pairs
struct H4Consts { f2 n2, pa, pb; float nab2, nb2, na2, a, b; };

// t = B^T d on TWO independent lines at once (component x: one row of the patch, y: the next), the arithmetic of f4_bt
__device__ __forceinline__ void h4_bt_rows(const H4Consts &k, const f2 d0, const f2 d1, const f2 d2, const f2 d3, const f2 d4, const f2 d5,
                                           f2 &t0, f2 &t1, f2 &t2, f2 &t3, f2 &t4, f2 &t5) {
  const f2 nb2 = {k.nb2, k.nb2}, na2 = {k.na2, k.na2}, nab2 = {k.nab2, k.nab2}, a = {k.a, k.a}, b = {k.b, k.b};
  const f2 pe = __builtin_elementwise_fma(nb2, d2, d4), po = __builtin_elementwise_fma(nb2, d1, d3);
  const f2 re = __builtin_elementwise_fma(na2, d2, d4), ro = __builtin_elementwise_fma(na2, d1, d3);
  t0 = __builtin_elementwise_fma(nab2, d2, d0 + d4);
  t1 = __builtin_elementwise_fma(a, po, pe); t2 = __builtin_elementwise_fma(-a, po, pe);
  t3 = __builtin_elementwise_fma(b, ro, re); t4 = __builtin_elementwise_fma(-b, ro, re);
  t5 = __builtin_elementwise_fma(nab2, d3, d1 + d5);
}
// t = B^T d on ONE line held as the pairs P0 = (d0, d1), P1 = (d2, d3), P2 = (d4, d5): six packed instructions instead of twelve
// scalar ones (every source operand picks its halves by op_sel); results as the pairs (t0, t5), (t1, t2), (t3, t4).  Same operations on
// the same values as f4_bt, so the same bits.
__device__ __forceinline__ void h4_bt_pairs(const H4Consts &k, const f2 P0, const f2 P1, const f2 P2, f2 &t05, f2 &t12, f2 &t34) {
  const f2 d1 = {P0.y, P0.y}, d2 = {P1.x, P1.x}, d3 = {P1.y, P1.y}, d4 = {P2.x, P2.x};
  const f2 odd = __builtin_elementwise_fma(k.n2, d1, d3);          // (po, ro) = (d3 - b^2 d1, d3 - a^2 d1)
  const f2 even = __builtin_elementwise_fma(k.n2, d2, d4);         // (pe, re)
  const f2 po = {odd.x, odd.x}, ro = {odd.y, odd.y}, pe = {even.x, even.x}, re = {even.y, even.y};
  t12 = __builtin_elementwise_fma(k.pa, po, pe);
  t34 = __builtin_elementwise_fma(k.pb, ro, re);
  const f2 nab2 = {k.nab2, k.nab2};
  t05 = __builtin_elementwise_fma(nab2, P1, P0 + P2);
}

__global__ void __launch_bounds__(F4_THREADS, 2)
winograd43h_kernel(const Wino43Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef IDIFF_W43H_STAMP   // diagnostic build (scripts/wino43h_stamps.py): 100 MHz ticks at the phases of a workgroup's life; buffer in p.coef
  const uint64_t st_start = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int BRING = H4_BRING;                // positions of U requested ahead; must divide 9 (the ring's phase then repeats every step)
  static_assert(9 % H4_BRING == 0, "the ring of U registers must divide the nine positions of a step");
  char *const ldsb = reinterpret_cast<char *>(lds);
  const int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int per_group = p.tiles_m * p.ngroup;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int tile_n = grp * p.ngroup + in_grp % p.ngroup, tile_m = in_grp / p.ngroup;
  const int tile0 = tile_m * F4_TILES, n0 = tile_n * F4_COUT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wh = wave >> 2, wq = (wave & 3) ^ wh, wa = wq >> 1, wb = wq & 1;       // as winograd43_kernel

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);

  // ---------------------------------------------------------------- loader: thread = (tile, channel)
  // 16 consecutive lanes = the 16 channels of one tile (64 contiguous bytes per pixel); lanes 2m and 2m + 1 hold the two halves of
  // one stage dword and exchange them by DPP (see stage()).
  const int lch = tid & 15, ltile = tid >> 4;
  const uint32_t cin4 = (uint32_t)p.Cin * 4u;
  // Vector offsets of my channel at pixel column 4 tx: v_mid for patch rows 1 .. 5 (image row 4 ty, the scalar offset adds (i - 1) rows:
  // rows 1 .. 4 of a patch always lie inside the image), v_top for row 0 and v_bot (= v_mid) for row 5, F4_INVALID where the row is
  // outside the image or the tile beyond the last one -- three registers where six offsets would be held.
  uint32_t v_top, v_mid, v_bot;
  bool c0ok, c5ok;                                    // columns 4 tx - 1 and 4 tx + 4 inside the image
  {
    const int T = tile0 + ltile;
    const bool tv = T < p.total_tiles;
    int img, ty, tx;
    f4_split_tile(p, tv ? T : 0, img, ty, tx);
    c0ok = tx > 0; c5ok = tx + 1 < p.tiles_x;
    const uint32_t at_row1 = (uint32_t)(((img * p.H + 4 * ty) * p.W + 4 * tx) * p.Cin + lch) * 4u;
    v_mid = tv ? at_row1 : F4_INVALID;
    v_top = (tv && ty > 0) ? at_row1 - (uint32_t)p.W * cin4 : F4_INVALID;
    v_bot = (tv && ty + 1 < p.tiles_y) ? at_row1 : F4_INVALID;
  }
  const int row4 = p.W * (int)cin4;                   // bytes per image row
  const int nsteps = p.Cin / H4_KC;
  // The patch in PAIRS of rows: dp[r][j] = (d[2r][j], d[2r+1][j]) -- one packed instruction (v_pk_fma_f32 / v_pk_add_f32) then
  // transforms two rows along x at once, and along y the three pairs of a column ARE the operand pairs of h4_bt_pairs.
  f2 dp[3][6];
  // column j of the patch for K step `step` (clamped: the last stage requests its own step again and nobody reads it).  The
  // range check of a buffer load covers the vector offset only, so columns 1 .. 4 take the row's offset (inside the tensor, or
  // F4_INVALID) plus a scalar, and the edge columns 0 and 5 go through descriptors of their own whose base is one pixel to the
  // left / four to the right (their extent shrinks with it: column 5 can never run past the tensor's end).
  const __amdgpu_buffer_rsrc_t rX0 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x - p.Cin), 0, (int)(p.x_bytes + cin4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rX5 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + 4 * p.Cin), 0, (int)(p.x_bytes - 4u * cin4), 0x00020000);
  auto fetch_col = [&](int j, int step) __attribute__((always_inline)) {
    const int choff = min(step, nsteps - 1) * (H4_KC * 4);
    uint32_t invalid = F4_INVALID;
    asm volatile("" : "+s"(invalid));                 // the edge offsets are formed per use: hoisted they would hold 12 registers
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float v;
#ifdef IDIFF_W43H_DIAG_NO_COL45   // timing-only builds: what the loop costs without the requests that repeat a neighbouring tile's (columns 4, 5 = the right
      if (j >= 4) { dp[i >> 1][j][i & 1] = dp[i >> 1][j - 4][i & 1]; continue; }   // neighbour's 0, 1; rows 4, 5 = the lower neighbour's 0, 1)
#endif
#ifdef IDIFF_W43H_DIAG_NO_ROW45
      if (i >= 4) { dp[i >> 1][j][i & 1] = dp[(i - 4) >> 1][j][i & 1]; continue; }
#endif
      const uint32_t vo = i == 0 ? v_top : (i == 5 ? v_bot : v_mid);
      const int ro = i == 0 ? 0 : (i - 1) * row4;
      if (j == 0) v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX0, (int)(c0ok ? vo : invalid), choff + ro, IDIFF_W43H_X_AUX));
      else if (j == 5) v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX5, (int)(c5ok ? vo : invalid), choff + ro, IDIFF_W43H_X_AUX));
      else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, (int)vo, choff + ro + (j - 1) * (int)cin4, IDIFF_W43H_X_AUX));
      dp[i >> 1][j][i & 1] = v;
    }
  };
  // Look-ahead touches -- an experiment that did NOT pay (A/B builds only, -DIDIFF_W43H_TOUCH): one dword per 64-byte chunk is 64 separate
  // line requests per instruction, 192 per wave and step beside the 144 of the real patch requests: the texture-address path, which the
  // kernel already keeps busy, pays more than the earlier arrival of the lines gives back.  The idea: the K loop's pace is set by the patch requests: they miss to HBM, and a wave's vector-memory operations retire in
  // order, so every U request (an L2 hit) issued behind one of them waits out an HBM round trip -- at six of a step's nine positions
  // (profiles/r04_wino43h_loop_experiments.txt).  Here each lane touches ONE dword of up to three 64-byte pixel chunks of the patch
  // TOUCH_AHEAD steps ahead -- between them the lanes of a wave cover the 4 x 36 chunks their tiles will request -- so that the L2 has the
  // lines when the real requests come: those then retire in an L2 round trip, and the one HBM-latency wait per step sits behind the
  // touches, at one position.  The touches are LDS-DMA loads into a landing zone nobody reads: no destination registers.
  auto touch = [&](int step) __attribute__((always_inline)) {
#ifdef IDIFF_W43H_TOUCH           // OFF: measured 87.2 against 76.9 ms per forward (profiles/r05_touch_ab.txt)
    typedef __attribute__((address_space(3))) void lds_void;
    lds_void *zone = (lds_void *)(ldsb + 2 * H4_STAGE_BYTES + (wave & 3) * 256);
    const int choff = min(step, nsteps - 1) * (H4_KC * 4);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int pix = (tid & 15) + 16 * q;             // this lane's pixel of its tile's 6 x 6 patch: pix = 6 i + j
      const int i = (pix * 43) >> 8, j = pix - 6 * i;
      // rows / columns outside the image are not special-cased: the address is then a neighbouring pixel's (a line some tile requests
      // anyway) or beyond the tensor (no request at all: the descriptor's range check)
      const bool ok = pix < 36 && v_mid != F4_INVALID;
      const uint32_t off = v_mid + (uint32_t)((i - 1) * row4 + (j - 1) * (int)cin4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, zone, 4, (int)(ok ? off : F4_INVALID), choff, 0, 0);
    }
#endif
  };
  // The dword (channels 2m, 2m + 1) this lane writes for every position: even lanes that of plane 0 (hi), odd lanes that of plane 1
  // (lo), which lies 32 bytes from it (bit 5 of the offset flipped)
  const bool odd = (tid & 1) != 0;
  const int w_off = ltile * H4_TILE_BYTES + (((2 * (tid & 1) + (lch >> 3)) ^ ((ltile >> 2) & 3)) << 4) + ((lch & 7) >> 1) * 4;
  // byte selectors of the two stored dwords [even channel | odd channel] from (partner's register, my register): mine is the low
  // half in even lanes, the high half in odd lanes
  const uint32_t sel0 = odd ? 0x01000504u : 0x05040100u, sel1 = sel0 + 0x02020202u;
  const H4Consts kc = {{p.c_nb2, p.c_na2}, {p.c_a, -p.c_a}, {p.c_b, -p.c_b}, p.c_nab2, p.c_nb2, p.c_na2, p.c_a, p.c_b};
  int f_step = 0;
#ifdef IDIFF_W43H_DIAG_NO_VWRITE
  uint32_t diag_sink = 0;
#endif
  // The stage of a step in seven parts -- the transform along x (all rows), then one column at a time: transform along y, cut into pairs,
  // store, request the next step's column.  step() spreads them over the wave's positions (see there).
  auto stage_rows = [&]() __attribute__((always_inline)) {
#ifdef IDIFF_W43H_DIAG_NO_STAGE   // timing-only build (scripts/wino43h_ab.py): no transform, no stage writes, no input loads
    return;
#endif
    // along x: two rows per instruction, in place
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      f2 t0, t1, t2, t3, t4, t5;
      h4_bt_rows(kc, dp[r][0], dp[r][1], dp[r][2], dp[r][3], dp[r][4], dp[r][5], t0, t1, t2, t3, t4, t5);
      dp[r][0] = t0; dp[r][1] = t1; dp[r][2] = t2; dp[r][3] = t3; dp[r][4] = t4; dp[r][5] = t5;
      __builtin_amdgcn_sched_barrier(0);              // one row pair / column at a time: interleaved they need registers that do not exist
    }
  };
  auto stage_col = [&](int buf, int j, bool last_col) __attribute__((always_inline)) {
#ifdef IDIFF_W43H_DIAG_NO_STAGE
    if (last_col) ++f_step;
    (void)buf; return;
#endif
    char *Vd = ldsb + buf * H4_STAGE_BYTES + w_off;
    // along y: V(0, j), V(5, j) | V(1, j), V(2, j) | V(3, j), V(4, j); each pair is cut into its fp16 pairs and written
    f2 v[3];
    h4_bt_pairs(kc, dp[0][j], dp[1][j], dp[2][j], v[0], v[1], v[2]);
    constexpr int row_lo[3] = {0, 1, 3}, row_hi[3] = {5, 2, 4};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      // (hi, lo) of the two positions, two halves per register; neighbouring lanes -- channels 2m and 2m + 1 -- trade by DPP: the even
      // lane collects both channels' hi parts, the odd lane both lo parts, and each stores whole dwords
      const uint32_t xh = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[q], h2));
      // v - hi in one mixed-precision instruction per component (fp32 + fp16 * -1: exact), not a conversion back and a subtraction
      f2 rest;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest.x) : "v"(xh), "v"(v[q].x));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest.y) : "v"(xh), "v"(v[q].y));
      const uint32_t xl = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, h2));
      const uint32_t give = odd ? xh : xl, keep = odd ? xl : xh;
      const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp((int)give, (int)give, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]: every lane receives
      char *qa = Vd + (6 * row_lo[q] + j) * H4_POS_BYTES, *qb = Vd + (6 * row_hi[q] + j) * H4_POS_BYTES;
#ifdef IDIFF_W43H_DIAG_NO_VWRITE  // timing-only build: the pairs are summed into one register instead of written
      diag_sink += keep + got; (void)qa; (void)qb;
#else
      *reinterpret_cast<uint32_t *>(qa) = __builtin_amdgcn_perm(got, keep, sel0);
      *reinterpret_cast<uint32_t *>(qb) = __builtin_amdgcn_perm(got, keep, sel1);
#endif
    }
#ifdef IDIFF_W43H_DIAG_NO_XLOAD   // timing-only build: the input is loaded for the first step only
    if (f_step == 0)
#endif
    fetch_col(j, f_step + 1);                         // the column's registers are free: the next step's column moves in
    if (last_col) ++f_step;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {       // all of it at once: the first step's, before the loop
    stage_rows();
#pragma unroll
    for (int q = 0; q < 6; ++q) stage_col(buf, H4_COL[q], q == 5);
  };

#ifdef IDIFF_W43H_DIAG_DMA
  // Timing-only build (scripts/wino43h_ab.py, results wrong by construction): what the K loop would cost if the transformed, pair-cut
  // patches V came READY from HBM (written by the producing GroupNorm pass) and went global -> LDS by LDS-DMA -- no patch registers, no
  // transform, no LDS stores; each wave moves its ninth of the 73,728-byte stage with nine 1 KB requests.  The source is the input tensor
  // itself, read as a stream of stage-sized blocks (one per workgroup row and step, shared by the workgroups of the other cout tiles).
  auto dma_stage = [&](int buf, int step) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) void lds_void;
    const uint32_t blocks = (p.x_bytes - H4_STAGE_BYTES) / H4_STAGE_BYTES;
    const uint32_t blk = ((uint32_t)tile_m * (uint32_t)nsteps + (uint32_t)min(step, nsteps - 1)) % (blocks ? blocks : 1u);
    const int base = (int)(blk * (uint32_t)H4_STAGE_BYTES) + wave * (H4_STAGE_BYTES / 8);
#pragma unroll
    for (int i = 0; i < 9; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void *)(ldsb + buf * H4_STAGE_BYTES + wave * (H4_STAGE_BYTES / 8) + i * 1024), 16,
                                               lane * 16, base + i * 1024, 0, 0);
  };
#endif

  // ---------------------------------------------------------------- contraction
  floatx16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  const int pos0 = (3 * wa) * 6 + 3 * wb;            // slot of this wave's first position
  // V fragment of tile row fr: chunk fh of plane 0, chunk 2 + fh of plane 1 (bit 5 flipped), swizzled as the writer does
  const int a_off = pos0 * H4_POS_BYTES + fr * H4_TILE_BYTES + ((fh ^ ((fr >> 2) & 3)) << 4);
  const uint32_t u_lane = (uint32_t)((wh * 32 + fr) * (H4_KC * 2) + fh * 16);
  halfx8 bh[BRING], bl[BRING];
  auto load_b = [&](int pp, int step) __attribute__((always_inline)) {
    const int slot = pos0 + (pp / 3) * 6 + (pp % 3);
    const int soff = ((step * p.tiles_n + tile_n) * F4_NPOS + slot) * H4_SLOT_BYTES;
    bh[pp % BRING] = __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_lane, soff, IDIFF_W43H_U_AUX));
    bl[pp % BRING] = __builtin_bit_cast(halfx8, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_lane + H4_PLANE_BYTES, soff, IDIFF_W43H_U_AUX));
  };
  // A step: nine positions, each three matrix instructions whose U operands were requested three positions earlier -- about one L2 round
  // trip (~1600 clocks) per three positions, so the contraction alone is bound by that latency (stamps: 5200 clocks of a 9000-clock step
  // when the staging was done in one block of 3000-3600 clocks beside it).  The next step's staging is therefore cut into seven parts and
  // one part follows each of the first seven positions' matrix instructions: the wave does its vector work while its own loads fly.
#ifdef IDIFF_W43H_STAMP
  uint32_t ph_stage = 0, ph_wait = 0;                 // shader-clock ticks this wave spent in its staging parts / at the step barrier
#define IDIFF_PH_T() ({ __builtin_amdgcn_sched_barrier(0); const uint64_t t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); t_; })
#endif
  auto step = [&](int s, auto last) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last)::value;
    const int buf = s & 1;
    const char *S = ldsb + buf * H4_STAGE_BYTES + a_off;
    auto a_hi = [&](int pp) { return *reinterpret_cast<const halfx8 *>(S + ((pp / 3) * 6 + (pp % 3)) * H4_POS_BYTES); };
    auto a_lo = [&](int pp) { return *reinterpret_cast<const halfx8 *>(S + ((pp / 3) * 6 + (pp % 3)) * H4_POS_BYTES + 32 - 2 * (a_off & 32)); };
    halfx8 ah[2], al[2];
    ah[0] = a_hi(0); al[0] = a_lo(0);
#pragma unroll
    for (int pp = 0; pp < 9; ++pp) {
      __builtin_amdgcn_sched_barrier(0);
      if (pp + 1 < 9) { ah[(pp + 1) & 1] = a_hi(pp + 1); al[(pp + 1) & 1] = a_lo(pp + 1); }
      const halfx8 xh = ah[pp & 1], xl = al[pp & 1], yh = bh[pp % BRING], yl = bl[pp % BRING];
#ifdef IDIFF_W43H_DIAG_NO_MFMA    // timing-only build: the operands are consumed by one vector instruction each instead
      acc[pp][0] += (float)xh[0] + (float)xl[0] + (float)yh[0] + (float)yl[0];
#else
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[pp], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[pp], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[pp], 0, 0, 0);
#endif
#ifdef IDIFF_W43H_DIAG_NO_BLOAD   // timing-only build: U is loaded for the first step only
      if (s == 0) { if (pp + BRING < 9) load_b(pp + BRING, s); }
#else
      if (pp + BRING < 9) load_b(pp + BRING, s); else if (!LAST) load_b(pp + BRING - 9, s + 1);
#endif
      if (!LAST && pp == IDIFF_W43H_TOUCH_AT) touch(s + IDIFF_W43H_TOUCH_AHEAD);
#ifdef IDIFF_W43H_DIAG_DMA
      if (!LAST && pp == IDIFF_W43H_DMA_AT) dma_stage(buf ^ 1, s + 1);
#endif
      if (!LAST) {
        __builtin_amdgcn_sched_barrier(0);
#ifdef IDIFF_W43H_STAMP
        const uint64_t a_ = IDIFF_PH_T();
#endif
        if (pp == H4_STAGE_AT) stage_rows();
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (pp == H4_STAGE_AT + 1 + q / H4_COLS_PER_PART) stage_col(buf ^ 1, H4_COL[q], q == 5);
#ifdef IDIFF_W43H_STAMP
        ph_stage += (uint32_t)(IDIFF_PH_T() - a_);
#endif
      }
    }
#ifdef IDIFF_W43H_STAMP
    { const uint64_t a_ = IDIFF_PH_T(); __syncthreads(); ph_wait += (uint32_t)(IDIFF_PH_T() - a_); }
#else
    __syncthreads();
#endif
  };

#ifdef IDIFF_W43H_STAMP
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t st_pro0 = __builtin_amdgcn_s_memrealtime();     // set-up done, nothing requested yet
  __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
  for (int pp = 0; pp < BRING; ++pp) load_b(pp, 0);
#pragma unroll
  for (int q = 0; q < 6; ++q) fetch_col(H4_COL[q], 0);
#ifdef IDIFF_W43H_STAMP
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t st_pro1 = __builtin_amdgcn_s_memrealtime();     // first operands arrived
  __builtin_amdgcn_sched_barrier(0);
#endif
  stage(0);
#ifdef IDIFF_W43H_DIAG_DMA
  dma_stage(0, 0);
#endif
#ifdef IDIFF_W43H_STAMP
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t st_pro2 = __builtin_amdgcn_s_memrealtime();     // first stage written by this wave
  __builtin_amdgcn_sched_barrier(0);
#endif
  __syncthreads();
#ifdef IDIFF_W43H_STAMP
  const uint64_t st_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
  {
    int s = 0;                                        // at least two steps (Cin >= 32, checked by the launcher)
    do step(s, std::false_type()); while (++s + 1 < nsteps);
  }
  step(nsteps - 1, std::true_type());
#ifdef IDIFF_W43H_STAMP
  const uint64_t st_loop1 = __builtin_amdgcn_s_memrealtime();
  uint64_t *st_out = p.stamps ? p.stamps + 8 * (int64_t)blockIdx.x : nullptr;
  if (tid == 0 && st_out) { st_out[0] = st_start; st_out[1] = st_loop0; st_out[2] = st_loop1; }
  // phase clocks of waves 0 and 4 (the two waves of one SIMD), lane 0 each
  if ((tid == 0 || tid == 256) && p.stamps) {
    uint64_t *q = p.stamps + 8 * (int64_t)gridDim.x + 4 * ((int64_t)blockIdx.x * 2 + (tid >> 8));
    q[0] = ph_stage; q[1] = ph_wait; q[2] = ((st_pro0 - st_start) << 32) | ((st_pro1 - st_pro0) << 16) | (st_pro2 - st_pro1); q[3] = st_loop0 - st_pro2;
  }
#endif

#ifdef IDIFF_W43H_DIAG_NO_VWRITE
  if (diag_sink == 12345u) ldsb[tid] = 1;
#endif
  const float descale = p.u[(int64_t)36 * p.Cin * p.Cout];
#ifdef IDIFF_W43H_DIAG_NO_TAIL     // timing-only build: one store per lane instead of the tail
  float diag_sum = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) diag_sum += acc[i][r];
  if (diag_sum == 12345.f) p.out[tid] = descale;
#else
#ifdef IDIFF_W43H_STAMP
  f4_tail<true>(p, lds, acc, tile0, tile_m, n0, wh, wa, wb, descale, st_out);
#else
  f4_tail<true>(p, lds, acc, tile0, tile_m, n0, wh, wa, wb, descale);
#endif
#endif
}

// pass 1: max |U| over the layer (bits of a non-negative float order like unsigned integers; the word was zeroed by the launcher)
__global__ void winograd43h_absmax_kernel(const float *wt, unsigned int *absmax_bits, int Cin, int Cout) {
  const int64_t total = (int64_t)Cin * Cout;
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    double U[36];
    f4_u_of_pair(wt, Cin, (int)(idx % Cin), (int)(idx / Cin), U);
    for (int k = 0; k < 36; ++k) m = fmaxf(m, fabsf((float)U[k]));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(absmax_bits, __float_as_uint(m));
}

// pass 2: the pairs.  The scale 2^k brings max |U| into [2^11, 2^12); header[0] receives 2^-k.
__global__ void winograd43h_pack_kernel(const float *wt, _Float16 *u, float *header, int Cin, int Cout) {
  const float amax = __uint_as_float(*reinterpret_cast<const unsigned int *>(header + 1));
  int e = 0;
  if (amax > 0.f && isfinite(amax)) { (void)frexpf(amax, &e); }          // amax = f 2^e, f in [0.5, 1)
  const int k = (amax > 0.f && isfinite(amax)) ? 12 - e : 0;
  const double scale = ldexp(1.0, k);
  const int64_t total = (int64_t)Cin * Cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
    double U[36];
    f4_u_of_pair(wt, Cin, cin, cout, U);
    const int s = cin / H4_KC, c16 = cin % H4_KC, nt = cout / F4_COUT, co = cout % F4_COUT;
    _Float16 *dst = u + ((int64_t)(s * (Cout / F4_COUT) + nt) * F4_NPOS) * (H4_SLOT_BYTES / 2) + co * H4_KC + c16;
    for (int q = 0; q < 36; ++q) {
      const float v = (float)(U[q] * scale);                                   // rounded once to fp32, as the fp32 kernel's U
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      dst[(int64_t)q * (H4_SLOT_BYTES / 2)] = hi;
      dst[(int64_t)q * (H4_SLOT_BYTES / 2) + H4_PLANE_BYTES / 2] = lo;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) header[0] = (float)ldexp(1.0, -k);
}

bool h4_geometry_ok(int B, int H, int W, int Cin, int Cout) {
  if (!f4_geometry_ok(B, H, W, Cin, Cout)) return false;
  if (Cin % H4_KC || Cin < 2 * H4_KC || Cin > 1024) return false;                                // edge-column offsets must not wrap past F4_INVALID
  if ((int64_t)B * H * W * Cin * 4 >= 0xFFFF0000ll - 0x4000) return false;
  return true;
}
}  // namespace

// ---------------------------------------------------------------- the fp16-pair form (winograd43h_kernel)
IDIFF_API int idiff_conv2d_winograd43h_ok(int B, int H, int W, int Cin, int Cout) {
  if (idiff::option(idiff::OPT_NO_WINOGRAD) || idiff::option(idiff::OPT_NO_WINO43) || idiff::option(idiff::OPT_NO_WINO43H)) return 0;
  return h4_geometry_ok(B, H, W, Cin, Cout) ? 1 : 0;
}

// floats of a packed bank: the fp16 pairs take the room of one float per weight-domain element, + 4 floats of header at the end
IDIFF_API int64_t idiff_winograd43h_weight_floats(int Cin, int Cout) { return (int64_t)36 * Cin * Cout + 4; }

IDIFF_API int idiff_winograd43h_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % H4_KC || Cout % F4_COUT)
    return fail("winograd43h_pack: Cin must be a multiple of %d and Cout of %d (got %d, %d)", H4_KC, F4_COUT, Cin, Cout);
  if (!wt || !u) return fail("winograd43h_pack: null pointer");
  if ((uintptr_t)u & 15) return fail("winograd43h_pack: u must be 16-byte aligned");
  const int64_t total = (int64_t)Cin * Cout;
  float *header = u + (int64_t)36 * Cin * Cout;
  hipError_t e = hipMemsetAsync(header, 0, 16, (hipStream_t)stream);
  if (e != hipSuccess) return fail("winograd43h_pack: hipMemsetAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(winograd43h_absmax_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<unsigned int *>(header + 1), Cin, Cout);
  hipLaunchKernelGGL(winograd43h_pack_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<_Float16 *>(u), header, Cin, Cout);
  return launch_status("winograd43h_pack");
}

IDIFF_API int idiff_conv2d_winograd43h_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                           const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (B == 0) return 0;
  if (!h4_geometry_ok(B, H, W, Cin, Cout))
    return fail("conv2d_winograd43h: geometry B=%d H=%d W=%d Cin=%d Cout=%d not supported (ask idiff_conv2d_winograd43h_ok)", B, H, W, Cin, Cout);
  if (!x || !u || !out) return fail("conv2d_winograd43h: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)out & 15)) return fail("conv2d_winograd43h: x, u and out must be 16-byte aligned");
  if (ep && ep->colstats && idiff_conv2d_winograd43_colstats_split(B, H, W, Cin, Cout) <= 0)
    return fail("conv2d_winograd43h: colstats needs whole workgroups per sample or whole samples per workgroup "
                "(ask idiff_conv2d_winograd43_colstats_split)");
  if (ep && (ep->rowbias || ep->rowscale) && ep->rows_per_group != H * W)
    return fail("conv2d_winograd43h: per-row-group bias / scale only per image (rows_per_group = H * W = %d, got %d)", H * W, ep->rows_per_group);
  if (ep && ep->residual && (((uintptr_t)ep->residual & 15) || ep->ld_residual % 4 || ep->ld_residual < Cout || ep->ld_residual > 0x7fffffff / 4))
    return fail("conv2d_winograd43h: residual must be 16-byte aligned with a row pitch >= Cout that is a multiple of 4");
  const int64_t res_bytes = (ep && ep->residual) ? (int64_t)B * H * W * ep->ld_residual * 4 : 0;
  if (res_bytes >= F4_X_LIMIT) return fail("conv2d_winograd43h: residual beyond one buffer descriptor");
  Wino43Params p = {};
  p.x = x; p.u = u; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tiles_x = W / 4; p.tiles_y = H / 4; p.tiles_per_img = p.tiles_x * p.tiles_y; p.total_tiles = B * p.tiles_per_img;
  p.tx_shift = p.tpi_shift = -1;
  if ((p.tiles_x & (p.tiles_x - 1)) == 0 && (p.tiles_per_img & (p.tiles_per_img - 1)) == 0) {
    p.tx_shift = __builtin_ctz((unsigned)p.tiles_x); p.tpi_shift = __builtin_ctz((unsigned)p.tiles_per_img);
  }
  p.tiles_m = ceil_div(p.total_tiles, F4_TILES); p.tiles_n = Cout / F4_COUT;
  {
    const int want = option_value(OPT_WINO_NGROUP);
    p.ngroup = (want > 0 && p.tiles_n % want == 0) ? want : ((p.tiles_n > 2 && p.tiles_n % 2 == 0) ? 2 : p.tiles_n);
  }
  p.x_bytes = (uint32_t)((int64_t)B * H * W * Cin * 4); p.u_bytes = (uint32_t)((int64_t)36 * Cin * Cout * 4);
  p.out_bytes = (uint32_t)((int64_t)B * H * W * Cout * 4); p.res_bytes = (uint32_t)res_bytes;
  if (ep) {
    p.ep = *ep; p.has_ep = 1;
    if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1;
  } else {
    p.has_ep = 0; p.ep.rows_per_group = 1; p.ep.out_scale = 1.f;
  }
  p.c_nb2 = -F4_b2; p.c_na2 = -F4_a2; p.c_nab2 = -F4_ab2; p.c_a = F4_a; p.c_b = F4_b;
#ifdef IDIFF_W43H_STAMP
  { const char *e = getenv("IDIFF_W43H_STAMP_PTR"); p.stamps = e ? reinterpret_cast<uint64_t *>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  static AttrGuard guard;
  const void *fn = reinterpret_cast<const void *>(winograd43h_kernel);
  if (int rc = set_dynamic_lds_once(guard, &fn, 1, (int)H4_LDS_BYTES, "conv2d_winograd43h")) return rc;
  hipLaunchKernelGGL(winograd43h_kernel, dim3(p.tiles_m * p.tiles_n), dim3(F4_THREADS), H4_LDS_BYTES, (hipStream_t)stream, p);
  return launch_status("conv2d_winograd43h");
}
