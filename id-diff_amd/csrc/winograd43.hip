// 3x3 / stride 1 / pad 1 convolutions by Winograd's F(4x4, 3x3) on the fp32 matrix cores of gfx950: 36 multiplications per
// 4x4 output tile and (cin, cout) pair = 2.25 per output, where F(2x2, 3x3) (winograd.hip) spends 4 and the implicit GEMM 9.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d: 6x6 input patch at (4ty-1, 4tx-1), Y: 4x4 outputs at (4ty, 4tx)
//
// Interpolation points 0, +-a, +-b, infinity with a = 2/3, b = 3/2.  Measured before this file was written
// (scripts/f43_emulation.py, CPU emulation of the whole nf = 128 NCSN++ with every eligible convolution in this form, fp32 transforms
// and contraction): per layer 0.8-1.3e-6 against an fp64 convolution (F(2x2,3x3): 4e-7; Lavin's points 0, +-1, +-2: 1.6-2.7e-6),
// rel_err(S) 3.3e-6 against an fp64 network (bar 2e-5), every singular value above 2e-5 sigma_max within 1.4e-5 (bar 1e-4) -- the
// round-1 estimate "loses two decimal digits" was wrong.
//
//   B^T rows (input transform, t = B^T d):                 A^T rows (output transform, y = A^T m):
//     t0 = d0 - (a^2 + b^2) d2 + d4                           y0 = m0 + (m1 + m2) + (m3 + m4)
//     t1 = (d4 - b^2 d2) + a (d3 - b^2 d1)                    y1 = a (m1 - m2) + b (m3 - m4)
//     t2 = (d4 - b^2 d2) - a (d3 - b^2 d1)                    y2 = a^2 (m1 + m2) + b^2 (m3 + m4)
//     t3 = (d4 - a^2 d2) + b (d3 - a^2 d1)                    y3 = a^3 (m1 - m2) + b^3 (m3 - m4) + m5
//     t4 = (d4 - a^2 d2) - b (d3 - a^2 d1)
//     t5 = d1 - (a^2 + b^2) d3 + d5                        G rows: (1, p, p^2) / prod_{q != p} (p - q) for p = 0, +-a, +-b; (0, 0, 1)
//
//   U = G g G^T is packed once per layer (idiff_winograd43_pack_f32, fp64, rounded once) as [Cin/8][Cout/64][36 slots][64 cout]
//   [8 cin]; V = B^T d B is formed by the loader between L2 and LDS; M_p = V_p U_p^T for the 36 positions p = (i, j) (slot 6 i + j)
//   are 36 independent [tiles x Cin] x [Cin x Cout] contractions on v_mfma_f32_32x32x2_f32.
//
// Workgroup: 512 threads = 8 waves, 32 tiles x 64 output channels of ALL 36 positions (73,728 accumulators: more than half of a
//   CU's register file, so ONE workgroup per CU, two waves per SIMD).  Wave (h, q): output-channel half h and the 3x3 block of
//   positions rows 3 (q >> 1) .., columns 3 (q & 1) ..: 9 positions x [32 tiles x 32 channels] = 144 accumulators.
// K loop: 8 input channels per step; per wave 36 MFMAs from 9 sixteen-byte LDS reads of V and 9 sixteen-byte global loads of U
//   (a ring of three register sets, each re-requested three positions ahead), two LDS stages of V, one barrier per step.
// Loader (all 8 waves): lanes in groups of 8 = the rows r of the 6x6 patch of one (tile, 4-channel quad) unit (lanes 6, 7 of a
//   group idle): six 16-byte loads (the row), the 6-point transform along x in registers (12 ops per component), the x-mixed row
//   to a wave-private LDS scratch, then lane c of the group reads COLUMN c of the unit back (6 x 16 bytes), transforms along y
//   and writes V(0..5, c) to the stage.  The scratch exchange needs no workgroup barrier: DS operations of a wave are in order.
//   Waves w and w + 4 share a SIMD: the first four transform at the start of a step, the others before their last positions.
// Tail: a wave's block holds 3 of the 6 columns of a transform row, so the row mixing z_{i,b} = sum_j A^T[b][j] m_{i,j} is done in
//   two parts: the waves of column block 1 park their part in LDS, those of block 0 add theirs on top ([6 i][32 tiles][2 b][64 cout]
//   fp32 = 96 KB per pass over the dead stages, two passes: output columns b = 0, 1 and b = 2, 3); then every thread owns one
//   (tile, 4-channel group) and finishes y_{a,b} = sum_i A^T[a][i] z_{i,b} with the fused epilogue and 16-byte stores.
#include "winograd43_shared.h"

namespace {

#ifndef IDIFF_W43_EARLY_WAVES
#define IDIFF_W43_EARLY_WAVES 2
#endif
constexpr int F4_EARLY_WAVES = IDIFF_W43_EARLY_WAVES;   // waves 0 .. n-1 transform at the start of a step, the others late (A/B)
#ifndef IDIFF_W43_MID_WAVES
#define IDIFF_W43_MID_WAVES IDIFF_W43_EARLY_WAVES
#endif
#ifndef IDIFF_W43_MID_AT
#define IDIFF_W43_MID_AT 3
#endif
constexpr int F4_MID_WAVES = IDIFF_W43_MID_WAVES, F4_MID_AT = IDIFF_W43_MID_AT;   // waves EARLY .. MID-1 transform in front of position MID_AT
#ifndef IDIFF_W43_VRING
#define IDIFF_W43_VRING 2
#endif
constexpr int F4_VRING = IDIFF_W43_VRING;   // V fragments in flight ahead of their MFMAs (register sets)
#ifndef IDIFF_W43_LATE_AT
#define IDIFF_W43_LATE_AT 6
#endif
constexpr int F4_LATE_AT = IDIFF_W43_LATE_AT;   // the late role transforms in front of this position of its step (A/B: scripts/wino43_ab.py)
constexpr int F4_VSLOT = F4_TILES * F4_KC + 4;           // 260 = 4 mod 32: the six columns a lane group writes fall on distinct banks
constexpr int F4_STAGE = F4_NPOS * F4_VSLOT;             // 9360 floats
constexpr int F4_SCR_ROW = 7 * 4;                        // scratch row: 6 float4 + one of padding (28 dwords: rows on distinct banks)
#ifndef IDIFF_W43_SCR_UNIT
#define IDIFF_W43_SCR_UNIT 224
#endif
constexpr int F4_SCR_UNIT = IDIFF_W43_SCR_UNIT;                       // floats per unit (6 rows used): with this pitch the column reads of the
                                                         // four 16-lane groups of a ds_read_b128 fall on distinct banks as well (PMC:
                                                         // SQ_LDS_BANK_CONFLICT was 21 % of the LDS cycles at the dense pitch of 168)
constexpr int F4_SCR_WAVE = 8 * F4_SCR_UNIT;             // 1792 floats per wave
constexpr int F4_SCRATCH = 8 * F4_SCR_WAVE;              // 14336 floats
constexpr int F4_LOOP_FLOATS = 2 * F4_STAGE + F4_SCRATCH + 8;   // + 8: lane 7 of the last group reads one float4 past its unit
constexpr size_t F4_LDS_BYTES = sizeof(float) * (size_t)(F4_LOOP_FLOATS > F4_Z_FLOATS ? F4_LOOP_FLOATS : F4_Z_FLOATS);   // 132,256 B
__device__ __forceinline__ void f4_bt4(const F4Consts &k, const float4 (&d)[6], float4 (&t)[6]) {
  f4_bt(k, d[0].x, d[1].x, d[2].x, d[3].x, d[4].x, d[5].x, t[0].x, t[1].x, t[2].x, t[3].x, t[4].x, t[5].x);
  f4_bt(k, d[0].y, d[1].y, d[2].y, d[3].y, d[4].y, d[5].y, t[0].y, t[1].y, t[2].y, t[3].y, t[4].y, t[5].y);
  f4_bt(k, d[0].z, d[1].z, d[2].z, d[3].z, d[4].z, d[5].z, t[0].z, t[1].z, t[2].z, t[3].z, t[4].z, t[5].z);
  f4_bt(k, d[0].w, d[1].w, d[2].w, d[3].w, d[4].w, d[5].w, t[0].w, t[1].w, t[2].w, t[3].w, t[4].w, t[5].w);
}

// one wave exchanging data through LDS with itself: DS operations of a wave execute in order; the compiler must neither
// reorder nor cache across this point
__device__ __forceinline__ void f4_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(F4_THREADS, 2)   // two waves per SIMD: one 512-thread workgroup per CU
winograd43_kernel(const Wino43Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef IDIFF_W43_STAMP   // diagnostic build only (scripts/wino43_stamps.py): 100 MHz ticks at the phases of a workgroup's life
  const uint64_t st_start = __builtin_amdgcn_s_memrealtime();
#endif
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
  // cout half, row block, column block.  Waves w and w + 4 share a SIMD: they get DIFFERENT column blocks, so that in the tail's
  // two-part exchange (block 1 parks, block 0 adds) every SIMD has one working wave in each part instead of two or none
  const int wh = wave >> 2, wq = (wave & 3) ^ wh, wa = wq >> 1, wb = wq & 1;

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, 0x00020000);

  // ---------------------------------------------------------------- loader state
  const int lr = lane & 7, lg = lane >> 3;            // patch row / column index, unit within the wave
  const bool lact = lr < 6;
  const int unit = wave * 8 + lg, ltile = unit >> 1, lq = unit & 1;
  uint32_t v_src[6];
  {
    const int T = tile0 + ltile;
    const bool tv = lact && T < p.total_tiles;
    int img, ty, tx;
    f4_split_tile(p, tv ? T : 0, img, ty, tx);
    const int y = 4 * ty - 1 + lr, x0 = 4 * tx - 1;
    const bool yok = tv && y >= 0 && y < p.H;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = x0 + c;
      v_src[c] = (yok && xx >= 0 && xx < p.W) ? (uint32_t)(((img * p.H + y) * p.W + xx) * p.Cin + lq * 4) * 4u : F4_INVALID;
    }
  }
  float *scr_row = lds + 2 * F4_STAGE + wave * F4_SCR_WAVE + lg * F4_SCR_UNIT + lr * F4_SCR_ROW;   // my row, as the writer
  const float *scr_col = lds + 2 * F4_STAGE + wave * F4_SCR_WAVE + lg * F4_SCR_UNIT + lr * 4;       // my column, as the reader
  // V(i, c = lr) of (ltile, quad lq): slot 6 i + lr, row ltile, 16-byte half lq swapped when bit 3 of the tile is set
  const int v_dst = lr * F4_VSLOT + ltile * F4_KC + 4 * (lq ^ ((ltile >> 3) & 1));

  const F4Consts kc = {p.c_nb2, p.c_na2, p.c_nab2, p.c_a, p.c_b};
  float4 ldv[6];
  const int nsteps = p.Cin / F4_KC;
  int f_step = 0;
  auto fetch = [&]() {
    const int choff = min(f_step, nsteps - 1) * (F4_KC * 4);
#pragma unroll
    for (int c = 0; c < 6; ++c) ldv[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)v_src[c], choff, 0));
    ++f_step;
  };
  auto stage = [&](int buf) {
    float4 t[6];
    f4_bt4(kc, ldv, t);                               // along x
    if (lact) {
#pragma unroll
      for (int c = 0; c < 6; ++c) *reinterpret_cast<float4 *>(scr_row + 4 * c) = t[c];
    }
    f4_wave_lds_sync();
    float4 d[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) d[r] = *reinterpret_cast<const float4 *>(scr_col + r * F4_SCR_ROW);
    f4_bt4(kc, d, t);                                 // along y
    f4_wave_lds_sync();                               // the scratch is free again before this wave's next stage()
    if (lact) {
      float *Vd = lds + buf * F4_STAGE + v_dst;
#pragma unroll
      for (int i = 0; i < 6; ++i) *reinterpret_cast<float4 *>(Vd + 6 * i * F4_VSLOT) = t[i];
    }
  };

  floatx16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int frag = fr * F4_KC + 4 * (fh ^ ((fr >> 3) & 1));
  const int pos0 = (3 * wa) * 6 + 3 * wb;            // slot of this wave's first position
  const int a_frag = pos0 * F4_VSLOT + frag;
  // U: [step][cout tile][slot][64 cout][8 cin]; lane (fr, fh) takes cout wh * 32 + fr, the 16-byte half fh (swapped like V)
  const uint32_t u_lane = (uint32_t)(((wh * 32 + fr) * 8 + 4 * (fh ^ ((fr >> 3) & 1))) * 4);
  float4 bfr[3];
  auto load_b = [&](int pp, int step) {              // pp = 0 .. 8: position (pp / 3, pp % 3) of the block
    const int slot = pos0 + (pp / 3) * 6 + (pp % 3);
    const int soff = ((step * p.tiles_n + tile_n) * F4_NPOS + slot) * (64 * F4_KC * 4);
    bfr[pp % 3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rU, (int)u_lane, soff, 0));
  };
  auto mfma4 = [&](int pp, const float4 a) {
    const float4 b = bfr[pp % 3];
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[pp], 0, 0, 0);
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[pp], 0, 0, 0);
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[pp], 0, 0, 0);
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[pp], 0, 0, 0);
  };
  auto v_at = [&](const float *S, int pp) { return *reinterpret_cast<const float4 *>(S + ((pp / 3) * 6 + (pp % 3)) * F4_VSLOT); };
  // LAST: the final step requests nothing (no next slab, no next input) and writes no stage -- the tail would only wait for
  // loads nobody uses
  // AT: the position of the step in front of which this wave transforms and stages the NEXT step's input (0: at the start of
  // the step, "early"; 9: never, the last step).  Waves get different AT (compile-time per loop copy) so that the transforms of
  // a CU are spread over the step instead of piling up behind the barrier.
  auto compute = [&](int buf, int s, auto at, auto last) {
    constexpr bool LAST = decltype(last)::value;
    constexpr int AT = decltype(at)::value;
    const float *S = lds + buf * F4_STAGE + a_frag;
    float4 f[F4_VRING];
#pragma unroll
    for (int q = 0; q < F4_VRING; ++q) f[q] = v_at(S, q);
#pragma unroll
    for (int pp = 0; pp < 9; ++pp) {
      __builtin_amdgcn_sched_barrier(0);
      if (pp == AT && AT > 0) { stage(buf ^ 1); fetch(); __builtin_amdgcn_sched_barrier(0); }
      mfma4(pp, f[pp % F4_VRING]);
      // the register set is free once these MFMAs have read it: request the position three ahead (of the next step beyond 8)
      if (pp + 3 < 9) load_b(pp + 3, s); else if (!LAST) load_b(pp + 3 - 9, s + 1);
      if (pp + F4_VRING < 9) f[pp % F4_VRING] = v_at(S, pp + F4_VRING);
    }
  };
  auto step = [&](int s, auto at, auto last) {
    constexpr bool LAST = decltype(last)::value;
    constexpr int AT = decltype(at)::value;
    const int buf = s & 1;
    if constexpr (!LAST && AT == 0) { stage(buf ^ 1); fetch(); }
    compute(buf, s, at, last);
    __syncthreads();
  };
  auto run = [&](auto at) { for (int s = 0; s + 1 < nsteps; ++s) step(s, at, std::false_type()); };

  load_b(0, 0); load_b(1, 0); load_b(2, 0);
  fetch();
  stage(0);
  fetch();
  __syncthreads();
#ifdef IDIFF_W43_STAMP
  const uint64_t st_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (wave < F4_EARLY_WAVES) run(std::integral_constant<int, 0>());
  else if (wave < F4_MID_WAVES) run(std::integral_constant<int, F4_MID_AT>());
  else run(std::integral_constant<int, F4_LATE_AT>());
  step(nsteps - 1, std::integral_constant<int, 9>(), std::true_type());
#ifdef IDIFF_W43_STAMP
  const uint64_t st_loop1 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---------------------------------------------------------------- tail
  const idiff_epilogue &ep = p.ep;
  const bool has_ep = p.has_ep != 0;
  const int cq = tid & 15, tl = tid >> 4;            // this thread finishes channels n .. n + 3 of tile tl
  const int n = n0 + 4 * cq;
  const bool has_res = has_ep && ep.residual != nullptr;
  const bool scaled = has_ep && (ep.out_scale != 1.f || ep.rowscale != nullptr);
  const int act = has_ep ? ep.act : (int)IDIFF_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)ep.residual, 0, (int)p.res_bytes, 0x00020000);
  const int ld_res = (int)ep.ld_residual;
  // this thread's output addresses and per-tile epilogue operands: formed again at the start of each output phase (about 30
  // instructions) rather than kept in registers under the accumulators
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

#ifdef IDIFF_W43_STAMP
  const bool want_stats = false;            // epilogue.colstats carries the stamp buffer in this build
#else
  const bool want_stats = has_ep && ep.colstats != nullptr;
#endif
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};    // column sums of this thread's 16 pixels x 4 channels
  // accumulator register `reg` of lane l is tile row (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), cout wh * 32 + (l & 31)
  float *zbase = lds + (size_t)(4 * (lane >> 5)) * 2 * F4_COUT + wh * 32 + (lane & 31);
#ifdef IDIFF_W43_STAMP
  uint64_t st_tail[4] = {0, 0, 0, 0};
#endif
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    // z_{i,b} for b = 2 pass, 2 pass + 1 and the three rows i of this wave's block: column block 1 parks, block 0 adds
    if (pass) __syncthreads();                       // every z of the first pass has been read
    // the residual operands of this pass's two output columns are requested HERE, in front of the exchange that covers their
    // latency (requested at their point of use they cost 3 us per workgroup: four exposed round trips)
    float4 res[2][4];
    if (has_res) {
      prep();
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          res[bb][a] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)roff, (a * p.W + 2 * pass + bb) * ld_res * 4, 0));
    }
    if (wb == 1) {
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float m3 = acc[ii * 3][reg], m4 = acc[ii * 3 + 1][reg], m5 = acc[ii * 3 + 2][reg];
          const int trow = (reg & 3) + 8 * (reg >> 2);
          float *zp = zbase + (((3 * wa + ii) * F4_TILES + trow) * 2) * F4_COUT;
          if (pass == 0) { zp[0] = m3 + m4; zp[F4_COUT] = F4_b * (m3 - m4); }
          else { zp[0] = F4_b2 * (m3 + m4); zp[F4_COUT] = fmaf(F4_b3, m3 - m4, m5); }
        }
    }
    __syncthreads();
    if (wb == 0) {
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float m0 = acc[ii * 3][reg], m1 = acc[ii * 3 + 1][reg], m2 = acc[ii * 3 + 2][reg];
          const int trow = (reg & 3) + 8 * (reg >> 2);
          float *zp = zbase + (((3 * wa + ii) * F4_TILES + trow) * 2) * F4_COUT;
          if (pass == 0) { zp[0] += m0 + (m1 + m2); zp[F4_COUT] = fmaf(F4_a, m1 - m2, zp[F4_COUT]); }
          else { zp[0] = fmaf(F4_a2, m1 + m2, zp[0]); zp[F4_COUT] = fmaf(F4_a3, m1 - m2, zp[F4_COUT]); }
        }
    }
    __syncthreads();
#ifdef IDIFF_W43_STAMP
    st_tail[2 * pass] = __builtin_amdgcn_s_memrealtime();       // z of this pass exchanged
#endif
    __builtin_amdgcn_sched_barrier(0);
    prep();
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
      // the four rows are stored together from registers nothing writes again before the next column: a store whose data
      // registers were recomputed for the next row a few instructions later wrote that row's first component in some lanes
      const int so = b * p.Cout * 4, rp = p.W * p.Cout * 4;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, make_float4(y[a][0], y[a][1], y[a][2], y[a][3])), rO, (int)ooff,
                                               so + a * rp, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef IDIFF_W43_STAMP
    st_tail[2 * pass + 1] = __builtin_amdgcn_s_memrealtime();   // outputs of this pass stored (issued)
#endif
  }
  if (want_stats) {
    // Per-tile partial sums meet in LDS ([32 tiles][64 channels][2] fp64 = 32 KB over the dead z area); one thread per (sample
    // or workgroup, channel) adds the tiles up in a fixed order.  Layout of epilogue.colstats as for the 2x2 form:
    // [samples][nsplit][Cout][2], nsplit = tiles_per_img / 32 workgroups per sample, or -- maps of fewer than 32 tiles -- one slot
    // per sample with 32 / tiles_per_img whole samples per workgroup.
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
    for (int o = tid; o < slots * F4_COUT; o += F4_THREADS) {
      const int smp = o / F4_COUT, ch = o - smp * F4_COUT;
      const int64_t slot = (int64_t)tile_m * slots + smp;                           // sample, or (sample, split) = workgroup row
      if (slots > 1 && slot >= p.B) continue;
      double a = 0.0, b = 0.0;
      for (int k = 0; k < per; ++k) { a += red[((smp * per + k) * F4_COUT + ch) * 2]; b += red[((smp * per + k) * F4_COUT + ch) * 2 + 1]; }
      double *dst = ep.colstats + (slot * p.Cout + n0 + ch) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
#ifdef IDIFF_W43_STAMP
  if (has_ep && ep.colstats && tid == 0) {
    uint64_t *st = reinterpret_cast<uint64_t *>(ep.colstats) + 10 * (int64_t)blockIdx.x;
    st[6] = st_tail[0]; st[7] = st_tail[1]; st[8] = st_tail[2]; st[9] = st_tail[3];
    st[0] = st_start; st[1] = st_loop0; st[2] = st_loop1; st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
    st[5] = (uint64_t)nsteps;
  }
#endif
}

// U = G g G^T in fp64, rounded once; g[ky][kx] = wt[cout][ky][kx][cin] (the K-contiguous panel of the direct kernel)
__global__ void winograd43_pack_kernel(const float *wt, float *u, int Cin, int Cout) {
  // G row of point p: (1, p, p^2) / N(p), N(p) = prod over the other finite points (p - q); N(0) = a^2 b^2 = 1,
  // N(+-a) = 2 a^2 (a^2 - b^2), N(+-b) = 2 b^2 (b^2 - a^2); the point at infinity picks g[2]
  const double a = F4_A, b = F4_B, na = 1.0 / (2.0 * a * a * (a * a - b * b)), nb = 1.0 / (2.0 * b * b * (b * b - a * a)), n0 = 1.0 / (a * a * b * b);
  const double G[6][3] = {{n0, 0.0, 0.0},
                          {na, a * na, a * a * na},
                          {na, -a * na, a * a * na},
                          {nb, b * nb, b * b * nb},
                          {nb, -b * nb, b * b * nb},
                          {0.0, 0.0, 1.0}};
  const int64_t total = (int64_t)Cin * Cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)wt[((int64_t)cout * 9 + ky * 3 + kx) * Cin + cin];
    double gg[6][3];   // G g
    for (int i = 0; i < 6; ++i)
      for (int kx = 0; kx < 3; ++kx) gg[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
    const int s = cin / F4_KC, c8 = cin % F4_KC, nt = cout / F4_COUT, co = cout % F4_COUT;
    const int slot8 = 4 * ((c8 >> 2) ^ ((co >> 3) & 1)) + (c8 & 3);
    float *dst = u + ((int64_t)(s * (Cout / F4_COUT) + nt) * F4_NPOS * 64 + co) * F4_KC + slot8;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const double v = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
        dst[(int64_t)(6 * i + j) * 64 * F4_KC] = (float)v;
      }
  }
}

}  // namespace

IDIFF_API int idiff_conv2d_winograd43_ok(int B, int H, int W, int Cin, int Cout) {
  if (idiff::option(idiff::OPT_NO_WINOGRAD) || idiff::option(idiff::OPT_NO_WINO43)) return 0;
  return f4_geometry_ok(B, H, W, Cin, Cout) ? 1 : 0;
}

// nsplit of epilogue.colstats ([samples, nsplit, Cout, 2]) or 0 when this form cannot produce the statistics
IDIFF_API int idiff_conv2d_winograd43_colstats_split(int B, int H, int W, int Cin, int Cout) {
  if (!idiff_conv2d_winograd43_ok(B, H, W, Cin, Cout) || idiff::option(idiff::OPT_NO_COLSTATS)) return 0;
  const int tpi = (H / 4) * (W / 4);
  if (tpi % F4_TILES == 0) return tpi / F4_TILES;
  return (F4_TILES % tpi == 0) ? 1 : 0;              // whole samples per workgroup
}

IDIFF_API int64_t idiff_winograd43_weight_floats(int Cin, int Cout) { return (int64_t)36 * Cin * Cout; }

IDIFF_API int idiff_winograd43_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream) {
  using namespace idiff;
  if (Cin <= 0 || Cout <= 0 || Cin % F4_KC || Cout % F4_COUT)
    return fail("winograd43_pack: Cin must be a multiple of %d and Cout of %d (got %d, %d)", F4_KC, F4_COUT, Cin, Cout);
  if (!wt || !u) return fail("winograd43_pack: null pointer");
  const int64_t total = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(winograd43_pack_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wt, u, Cin, Cout);
  return launch_status("winograd43_pack");
}

IDIFF_API int idiff_conv2d_winograd43_colstats_split(int B, int H, int W, int Cin, int Cout);
IDIFF_API int idiff_conv2d_winograd43_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                          const idiff_epilogue *ep, void *stream) {
  using namespace idiff;
  if (B == 0) return 0;
  if (!f4_geometry_ok(B, H, W, Cin, Cout))
    return fail("conv2d_winograd43: geometry B=%d H=%d W=%d Cin=%d Cout=%d not supported (ask idiff_conv2d_winograd43_ok)", B, H, W, Cin, Cout);
  if (!x || !u || !out) return fail("conv2d_winograd43: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)out & 15)) return fail("conv2d_winograd43: x, u and out must be 16-byte aligned");
#ifndef IDIFF_W43_STAMP
  if (ep && ep->colstats && idiff_conv2d_winograd43_colstats_split(B, H, W, Cin, Cout) <= 0)
    return fail("conv2d_winograd43: colstats needs whole workgroups per sample or whole samples per workgroup "
                "(ask idiff_conv2d_winograd43_colstats_split)");
#endif
  if (ep && (ep->rowbias || ep->rowscale) && ep->rows_per_group != H * W)
    return fail("conv2d_winograd43: per-row-group bias / scale only per image (rows_per_group = H * W = %d, got %d)", H * W, ep->rows_per_group);
  if (ep && ep->residual && (((uintptr_t)ep->residual & 15) || ep->ld_residual % 4 || ep->ld_residual < Cout || ep->ld_residual > 0x7fffffff / 4))
    return fail("conv2d_winograd43: residual must be 16-byte aligned with a row pitch >= Cout that is a multiple of 4");
  const int64_t res_bytes = (ep && ep->residual) ? (int64_t)B * H * W * ep->ld_residual * 4 : 0;
  if (res_bytes >= F4_X_LIMIT) return fail("conv2d_winograd43: residual beyond one buffer descriptor");
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
  static AttrGuard guard;
  const void *fn = reinterpret_cast<const void *>(winograd43_kernel);
  if (int rc = set_dynamic_lds_once(guard, &fn, 1, (int)F4_LDS_BYTES, "conv2d_winograd43")) return rc;
  hipLaunchKernelGGL(winograd43_kernel, dim3(p.tiles_m * p.tiles_n), dim3(F4_THREADS), F4_LDS_BYTES, (hipStream_t)stream, p);
  return launch_status("conv2d_winograd43");
}

