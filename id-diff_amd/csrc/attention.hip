// Single-head self-attention over 256 tokens with the logits kept on chip (reference: models/layerspp.py:75-91 AttnBlockpp.forward,
// models/BeatGANsblocks.py:466-491 QKVAttentionLegacy):
//     w = softmax_j(q_i . k_j * scale),   out_i = sum_j w_ij v_j (+ bias_v)
// replacing three launches (Q K^T -> [B, 256, 256] logits in HBM, row softmax, P V) and 2.3 GB of logits traffic per block at B = 2240.
//
// Arithmetic: both contractions on v_mfma_f32_16x16x32_f16 with every fp32 operand as a PAIR of fp16 values (hi = fp16(s v),
// lo = fp16(s v - hi), 22 significand bits; products hi hi + hi lo + lo hi, fp32 accumulation), the same form as igemm.hip's SPLIT == 2
// and winograd43h.hip.  s is a power of two per tensor: for Q / K and for V the caller's (derived once per weight from the projections'
// row norms: their input is a GroupNorm's output), for the softmax rows 2^10 (they lie in [0, 1]).  Softmax itself in fp32 (expf).
// Range: |s q|, |s k|, |s v| must stay below 65504 or the outputs are NaN -- the drivers re-run such a point on the fp32 route.
//
// Work split: a workgroup = (sample, 128 queries), eight waves of 16 queries; all 256 keys / values of the sample pass through LDS once per
// workgroup in 32-wide chunks (K: 32 channels of all keys; V^T: 32 keys of all channels), converted to pairs on the way in.
//   phase 1  S^T[key, q] = sum_c K[key, c] Q[q, c]      A = K chunk (LDS), B = Q (registers, straight from global), 16 key blocks
//   softmax  the 256 logits of query q sit in lanes q, q + 16, q + 32, q + 48 (64 registers each): in-lane reductions + two shuffles;
//            the probabilities are cut into pairs IN PLACE and are already the B operand of phase 2 (a matrix instruction's k index may
//            be any fixed permutation of the keys as long as A uses the same one: k slot (g, j) = key 16 (2 m + j / 4) + 4 g + j % 4)
//   phase 2  O^T[c, q] = sum_key V^T[c, key] P[key, q]   A = V^T chunk (LDS, keys stored in that slot order), B = P (registers)
// LDS: two chunk buffers of 32 KB (hi plane | lo plane, 64 B per row, 16-byte pieces XOR-swizzled by at_swz(row): conflict-free
// ds_read_b128); 64 KB per workgroup.  The operands' conversion to pairs is vector-ALU work every workgroup of a sample repeats, so a
// workgroup takes as many queries as the register file allows (8 waves x 224 registers).
#include "common.h"

namespace {

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int AT_T = 256;                       // tokens
constexpr int AT_BQ = 128;                      // queries per workgroup (eight waves of 16)
constexpr int AT_THREADS = 512;
constexpr int AT_PLANE = AT_T * 64;             // bytes: 256 rows x 32 halves
constexpr int AT_CHUNK = 2 * AT_PLANE;          // hi | lo
constexpr int AT_LDS = 2 * AT_CHUNK;            // two buffers: 65,536 B
constexpr float AT_PSCALE = 1024.0f;
#ifndef IDIFF_AT_LEAD
#define IDIFF_AT_LEAD 3
#endif
constexpr int AT_LEAD = IDIFF_AT_LEAD;         // operand blocks requested from LDS ahead of their matrix instructions

struct AttnParams {
  const float *qk;        // [B * 256, ld_qk]: q in columns [0, C), k in [C, 2C)
  const float *vt;        // [B, C, 256]
  const float *bias_v;    // [C] or null
  const float *s_qk;      // device {s, 1 / s}
  const float *s_v;       // device {s, 1 / s}
  float *out;             // [B * 256, C]
  int64_t ld_qk;
  int B;
  float scale;            // softmax scale (C^-1/2)
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Where the 16-byte piece `c` of a 64-byte operand row lives: c ^ at_swz(row).  A ds_read_b128 is served in four groups of 16 lanes that are
// NOT consecutive lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; MI355X_MICROARCH.md, LDS): with the operand layout of the 16x16x32
// instruction (lane = row + 16 piece) a group holds rows 0-3 and 12-15 at one piece and rows 4-11 at its neighbour, so the four rows that
// share their banks (r, r + 4, r + 8, r + 12) must be shifted by 0, 3, 2, 1 -- the plain (row >> 2) & 3 leaves every read two-way conflicted.
__device__ __forceinline__ int at_swz(int row) { return (0 - (row >> 2)) & 3; }

// s v -> (hi, lo) for two values: one packed conversion, the residual s v - hi in ONE mixed-precision instruction per value (fp32 x fp32 +
// fp16, exact), one more packed conversion -- five vector instructions per two values where the plain form takes eight
__device__ __forceinline__ void cut2(const float a, const float b, const float s, uint32_t &hi, uint32_t &lo) {
  const f2 x = {a * s, b * s};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, h2));
  f2 rest;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest.x) : "v"(hi), "v"(x.x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest.y) : "v"(hi), "v"(x.y));
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, h2));
}
typedef uint32_t uintx2 __attribute__((ext_vector_type(2)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cut4(const float4 v, const float s, uintx2 &hi, uintx2 &lo) {
  uint32_t h0, l0, h1, l1;
  cut2(v.x, v.y, s, h0, l0);
  cut2(v.z, v.w, s, h1, l1);
  hi = uintx2{h0, h1}; lo = uintx2{l0, l1};
}

template <int C>
__global__ void __launch_bounds__(AT_THREADS, 2)
attention256_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char at_lds[];
  constexpr int NKC = C / 32;                   // channel chunks of phase 1
  constexpr int NVC = AT_T / 32;                // key chunks of phase 2
  constexpr int NCB = C / 16;                   // channel blocks of the output
  static_assert(C % 32 == 0 && C >= 64 && C <= 256, "channels: a multiple of 32 in [64, 256]");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the two workgroups of a sample read the same K and V: give them consecutive slots on ONE XCD (workgroups are dealt round-robin
  // over the 8 XCDs, each with its own L2), so that one of them pulls the sample's rows from HBM and the others find them in that L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int b = bid >> 1, q0 = (bid & 1) * AT_BQ + wave * 16;
  const int l15 = lane & 15, g = lane >> 4;
  const float s_qk = p.s_qk[0], inv_qk = p.s_qk[1], s_v = p.s_v[0], inv_v = p.s_v[1];

  // ---- staging: thread (row r0 = tid / 8, piece = tid % 8) moves 16 bytes = 4 fp32 of rows r0, r0 + 64, ... per chunk
  const int piece = tid & 7, r0 = tid >> 3;
  const float *k_src = p.qk + ((int64_t)b * AT_T + r0) * p.ld_qk + C + 4 * piece;          // + 32 kc, rows step 64 ld
  const float *v_src = p.vt + ((int64_t)b * C + r0) * AT_T + 4 * piece;                     // + 32 m, rows step 64 * 256
  // st: the chunk of the NEXT step (requested one step ago), st2: the one after it (requested at the start of this step): a request has
  // about 1.75 steps to arrive.  With one step of lead the waves sat at s_waitcnt for two thirds of their life (SQ_WAIT_ANY 66 %).
  float4 st[4], st2[4];
  // Branch-free on purpose: behind a conditional request the compiler's wait-count bookkeeping falls back to "wait for everything", which
  // exposes the latency of the request made at the start of the step.  Requests past the last chunk (and V rows beyond C) re-read a valid
  // address and are never stored.
  auto fetch = [&](int step, float4 (&st)[4]) __attribute__((always_inline)) {   // chunk `step`: K chunks 0 .. NKC - 1, then V chunks
    const int sc = step < NKC + NVC ? step : NKC + NVC - 1;
    const bool is_k = sc < NKC;
    const float *src = is_k ? k_src + 32 * sc : v_src + 32 * (sc - NKC);
    const int64_t stride = is_k ? 64 * p.ld_qk : (int64_t)64 * AT_T;
#pragma unroll
    for (int r = 0; r < 4; ++r) st[r] = *reinterpret_cast<const float4 *>(src + (is_k ? r : r % (C / 64)) * stride);
  };
  // K rows: pieces 2 c16, 2 c16 + 1 make the 16-byte piece c16 (channels 8 c16 .. 8 c16 + 7).  V^T rows: the row's 32 keys are stored in
  // matrix-instruction slot order: key 16 h + 4 gq + i (piece = 4 h + gq) -> 16-byte piece gq, half h.
  const int k_off = (((piece >> 1) ^ at_swz(r0)) << 4) + (piece & 1) * 8;
  const int v_off = (((piece & 3) ^ at_swz(r0)) << 4) + (piece >> 2) * 8;
  // pins a requested value to the point where it is consumed: the conversions are pure register arithmetic, and without this the compiler
  // moves them (and with them the wait for the request) up to right behind the request, a whole step early
  auto pin = [](float4 &v) __attribute__((always_inline)) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  auto stash = [&](int step, float4 (&st)[4]) __attribute__((always_inline)) {
    char *dst = at_lds + (step & 1) * AT_CHUNK + r0 * 64 + (step < NKC ? k_off : v_off);
    const float s = step < NKC ? s_qk : s_v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (step < NKC || r < C / 64) {            // (row + 64 r) >> 2 & 3 == r0 >> 2 & 3: the swizzle does not change with r
        pin(st[r]);
        uintx2 hi, lo;
        cut4(st[r], s, hi, lo);
        *reinterpret_cast<uintx2 *>(dst + r * 64 * 64) = hi;
        *reinterpret_cast<uintx2 *>(dst + r * 64 * 64 + AT_PLANE) = lo;
      }
    }
  };

  // ---- this wave's queries as the B operand of phase 1: lane (query l15, k group g) holds channels 32 kc + 8 g .. + 7
  const float *q_src = p.qk + ((int64_t)b * AT_T + q0 + l15) * p.ld_qk + 8 * g;
  floatx4 sacc[16];                              // S^T: key block kb, rows 4 g + i, column l15; later the probabilities as pairs
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) sacc[kb] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int a_off = l15 * 64;                    // operand rows of block rb: row 16 rb + l15, piece g ^ swizzle(row)
  auto a_addr = [&](int buf, int rb) __attribute__((always_inline)) {
    const int row = 16 * rb + l15;
    return at_lds + buf * AT_CHUNK + rb * 16 * 64 + a_off + ((g ^ at_swz(row)) << 4);
  };

  halfx8 qh, ql;
  float4 qa, qb, qa2, qb2;                       // the queries of the next chunk and of the one after it, like st / st2
  auto q_fetch = [&](int kc, float4 &a, float4 &c) __attribute__((always_inline)) {
    const int kk = kc < NKC ? kc : NKC - 1;      // branch-free, as fetch
    a = *reinterpret_cast<const float4 *>(q_src + 32 * kk); c = *reinterpret_cast<const float4 *>(q_src + 32 * kk + 4);
  };
  auto q_cut = [&]() __attribute__((always_inline)) {
    uintx2 h0, l0, h1, l1;
    cut4(qa, s_qk, h0, l0); cut4(qb, s_qk, h1, l1);
    qh = __builtin_bit_cast(halfx8, uintx4{h0.x, h0.y, h1.x, h1.y});
    ql = __builtin_bit_cast(halfx8, uintx4{l0.x, l0.y, l1.x, l1.y});
  };
  fetch(0, st);
  q_fetch(0, qa, qb);
  stash(0, st);
  q_cut();
  fetch(1, st);
  q_fetch(1, qa, qb);
  __syncthreads();
  // ---- phase 1.  The scheduling fences keep the compiler from sinking the requests down to their first use.  Two steps per trip: the
  // register sets swap ROLES (a copy st = st2 at the end of a step would wait for the request made at its start).
  auto qk_step = [&](int kc, float4 (&cur)[4], float4 (&far)[4], float4 &ca, float4 &cb, float4 &fa, float4 &fb) __attribute__((always_inline)) {
    fetch(kc + 2, far);
    q_fetch(kc + 2, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    const int buf = kc & 1;
    // the conversion of the next chunk goes in front of the last quarter of the matrix block: its vector instructions issue in the
    // shadow of the matrix instructions around them (a 16x16x32 instruction holds the SIMD's issue for 8 of its 16 cycles)
    const halfx8 qh0 = qh, ql0 = ql;
    // the K fragments of block kb + 3 are requested from LDS as soon as block kb's registers are free: three blocks (144 matrix-core cycles)
    // of lead for an LDS round trip under load; the fences keep that distance (left alone the compiler reads one block ahead)
    halfx8 ah[AT_LEAD], al[AT_LEAD];
#pragma unroll
    for (int i = 0; i < AT_LEAD; ++i) {
      const char *ap = a_addr(buf, i);
      ah[i] = *reinterpret_cast<const halfx8 *>(ap); al[i] = *reinterpret_cast<const halfx8 *>(ap + AT_PLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      if (kb == 12) {
        stash(kc + 1, cur);
        if (kc + 1 < NKC) {
          pin(ca); pin(cb);
          uintx2 h0, l0, h1, l1;
          cut4(ca, s_qk, h0, l0); cut4(cb, s_qk, h1, l1);
          qh = __builtin_bit_cast(halfx8, uintx4{h0.x, h0.y, h1.x, h1.y});
          ql = __builtin_bit_cast(halfx8, uintx4{l0.x, l0.y, l1.x, l1.y});
        }
      }
      const halfx8 kh = ah[kb % AT_LEAD], kl = al[kb % AT_LEAD];
      sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh0, sacc[kb], 0, 0, 0);
      sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql0, sacc[kb], 0, 0, 0);
      sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh0, sacc[kb], 0, 0, 0);
      if (kb + AT_LEAD < 16) {
        const char *ap = a_addr(buf, kb + AT_LEAD);
        ah[kb % AT_LEAD] = *reinterpret_cast<const halfx8 *>(ap); al[kb % AT_LEAD] = *reinterpret_cast<const halfx8 *>(ap + AT_PLANE);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  static_assert(NKC % 2 == 0 && NVC % 2 == 0, "the register sets alternate: an even number of steps per phase");
#pragma unroll 1
  for (int kc = 0; kc < NKC; kc += 2) {
    qk_step(kc, st, st2, qa, qb, qa2, qb2);
    qk_step(kc + 1, st2, st, qa2, qb2, qa, qb);
  }

  // ---- softmax over the 256 keys of query l15 (this lane: 64 of them; lanes l15 + 16 g' the rest)
  halfx8 ph[NVC], pl[NVC];
  {
    // logits = S / s^2 * scale; exp(x) = 2^(x log2 e) on v_exp_f32 (1 ulp; the rounding of the product matters only where the result is
    // tiny: 6e-8 |x| relative)
    const float sc = p.scale * inv_qk * inv_qk * 1.44269504088896340736f;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) m = fmaxf(m, sacc[kb][i]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f((sacc[kb][i] - m) * sc); sacc[kb][i] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = AT_PSCALE / sum;
#pragma unroll
    for (int m2 = 0; m2 < NVC; ++m2) {
      uintx2 h0, l0, h1, l1;
      cut4(make_float4(sacc[2 * m2][0], sacc[2 * m2][1], sacc[2 * m2][2], sacc[2 * m2][3]), inv, h0, l0);
      cut4(make_float4(sacc[2 * m2 + 1][0], sacc[2 * m2 + 1][1], sacc[2 * m2 + 1][2], sacc[2 * m2 + 1][3]), inv, h1, l1);
      ph[m2] = __builtin_bit_cast(halfx8, uintx4{h0.x, h0.y, h1.x, h1.y});
      pl[m2] = __builtin_bit_cast(halfx8, uintx4{l0.x, l0.y, l1.x, l1.y});
    }
  }

  // ---- phase 2
  floatx4 oacc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) oacc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto pv_step = [&](int m2, float4 (&cur)[4], float4 (&far)[4]) __attribute__((always_inline)) {
    const int step = NKC + m2, buf = step & 1;
    fetch(step + 2, far);
    halfx8 ah[AT_LEAD], al[AT_LEAD];
#pragma unroll
    for (int i = 0; i < AT_LEAD; ++i) {
      const char *ap = a_addr(buf, i);
      ah[i] = *reinterpret_cast<const halfx8 *>(ap); al[i] = *reinterpret_cast<const halfx8 *>(ap + AT_PLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (cb == (3 * NCB) / 4 && m2 + 1 < NVC) stash(step + 1, cur);
      const halfx8 vh = ah[cb % AT_LEAD], vl = al[cb % AT_LEAD];
      oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[m2], oacc[cb], 0, 0, 0);
      oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[m2], oacc[cb], 0, 0, 0);
      oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[m2], oacc[cb], 0, 0, 0);
      if (cb + AT_LEAD < NCB) {
        const char *ap = a_addr(buf, cb + AT_LEAD);
        ah[cb % AT_LEAD] = *reinterpret_cast<const halfx8 *>(ap); al[cb % AT_LEAD] = *reinterpret_cast<const halfx8 *>(ap + AT_PLANE);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (m2 + 1 < NVC) __syncthreads();
  };
#pragma unroll
  for (int m2 = 0; m2 < NVC; m2 += 2) {
    pv_step(m2, st, st2);
    pv_step(m2 + 1, st2, st);
  }

  // ---- output: lane (query l15, g) holds channels 16 cb + 4 g + i
  const float descale = inv_v * (1.0f / AT_PSCALE);
  float *o = p.out + ((int64_t)b * AT_T + q0 + l15) * C + 4 * g;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    float4 y = make_float4(oacc[cb][0] * descale, oacc[cb][1] * descale, oacc[cb][2] * descale, oacc[cb][3] * descale);
    if (p.bias_v) {
      const float4 bv = *reinterpret_cast<const float4 *>(p.bias_v + 16 * cb + 4 * g);
      y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
    }
    *reinterpret_cast<float4 *>(o + 16 * cb) = y;
  }
}

template <int C>
int launch_attention(const AttnParams &p, hipStream_t st) {
  static idiff::AttrGuard guard;
  const void *fn = reinterpret_cast<const void *>(attention256_kernel<C>);
  if (int rc = idiff::set_dynamic_lds_once(guard, &fn, 1, AT_LDS, "attention256")) return rc;
  hipLaunchKernelGGL(attention256_kernel<C>, dim3(p.B * (AT_T / AT_BQ)), dim3(AT_THREADS), AT_LDS, st, p);
  return idiff::launch_status("attention256");
}

}  // namespace

IDIFF_API int idiff_attention256_ok(int B, int tokens, int C) {
  if (idiff::option(idiff::OPT_NO_PAIRS) || idiff::option(idiff::OPT_NO_SPLIT) || idiff::option(idiff::OPT_NO_FUSED_ATTN)) return 0;
  return (B > 0 && B <= (1 << 20) && tokens == AT_T && (C == 128 || C == 256)) ? 1 : 0;
}

IDIFF_API int idiff_attention256_f32(const float *qk, int64_t ld_qk, const float *vt, const float *bias_v, const float *s_qk, const float *s_v,
                                     float *out, int B, int tokens, int C, float scale, void *stream) {
  using namespace idiff;
  if (B == 0) return 0;
  if (tokens != AT_T || (C != 128 && C != 256) || B < 0 || B > (1 << 20))
    return fail("attention256: tokens must be 256 and channels 128 or 256 (got %d tokens, %d channels, batch %d)", tokens, C, B);
  if (!qk || !vt || !out || !s_qk || !s_v) return fail("attention256: null pointer");
  if (ld_qk < 2 * C || ld_qk % 4) return fail("attention256: the q|k row pitch must be >= 2 C and a multiple of 4 (got %lld)", (long long)ld_qk);
  if (((uintptr_t)qk & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 15) || (bias_v && ((uintptr_t)bias_v & 15)))
    return fail("attention256: qk, vt, bias_v and out must be 16-byte aligned");
  AttnParams p = {qk, vt, bias_v, s_qk, s_v, out, ld_qk, B, scale};
  return C == 256 ? launch_attention<256>(p, (hipStream_t)stream) : launch_attention<128>(p, (hipStream_t)stream);
}
