// 3x3 / stride 1 / pad 1 convolution to a HANDFUL of output channels (the 128 -> 3 image heads of the score networks,
// models/ncsnpp.py:397-399, models/ddpm.py:138, models/BeatGANsUNET.py `out`), NHWC in, NHWC out.
//
// On the matrix cores this layer pads its 3 output channels to a 32-wide MFMA column: 10x the useful multiplications,
// 1.57 ms for [2240,32,32,128] -> 3 (10 TFLOP/s, 0.76 TB/s) although it only has to read its input once (1.17 GB, 0.2 ms).
// Here it runs on the vector ALUs with the reduction over channels spread across the 64 lanes of a wave:
//   * a wave owns one image row; lane l holds channels 2l, 2l+1 of the 3x3 window around the current pixel (18 registers)
//     and its 2 x 9 x COUT filter taps (54 registers); each pixel costs 9 x COUT packed fmas (v_pk_fma_f32: no MFMA runs
//     beside them here) per lane and one new column (three 8-byte loads per lane, 512 contiguous bytes per
//     wave-instruction; the columns of the next eight pixels are requested together);
//   * per eight pixels the 8 x COUT per-lane partial sums are folded across the wave by a halving butterfly (xor 32, 16, 8
//     keep half of the pixels each, xor 4, 2, 1 finish): 30 shuffle-adds per eight pixels instead of 6 per value;
//   * every input row is read by the three waves of the rows around it, which run next to each other (one workgroup =
//     four consecutive rows), so the re-reads are L2 / L1 hits.
// Measured: [2240,32,32,128] -> 3 in 0.51 ms (1.57 ms on the MFMA column); without the fmas the same kernel streams its
// input at 5.0 TB/s (0.235 ms), the 27 packed fmas per pixel and lane cost the rest (they do not overlap the loads of
// the same wave, and three waves per SIMD are not enough to hide them all).
// Epilogue: bias, activation, out_scale, per-row-group scale (the -1/std of the score function).  Anything else (residual,
// per-group bias, column statistics, other channel counts) stays on the implicit-GEMM path: conv3x3_narrow_ok().
#include "common.h"

namespace {

typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
constexpr uint32_t NARROW_OOB = 0xFFFFFFF0u;
constexpr int NPIX = 8;            // pixels per butterfly
constexpr int ROWS_PER_WAVE = 1;   // image rows a wave walks with its filter taps in registers (2 and 4 measured no faster)

struct NarrowParams {
  const float *x, *wt;
  float *out;
  int B, H, W, Cin;
  uint32_t x_bytes;
  idiff_epilogue ep;
  int has_ep;
};

template <int COUT>
__global__ void __launch_bounds__(256)
conv3x3_narrow_kernel(const NarrowParams p) {
  const int lane = threadIdx.x & 63;
  const int row_first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;      // wave-uniform
  const int nrows = p.B * p.H;
  if (row_first >= nrows) return;                              // no workgroup barrier below
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);

  floatx2 w[COUT][9];            // this lane's two channels of every tap: v_pk_fma_f32 operands
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float2 v = *reinterpret_cast<const float2 *>(p.wt + ((int64_t)co * 9 + t) * p.Cin + 2 * lane);
      w[co][t] = (floatx2){v.x, v.y};
    }
  const idiff_epilogue &ep = p.ep;
  const int pix_l = lane >> 3, ch_l = lane & 7;                // after the butterfly: this lane holds pixel pix_l, all channels
  float bias = 0.f;
  if (p.has_ep && ep.bias && ch_l < COUT) bias = ep.bias[ch_l];
  const bool has_rs = p.has_ep && ep.rowscale != nullptr;
  const bool row_groups = has_rs && ep.rows_per_group % p.W == 0;      // a row group never ends inside an image row
  const uint32_t px_bytes = (uint32_t)p.Cin * 4u;

  for (int row_id = row_first; row_id < min(row_first + ROWS_PER_WAVE, nrows); ++row_id) {
    const int b = row_id / p.H, y = row_id - b * p.H;
    // byte offset of (row y + ky - 1, column 0, this lane's channels), or out of range for the rows of the zero padding
    uint32_t rbase[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      rbase[ky] = (yy >= 0 && yy < p.H) ? (uint32_t)((b * p.H + yy) * p.W) * (uint32_t)p.Cin * 4u + (uint32_t)lane * 8u : NARROW_OOB;
    }
    // the column offset is wave-uniform and rides in the scalar offset (not part of the range check: a padding row stays
    // out of range); a column beyond the row is replaced by an out-of-range offset as a whole
    auto load_col = [&](int xx, floatx2 (&dst)[3]) {
      const bool ok = xx >= 0 && xx < p.W;                     // wave-uniform
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const uintx2 v = __builtin_amdgcn_raw_buffer_load_b64(rX, (int)(ok ? rbase[ky] : NARROW_OOB), ok ? xx * (int)px_bytes : 0, 0);
        dst[ky] = __builtin_bit_cast(floatx2, v);
      }
    };
    float row_scale = p.has_ep ? ep.out_scale : 1.f;
    if (row_groups) row_scale *= ep.rowscale[(int)(((int64_t)row_id * p.W) / ep.rows_per_group)];

    floatx2 cols[NPIX + 2][3];   // columns x0 - 1 .. x0 + NPIX of the three rows: pixel q reads cols[q .. q + 2]
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) cols[0][ky] = (floatx2){0.f, 0.f};
    load_col(0, cols[1]);
    for (int x0 = 0; x0 < p.W; x0 += NPIX) {
      // (requesting the next eight columns before the butterfly instead measured 12 % slower: 148 registers)
#pragma unroll
      for (int q = 0; q < NPIX; ++q) load_col(x0 + 1 + q, cols[q + 2]);
      float part[NPIX][COUT];
#pragma unroll
      for (int q = 0; q < NPIX; ++q) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          floatx2 a = (floatx2){0.f, 0.f};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = __builtin_elementwise_fma(cols[q + kx][ky], w[co][ky * 3 + kx], a);
          part[q][co] = a.x + a.y;
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) { cols[0][ky] = cols[NPIX][ky]; cols[1][ky] = cols[NPIX + 1][ky]; }
      // halving butterfly: after the stages xor 32 / 16 / 8 a lane holds the COUT partial sums of pixel (lane >> 3)
      float s4[4][COUT], s2[2][COUT], s1[COUT];
      {
        const bool hi = (lane & 32) != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float give = hi ? part[q][co] : part[q + 4][co];
            const float keep = hi ? part[q + 4][co] : part[q][co];
            s4[q][co] = keep + __shfl_xor(give, 32, 64);
          }
      }
      {
        const bool hi = (lane & 16) != 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float give = hi ? s4[q][co] : s4[q + 2][co];
            const float keep = hi ? s4[q + 2][co] : s4[q][co];
            s2[q][co] = keep + __shfl_xor(give, 16, 64);
          }
      }
      {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float give = hi ? s2[0][co] : s2[1][co];
          const float keep = hi ? s2[1][co] : s2[0][co];
          s1[co] = keep + __shfl_xor(give, 8, 64);
        }
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        s1[co] += __shfl_xor(s1[co], 4, 64);
        s1[co] += __shfl_xor(s1[co], 2, 64);
        s1[co] += __shfl_xor(s1[co], 1, 64);
      }
      // lane (pixel << 3 | channel) finishes and stores one value: NPIX x COUT contiguous floats per wave
      const int xo = x0 + pix_l;
      if (ch_l < COUT && xo < p.W) {
        float v = s1[0];
#pragma unroll
        for (int co = 1; co < COUT; ++co) v = ch_l == co ? s1[co] : v;
        v += bias;
        const int m = row_id * p.W + xo;                       // < 2^31: checked by the caller
        if (p.has_ep) {
          v = idiff::act_apply(v, ep.act);
          v *= row_scale;
          if (has_rs && !row_groups) v *= ep.rowscale[m / ep.rows_per_group];
        }
        p.out[(int64_t)m * COUT + ch_l] = v;
      }
    }
  }
}

}  // namespace

namespace idiff {

// Whether conv3x3_narrow() takes this problem (the caller then skips the implicit GEMM).
bool conv3x3_narrow_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_lo, int pad_hi,
                       const idiff_epilogue *ep) {
  if (KH != 3 || KW != 3 || stride != 1 || pad_lo != 1 || pad_hi != 1 || Cin != 128 || Cout < 1 || Cout > 4) return false;
  if ((int64_t)B * H * W * Cin * 4 >= (int64_t)NARROW_OOB || (int64_t)B * H > 0x7fffffff / 4) return false;
  if (ep && (ep->rowbias || ep->residual || ep->colstats)) return false;
  return !option(OPT_NO_PIPE);
}

int conv3x3_narrow(const float *x, const float *wt, float *out, int B, int H, int W, int Cin, int Cout,
                   const idiff_epilogue *ep, hipStream_t stream) {
  NarrowParams p = {};
  p.x = x; p.wt = wt; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin;
  p.x_bytes = (uint32_t)((int64_t)B * H * W * Cin * 4);
  if (ep) { p.ep = *ep; p.has_ep = 1; if (p.ep.rows_per_group <= 0) p.ep.rows_per_group = 1; }
  const dim3 grid(ceil_div(B * H, 4 * ROWS_PER_WAVE)), block(256);
  switch (Cout) {
    case 1: hipLaunchKernelGGL(conv3x3_narrow_kernel<1>, grid, block, 0, stream, p); break;
    case 2: hipLaunchKernelGGL(conv3x3_narrow_kernel<2>, grid, block, 0, stream, p); break;
    case 3: hipLaunchKernelGGL(conv3x3_narrow_kernel<3>, grid, block, 0, stream, p); break;
    default: hipLaunchKernelGGL(conv3x3_narrow_kernel<4>, grid, block, 0, stream, p); break;
  }
  return launch_status("conv3x3_narrow");
}

}  // namespace idiff
