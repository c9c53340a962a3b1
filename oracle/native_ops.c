/*
 * Plain-C restatement of the reference's two native ops -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * oracle_upfirdn2d_f32   follows the index algebra of op/upfirdn2d_kernel.cu:49-105 (the generic kernel): for
 *                        every output pixel walk the input samples whose zero-stuffed, padded position falls
 *                        under the FIR window; taps are read flipped (true convolution, :137).
 * oracle_fused_bias_act_f32 follows op/fused_bias_act_kernel.cu:18-49.
 * Scalar loops, fp32 accumulation in the order ky-major then kx (the CUDA kernel's order).
 */
#include <stdint.h>

static int floor_div(int a, int b) { int q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; }

int oracle_upfirdn2d_f32(const float *x, const float *k, float *out, int major, int in_h, int in_w, int minor, int kh,
                         int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                         int pad_y1) {
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  if (out_h <= 0 || out_w <= 0) return 1;
  for (int m = 0; m < major; ++m)
    for (int oy = 0; oy < out_h; ++oy)
      for (int ox = 0; ox < out_w; ++ox) {
        /* position of the window's last row/col in the stuffed signal, as the CUDA kernel computes it */
        const int mid_y = oy * down_y + up_y - 1 - pad_y0, mid_x = ox * down_x + up_x - 1 - pad_x0;
        int iy0 = floor_div(mid_y, up_y); if (iy0 < 0) iy0 = 0; if (iy0 > in_h) iy0 = in_h;
        int iy1 = floor_div(mid_y + kh, up_y); if (iy1 < 0) iy1 = 0; if (iy1 > in_h) iy1 = in_h;
        int ix0 = floor_div(mid_x, up_x); if (ix0 < 0) ix0 = 0; if (ix0 > in_w) ix0 = in_w;
        int ix1 = floor_div(mid_x + kw, up_x); if (ix1 < 0) ix1 = 0; if (ix1 > in_w) ix1 = in_w;
        for (int c = 0; c < minor; ++c) {
          float v = 0.0f;
          for (int iy = iy0; iy < iy1; ++iy) {
            const int ky = mid_y + kh - (iy + 1) * up_y;
            for (int ix = ix0; ix < ix1; ++ix) {
              const int kx = mid_x + kw - (ix + 1) * up_x;
              v += x[(((int64_t)m * in_h + iy) * in_w + ix) * minor + c] * k[ky * kw + kx];
            }
          }
          out[(((int64_t)m * out_h + oy) * out_w + ox) * minor + c] = v;
        }
      }
  return 0;
}

int oracle_fused_bias_act_f32(const float *x, const float *b, const float *ref, float *out, int64_t n, int step_b,
                              int size_b, int act, int grad, float alpha, float scale) {
  for (int64_t i = 0; i < n; ++i) {
    float v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    const float r = ref ? ref[i] : 0.0f;
    float y;
    switch (act * 10 + grad) {
      case 10: case 11: y = v; break;
      case 12: y = 0.0f; break;
      case 30: y = v > 0.0f ? v : v * alpha; break;
      case 31: y = r > 0.0f ? v : v * alpha; break;
      case 32: y = 0.0f; break;
      default: return 1;
    }
    out[i] = y * scale;
  }
  return 0;
}
