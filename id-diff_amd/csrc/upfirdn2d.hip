// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR (true convolution) -> decimate.
//
// Semantics follow the reference op (op/upfirdn2d.py:159-200, op/upfirdn2d_kernel.cu:49-105):
//   out[p, oy, ox, c] = sum_{ky,kx} U[p, oy*down_y + ky, ox*down_x + kx, c] * k[kh-1-ky, kw-1-kx]
// where U is x with (up-1) zeros inserted after every sample, then padded by pad_*0 / pad_*1 (negative
// pads crop).  Only taps that land on a real sample contribute, so each thread walks INPUT rows/cols:
//   iy in [ceil((oy*down_y - pad_y0)/up_y), floor((oy*down_y + kh-1 - pad_y0)/up_y)] clipped to [0, in_h),
//   ky = iy*up_y + pad_y0 - oy*down_y.
//
// Three kernels, chosen by layout:
//   planes_lds   minor == 1 (the reference's NCHW view): W is the contiguous axis; a workgroup stages the
//                input window of a [planes x TOH x TOW] output tile in LDS with coalesced row loads, then
//                every output pixel reads its taps from LDS.  HBM traffic = algorithmic bytes.
//   nhwc_vec4    minor % 4 == 0 (NHWC activations of the score networks): a thread owns 4 channels of one
//                output pixel; every tap is one coalesced float4; neighbours re-read through L1/L2.
//   generic      anything else, scalar.
#include "common.h"

namespace {

struct UfdParams {
  int major, in_h, in_w, minor, kh, kw;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int out_h, out_w;
};

__device__ __forceinline__ int floor_div(int a, int b) {  // b > 0
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int ceil_div_s(int a, int b) { return -floor_div(-a, b); }

constexpr int kMaxTaps = 64;  // kh*kw kept in LDS for the fast kernels

// ---------------------------------------------------------------- minor == 1
// grid.x = plane groups, grid.y = tiles over (out_h, out_w).  Dynamic LDS: taps + [PPB][tih][tiw+1] window.
__global__ void __launch_bounds__(256)
upfirdn2d_planes_lds(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                     int toh, int tow, int tih, int tiw, int ppb, int tiles_x) {
  extern __shared__ float lds[];
  float *taps = lds;                 // [kh*kw], already flipped
  float *win = lds + kMaxTaps;       // [ppb][tih][tiw + 1]
  const int tid = threadIdx.x;
  const int ntap = p.kh * p.kw;
  for (int i = tid; i < ntap; i += blockDim.x) {
    int ky = i / p.kw, kx = i - ky * p.kw;
    taps[i] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  const int tile_y = blockIdx.y / tiles_x, tile_x = blockIdx.y - tile_y * tiles_x;
  const int oy0 = tile_y * toh, ox0 = tile_x * tow;
  const int plane0 = blockIdx.x * ppb;
  const int nplanes = min(ppb, p.major - plane0);
  // first input row/col any output of this tile can touch
  const int iy0 = ceil_div_s(oy0 * p.down_y - p.pad_y0, p.up_y);
  const int ix0 = ceil_div_s(ox0 * p.down_x - p.pad_x0, p.up_x);
  const int pitch = tiw + 1;
  const int win_elems = tih * tiw;
  for (int i = tid; i < nplanes * win_elems; i += blockDim.x) {
    int pl = i / win_elems, r = i - pl * win_elems;
    int wy = r / tiw, wx = r - wy * tiw;
    int iy = iy0 + wy, ix = ix0 + wx;
    float v = 0.f;
    if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w)
      v = x[((int64_t)(plane0 + pl) * p.in_h + iy) * p.in_w + ix];
    win[(pl * tih + wy) * pitch + wx] = v;
  }
  __syncthreads();
  const int tile_elems = toh * tow;
  for (int i = tid; i < nplanes * tile_elems; i += blockDim.x) {
    int pl = i / tile_elems, r = i - pl * tile_elems;
    int ty = r / tow, tx = r - ty * tow;
    int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= p.out_h || ox >= p.out_w) continue;
    const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
    const int iy_lo = max(ceil_div_s(by, p.up_y), 0), iy_hi = min(floor_div(by + p.kh - 1, p.up_y), p.in_h - 1);
    const int ix_lo = max(ceil_div_s(bx, p.up_x), 0), ix_hi = min(floor_div(bx + p.kw - 1, p.up_x), p.in_w - 1);
    float acc = 0.f;
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
      const int ky = iy * p.up_y - by;
      const float *wrow = win + (pl * tih + (iy - iy0)) * pitch - ix0;
      const float *trow = taps + ky * p.kw;
      for (int ix = ix_lo; ix <= ix_hi; ++ix) acc += wrow[ix] * trow[ix * p.up_x - bx];
    }
    out[((int64_t)(plane0 + pl) * p.out_h + oy) * p.out_w + ox] = acc;
  }
}

// ---------------------------------------------------------------- minor % 4 == 0
__global__ void __launch_bounds__(256)
upfirdn2d_nhwc_vec4(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                    int64_t total_vec) {
  __shared__ float taps[kMaxTaps];
  const int ntap = p.kh * p.kw;
  for (int i = threadIdx.x; i < ntap; i += blockDim.x) {
    int ky = i / p.kw, kx = i - ky * p.kw;
    taps[i] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  __syncthreads();
  const int cv = p.minor >> 2;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * blockDim.x) {
    int c4 = (int)(v % cv);
    int64_t pix = v / cv;
    int ox = (int)(pix % p.out_w);
    int64_t t = pix / p.out_w;
    int oy = (int)(t % p.out_h);
    int64_t plane = t / p.out_h;
    const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
    const int iy_lo = max(ceil_div_s(by, p.up_y), 0), iy_hi = min(floor_div(by + p.kh - 1, p.up_y), p.in_h - 1);
    const int ix_lo = max(ceil_div_s(bx, p.up_x), 0), ix_hi = min(floor_div(bx + p.kw - 1, p.up_x), p.in_w - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *xp = reinterpret_cast<const float4 *>(x) + plane * p.in_h * p.in_w * cv + c4;
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
      const float *trow = taps + (iy * p.up_y - by) * p.kw;
      for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        const float w = trow[ix * p.up_x - bx];
        const float4 s = xp[((int64_t)iy * p.in_w + ix) * cv];
        acc.x += s.x * w; acc.y += s.y * w; acc.z += s.z * w; acc.w += s.w * w;
      }
    }
    reinterpret_cast<float4 *>(out)[v] = acc;
  }
}

// ---------------------------------------------------------------- anything else
__global__ void __launch_bounds__(256)
upfirdn2d_generic(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                  int64_t total) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(v % p.minor);
    int64_t pix = v / p.minor;
    int ox = (int)(pix % p.out_w);
    int64_t t = pix / p.out_w;
    int oy = (int)(t % p.out_h);
    int64_t plane = t / p.out_h;
    const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
    const int iy_lo = max(ceil_div_s(by, p.up_y), 0), iy_hi = min(floor_div(by + p.kh - 1, p.up_y), p.in_h - 1);
    const int ix_lo = max(ceil_div_s(bx, p.up_x), 0), ix_hi = min(floor_div(bx + p.kw - 1, p.up_x), p.in_w - 1);
    float acc = 0.f;
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
      const int ky = iy * p.up_y - by;
      for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        const int kx = ix * p.up_x - bx;
        acc += x[((plane * p.in_h + iy) * p.in_w + ix) * p.minor + c] * k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    out[v] = acc;
  }
}

}  // namespace

IDIFF_API int idiff_upfirdn2d_f32(const float *x, const float *k, float *out, int major, int in_h, int in_w,
                                  int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream) {
  using namespace idiff;
  if (!x || !k || !out) return fail("upfirdn2d: null pointer");
  if (major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0)
    return fail("upfirdn2d: bad shape major=%d in=%dx%d minor=%d k=%dx%d", major, in_h, in_w, minor, kh, kw);
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return fail("upfirdn2d: up/down factors must be >= 1");
  UfdParams p;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  const int span_h = in_h * up_y + pad_y0 + pad_y1 - kh, span_w = in_w * up_x + pad_x0 + pad_x1 - kw;
  if (span_h < 0 || span_w < 0) return fail("upfirdn2d: kernel larger than padded input");
  p.out_h = span_h / down_y + 1;
  p.out_w = span_w / down_x + 1;
  if (major == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)major * p.out_h * p.out_w * minor;

  if (minor == 1 && kh * kw <= kMaxTaps) {
    // output tile: up to 32 x 64 pixels, several planes per workgroup when planes are small
    const int tow = min(p.out_w, 64), toh = min(p.out_h, 32);
    const int tih = ((toh - 1) * down_y + kh - 1) / up_y + 2;
    const int tiw = ((tow - 1) * down_x + kw - 1) / up_x + 2;
    const int tile_elems = toh * tow;
    int ppb = max(1, 2048 / tile_elems);                      // ~8 outputs per thread
    const int max_ppb_lds = max(1, (int)((48 * 1024 / sizeof(float) - kMaxTaps) / (tih * (tiw + 1))));
    ppb = min(min(ppb, max_ppb_lds), major);
    const size_t lds_bytes = (kMaxTaps + (size_t)ppb * tih * (tiw + 1)) * sizeof(float);
    if (lds_bytes <= 64 * 1024) {
      const int tiles_x = ceil_div(p.out_w, tow), tiles_y = ceil_div(p.out_h, toh);
      dim3 grid(ceil_div(major, ppb), tiles_x * tiles_y);
      if (grid.y <= 65535) {
        hipLaunchKernelGGL(upfirdn2d_planes_lds, grid, dim3(256), lds_bytes, st, x, k, out, p, toh, tow, tih, tiw,
                           ppb, tiles_x);
        return launch_status("upfirdn2d_planes_lds");
      }
    }
  }
  if (minor % 4 == 0 && kh * kw <= kMaxTaps && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    const int64_t total_vec = total / 4;
    hipLaunchKernelGGL(upfirdn2d_nhwc_vec4, dim3(streaming_grid(total_vec, 256)), dim3(256), 0, st, x, k, out, p,
                       total_vec);
    return launch_status("upfirdn2d_nhwc_vec4");
  }
  hipLaunchKernelGGL(upfirdn2d_generic, dim3(streaming_grid(total, 256)), dim3(256), 0, st, x, k, out, p, total);
  return launch_status("upfirdn2d_generic");
}
