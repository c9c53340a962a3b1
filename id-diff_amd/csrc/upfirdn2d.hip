// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR (true convolution) -> decimate.
//
// Semantics follow the reference op (op/upfirdn2d.py:159-200, op/upfirdn2d_kernel.cu:49-105):
//   out[p, oy, ox, c] = sum_{ky,kx} U[p, oy*down_y + ky, ox*down_x + kx, c] * k[kh-1-ky, kw-1-kx]
// where U is x with (up-1) zeros inserted after every sample, then padded by pad_*0 / pad_*1 (negative
// pads crop).  Only taps that land on a real sample contribute, so each thread walks INPUT rows/cols:
//   iy in [ceil((oy*down_y - pad_y0)/up_y), floor((oy*down_y + kh-1 - pad_y0)/up_y)] clipped to [0, in_h),
//   ky = iy*up_y + pad_y0 - oy*down_y.
//
// Kernels, chosen by layout and size:
//   planes_whole minor == 1 (the reference's NCHW view), planes <= 32 KB: `ppb` consecutive planes are copied to
//                LDS as one contiguous 16-byte-per-lane stream, every output pixel reads its taps from LDS.
//   planes_lds   minor == 1, large planes: a workgroup stages the input window of a [TOH x TOW] output tile.
//   nhwc_rows    minor % 4 == 0 (NHWC activations of the score networks): grid.y = (plane, output row), a thread
//                owns 4 channels; every tap is one coalesced float4; neighbours re-read through L1/L2.
//   nhwc_vec4    same layout, flat grid-stride form for shapes the row form cannot index.
//   generic      anything else, scalar.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

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

// How many planes a workgroup of the small-plane kernels takes per group, and how many workgroups to launch.  These launches last
// 5 - 20 us: a grid of 1,093 workgroups on 1,024 resident slots runs as two rounds, the second 7 % full (measured: [128, 128, 16, 16]
// plain FIR 15.4 us at 15 planes per group).  So: among the admissible group sizes pick the one with the least (rounds x planes per
// group), launch ceil(groups / rounds) workgroups and let each walk its groups (grid-stride), so that every workgroup does the same
// number of groups +- 1 and all of them are resident from the start.  256 CUs x min(8, LDS) workgroups of 256 threads are resident.
struct GroupPlan { int ppb, grid, ngroups; };
template <typename LdsBytes, typename Admissible>
GroupPlan plan_plane_groups(int major, int ppb_max, LdsBytes lds_bytes_of, Admissible admissible) {
  GroupPlan best = {1, major, major};
  double best_cost = 1e300;
  for (int ppb = 1; ppb <= ppb_max; ++ppb) {
    if (!admissible(ppb)) continue;
    const size_t lds = lds_bytes_of(ppb);
    if (lds > 64 * 1024) continue;
    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    const int slots = 256 * std::max(per_cu, 1), ngroups = idiff::ceil_div(major, ppb), rounds = idiff::ceil_div(ngroups, slots);
    // per-group cost ~ planes + a fixed part (frame / taps / barriers), less latency hidden when few workgroups share a CU
    const double cost = rounds * (ppb + 1.5) * (per_cu >= 4 ? 1.0 : 1.0 + 0.15 * (4 - per_cu));
    if (cost < best_cost) { best_cost = cost; best = {ppb, idiff::ceil_div(ngroups, rounds), ngroups}; }
  }
  return best;
}

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


// ---------------------------------------------------------------- minor == 1, whole planes in LDS
// The score networks' planes are small (4 KB at 32x32): a workgroup copies `ppb` consecutive planes -- one
// contiguous chunk of HBM -- into LDS with 16-byte loads, then thread (ty, tx) produces output pixel (oy, ox) of
// every plane.  No index arithmetic beyond adds in the loops (the tap ranges are clipped once per pixel), the
// input is read exactly once and the output written exactly once.
// UPLOG >= 0: up_x == up_y == 1 << UPLOG and kh, kw <= 4 (every FIR the score networks use): the 4x4 tap loop is
// fully unrolled with the taps in registers, and a tap is "on" when its stuffed coordinate is a multiple of up
// and inside the image -- per-pixel validity bits and LDS offsets are computed once and reused for all planes.
template <int UPLOG>
__global__ void __launch_bounds__(256)
upfirdn2d_planes_whole(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                       int ppb, int tx_log2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *taps = lds;            // [kMaxTaps] flipped
  float *pl = lds + kMaxTaps;   // [ppb][in_h * in_w]
  const int tid = threadIdx.x;
  const int ntap = p.kh * p.kw;
  for (int i = tid; i < ntap; i += 256) {
    int ky = i / p.kw, kx = i - ky * p.kw;
    taps[i] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  const int plane0 = blockIdx.x * ppb;
  const int nplanes = min(ppb, p.major - plane0);
  const int psz = p.in_h * p.in_w;
  const int total = nplanes * psz;
  const float *src = x + (int64_t)plane0 * psz;
  if (((psz & 3) == 0) && ((((uintptr_t)src) & 15) == 0)) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(pl);
    for (int i = tid; i < (total >> 2); i += 256) d4[i] = s4[i];
  } else {
    for (int i = tid; i < total; i += 256) pl[i] = src[i];
  }
  __syncthreads();
  // thread -> (tq, ty, tx): TX x TY covers the output tile (powers of two); when the whole plane has fewer than
  // 256 pixels the remaining thread bits index planes (TQ planes in flight), so no lane idles on 8x8 outputs
  const int TX = 1 << tx_log2;
  int ty_log2 = 0;
  while ((1 << ty_log2) < p.out_h && (TX << ty_log2) < 256) ++ty_log2;
  const int TY = 1 << ty_log2, TQ = 256 >> (tx_log2 + ty_log2);
  const int tx = tid & (TX - 1), ty = (tid >> tx_log2) & (TY - 1), tq = tid >> (tx_log2 + ty_log2);
  const int osz = p.out_h * p.out_w;
  float *dst = out + (int64_t)plane0 * osz;
  for (int oy = ty; oy < p.out_h; oy += TY) {
    const int by = oy * p.down_y - p.pad_y0;
    const int iy_lo = max(ceil_div_s(by, p.up_y), 0), iy_hi = min(floor_div(by + p.kh - 1, p.up_y), p.in_h - 1);
    for (int ox = tx; ox < p.out_w; ox += TX) {
      const int bx = ox * p.down_x - p.pad_x0;
      const int ix_lo = max(ceil_div_s(bx, p.up_x), 0), ix_hi = min(floor_div(bx + p.kw - 1, p.up_x), p.in_w - 1);
      if (UPLOG >= 0) {
        // only the taps whose stuffed coordinate is a multiple of up can be on: ky = ky0 + up*a, ky0 = (-by) mod up
        constexpr int UL = UPLOG >= 0 ? UPLOG : 0, UP = 1 << UL, UM = UP - 1, NI = 4 >> UL;
        const int ky0 = (UP - (by & UM)) & UM, kx0 = (UP - (bx & UM)) & UM;
        float T[NI * NI];
        int off[NI * NI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int b = 0; b < NI; ++b) {
            const int ky = ky0 + UP * a, kx = kx0 + UP * b, iy = (by + ky) >> UL, ix = (bx + kx) >> UL;
            const bool ok = ky < p.kh && kx < p.kw && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            T[a * NI + b] = ok ? taps[ky * p.kw + kx] : 0.f;    // an "off" tap reads pixel 0 with weight 0
            off[a * NI + b] = ok ? iy * p.in_w + ix : 0;
          }
        for (int q = tq; q < nplanes; q += TQ) {
          const float *plane = pl + q * psz;
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < NI * NI; ++i) acc += plane[off[i]] * T[i];
          dst[(int64_t)q * osz + oy * p.out_w + ox] = acc;
        }
      } else {
        for (int q = tq; q < nplanes; q += TQ) {
          const float *plane = pl + q * psz;
          float acc = 0.f;
          for (int iy = iy_lo; iy <= iy_hi; ++iy) {
            const float *trow = taps + (iy * p.up_y - by) * p.kw - bx;
            const float *prow = plane + iy * p.in_w;
            for (int ix = ix_lo; ix <= ix_hi; ++ix) acc += prow[ix] * trow[ix * p.up_x];
          }
          dst[(int64_t)q * osz + oy * p.out_w + ox] = acc;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- minor == 1, FIR + decimation by 2 on small planes
// up == 1, down == 2, kh, kw <= 4 (downsample_2d of the score networks, up_or_down_sampling.py:227-257): a thread produces a 2 x 2 block
// of outputs from its 6 x 6 input window, read from LDS as one ds_read_b128 + one ds_read_b64 per row -- 3 LDS instructions per output
// where the per-pixel form above issues 16 four-byte reads (which made that form LDS-bound: 2.1 MB of reads per CU at [128, 256, 16, 16]).
// Planes sit in LDS inside a frame of zeros (pad_y0 rows on top, pad_x0 columns on the left, enough on the other two sides for the last
// window), so no tap needs a validity test, and the frame puts column 4 X of the framed plane at the start of block X's window: 16-byte
// aligned.  Stores are float2 (the two outputs of a block row), contiguous over the lanes.
__global__ void __launch_bounds__(256)
upfirdn2d_planes_down2(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p, int ppb,
                       int rows, int pitch, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int psz = p.in_h * p.in_w, fsz = rows * pitch;            // fsz % 4 == 0 (pitch % 4 == 0)
  // flipped taps, zero beyond kh x kw
  float T[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) T[a][b] = (a < p.kh && b < p.kw) ? k[(p.kh - 1 - a) * p.kw + (p.kw - 1 - b)] : 0.f;
  // the frame of zeros is written once: every group overwrites the same interior cells
  {
    float4 *z4 = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < (ppb * fsz) >> 2; i += 256) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const int plane0 = grp * ppb;
  const int nplanes = min(ppb, p.major - plane0);
  {
    const float4 *s4 = reinterpret_cast<const float4 *>(x + (int64_t)plane0 * psz);
    const int w4 = p.in_w >> 2, per_plane = psz >> 2;
    for (int i = tid; i < nplanes * per_plane; i += 256) {
      const int q = i / per_plane, r = i - q * per_plane, iy = r / w4, c4 = r - iy * w4;
      const float4 v = s4[i];
      float *d = lds + q * fsz + (iy + p.pad_y0) * pitch + p.pad_x0 + 4 * c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  __syncthreads();
  const int bw = p.out_w >> 1, nb = (p.out_h >> 1) * bw, osz = p.out_h * p.out_w;
  float *dst = out + (int64_t)plane0 * osz;
  for (int i = tid; i < nplanes * nb; i += 256) {
    const int q = i / nb, r = i - q * nb, Y = r / bw, X = r - Y * bw;
    const float *w = lds + q * fsz + 4 * Y * pitch + 4 * X;
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const float4 a = *reinterpret_cast<const float4 *>(w + rr * pitch);
      const float2 b = *reinterpret_cast<const float2 *>(w + rr * pitch + 4);
      const float c[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
      if (rr < 4) {                                                // output row 2 Y: window rows 0 .. 3
#pragma unroll
        for (int t = 0; t < 4; ++t) { o00 = fmaf(c[t], T[rr][t], o00); o01 = fmaf(c[t + 2], T[rr][t], o01); }
      }
      if (rr >= 2) {                                               // output row 2 Y + 1: window rows 2 .. 5
#pragma unroll
        for (int t = 0; t < 4; ++t) { o10 = fmaf(c[t], T[rr - 2][t], o10); o11 = fmaf(c[t + 2], T[rr - 2][t], o11); }
      }
    }
    float *o = dst + (int64_t)q * osz + (2 * Y) * p.out_w + 2 * X;
    *reinterpret_cast<float2 *>(o) = make_float2(o00, o01);
    *reinterpret_cast<float2 *>(o + p.out_w) = make_float2(o10, o11);
  }
  __syncthreads();                                                 // the next group's planes overwrite the interior
  }
}

// ---------------------------------------------------------------- minor == 1, plain FIR on small planes, strip form
// up == down == 1, kh, kw <= 4 (the FIR in front of a stride-2 convolution, up_or_down_sampling.py:144-178; out = in + 1 with pads (2, 2)):
// the framed-plane layout of upfirdn2d_planes_down2; a thread produces a strip of four outputs of one row from its 4 x 7 window, read as
// two ds_read_b128 per row (2 LDS instructions per output).  The row-walking kernel below -- one thread per output ROW, a serial chain of
// out_w steps -- sat at 2.2 TB/s on [128, 128, 16, 16].
__global__ void __launch_bounds__(256)
upfirdn2d_planes_fir4(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p, int ppb, int rows,
                      int pitch, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int psz = p.in_h * p.in_w, fsz = rows * pitch;
  float T[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) T[a][b] = (a < p.kh && b < p.kw) ? k[(p.kh - 1 - a) * p.kw + (p.kw - 1 - b)] : 0.f;
  {
    float4 *z4 = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < (ppb * fsz) >> 2; i += 256) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int sx = (p.out_w + 3) >> 2, ns = p.out_h * sx, osz = p.out_h * p.out_w;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int plane0 = grp * ppb;
    const int nplanes = min(ppb, p.major - plane0);
    {
      const float4 *s4 = reinterpret_cast<const float4 *>(x + (int64_t)plane0 * psz);
      const int w4 = p.in_w >> 2, per_plane = psz >> 2;
      for (int i = tid; i < nplanes * per_plane; i += 256) {
        const int q = i / per_plane, r = i - q * per_plane, iy = r / w4, c4 = r - iy * w4;
        const float4 v = s4[i];
        float *d = lds + q * fsz + (iy + p.pad_y0) * pitch + p.pad_x0 + 4 * c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();
    float *dst = out + (int64_t)plane0 * osz;
    for (int i = tid; i < nplanes * ns; i += 256) {
      const int q = i / ns, r = i - q * ns, oy = r / sx, S = r - oy * sx;
      const float *w = lds + q * fsz + oy * pitch + 4 * S;
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 a = *reinterpret_cast<const float4 *>(w + rr * pitch), b = *reinterpret_cast<const float4 *>(w + rr * pitch + 4);
        const float c[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t) o[e] = fmaf(c[e + t], T[rr][t], o[e]);
      }
      float *op = dst + (int64_t)q * osz + oy * p.out_w + 4 * S;
      const int left = p.out_w - 4 * S;
      op[0] = o[0];
      if (left > 1) op[1] = o[1];
      if (left > 2) op[2] = o[2];
      if (left > 3) op[3] = o[3];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- minor == 1, plain FIR on small planes
// up == down == 1, kh, kw <= 4 (the FIR in front of a stride-2 convolution, up_or_down_sampling.py:144-178): thread =
// (plane, output row).  It walks its row with the three previous inputs of each of the four input rows in registers:
// four LDS reads per output instead of sixteen, every lane busy whatever the row length (a 17 x 17 output on a 32 x 8
// thread tile left 62 % of the lanes idle).  Planes sit in LDS with a pitch of in_w + 1 words (rows of one plane on
// distinct banks); results go through LDS once more so that the stores are contiguous.
__global__ void __launch_bounds__(256)
upfirdn2d_planes_rowslide(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p, int ppb,
                          int ngroups) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *taps = lds;                        // [4][4] flipped, zero beyond kh x kw
  float *pl = lds + 16;                     // [ppb][in_h][in_w + 1]
  const int tid = threadIdx.x;
  const int pitch = p.in_w + 1, psz = p.in_h * p.in_w, lpsz = p.in_h * pitch, osz = p.out_h * p.out_w;
  float *ob = pl + ppb * lpsz;              // [ppb][out_h * out_w]
  if (tid < 16) {
    const int ky = tid >> 2, kx = tid & 3;
    taps[tid] = (ky < p.kh && kx < p.kw) ? k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : 0.f;
  }
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  if (grp != (int)blockIdx.x) __syncthreads();                     // the previous group's output staging has been stored
  const int plane0 = grp * ppb;
  const int nplanes = min(ppb, p.major - plane0);
  const float *src = x + (int64_t)plane0 * psz;
  if ((p.in_w & 3) == 0 && ((((uintptr_t)src) & 15) == 0)) {
    // 16-byte loads (the planes of a workgroup are one contiguous run; a row is a whole number of float4), and the
    // (plane, row, column) of an element by one division per thread and running sums instead of two divisions per element:
    // the fill was a third of this kernel's time on 16 x 16 planes
    const int row4 = p.in_w >> 2, n4 = (nplanes * psz) >> 2;
    int i4 = tid;
    int rowi = i4 / row4, c4 = i4 - rowi * row4;          // global row index over the workgroup's planes, float4 column
    int q = rowi / p.in_h, iy = rowi - q * p.in_h;
    const int drow = 256 / row4, dc4 = 256 - drow * row4;  // what 256 float4 further means in (rows, columns)
    for (; i4 < n4; i4 += 256) {
      const float4 v = reinterpret_cast<const float4 *>(src)[i4];
      float *d = pl + q * lpsz + iy * pitch + 4 * c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      c4 += dc4; iy += drow;
      if (c4 >= row4) { c4 -= row4; ++iy; }
      while (iy >= p.in_h) { iy -= p.in_h; ++q; }
    }
  } else {
    for (int i = tid; i < nplanes * psz; i += 256) {
      const int q = i / psz, r = i - q * psz, iy = r / p.in_w, ix = r - iy * p.in_w;
      pl[q * lpsz + iy * pitch + ix] = src[i];
    }
  }
  __syncthreads();
  const int q = tid / p.out_h, oy = tid - q * p.out_h;
  if (q < nplanes) {
    float t[4][4], c[4][3];
    const float *row[4];
    bool rv[4];
    const int by = oy - p.pad_y0, bx0 = -p.pad_x0;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = by + ky;
      rv[ky] = ky < p.kh && iy >= 0 && iy < p.in_h;
      row[ky] = pl + q * lpsz + (rv[ky] ? iy : 0) * pitch;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) t[ky][kx] = rv[ky] ? taps[ky * 4 + kx] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) { const int ix = bx0 + j; c[ky][j] = (rv[ky] && ix >= 0 && ix < p.in_w) ? row[ky][ix] : 0.f; }
    }
    float *orow = ob + q * osz + oy * p.out_w;
    for (int ox = 0; ox < p.out_w; ++ox) {
      const int ix3 = bx0 + ox + 3;
      const bool in3 = ix3 >= 0 && ix3 < p.in_w;
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const float c3 = (in3 && rv[ky]) ? row[ky][ix3] : 0.f;    // a row outside the image contributes exact zeros
        acc += c[ky][0] * t[ky][0] + c[ky][1] * t[ky][1] + c[ky][2] * t[ky][2] + c3 * t[ky][3];
        c[ky][0] = c[ky][1]; c[ky][1] = c[ky][2]; c[ky][2] = c3;
      }
      orow[ox] = acc;
    }
  }
  __syncthreads();
  float *dst = out + (int64_t)plane0 * osz;
  for (int i = tid; i < nplanes * osz; i += 256) dst[i] = ob[i];
  }
}

// Workgroups are dealt to the 8 XCDs round-robin, and every XCD has an L2 of its own: consecutive block indices -- here
// neighbouring rows of one plane, which read the same input rows -- would each pull those rows through a different L2.
// This bijection gives every XCD a contiguous run of the grid instead (blocks b and b + 8 share an XCD).
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// ---------------------------------------------------------------- minor % 4 == 0, row-structured
// grid.x = (plane, oy); a thread keeps its channel vector c4 and walks output columns, so the only integer
// divisions are the two that split blockIdx.x (uniform) -- the per-element 64-bit div/mod of a flat grid-stride
// loop made this kernel VALU-bound.
template <int UPLOG>
__global__ void __launch_bounds__(256)
upfirdn2d_nhwc_rows(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                    int cv, int col_step) {
  __shared__ float taps[kMaxTaps];
  const int ntap = p.kh * p.kw;
  for (int i = threadIdx.x; i < ntap; i += 256) {
    int ky = i / p.kw, kx = i - ky * p.kw;
    taps[i] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  __syncthreads();
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int plane = bid / p.out_h, oy = bid - plane * p.out_h;   // rows on grid.x (no 65535 limit)
  const int c4 = threadIdx.x % cv, col0 = threadIdx.x / cv;
  if (col0 >= col_step) return;
  const int by = oy * p.down_y - p.pad_y0;
  const int iy_lo = max(ceil_div_s(by, p.up_y), 0), iy_hi = min(floor_div(by + p.kh - 1, p.up_y), p.in_h - 1);
  const float4 *xp = reinterpret_cast<const float4 *>(x) + (int64_t)plane * p.in_h * p.in_w * cv + c4;
  float4 *op = reinterpret_cast<float4 *>(out) + ((int64_t)plane * p.out_h + oy) * p.out_w * cv + c4;
  for (int ox = blockIdx.y * col_step + col0; ox < p.out_w; ox += gridDim.y * col_step) {
    const int bx = ox * p.down_x - p.pad_x0;
    const int ix_lo = max(ceil_div_s(bx, p.up_x), 0), ix_hi = min(floor_div(bx + p.kw - 1, p.up_x), p.in_w - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (UPLOG >= 0) {
      constexpr int UL = UPLOG >= 0 ? UPLOG : 0, UP = 1 << UL, UM = UP - 1, NI = 4 >> UL;
      const int ky0 = (UP - (by & UM)) & UM, kx0 = (UP - (bx & UM)) & UM;
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int ky = ky0 + UP * a, iy = (by + ky) >> UL;
        const bool vy = ky < p.kh && (unsigned)iy < (unsigned)p.in_h;
        const float4 *row = xp + (int64_t)(vy ? iy : 0) * p.in_w * cv;
#pragma unroll
        for (int b = 0; b < NI; ++b) {
          const int kx = kx0 + UP * b, ix = (bx + kx) >> UL;
          const bool ok = vy && kx < p.kw && (unsigned)ix < (unsigned)p.in_w;
          const float w = ok ? taps[ky * p.kw + kx] : 0.f;       // an "off" tap reads pixel 0 with weight 0
          const float4 s = row[(int64_t)(ok ? ix : 0) * cv];
          acc.x += s.x * w; acc.y += s.y * w; acc.z += s.z * w; acc.w += s.w * w;
        }
      }
    } else {
      for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        const float *trow = taps + (iy * p.up_y - by) * p.kw - bx;
        const float4 *row = xp + (int64_t)iy * p.in_w * cv;
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
          const float w = trow[ix * p.up_x];
          const float4 s = row[(int64_t)ix * cv];
          acc.x += s.x * w; acc.y += s.y * w; acc.z += s.z * w; acc.w += s.w * w;
        }
      }
    }
    op[(int64_t)ox * cv] = acc;
  }
}

// ---------------------------------------------------------------- NHWC, FIR upsampling by 2 with a 4x4 kernel
// (upsample_2d of the score networks: up 2, pads (2, 1), out = 2 x in).  Output rows 2i / 2i+1 use input rows
// (i-1, i) with kernel rows (0, 2) / (i, i+1) with (1, 3) of the flipped kernel, columns alike: a thread produces the
// 2x2 output block of input pixel (i, j) from its 3x3 neighbourhood -- 9 loads per 4 outputs instead of 16, all nine
// in flight at once (the row-walking kernel above sat at 3.0 TB/s, below 40 % of the HBM peak).
__global__ void __launch_bounds__(256)
upfirdn2d_nhwc_up2_block(const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out, UfdParams p,
                         int cv, int col_step) {
  __shared__ float taps[16];
  if (threadIdx.x < 16) {
    const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
    taps[threadIdx.x] = k[(3 - ky) * 4 + (3 - kx)];
  }
  __syncthreads();
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int plane = bid / p.in_h, i = bid - plane * p.in_h;
  const int c4 = threadIdx.x % cv, col0 = threadIdx.x / cv;
  if (col0 >= col_step) return;
  // the sixteen taps are wave-uniform: scalar registers (the array form was kept in a 32-byte scratch slot by the compiler)
  float w[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) w[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[t])));
  const float4 *xp = reinterpret_cast<const float4 *>(x) + (int64_t)plane * p.in_h * p.in_w * cv + c4;
  float4 *op = reinterpret_cast<float4 *>(out) + ((int64_t)plane * p.out_h + 2 * i) * p.out_w * cv + c4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = col0; j < p.in_w; j += col_step) {
    float4 in[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = i - 1 + r;
      const bool vy = (unsigned)iy < (unsigned)p.in_h;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int ix = j - 1 + c;
        const bool ok = vy && (unsigned)ix < (unsigned)p.in_w;
        in[r][c] = ok ? xp[((int64_t)iy * p.in_w + ix) * cv] : zero;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // rows: (kernel row a, input row a) and (a + 2, a + 1) of the 3x3 window; columns alike
        float4 acc = zero;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const float wt = w[(a + 2 * u) * 4 + (b + 2 * v)];
            const float4 sv = in[a + u][b + v];
            acc.x += sv.x * wt; acc.y += sv.y * wt; acc.z += sv.z * wt; acc.w += sv.w * wt;
          }
        op[((int64_t)a * p.out_w + 2 * j + b) * cv] = acc;
      }
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

// The same walk for the other two dtypes of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// op/upfirdn2d_kernel.cu:311): T = _Float16 with fp32 products and accumulation, rounded ONCE to half on the way out, and
// T = double with fp64 products and accumulation.  (The reference's CUDA kernels are not one arithmetic: its tiled kernels
// stage samples and taps in `float` LDS arrays, .cu:115-116, and add each fp32 product to a scalar_t accumulator -- per-tap
// rounding to half, fp32 products under an fp64 accumulator -- while its generic kernel multiplies in scalar_t, .cu:85.  The
// parity target is the CPU path, upfirdn2d_native in the tensor's dtype, op/upfirdn2d.py:159-200, which this form matches to
// the last half ulp / to fp64 rounding.)
template <typename T, typename ACC>
__global__ void __launch_bounds__(256)
upfirdn2d_generic_t(const T *__restrict__ x, const T *__restrict__ k, T *__restrict__ out, UfdParams p, int64_t total) {
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
    ACC acc = (ACC)0;
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
      const int ky = iy * p.up_y - by;
      for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        const int kx = ix * p.up_x - bx;
        acc += (ACC)x[((plane * p.in_h + iy) * p.in_w + ix) * p.minor + c] * (ACC)k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    out[v] = (T)acc;
  }
}

// argument checks and output geometry shared by the three dtypes; returns 0 and fills p, or an error code
int ufd_geometry(const void *x, const void *k, const void *out, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                 int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, UfdParams &p) {
  using namespace idiff;
  if (!x || !k || !out) return fail("upfirdn2d: null pointer");
  if (major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0)
    return fail("upfirdn2d: bad shape major=%d in=%dx%d minor=%d k=%dx%d", major, in_h, in_w, minor, kh, kw);
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return fail("upfirdn2d: up/down factors must be >= 1");
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  const int span_h = in_h * up_y + pad_y0 + pad_y1 - kh, span_w = in_w * up_x + pad_x0 + pad_x1 - kw;
  if (span_h < 0 || span_w < 0) return fail("upfirdn2d: kernel larger than padded input");
  p.out_h = span_h / down_y + 1;
  p.out_w = span_w / down_x + 1;
  return 0;
}

template <typename T, typename ACC>
int ufd_launch_t(const void *x, const void *k, void *out, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y,
                 int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream, const char *what) {
  UfdParams p;
  if (int rc = ufd_geometry(x, k, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, p))
    return rc;
  if (major == 0) return 0;
  const int64_t total = (int64_t)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL((upfirdn2d_generic_t<T, ACC>), dim3(idiff::streaming_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const T *)x, (const T *)k, (T *)out, p, total);
  return idiff::launch_status(what);
}

}  // namespace

IDIFF_API int idiff_upfirdn2d_f16(const void *x, const void *k, void *out, int major, int in_h, int in_w, int minor, int kh, int kw,
                                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                  void *stream) {
  return ufd_launch_t<_Float16, float>(x, k, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                                       pad_y1, stream, "upfirdn2d_f16");
}

IDIFF_API int idiff_upfirdn2d_f64(const double *x, const double *k, double *out, int major, int in_h, int in_w, int minor, int kh,
                                  int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                  void *stream) {
  return ufd_launch_t<double, double>(x, k, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                                      pad_y1, stream, "upfirdn2d_f64");
}

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

  if (minor == 1 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh <= 4 && kw <= 4 && pad_x0 >= 0 && pad_y0 >= 0 &&
      pad_x0 <= 4 && pad_y0 <= 4 && in_w % 4 == 0 && (int64_t)in_h * in_w <= 4096 && ((uintptr_t)x & 15) == 0 && !option(OPT_UFD_ROWS)) {
    // plain FIR, strip form: frame rows -pad_y0 .. out_h + 2 - pad_y0, columns -pad_x0 .. 4 ceil(out_w / 4) + 3 - pad_x0
    const int rows = max(in_h + pad_y0, p.out_h + 3), pitch = (max(in_w + pad_x0, 4 * ((p.out_w + 3) / 4) + 4) + 3) & ~3;
    const int fsz = rows * pitch;
    auto lds_of = [&](int ppb) { return (size_t)ppb * fsz * sizeof(float); };
    const GroupPlan g = plan_plane_groups(major, std::min(major, 64), lds_of, [](int) { return true; });
    hipLaunchKernelGGL(upfirdn2d_planes_fir4, dim3(g.grid), dim3(256), lds_of(g.ppb), st, x, k, out, p, g.ppb, rows, pitch, g.ngroups);
    return launch_status("upfirdn2d_planes_fir4");
  }
  if (minor == 1 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh <= 4 && kw <= 4 && p.out_h <= 128 &&
      (int64_t)in_h * in_w <= 4096) {
    // plain FIR on small planes: one thread per (plane, output row)
    const int lpsz = in_h * (in_w + 1), osz = p.out_h * p.out_w;
    auto lds_of = [&](int ppb) { return (16 + (size_t)ppb * (lpsz + osz)) * sizeof(float); };
    const GroupPlan g = plan_plane_groups(major, max(1, 256 / p.out_h), lds_of, [](int) { return true; });
    hipLaunchKernelGGL(upfirdn2d_planes_rowslide, dim3(g.grid), dim3(256), lds_of(g.ppb), st, x, k, out, p, g.ppb, g.ngroups);
    return launch_status("upfirdn2d_planes_rowslide");
  }
  if (minor == 1 && up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2 && kh <= 4 && kw <= 4 && pad_x0 >= 0 && pad_y0 >= 0 &&
      pad_x0 <= 4 && pad_y0 <= 4 && p.out_h % 2 == 0 && p.out_w % 2 == 0 && in_w % 4 == 0 && (int64_t)in_h * in_w <= 4096 &&
      (((uintptr_t)x | (uintptr_t)out) & 15) == 0 && !option(OPT_UFD_ROWS)) {
    // FIR + decimation by 2: 2 x 2 output blocks from framed planes in LDS.  Frame: rows -pad_y0 .. 2 out_h + 1 - pad_y0, columns
    // -pad_x0 .. 2 out_w + 1 - pad_x0 (the last block's window), pitch a multiple of 4 floats
    const int rows = max(in_h + pad_y0, 2 * p.out_h + 2), pitch = (max(in_w + pad_x0, 2 * p.out_w + 2) + 3) & ~3;
    const int fsz = rows * pitch, nb = (p.out_h / 2) * (p.out_w / 2);
    auto lds_of = [&](int ppb) { return (size_t)ppb * fsz * sizeof(float); };
    // whole rounds of 256 output blocks per group where the shape allows it (no idle lanes in the last round of a group)
    auto full_rounds = [&](int ppb) {
      if (nb >= 256 || 256 % nb != 0) return true;
      const int unit = 256 / nb;                                     // planes per round of 256 blocks
      return ppb % unit == 0 || (major < unit && ppb == major);
    };
    const GroupPlan g = plan_plane_groups(major, std::min(major, 64), lds_of, full_rounds);
    hipLaunchKernelGGL(upfirdn2d_planes_down2, dim3(g.grid), dim3(256), lds_of(g.ppb), st, x, k, out, p, g.ppb, rows, pitch, g.ngroups);
    return launch_status("upfirdn2d_planes_down2");
  }
  if (minor == 1 && kh * kw <= kMaxTaps && (int64_t)in_h * in_w <= 8192 && (int64_t)p.out_h * p.out_w <= 16384) {
    // whole planes in LDS: <= 32 KB per plane; as many planes per workgroup as fit 32 KB / ~16 outputs per thread
    const int psz = in_h * in_w, osz = p.out_h * p.out_w;
    int ppb = max(1, min((32 * 1024 / 4) / psz, 4096 / max(osz, 1)));
    ppb = max(1, min(ppb, max(1, major / 1024)));  // keep >= ~4 workgroups per CU when there are few planes
    int tx_log2 = 0;
    while ((1 << tx_log2) < min(p.out_w, 64)) ++tx_log2;
    const size_t lds_bytes = (kMaxTaps + (size_t)ppb * psz) * sizeof(float);
    const int uplog = (up_x == up_y && kh <= 4 && kw <= 4) ? (up_x == 1 ? 0 : up_x == 2 ? 1 : -1) : -1;
    const dim3 grid(ceil_div(major, ppb));
    if (uplog == 0)
      hipLaunchKernelGGL(upfirdn2d_planes_whole<0>, grid, dim3(256), lds_bytes, st, x, k, out, p, ppb, tx_log2);
    else if (uplog == 1)
      hipLaunchKernelGGL(upfirdn2d_planes_whole<1>, grid, dim3(256), lds_bytes, st, x, k, out, p, ppb, tx_log2);
    else
      hipLaunchKernelGGL(upfirdn2d_planes_whole<-1>, grid, dim3(256), lds_bytes, st, x, k, out, p, ppb, tx_log2);
    return launch_status("upfirdn2d_planes_whole");
  }
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
  if (minor % 4 == 0 && minor <= 1024 && kh * kw <= kMaxTaps && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
      (int64_t)major * p.out_h <= 0x7fffffff) {
    const int cv = minor / 4, col_step = 256 / cv;
    // enough rows to fill the chip: one workgroup per output row (each thread walks out_w / col_step pixels);
    // otherwise split rows over up to 8 workgroups
    const int gx = (int64_t)major * p.out_h >= 2048 ? 1 : max(1, min(ceil_div(p.out_w, col_step), 8));
    if (up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 && pad_x0 == 2 && pad_y0 == 2 &&
        p.out_h == 2 * in_h && p.out_w == 2 * in_w && (int64_t)major * in_h <= 0x7fffffff && !idiff::option(idiff::OPT_UFD_ROWS)) {
      hipLaunchKernelGGL(upfirdn2d_nhwc_up2_block, dim3(major * in_h), dim3(256), 0, st, x, k, out, p, cv, col_step);
      return launch_status("upfirdn2d_nhwc_up2_block");
    }
    const int uplog = (up_x == up_y && kh <= 4 && kw <= 4) ? (up_x == 1 ? 0 : up_x == 2 ? 1 : -1) : -1;
    const dim3 grid(major * p.out_h, gx);
    if (uplog == 0)
      hipLaunchKernelGGL(upfirdn2d_nhwc_rows<0>, grid, dim3(256), 0, st, x, k, out, p, cv, col_step);
    else if (uplog == 1)
      hipLaunchKernelGGL(upfirdn2d_nhwc_rows<1>, grid, dim3(256), 0, st, x, k, out, p, cv, col_step);
    else
      hipLaunchKernelGGL(upfirdn2d_nhwc_rows<-1>, grid, dim3(256), 0, st, x, k, out, p, cv, col_step);
    return launch_status("upfirdn2d_nhwc_rows");
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
