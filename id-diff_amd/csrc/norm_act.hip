// HBM-bound passes of the score networks on gfx950: GroupNorm (+activation), row softmax, pointwise maps,
// embeddings and layout changes.  All kernels move 16 bytes per lane per access where the shape allows
// and grid-stride over at most 2048 workgroups (256 CUs x 8).
#include "common.h"
#include <math.h>

namespace {

using idiff::act_apply;

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics, NHWC.  Pass 1: per (sample, row-split) per-channel sum / sum of squares in fp64.
// A thread owns one float4 channel column and walks rows; the RP threads that share a column are
// combined through LDS.  Pass 2: per (sample, group) mean and rstd.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gn_partial_kernel(const float *__restrict__ x, int C, int c_off, int Ctot, int HW, int nsplit,
                  double *__restrict__ ws /* [B][nsplit][Ctot][2] */) {
  __shared__ double red[256 * 8];
  const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int CV = C >> 2;
  const int RP = 256 / CV;  // rows in flight per workgroup (CV <= 256 checked on the host)
  const int tid = threadIdx.x;
  const int r0 = tid / CV, c4 = tid - r0 * CV;
  const int rows_per = (HW + nsplit - 1) / nsplit;
  const int row_lo = sp * rows_per, row_hi = min(HW, row_lo + rows_per);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r0 < RP) {
    const float4 *xp = reinterpret_cast<const float4 *>(x) + (int64_t)b * HW * CV + c4;
    for (int r = row_lo + r0; r < row_hi; r += RP) {
      const float4 v = xp[(int64_t)r * CV];
      s[0] += v.x; q[0] += (double)v.x * v.x;
      s[1] += v.y; q[1] += (double)v.y * v.y;
      s[2] += v.z; q[2] += (double)v.z * v.z;
      s[3] += v.w; q[3] += (double)v.w * v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = q[j]; }
  __syncthreads();
  // thread c (< C) reduces channel c over the RP row-threads
  for (int c = tid; c < C; c += 256) {
    const int col = c >> 2, j = c & 3;
    double ss = 0, qq = 0;
    for (int r = 0; r < RP; ++r) { ss += red[(r * CV + col) * 8 + j]; qq += red[(r * CV + col) * 8 + 4 + j]; }
    double *o = ws + (((int64_t)b * nsplit + sp) * Ctot + c_off + c) * 2;
    o[0] = ss; o[1] = qq;
  }
}

__global__ void gn_finalize_kernel(const double *__restrict__ ws, int B, int nsplit, int Ctot, int G, int HW,
                                   float eps, float *__restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i - b * G, cpg = Ctot / G;
  double s = 0, q = 0;
  for (int sp = 0; sp < nsplit; ++sp) {
    const double *w = ws + (((int64_t)b * nsplit + sp) * Ctot + g * cpg) * 2;
    for (int c = 0; c < cpg; ++c) { s += w[2 * c]; q += w[2 * c + 1]; }
  }
  const double n = (double)cpg * HW;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// Statistics from per-tile column sums written by the producing contraction's epilogue (idiff_epilogue.colstats,
// layout [B][nsplit][C][2] per source): no pass over the activations at all.
__global__ void gn_finalize2_kernel(const double *__restrict__ ws1, int ns1, int C1, const double *__restrict__ ws2,
                                    int ns2, int C2, int B, int G, int HW, float eps, float *__restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i - b * G, Ctot = C1 + C2, cpg = Ctot / G;
  double s = 0, q = 0;
  for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
    const bool first = c < C1;
    const double *w = first ? ws1 + ((int64_t)b * ns1 * C1 + c) * 2 : ws2 + ((int64_t)b * ns2 * C2 + (c - C1)) * 2;
    const int ns = first ? ns1 : ns2, Cs = first ? C1 : C2;
    for (int sp = 0; sp < ns; ++sp) { s += w[(int64_t)sp * Cs * 2]; q += w[(int64_t)sp * Cs * 2 + 1]; }
  }
  const double n = (double)cpg * HW;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void __launch_bounds__(256)
gn_apply_kernel(const float *__restrict__ x, int C, const float *__restrict__ x2, int C2, int HW, int G,
                const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                const float *__restrict__ mod, int64_t ld_mod, int act, float *__restrict__ y, int64_t total_vec) {
  const int Ctot = C + C2, CVt = Ctot >> 2, cpg = Ctot / G;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CVt) * 4;
    const int64_t pix = v / CVt;
    const int b = (int)(pix / HW);
    float4 in;
    if (c < C) in = *reinterpret_cast<const float4 *>(x + pix * C + c);
    else in = *reinterpret_cast<const float4 *>(x2 + pix * C2 + (c - C));
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 be = *reinterpret_cast<const float4 *>(beta + c);
    float o[4] = {in.x, in.y, in.z, in.w};
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float *st = stats + 2 * ((int64_t)b * G + (c + j) / cpg);
      float v = (o[j] - st[0]) * st[1] * gg[j] + bb[j];
      // scale-shift conditioning h * (1 + scale) + shift between norm and activation (BeatGANsblocks.py:316-321)
      if (mod) v = v * (1.0f + mod[(int64_t)b * ld_mod + c + j]) + mod[(int64_t)b * ld_mod + Ctot + c + j];
      o[j] = act_apply(v, act);
    }
    reinterpret_cast<float4 *>(y)[v] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Row-structured apply: grid.y = sample, a thread owns one float4 channel column, so mean/rstd/gamma/beta (and
// the optional scale-shift modulation) collapse into per-thread constants y = act(x * s + t) and the pixel loop
// is a pure 16-byte stream.  (The flat form above spends ~8 integer divisions per element on index
// decomposition and ran at 39-44 % of HBM peak.)
__global__ void __launch_bounds__(256)
gn_apply_rows_kernel(const float *__restrict__ x, int C, const float *__restrict__ x2, int C2, int HW, int G,
                     const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                     const float *__restrict__ mod, int64_t ld_mod, int act, float *__restrict__ y, int rows_per_block,
                     const double *__restrict__ ws1, int ns1, const double *__restrict__ ws2, int ns2, float eps) {
  const int Ctot = C + C2, CVt = Ctot >> 2, cpg = Ctot / G;
  const int RP = 256 / CVt;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  // ws1 != nullptr: the statistics come straight from the per-tile column sums the producing contractions wrote
  // (gn_finalize2_kernel's arithmetic in the same order, so the same bits), which saves the separate finalize launch per
  // GroupNorm.  Thread g < G finishes group g of this sample once per workgroup and hands it over in LDS (every thread
  // redoing the sums of its own group read as many bytes as the workgroup streams when a sample has 32 row tiles).
  __shared__ float sh_stats[2 * 256];
  const bool shared_stats = ws1 != nullptr && G <= 256;
  if (shared_stats) {
    if (tid < G) {
      double sum = 0, sq = 0;
      for (int cc = tid * cpg; cc < (tid + 1) * cpg; ++cc) {
        const bool first = cc < C;
        const double *w = first ? ws1 + ((int64_t)b * ns1 * C + cc) * 2 : ws2 + ((int64_t)b * ns2 * C2 + (cc - C)) * 2;
        const int ns = first ? ns1 : ns2, Cs = first ? C : C2;
        for (int sp = 0; sp < ns; ++sp) { sum += w[(int64_t)sp * Cs * 2]; sq += w[(int64_t)sp * Cs * 2 + 1]; }
      }
      const double n = (double)cpg * HW;
      const double mean = sum / n;
      double var = sq / n - mean * mean;
      if (var < 0) var = 0;
      sh_stats[2 * tid] = (float)mean;
      sh_stats[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  const int r0 = tid / CVt, c4 = tid - r0 * CVt;
  if (r0 >= RP) return;
  const int c = c4 * 4;
  float mu[4], s[4], t[4];   // y = act((x - mu) * s + t): the mean is subtracted first, as torch does
  float st_mean[4], st_rstd[4];
  if (shared_stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { st_mean[j] = sh_stats[2 * ((c + j) / cpg)]; st_rstd[j] = sh_stats[2 * ((c + j) / cpg) + 1]; }
  } else if (ws1) {          // more than 256 groups: every thread finishes the group(s) of its own four channels
    int g_prev = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (c + j) / cpg;
      if (g != g_prev) {
        double sum = 0, sq = 0;
        for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) {
          const bool first = cc < C;
          const double *w = first ? ws1 + ((int64_t)b * ns1 * C + cc) * 2 : ws2 + ((int64_t)b * ns2 * C2 + (cc - C)) * 2;
          const int ns = first ? ns1 : ns2, Cs = first ? C : C2;
          for (int sp = 0; sp < ns; ++sp) { sum += w[(int64_t)sp * Cs * 2]; sq += w[(int64_t)sp * Cs * 2 + 1]; }
        }
        const double n = (double)cpg * HW;
        const double mean = sum / n;
        double var = sq / n - mean * mean;
        if (var < 0) var = 0;
        st_mean[j] = (float)mean;
        st_rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
        g_prev = g;
      } else {
        st_mean[j] = st_mean[j - 1]; st_rstd[j] = st_rstd[j - 1];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float *st = stats + 2 * ((int64_t)b * G + (c + j) / cpg);
      st_mean[j] = st[0]; st_rstd[j] = st[1];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float st[2] = {st_mean[j], st_rstd[j]};
    float sc = st[1] * gamma[c + j], sh = beta[c + j];
    if (mod) {
      const float m1 = 1.0f + mod[(int64_t)b * ld_mod + c + j];
      sc *= m1;
      sh = sh * m1 + mod[(int64_t)b * ld_mod + Ctot + c + j];
    }
    mu[j] = st[0]; s[j] = sc; t[j] = sh;
  }
  const bool first = c < C;
  const float4 *src = first ? reinterpret_cast<const float4 *>(x + (int64_t)b * HW * C + c)
                            : reinterpret_cast<const float4 *>(x2 + (int64_t)b * HW * C2 + (c - C));
  const int src_stride = (first ? C : C2) >> 2;
  float4 *dst = reinterpret_cast<float4 *>(y + (int64_t)b * HW * Ctot + c);
  const int p_lo = blockIdx.x * rows_per_block, p_hi = min(HW, p_lo + rows_per_block);
  // four independent 16-byte loads in flight per thread: with one, a CU holds 32 KB in flight and the kernel sat at
  // 4.5 TB/s (latency-bound); the activation's exp/rcp then overlap the next loads
  int p = p_lo + r0;
  for (; p + 3 * RP < p_hi; p += 4 * RP) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = src[(int64_t)(p + u * RP) * src_stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 o;
      o.x = act_apply((v[u].x - mu[0]) * s[0] + t[0], act);
      o.y = act_apply((v[u].y - mu[1]) * s[1] + t[1], act);
      o.z = act_apply((v[u].z - mu[2]) * s[2] + t[2], act);
      o.w = act_apply((v[u].w - mu[3]) * s[3] + t[3], act);
      dst[(int64_t)(p + u * RP) * CVt] = o;
    }
  }
  for (; p < p_hi; p += RP) {
    const float4 v = src[(int64_t)p * src_stride];
    float4 o;
    o.x = act_apply((v.x - mu[0]) * s[0] + t[0], act);
    o.y = act_apply((v.y - mu[1]) * s[1] + t[1], act);
    o.z = act_apply((v.z - mu[2]) * s[2] + t[2], act);
    o.w = act_apply((v.w - mu[3]) * s[3] + t[3], act);
    dst[(int64_t)p * CVt] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Row softmax: one wave per row, values kept in registers for cols <= 1024.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void __launch_bounds__(256)
softmax_rows_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + row * cols;
  float *yr = y + row * cols;
  constexpr int PL = 16;
  if (cols <= 64 * PL) {
    float v[PL];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < cols ? xr[c] * scale : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < cols ? expf(v[i] - m) : 0.f;
      s += v[i];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int c = lane + 64 * i;
      if (c < cols) yr[c] = v[i] * inv;
    }
  } else {
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c] * scale);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(xr[c] * scale - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) yr[c] = expf(xr[c] * scale - m) * inv;
  }
}

// Rows of <= 1024 columns with 16-byte alignment (the attention logits: 256 or 64 keys): a lane holds one or more runs of
// four columns, a wave walks rows with the next row's loads in flight, and the exponential is v_exp_f32 (as the SiLU of
// act_apply; the libm expf of the general kernel is ~25 VALU instructions per element, which held this pass at 3.3 TB/s).
template <int RUNS>
__global__ void __launch_bounds__(256)
softmax_rows_vec_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const float k = scale * 1.44269504088896340736f;     // exp(s x - m) = 2^(k x - m')
  float4 v[RUNS], nx[RUNS];
  auto load = [&](int64_t row, float4 *dst) {
#pragma unroll
    for (int i = 0; i < RUNS; ++i) {
      const int c = (lane + 64 * i) * 4;
      dst[i] = c < cols ? *reinterpret_cast<const float4 *>(x + row * cols + c) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
  };
  if (wave0 < rows) load(wave0, nx);
  for (int64_t row = wave0; row < rows; row += nwaves) {
#pragma unroll
    for (int i = 0; i < RUNS; ++i) v[i] = nx[i];
    if (row + nwaves < rows) load(row + nwaves, nx);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < RUNS; ++i) {
      v[i].x *= k; v[i].y *= k; v[i].z *= k; v[i].w *= k;
      m = fmaxf(fmaxf(fmaxf(m, v[i].x), fmaxf(v[i].y, v[i].z)), v[i].w);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RUNS; ++i) {
      v[i].x = __builtin_amdgcn_exp2f(v[i].x - m); v[i].y = __builtin_amdgcn_exp2f(v[i].y - m);
      v[i].z = __builtin_amdgcn_exp2f(v[i].z - m); v[i].w = __builtin_amdgcn_exp2f(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < RUNS; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < cols) *reinterpret_cast<float4 *>(y + row * cols + c) = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pointwise maps
// ------------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ void __launch_bounds__(256)
affine_act_kernel(const float *__restrict__ a, float *__restrict__ y, int64_t n, float alpha, float beta, int act,
                  const float *__restrict__ rowscale, int64_t inner) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC4) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (n >> 2); v += stride) {
      float4 in = reinterpret_cast<const float4 *>(a)[v];
      const float rs = rowscale ? rowscale[(v << 2) / inner] : 1.0f;  // inner % 4 == 0 on this path
      float4 o;
      o.x = act_apply(in.x * alpha + beta, act) * rs;
      o.y = act_apply(in.y * alpha + beta, act) * rs;
      o.z = act_apply(in.z * alpha + beta, act) * rs;
      o.w = act_apply(in.w * alpha + beta, act) * rs;
      reinterpret_cast<float4 *>(y)[v] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      const float rs = rowscale ? rowscale[i / inner] : 1.0f;
      y[i] = act_apply(a[i] * alpha + beta, act) * rs;
    }
  }
}

template <bool VEC4>
__global__ void __launch_bounds__(256)
add_scale_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int64_t n,
                 float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC4) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (n >> 2); v += stride) {
      const float4 p = reinterpret_cast<const float4 *>(a)[v], q = reinterpret_cast<const float4 *>(b)[v];
      reinterpret_cast<float4 *>(y)[v] =
          make_float4((p.x + q.x) * scale, (p.y + q.y) * scale, (p.z + q.z) * scale, (p.w + q.w) * scale);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = (a[i] + b[i]) * scale;
  }
}

__global__ void fourier_embed_kernel(const float *__restrict__ t, const float *__restrict__ W, float *__restrict__ out,
                                     int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  // x[:, None] * W[None, :] * 2 * np.pi, each product rounded to fp32 (layerspp.py:40); 2*pi_f32 is exact
  const float proj = __fmul_rn(__fmul_rn(t[b], W[j]), 6.2831854820251465f);
  out[(int64_t)b * 2 * half + j] = sinf(proj);
  out[(int64_t)b * 2 * half + half + j] = cosf(proj);
}

__global__ void positional_embed_kernel(const float *__restrict__ t, float *__restrict__ out, int B, int dim,
                                        float log_max_over, float neg_log_max, int mode) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  if (mode == 0) {
    // DDPM (models/layers.py:524-538): exp(arange * -(log(max)/(half-1))), [sin, cos]
    const float freq = expf(__fmul_rn((float)j, -log_max_over));
    const float arg = __fmul_rn(t[b], freq);
    out[(int64_t)b * dim + j] = sinf(arg);
    out[(int64_t)b * dim + half + j] = cosf(arg);
  } else {
    // guided-diffusion / BeatGANs (models/BeatGANs_nn.py:107-125): exp(-log(max) * arange / half), [cos, sin]
    const float freq = expf(__fdiv_rn(__fmul_rn(neg_log_max, (float)j), (float)half));
    const float arg = __fmul_rn(t[b], freq);
    out[(int64_t)b * dim + j] = cosf(arg);
    out[(int64_t)b * dim + half + j] = sinf(arg);
  }
}

__global__ void __launch_bounds__(256)
concat_cols_kernel(const float *__restrict__ a, int Ca, const float *__restrict__ b, int Cb, float *__restrict__ out,
                   int64_t total) {
  const int Ct = Ca + Cb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Ct);
    const int64_t r = i / Ct;
    out[i] = c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)];
  }
}

__global__ void __launch_bounds__(256)
concat_cols_vec4_kernel(const float *__restrict__ a, int Ca, const float *__restrict__ b, int Cb,
                        float *__restrict__ out, int64_t total_vec) {
  const int CVt = (Ca + Cb) >> 2;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CVt) * 4;
    const int64_t r = v / CVt;
    reinterpret_cast<float4 *>(out)[v] = c < Ca ? *reinterpret_cast<const float4 *>(a + r * Ca + c)
                                                : *reinterpret_cast<const float4 *>(b + r * Cb + (c - Ca));
  }
}

__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int C, int HW, int Cpad, float alpha,
                    float beta) {
  const int64_t total = (int64_t)B * HW * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t pix = i / Cpad;
    const int p = (int)(pix % HW);
    const int64_t b = pix / HW;
    y[i] = c < C ? x[(b * C + c) * HW + p] * alpha + beta : 0.f;
  }
}

__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int C, int HW, int Cpad,
                    const float *__restrict__ rowscale) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int64_t t = i / HW;
    const int c = (int)(t % C);
    const int64_t b = t / C;
    float v = x[(b * HW + p) * Cpad + c];
    if (rowscale) v *= rowscale[b];
    y[i] = v;
  }
}

__global__ void __launch_bounds__(256)
resample2x_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C, int up,
                  int64_t total_vec) {
  const int CV = C >> 2;
  const int OH = up ? H * 2 : H / 2, OW = up ? W * 2 : W / 2;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(v % CV);
    int64_t pix = v / CV;
    const int ox = (int)(pix % OW);
    pix /= OW;
    const int oy = (int)(pix % OH);
    const int64_t b = pix / OH;
    const float4 *xp = reinterpret_cast<const float4 *>(x) + b * H * W * CV + c4;
    float4 o;
    if (up) {
      o = xp[((int64_t)(oy >> 1) * W + (ox >> 1)) * CV];
    } else {
      const float4 p00 = xp[((int64_t)(2 * oy) * W + 2 * ox) * CV], p01 = xp[((int64_t)(2 * oy) * W + 2 * ox + 1) * CV];
      const float4 p10 = xp[((int64_t)(2 * oy + 1) * W + 2 * ox) * CV], p11 = xp[((int64_t)(2 * oy + 1) * W + 2 * ox + 1) * CV];
      o = make_float4((p00.x + p01.x + p10.x + p11.x) * 0.25f, (p00.y + p01.y + p10.y + p11.y) * 0.25f,
                      (p00.z + p01.z + p10.z + p11.z) * 0.25f, (p00.w + p01.w + p10.w + p11.w) * 0.25f);
    }
    reinterpret_cast<float4 *>(y)[v] = o;
  }
}

// out[r, :] = mean_coeff[r] * x[:] + std[r] * z[r, :]  (dim_reduction.py:180-182: mean + std * randn_like)
__global__ void __launch_bounds__(256)
perturb_kernel(const float *__restrict__ x, const float *__restrict__ z, const float *__restrict__ std_,
               const float *__restrict__ mean_coeff, float *__restrict__ out, int64_t rows, int64_t D) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, c = i - r * D;
    const float m = mean_coeff ? mean_coeff[r] * x[c] : x[c];
    out[i] = m + std_[r] * z[i];
  }
}

bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

using namespace idiff;

IDIFF_API int idiff_groupnorm_nsplit(int B, int HW, int C) {
  (void)C;
  if (B <= 0 || HW <= 0) return 1;
  int ns = ceil_div(1024, B);
  const int max_ns = HW >= 64 ? HW / 64 : 1;  // keep >= 64 rows per split
  if (ns > max_ns) ns = max_ns;
  return ns < 1 ? 1 : ns;
}

IDIFF_API int idiff_groupnorm_stats_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G,
                                        float eps, double *workspace, float *stats, void *stream) {
  if (!x || !workspace || !stats) return fail("groupnorm_stats: null pointer");
  if (!x2) C2 = 0;
  const int Ctot = C + C2;
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || Ctot % G != 0) return fail("groupnorm_stats: bad shape");
  if (C % 4 || C2 % 4 || C > 1024 || C2 > 1024) return fail("groupnorm_stats: channels must be multiples of 4 and <= 1024");
  if (!al16(x) || (x2 && !al16(x2))) return fail("groupnorm_stats: inputs must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = idiff_groupnorm_nsplit(B, HW, Ctot);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(B * nsplit), dim3(256), 0, st, x, C, 0, Ctot, HW, nsplit, workspace);
  if (C2) hipLaunchKernelGGL(gn_partial_kernel, dim3(B * nsplit), dim3(256), 0, st, x2, C2, C, Ctot, HW, nsplit, workspace);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(B * G, 256)), dim3(256), 0, st, workspace, B, nsplit, Ctot, G,
                     HW, eps, stats);
  return launch_status("groupnorm_stats");
}

IDIFF_API int idiff_groupnorm_finalize_f32(const double *ws1, int nsplit1, int C1, const double *ws2, int nsplit2, int C2,
                                           int B, int HW, int G, float eps, float *stats, void *stream) {
  if (!ws1 || !stats || B <= 0 || HW <= 0 || C1 <= 0 || nsplit1 <= 0 || G <= 0) return fail("groupnorm_finalize: bad arguments");
  if (!ws2) { C2 = 0; nsplit2 = 0; }
  if (ws2 && (C2 <= 0 || nsplit2 <= 0)) return fail("groupnorm_finalize: second source needs C2, nsplit2 > 0");
  if ((C1 + C2) % G) return fail("groupnorm_finalize: channels not divisible by groups");
  hipLaunchKernelGGL(gn_finalize2_kernel, dim3(ceil_div(B * G, 256)), dim3(256), 0, (hipStream_t)stream, ws1, nsplit1, C1, ws2,
                     nsplit2, C2, B, G, HW, eps, stats);
  return launch_status("groupnorm_finalize");
}

namespace {
int groupnorm_apply_impl(const float *x, int C, const float *x2, int C2, int B, int HW, int G, const float *stats,
                         const double *ws1, int ns1, const double *ws2, int ns2, float eps, const float *gamma,
                         const float *beta, const float *mod, int64_t ld_mod, int act, float *y, void *stream);
}

IDIFF_API int idiff_groupnorm_apply_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G,
                                        const float *stats, const float *gamma, const float *beta, const float *mod,
                                        int64_t ld_mod, int act, float *y, void *stream) {
  if (!stats) return fail("groupnorm_apply: null pointer");
  return groupnorm_apply_impl(x, C, x2, C2, B, HW, G, stats, nullptr, 0, nullptr, 0, 0.f, gamma, beta, mod, ld_mod, act, y, stream);
}

IDIFF_API int idiff_groupnorm_apply_colstats_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G,
                                                 const double *ws1, int nsplit1, const double *ws2, int nsplit2, float eps,
                                                 const float *gamma, const float *beta, const float *mod, int64_t ld_mod,
                                                 int act, float *y, void *stream) {
  if (!ws1 || nsplit1 <= 0) return fail("groupnorm_apply_colstats: the first source's column sums are required");
  if (!x2) { ws2 = nullptr; nsplit2 = 0; }
  if (x2 && (!ws2 || nsplit2 <= 0)) return fail("groupnorm_apply_colstats: the second source needs its column sums too");
  if ((C + (x2 ? C2 : 0)) / 4 > 256 || B > 65535)
    return fail("groupnorm_apply_colstats: more than 1024 channels or 65535 samples (use idiff_groupnorm_finalize_f32 + idiff_groupnorm_apply_f32)");
  return groupnorm_apply_impl(x, C, x2, C2, B, HW, G, nullptr, ws1, nsplit1, ws2, nsplit2, eps, gamma, beta, mod, ld_mod, act, y, stream);
}

namespace {
int groupnorm_apply_impl(const float *x, int C, const float *x2, int C2, int B, int HW, int G, const float *stats,
                         const double *ws1, int ns1, const double *ws2, int ns2, float eps, const float *gamma,
                         const float *beta, const float *mod, int64_t ld_mod, int act, float *y, void *stream) {
  if (!x || !gamma || !beta || !y) return fail("groupnorm_apply: null pointer");
  if (!x2) C2 = 0;
  const int Ctot = C + C2;
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || Ctot % G != 0 || C % 4 || C2 % 4) return fail("groupnorm_apply: bad shape");
  if (!al16(x) || (x2 && !al16(x2)) || !al16(y) || !al16(gamma) || !al16(beta))
    return fail("groupnorm_apply: pointers must be 16-byte aligned");
  const int64_t total_vec = (int64_t)B * HW * (Ctot / 4);
  const int CVt = Ctot / 4;
  if (CVt <= 256 && B <= 65535) {
    // ~2048 workgroups in total, each a contiguous range of pixels of one sample
    const int RP = 256 / CVt;
    int xblocks = max(1, min(ceil_div(HW, RP), ceil_div(2048, B)));
    const int rows_per_block = ceil_div(ceil_div(HW, xblocks), RP) * RP;
    xblocks = ceil_div(HW, rows_per_block);
    hipLaunchKernelGGL(gn_apply_rows_kernel, dim3(xblocks, B), dim3(256), 0, (hipStream_t)stream, x, C, x2, C2, HW, G, stats,
                       gamma, beta, mod, ld_mod, act, y, rows_per_block, ws1, ns1, ws2, ns2, eps);
    return launch_status("groupnorm_apply_rows");
  }
  hipLaunchKernelGGL(gn_apply_kernel, dim3(streaming_grid(total_vec, 256)), dim3(256), 0, (hipStream_t)stream, x, C, x2,
                     C2, HW, G, stats, gamma, beta, mod, ld_mod, act, y, total_vec);
  return launch_status("groupnorm_apply");
}
}  // namespace

IDIFF_API int idiff_softmax_rows_f32(const float *x, float *y, int64_t rows, int cols, float scale, void *stream) {
  if (rows < 0 || cols <= 0) return fail("softmax: bad shape");
  if (rows == 0) return 0;
  if (!x || !y) return fail("softmax: null pointer");
  const int64_t blocks = ceil_div64(rows, 4);
  if (blocks > 0x7fffffff) return fail("softmax: too many rows");
  if (cols % 4 == 0 && cols <= 1024 && al16(x) && al16(y) && scale > 0.f) {
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);       // 16 four-wave workgroups per CU walk the rows
    if (cols <= 256) hipLaunchKernelGGL(softmax_rows_vec_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, scale);
    else if (cols <= 512) hipLaunchKernelGGL(softmax_rows_vec_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, scale);
    else hipLaunchKernelGGL(softmax_rows_vec_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, scale);
    return launch_status("softmax_rows");
  }
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, scale);
  return launch_status("softmax_rows");
}

IDIFF_API int idiff_affine_act_f32(const float *a, float *y, int64_t n, float alpha, float beta_const, int act,
                                   const float *rowscale, int64_t inner, void *stream) {
  if (n < 0) return fail("affine_act: negative size");
  if (n == 0) return 0;
  if (!a || !y) return fail("affine_act: null pointer");
  if (rowscale && inner <= 0) return fail("affine_act: rowscale needs inner > 0");
  if (!rowscale) inner = 4;
  hipStream_t st = (hipStream_t)stream;
  if (n % 4 == 0 && inner % 4 == 0 && al16(a) && al16(y))
    hipLaunchKernelGGL(affine_act_kernel<true>, dim3(streaming_grid(n / 4, 256)), dim3(256), 0, st, a, y, n, alpha,
                       beta_const, act, rowscale, inner);
  else
    hipLaunchKernelGGL(affine_act_kernel<false>, dim3(streaming_grid(n, 256)), dim3(256), 0, st, a, y, n, alpha,
                       beta_const, act, rowscale, inner);
  return launch_status("affine_act");
}

IDIFF_API int idiff_add_scale_f32(const float *a, const float *b, float *y, int64_t n, float scale, void *stream) {
  if (n < 0) return fail("add_scale: negative size");
  if (n == 0) return 0;
  if (!a || !b || !y) return fail("add_scale: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n % 4 == 0 && al16(a) && al16(b) && al16(y))
    hipLaunchKernelGGL(add_scale_kernel<true>, dim3(streaming_grid(n / 4, 256)), dim3(256), 0, st, a, b, y, n, scale);
  else
    hipLaunchKernelGGL(add_scale_kernel<false>, dim3(streaming_grid(n, 256)), dim3(256), 0, st, a, b, y, n, scale);
  return launch_status("add_scale");
}

IDIFF_API int idiff_fourier_embed_f32(const float *t, const float *W, float *out, int B, int half, void *stream) {
  if (B <= 0 || half <= 0 || !t || !W || !out) return fail("fourier_embed: bad arguments");
  hipLaunchKernelGGL(fourier_embed_kernel, dim3(ceil_div(B * half, 256)), dim3(256), 0, (hipStream_t)stream, t, W, out, B, half);
  return launch_status("fourier_embed");
}

IDIFF_API int idiff_positional_embed_f32(const float *t, float *out, int B, int dim, float max_positions, int mode,
                                         void *stream) {
  if (B <= 0 || dim < 4 || dim % 2 || !t || !out) return fail("positional_embed: bad arguments (dim must be even, >= 4)");
  if (mode != 0 && mode != 1) return fail("positional_embed: mode must be 0 (DDPM) or 1 (guided-diffusion)");
  const float neg_log_max = (float)(-log((double)max_positions));
  // math.log(max_positions) / (half_dim - 1) is a Python float, rounded to fp32 when it meets the tensor
  const float log_max_over = (float)(log((double)max_positions) / (double)(dim / 2 - 1));
  hipLaunchKernelGGL(positional_embed_kernel, dim3(ceil_div(B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, out, B,
                     dim, log_max_over, neg_log_max, mode);
  return launch_status("positional_embed");
}

IDIFF_API int idiff_concat_cols_f32(const float *a, int Ca, const float *b, int Cb, float *out, int64_t rows, void *stream) {
  if (rows < 0 || Ca <= 0 || Cb <= 0) return fail("concat_cols: bad shape");
  if (rows == 0) return 0;
  if (!a || !b || !out) return fail("concat_cols: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = rows * (Ca + Cb);
  if (Ca % 4 == 0 && Cb % 4 == 0 && al16(a) && al16(b) && al16(out))
    hipLaunchKernelGGL(concat_cols_vec4_kernel, dim3(streaming_grid(total / 4, 256)), dim3(256), 0, st, a, Ca, b, Cb, out, total / 4);
  else
    hipLaunchKernelGGL(concat_cols_kernel, dim3(streaming_grid(total, 256)), dim3(256), 0, st, a, Ca, b, Cb, out, total);
  return launch_status("concat_cols");
}

IDIFF_API int idiff_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int HW, int Cpad, float alpha, float beta,
                                     void *stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || Cpad < C || !x || !y) return fail("nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(streaming_grid((int64_t)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, C,
                     HW, Cpad, alpha, beta);
  return launch_status("nchw_to_nhwc");
}

IDIFF_API int idiff_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int HW, int Cpad, const float *rowscale,
                                     void *stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || Cpad < C || !x || !y) return fail("nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(streaming_grid((int64_t)B * HW * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, C,
                     HW, Cpad, rowscale);
  return launch_status("nhwc_to_nchw");
}

IDIFF_API int idiff_resample2x_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int up, void *stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || !x || !y) return fail("resample2x: bad arguments");
  if (!up && (H % 2 || W % 2)) return fail("resample2x: downsampling needs even H, W");
  if (!al16(x) || !al16(y)) return fail("resample2x: pointers must be 16-byte aligned");
  const int64_t total_vec = up ? (int64_t)B * H * W * C : (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(resample2x_kernel, dim3(streaming_grid(total_vec, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, up,
                     total_vec);
  return launch_status("resample2x");
}

IDIFF_API int idiff_perturb_f32(const float *x, const float *z, const float *std_, const float *mean_coeff, float *out,
                                int64_t rows, int64_t D, void *stream) {
  if (rows < 0 || D <= 0) return fail("perturb: bad shape");
  if (rows == 0) return 0;
  if (!x || !z || !std_ || !out) return fail("perturb: null pointer");
  hipLaunchKernelGGL(perturb_kernel, dim3(streaming_grid(rows * D, 256)), dim3(256), 0, (hipStream_t)stream, x, z, std_,
                     mean_coeff, out, rows, D);
  return launch_status("perturb");
}
