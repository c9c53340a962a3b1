// fused_bias_act for gfx950: out = f(x + bias[channel]) * scale in one pass over HBM.
//
// Semantics follow op/fused_bias_act_kernel.cu:18-49: channel = (i / step_b) % size_b with
// step_b = prod(dims[2:]); mode = act*10 + grad: 10/11 identity, 12/32 zero, 30 leaky-relu on x,
// 31 leaky-relu gated by the sign of `ref` (first-order gradient).
// Algorithmic traffic 4*(2*numel + C) bytes -> HBM-bound; each lane moves 16 B per access
// (global_load_dwordx4 / global_store_dwordx4), grid-stride over at most 2048 workgroups.
#include "common.h"

namespace {

__device__ __forceinline__ float fba_one(float x, float ref, int mode, float alpha) {
  switch (mode) {
    case 10: case 11: return x;
    case 30: return x > 0.f ? x : x * alpha;
    case 31: return ref > 0.f ? x : x * alpha;
    default: return 0.f;  // 12, 32: second derivative of a piecewise-linear map
  }
}

template <bool VEC4>
__global__ void __launch_bounds__(256)
fused_bias_act_kernel(const float *__restrict__ x, const float *__restrict__ b, const float *__restrict__ ref,
                      float *__restrict__ out, int64_t n, int step_b, int size_b, int mode, float alpha,
                      float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC4) {
    // step_b % 4 == 0: the four lanes of a float4 share one channel
    const int64_t nv = n >> 2;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
      float4 xv = reinterpret_cast<const float4 *>(x)[v];
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ref) rv = reinterpret_cast<const float4 *>(ref)[v];
      float bias = 0.f;
      if (b) bias = b[((v << 2) / step_b) % size_b];
      float4 o;
      o.x = fba_one(xv.x + bias, rv.x, mode, alpha) * scale;
      o.y = fba_one(xv.y + bias, rv.y, mode, alpha) * scale;
      o.z = fba_one(xv.z + bias, rv.z, mode, alpha) * scale;
      o.w = fba_one(xv.w + bias, rv.w, mode, alpha) * scale;
      reinterpret_cast<float4 *>(out)[v] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      float v = x[i];
      if (b) v += b[(i / step_b) % size_b];
      out[i] = fba_one(v, ref ? ref[i] : 0.f, mode, alpha) * scale;
    }
  }
}

// The other two dtypes of the reference's dispatch (op/fused_bias_act_kernel.cu:79).  The reference kernel computes in
// scalar_t, its float arguments alpha and scale converted to scalar_t at the call (.cu:19, 80-93): for at::Half every
// operation is an fp32 operation on the widened operands rounded back to half -- x = h(x + b), y = x > 0 ? x : h(x * h(alpha)),
// out = h(y * h(scale)) -- and that is what T = _Float16 does here, operation by operation; T = double widens alpha and scale.
// VEC elements (16 bytes) per lane and access when the geometry allows (step_b % VEC == 0: the lanes of a vector share a channel).
// (OP = the type an operation is carried out in before its result is rounded to T: float for half -- c10::Half's operators
// widen, operate and round -- and double for double.)
template <typename T> struct FbaVec;
template <> struct FbaVec<_Float16> { static constexpr int n = 8; typedef float op; };
template <> struct FbaVec<double> { static constexpr int n = 2; typedef double op; };

template <typename T>
__device__ __forceinline__ T fba_one_t(T x, T ref, int mode, T alpha) {
  typedef typename FbaVec<T>::op OP;
  switch (mode) {
    case 10: case 11: return x;
    case 30: return x > (T)0 ? x : (T)((OP)x * (OP)alpha);
    case 31: return ref > (T)0 ? x : (T)((OP)x * (OP)alpha);
    default: return (T)0;
  }
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
fused_bias_act_kernel_t(const T *__restrict__ x, const T *__restrict__ b, const T *__restrict__ ref, T *__restrict__ out, int64_t n,
                        int step_b, int size_b, int mode, T alpha, T scale) {
  constexpr int V = FbaVec<T>::n;
  typedef typename FbaVec<T>::op OP;
  struct alignas(16) Pack { T e[V]; };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int64_t nv = n / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
      const Pack xv = reinterpret_cast<const Pack *>(x)[v];
      Pack rv, o;
      if (ref) rv = reinterpret_cast<const Pack *>(ref)[v];
      const T bias = b ? b[((v * V) / step_b) % size_b] : (T)0;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T xe = b ? (T)((OP)xv.e[e] + (OP)bias) : xv.e[e];
        o.e[e] = (T)((OP)fba_one_t<T>(xe, ref ? rv.e[e] : (T)0, mode, alpha) * (OP)scale);
      }
      reinterpret_cast<Pack *>(out)[v] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      T v = x[i];
      if (b) v = (T)((OP)v + (OP)b[(i / step_b) % size_b]);
      out[i] = (T)((OP)fba_one_t<T>(v, ref ? ref[i] : (T)0, mode, alpha) * (OP)scale);
    }
  }
}

template <typename T>
int fba_launch_t(const void *x, const void *b, const void *ref, void *out, int64_t n, int step_b, int size_b, int act, int grad,
                 float alpha, float scale, void *stream, const char *what) {
  using namespace idiff;
  if (n < 0) return fail("fused_bias_act: negative size");
  if (n == 0) return 0;
  if (!x || !out) return fail("fused_bias_act: null pointer");
  if (act != 1 && act != 3) return fail("fused_bias_act: act must be 1 (linear) or 3 (lrelu), got %d", act);
  if (grad < 0 || grad > 2) return fail("fused_bias_act: grad must be 0, 1 or 2, got %d", grad);
  if (act == 3 && grad == 1 && !ref) return fail("fused_bias_act: grad=1 needs the reference tensor");
  if (b && (size_b <= 0 || step_b <= 0)) return fail("fused_bias_act: bias given but step_b/size_b not positive");
  if (!b) { size_b = 1; step_b = 1; }
  const int mode = act * 10 + grad;
  if (!(act == 3 && grad == 1)) ref = nullptr;
  constexpr int V = FbaVec<T>::n;
  const bool vec = (n % V == 0) && (step_b % V == 0 || !b) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                   (!ref || (uintptr_t)ref % 16 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL((fused_bias_act_kernel_t<T, true>), dim3(streaming_grid(n / V, 256)), dim3(256), 0, st, (const T *)x,
                       (const T *)b, (const T *)ref, (T *)out, n, step_b, size_b, mode, (T)alpha, (T)scale);
  else
    hipLaunchKernelGGL((fused_bias_act_kernel_t<T, false>), dim3(streaming_grid(n, 256)), dim3(256), 0, st, (const T *)x,
                       (const T *)b, (const T *)ref, (T *)out, n, step_b, size_b, mode, (T)alpha, (T)scale);
  return launch_status(what);
}

}  // namespace

IDIFF_API int idiff_fused_bias_act_f16(const void *x, const void *b, const void *ref, void *out, int64_t n, int step_b, int size_b,
                                       int act, int grad, float alpha, float scale, void *stream) {
  return fba_launch_t<_Float16>(x, b, ref, out, n, step_b, size_b, act, grad, alpha, scale, stream, "fused_bias_act_f16");
}

IDIFF_API int idiff_fused_bias_act_f64(const double *x, const double *b, const double *ref, double *out, int64_t n, int step_b,
                                       int size_b, int act, int grad, float alpha, float scale, void *stream) {
  return fba_launch_t<double>(x, b, ref, out, n, step_b, size_b, act, grad, alpha, scale, stream, "fused_bias_act_f64");
}

IDIFF_API int idiff_fused_bias_act_f32(const float *x, const float *b, const float *ref, float *out, int64_t n,
                                       int step_b, int size_b, int act, int grad, float alpha, float scale,
                                       void *stream) {
  using namespace idiff;
  if (n < 0) return fail("fused_bias_act: negative size");
  if (n == 0) return 0;
  if (!x || !out) return fail("fused_bias_act: null pointer");
  if (act != 1 && act != 3) return fail("fused_bias_act: act must be 1 (linear) or 3 (lrelu), got %d", act);
  if (grad < 0 || grad > 2) return fail("fused_bias_act: grad must be 0, 1 or 2, got %d", grad);
  if (act == 3 && grad == 1 && !ref) return fail("fused_bias_act: grad=1 needs the reference tensor");
  if (b && (size_b <= 0 || step_b <= 0)) return fail("fused_bias_act: bias given but step_b/size_b not positive");
  if (!b) { size_b = 1; step_b = 1; }
  const int mode = act * 10 + grad;
  if (!(act == 3 && grad == 1)) ref = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (n % 4 == 0) && (step_b % 4 == 0 || !b) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                   (!ref || (uintptr_t)ref % 16 == 0);
  if (vec) {
    hipLaunchKernelGGL(fused_bias_act_kernel<true>, dim3(streaming_grid(n / 4, 256)), dim3(256), 0, st, x, b, ref,
                       out, n, step_b, size_b, mode, alpha, scale);
  } else {
    hipLaunchKernelGGL(fused_bias_act_kernel<false>, dim3(streaming_grid(n, 256)), dim3(256), 0, st, x, b, ref, out,
                       n, step_b, size_b, mode, alpha, scale);
  }
  return launch_status("fused_bias_act");
}
