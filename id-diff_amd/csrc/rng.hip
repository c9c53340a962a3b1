// Counter-based Gaussian noise fused with the SDE perturbation (replaces `z = torch.randn_like(batch)` +
// `batch = mean + std * z` of dim_reduction.py:180-182 in one pass; the reference draws from the unseeded global
// torch generator, so any reproducible N(0, 1) stream is a valid stand-in).
//
// Philox4x32-10 (Salmon et al., SC'11): key = 64-bit seed, counter = index of the 4-element group inside the
// logical [total_rows, D] noise matrix of one data point, so the draw for element (r, c) does not depend on how
// rows are cut into launch sets or distributed over GPUs.  Four uniforms -> two Box-Muller pairs -> four normals
// -> one 16-byte store per lane.
#include "common.h"

namespace {

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 ctr, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = {hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
    k0 += W0; k1 += W1;
  }
  return ctr;
}

__device__ __forceinline__ float u01(uint32_t v) {   // (0, 1]: never feeds log(0)
  return ((float)(v >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256)
perturb_randn_kernel(const float *__restrict__ x, const float *__restrict__ std_, const float *__restrict__ mean_coeff,
                     float *__restrict__ out, int64_t rows, int64_t D, int64_t row0, uint32_t k0, uint32_t k1,
                     float *__restrict__ z_out) {
  const int64_t groups = rows * D / 4;   // D % 4 == 0
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = gidx * 4, r = e / D, c = e - r * D;
    const uint64_t ctr = (uint64_t)((row0 + r) * D + c) >> 2;
    const u4 rnd = philox4x32_10({(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u}, k0, k1);
    const float r0 = sqrtf(-2.0f * logf(u01(rnd.x))), r1 = sqrtf(-2.0f * logf(u01(rnd.z)));
    float s0, c0, s1, c1;
    sincosf(6.2831853071795864f * u01(rnd.y), &s0, &c0);
    sincosf(6.2831853071795864f * u01(rnd.w), &s1, &c1);
    const float4 z = make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
    const float4 xv = *reinterpret_cast<const float4 *>(x + c);
    const float m = mean_coeff ? mean_coeff[r] : 1.0f, sd = std_[r];
    *reinterpret_cast<float4 *>(out + e) = make_float4(m * xv.x + sd * z.x, m * xv.y + sd * z.y, m * xv.z + sd * z.z, m * xv.w + sd * z.w);
    if (z_out) *reinterpret_cast<float4 *>(z_out + e) = z;
  }
}

}  // namespace

IDIFF_API int idiff_perturb_randn_f32(const float *x, const float *std_, const float *mean_coeff, float *out, int64_t rows,
                                      int64_t D, int64_t row0, uint64_t seed, float *z_out, void *stream) {
  using namespace idiff;
  if (rows < 0 || D <= 0 || row0 < 0) return fail("perturb_randn: bad shape");
  if (rows == 0) return 0;
  if (!x || !std_ || !out) return fail("perturb_randn: null pointer");
  if (D % 4 != 0) return fail("perturb_randn: D must be a multiple of 4 (got %lld)", (long long)D);
  if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || (z_out && ((uintptr_t)z_out & 15)))
    return fail("perturb_randn: x, out and z_out must be 16-byte aligned");
  hipLaunchKernelGGL(perturb_randn_kernel, dim3(streaming_grid(rows * D / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, std_,
                     mean_coeff, out, rows, D, row0, (uint32_t)seed, (uint32_t)(seed >> 32), z_out);
  return launch_status("perturb_randn");
}
