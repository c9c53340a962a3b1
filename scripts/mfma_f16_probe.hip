// What v_mfma_f32_32x32x16_f16 does with fp16 SUBNORMAL operands, and how exact a contraction on fp16 pairs is (DESIGN.md 4.1:
// the F(4x4,3x3) kernel's contraction on hi/lo fp16 pairs).  fp32 v = hi + lo, hi = fp16(v), lo = fp16(v - hi); for |v| < 2^-3 the
// low part is an fp16 subnormal, so the whole scheme stands on the matrix core NOT flushing those.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe scripts/mfma_f16_probe.hip && ./mfma_f16_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// D[32][32] = sum over the planes listed: A_pa[32][16] * B_pb[16][32]; a, b: fp32 [32][16] and [16][32]; mode 0: hi*hi only,
// 1: hi*hi + hi*lo + lo*hi, 2: all four
__global__ void contract(const float *a, const float *b, float *d, int mode, float sa, float sb) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  halfx8 ah, al, bh, bl;
  for (int j = 0; j < 8; ++j) {
    const float va = a[r * 16 + 8 * h + j] * sa, vb = b[(8 * h + j) * 32 + r] * sb;
    const _Float16 x = (_Float16)va, y = (_Float16)vb;
    ah[j] = x; al[j] = (_Float16)(va - (float)x);
    bh[j] = y; bl[j] = (_Float16)(vb - (float)y);
  }
  floatx16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  if (mode >= 1) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  }
  if (mode >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
  const float inv = 1.0f / (sa * sb);
  for (int reg = 0; reg < 16; ++reg) d[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg] * inv;
}

int main() {
  float *a, *b, *d;
  CHECK(hipMalloc(&a, 32 * 16 * 4)); CHECK(hipMalloc(&b, 16 * 32 * 4)); CHECK(hipMalloc(&d, 32 * 32 * 4));
  std::vector<float> ha(512), hb(512), hd(1024);
  // 1. subnormal operands: A[.][0] = 2^-20 (an fp16 subnormal), B[0][.] = 1024 -> 2^-10 unless flushed
  for (auto &v : ha) v = 0; for (auto &v : hb) v = 0;
  for (int r = 0; r < 32; ++r) { ha[r * 16] = ldexpf(1.f, -20); hb[r] = 1024.f; }
  CHECK(hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice));
  contract<<<1, 64>>>(a, b, d, 0, 1.f, 1.f);
  CHECK(hipMemcpy(hd.data(), d, 4096, hipMemcpyDeviceToHost));
  printf("subnormal A (2^-20) x 1024: D = %.6g (exact %.6g): %s\n", hd[0], ldexp(1.0, -10), hd[0] == ldexpf(1.f, -10) ? "kept" : "FLUSHED or wrong");
  // the low part of 0.1 is subnormal in fp16 (0.1 - fp16(0.1) = 2.4e-5): A = 0.1, B = 1 over k = 0 with the pair scheme
  for (auto &v : ha) v = 0; for (auto &v : hb) v = 0;
  for (int r = 0; r < 32; ++r) { ha[r * 16] = 0.1f; hb[r] = 1.f; }
  CHECK(hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice));
  contract<<<1, 64>>>(a, b, d, 1, 1.f, 1.f);
  CHECK(hipMemcpy(hd.data(), d, 4096, hipMemcpyDeviceToHost));
  printf("0.1 as a pair x 1: D = %.9g, fp32 0.1 = %.9g, rel err %.2e (2^-11 = 4.9e-4 if the subnormal low part were dropped)\n", hd[0], 0.1f,
         fabs((double)hd[0] - (double)0.1f) / 0.1);
  // 2. random contractions: error of the pair scheme against fp64 of the same fp32 operands, beside an fp32 fma chain
  srand(1);
  for (int trial = 0; trial < 3; ++trial) {
    const float sa = trial == 2 ? 0.25f : 1.f, sb = trial == 0 ? 1.f : 4096.f;
    auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
    for (auto &v : ha) v = 2.f * rnd(); for (auto &v : hb) v = 0.05f * rnd();
    CHECK(hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; ++mode) {
      contract<<<1, 64>>>(a, b, d, mode, sa, sb);
      CHECK(hipMemcpy(hd.data(), d, 4096, hipMemcpyDeviceToHost));
      double num = 0, den = 0, num32 = 0;
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0; float f = 0;
        for (int k = 0; k < 16; ++k) { ref += (double)ha[i * 16 + k] * (double)hb[k * 32 + j]; f = fmaf(ha[i * 16 + k], hb[k * 32 + j], f); }
        num += (hd[i * 32 + j] - ref) * (hd[i * 32 + j] - ref); den += ref * ref; num32 += (f - ref) * (f - ref);
      }
      printf("A ~ 2 N(0,1) x %.2g, B ~ 0.05 N(0,1) x %g, K = 16, %s: rel err (Frobenius) %.2e   [fp32 fma chain: %.2e]\n", sa, sb,
             mode == 0 ? "hi*hi" : mode == 1 ? "hi*hi + hi*lo + lo*hi" : "all four products", sqrt(num / den), sqrt(num32 / den));
    }
  }
  return 0;
}
