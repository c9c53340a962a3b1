// Gate for the split-precision contraction: C[M x N] = A[M x K] . Bt[N x K]^T with fp32 inputs, each
// operand split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits: a = a1 + a2 + a3), the six products of weight
// >= 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; what is dropped is
// <= 2^-23 |a||b| per product, the size of one fp32 rounding.  B is split once (weights); A is split on the fly between its
// global load and the MFMA, which is the cost this probe has to show it can hide.
//   hipcc --offload-arch=gfx950 -O3 scripts/bf16x6_probe.hip -o /tmp/bf16x6_probe && /tmp/bf16x6_probe
// Prints: time and TFLOP/s (fp32-equivalent flops 2MNK) and GB/s of the kernel on [2240 * 1024 x 128] x [128 x 128], the
// error of a row sample against an fp64 contraction of the same fp32 inputs, and the same for plain fp32 summation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int K = 128, N = 128;
constexpr int BPITCH = K + 8;          // bf16 elements per LDS row of a B plane: 272 bytes = 17 x 16 (ds_read_b128 conflict-free)

// a = hi + lo exactly, hi = the top 8 mantissa bits of a (truncation keeps the split exact; rounding to nearest would too,
// but costs more instructions)
__device__ __forceinline__ unsigned top(float a) { return __float_as_uint(a) & 0xffff0000u; }

// eight fp32 -> three fragments of eight bf16 (element j of a fragment = bits 16 j .. 16 j + 15 of a 128-bit register quad)
__device__ __forceinline__ void split8(const float *a, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
  uintx4 q1, q2, q3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h1[2], h2[2], h3[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = a[2 * j + e];
      const unsigned t1 = top(x);
      const float r1 = x - __uint_as_float(t1);           // exact: the low 16 mantissa bits
      const unsigned t2 = top(r1);
      const float r2 = r1 - __uint_as_float(t2);          // exact: at most 8 significant bits -> a bf16 as it stands
      h1[e] = t1; h2[e] = t2; h3[e] = __float_as_uint(r2);
    }
    q1[j] = (h1[0] >> 16) | h1[1];                        // v_perm_b32 / v_and_or
    q2[j] = (h2[0] >> 16) | h2[1];
    q3[j] = (h3[0] >> 16) | (h3[1] & 0xffff0000u);
  }
  p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2); p3 = __builtin_bit_cast(bf16x8, q3);
}

__global__ void split_b_kernel(const float *__restrict__ Bt, unsigned short *__restrict__ planes /* [3][N][K] */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * K) return;
  const float x = Bt[i];
  const unsigned t1 = top(x);
  const float r1 = x - __uint_as_float(t1);
  const unsigned t2 = top(r1);
  const float r2 = r1 - __uint_as_float(t2);
  planes[i] = (unsigned short)(t1 >> 16);
  planes[N * K + i] = (unsigned short)(t2 >> 16);
  planes[2 * N * K + i] = (unsigned short)(__float_as_uint(r2) >> 16);
}

// Persistent workgroups of NW waves; a wave owns 32 rows per round and all N = 128 columns (4 accumulator tiles).  The rows
// of round i + 1 are requested before round i is computed (256 bytes per lane in flight per wave).
template <int NW>
__global__ void __launch_bounds__(64 * NW) gemm_bf16x6(const float *__restrict__ A, const unsigned short *__restrict__ planes,
                                                      float *__restrict__ C, int M, int rounds) {
  extern __shared__ unsigned short Bs[];                 // [3][N][BPITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int e = tid; e < 3 * N * (K / 8); e += 64 * NW) {     // 16-byte pieces
    const int p = e / (N * (K / 8)), rem = e % (N * (K / 8)), n = rem / (K / 8), k8 = rem % (K / 8);
    *reinterpret_cast<uintx4 *>(&Bs[(p * N + n) * BPITCH + 8 * k8]) =
        *reinterpret_cast<const uintx4 *>(&planes[(p * N + n) * K + 8 * k8]);
  }
  __syncthreads();
  auto row_of = [&](int rd) { return ((int64_t)(blockIdx.x + (int64_t)rd * gridDim.x) * NW + wave) * 32; };
  floatx4 nxt[16];
  auto fetch = [&](int rd) {
    const int64_t row0 = row_of(rd);
    const float *ap = A + (row0 < M ? row0 + r : 0) * K + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      nxt[2 * ks] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(ap + 16 * ks));
      nxt[2 * ks + 1] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(ap + 16 * ks + 4));
    }
  };
  fetch(0);
  for (int rd = 0; rd < rounds; ++rd) {
    const int64_t row0 = row_of(rd);
    if (row0 >= M) break;
    floatx4 cur[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    if (rd + 1 < rounds) fetch(rd + 1);
    floatx16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float av[8] = {cur[2 * ks].x, cur[2 * ks].y, cur[2 * ks].z, cur[2 * ks].w,
                           cur[2 * ks + 1].x, cur[2 * ks + 1].y, cur[2 * ks + 1].z, cur[2 * ks + 1].w};
      bf16x8 a1, a2, a3;
      split8(av, a1, a2, a3);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned short *bp = &Bs[(32 * t + r) * BPITCH + 16 * ks + 8 * h];
        const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(bp);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(bp + N * BPITCH);
        const bf16x8 b3 = *reinterpret_cast<const bf16x8 *>(bp + 2 * N * BPITCH);
        // smallest terms first
        // the weights as the A operand, the activations as the B operand: D = C^T tile, its rows (registers) are output
        // columns n, its columns (lanes) are rows m -- four consecutive n per register quad = one 16-byte store per lane
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[t], 0, 0, 0);
      }
    }
    // D[i][j]: j = lane & 31 = row m of C, i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = column n within the tile
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const floatx4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        __builtin_nontemporal_store(v, reinterpret_cast<floatx4 *>(&C[(row0 + r) * N + 32 * t + 8 * q + 4 * h]));
      }
  }
}

// the same contraction on the fp32 matrix cores (one product per element pair, k-ordered fp32 fma chain): the yardstick
__global__ void __launch_bounds__(256) gemm_f32(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M,
                                                int rounds) {
  extern __shared__ unsigned short raw[];
  float *Bsf = reinterpret_cast<float *>(raw);           // [N][K + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int e = tid; e < N * K; e += 256) Bsf[(e / K) * (K + 1) + e % K] = Bt[e];
  __syncthreads();
  for (int rd = 0; rd < rounds; ++rd) {
    const int64_t row0 = ((int64_t)(blockIdx.x + (int64_t)rd * gridDim.x) * 4 + wave) * 32;
    if (row0 >= M) break;
    floatx16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const float *ap = A + (row0 + r) * K;
#pragma unroll 4
    for (int k = 0; k < K; k += 2) {
      const float a = ap[k + h];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bsf[(32 * t + r) * (K + 1) + k + h], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[(row0 + row) * N + 32 * t + r] = acc[t][i];
      }
  }
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2240 * 1024;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs; C[%d x %d] = A[%d x %d] . Bt[%d x %d]^T\n", prop.name, prop.multiProcessorCount, M, N, M, K, N, K);
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
  for (auto &v : hA) v = rnd() * 3.0f;                   // activations of either sign
  for (auto &v : hB) v = rnd() * 0.09f;                  // weights ~ 1 / sqrt(K)
  float *A, *Bt, *C;
  unsigned short *planes;
  CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&Bt, hB.size() * 4)); CHECK(hipMalloc(&C, (size_t)M * N * 4));
  CHECK(hipMalloc(&planes, 3 * N * K * 2));
  CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Bt, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_b_kernel, dim3((N * K + 255) / 256), dim3(256), 0, 0, Bt, planes);
  const size_t lds6 = (size_t)3 * N * BPITCH * 2, lds32 = (size_t)N * (K + 1) * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x6<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x6<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> hC((size_t)4096 * N);
  for (int which = 0; which < 2; ++which) {
    for (int nw : {4, 8}) {
      if (which == 1 && nw == 8) continue;
      const int tiles = (M + 32 * nw - 1) / (32 * nw);
      const int grid = prop.multiProcessorCount, rounds = (tiles + grid - 1) / grid;
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (which == 0 && nw == 4) hipLaunchKernelGGL(gemm_bf16x6<4>, dim3(grid), dim3(256), lds6, 0, A, planes, C, M, rounds);
        else if (which == 0) hipLaunchKernelGGL(gemm_bf16x6<8>, dim3(grid), dim3(512), lds6, 0, A, planes, C, M, rounds);
        else hipLaunchKernelGGL(gemm_f32, dim3(grid), dim3(256), lds32, 0, A, Bt, C, M, rounds);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%-12s %d waves per CU: %8.3f ms  %7.1f TFLOP/s (2MNK)  %6.0f GB/s (A + C once)\n", which == 0 ? "bf16 x 6" : "fp32 MFMA",
             nw, best, 2.0 * M * N * K / (best * 1e-3) / 1e12, 4.0 * ((double)M * K + (double)M * N) / (best * 1e-3) / 1e9);
    }
    // accuracy on the first 4096 rows against an fp64 contraction of the same fp32 inputs
    CHECK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0;
    for (int i = 0; i < 4096; ++i)
      for (int n = 0; n < N; ++n) {
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)i * K + k] * (double)hB[(size_t)n * K + k]; ref += p; mag += fabs(p); }
        const double d = (double)hC[(size_t)i * N + n] - ref;
        num += d * d; den += ref * ref;
        if (fabs(d) / mag > worst) worst = fabs(d) / mag;
      }
    printf("%-12s error vs fp64: ||C - ref|| / ||ref|| = %.3e, max |C - ref| / sum|a b| = %.3e\n", which == 0 ? "bf16 x 6" : "fp32 MFMA",
           sqrt(num / den), worst);
  }
  return 0;
}
