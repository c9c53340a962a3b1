// How long a v_mfma_f32_32x32x16_bf16 takes when the next one needs its result: chains of dependent MFMAs on 1, 2, 4 or 8
// accumulators, one wave per SIMD on every CU.  cycles per MFMA = elapsed x clock / MFMAs per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_bf16_chain.hip -o /tmp/mfma_bf16_chain && /tmp/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) chain(float *sink, int iters, unsigned seed) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uintx4 qa = {seed + threadIdx.x, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u}, qb = {0x3f803f80u, seed * 3u + threadIdx.x, 0x3f803f80u, 0x3f003f00u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, qa), b = __builtin_bit_cast(bf16x8, qb);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        // volatile: keeps the round-robin order (the scheduler otherwise groups the MFMAs of one accumulator)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
  }
  asm volatile("s_nop 15\ns_nop 15\ns_nop 15");
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;
}

template <int NACC>
void run(float *sink, int cus, double mhz) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000 / NACC;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain<NACC>, dim3(cus), dim3(256), 0, 0, sink, iters, 12345u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double mfmas = (double)iters * 8 * NACC;
  printf("%d accumulator(s) in turn: %.1f ns per MFMA per SIMD = %.1f cycles at %.0f MHz; %.0f TFLOP/s\n", NACC, ms * 1e6 / mfmas,
         ms * 1e-3 / mfmas * mhz * 1e6, mhz, mfmas * 32768.0 * cus * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  float *sink; CHECK(hipMalloc(&sink, 64));
  const double mhz = 2400.0;
  run<1>(sink, prop.multiProcessorCount, mhz);
  run<2>(sink, prop.multiProcessorCount, mhz);
  run<4>(sink, prop.multiProcessorCount, mhz);
  run<8>(sink, prop.multiProcessorCount, mhz);
  return 0;
}
