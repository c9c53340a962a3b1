// What the matrix cores of THIS card sustain with nothing else going on: back-to-back independent MFMAs from registers,
// no memory traffic, two waves per SIMD on every CU, for a few milliseconds (long enough for the power management to
// settle).  The rate gives the clock the chip holds under that load: cycles = MFMAs per SIMD x (passes x 4).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Used to put the spec-sheet peaks bench.py divides by (157.3 TFLOP/s fp32, 78.6 fp64) next to a measured ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef double doublex4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 2) spin_f32(float *sink, int iters, float a0, float b0) {
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // a0 = 0: constant operands (1.0 x 1e-30: almost nothing toggles).  a0 != 0: per-lane pseudo-random operands of either
  // sign, a different pair for every MFMA of the iteration: accumulators that move like a real contraction's
  float a[8], b[8];
  unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = a0 == 0.f ? 1.0f : ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    h = h * 1664525u + 1013904223u; b[i] = a0 == 0.f ? b0 : ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
  }
  for (int it = 0; it < iters; it += 8) {      // unrolled by 8 so that every operand index is static
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + u) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;      // never true: keeps the loop alive without a store
}

__global__ void __launch_bounds__(256, 2) spin_f64(double *sink, int iters, double a0, double b0) {
  doublex4 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
  double a[8], b[8];
  unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = a0 == 0.0 ? 1.0 : ((int)(h >> 8) - (1 << 23)) * (1.0 / (1 << 23)) + h * 1e-19;
    h = h * 1664525u + 1013904223u; b[i] = a0 == 0.0 ? b0 : ((int)(h >> 8) - (1 << 23)) * (1.0 / (1 << 23)) + h * 1e-19;
  }
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + u) & 7], acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 12345.678) sink[0] = s;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, clockRate %.0f MHz\n", prop.name, cus, prop.clockRate / 1e3);
  float *sink; CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = cus * 2;            // two 4-wave workgroups per CU = two waves per SIMD
  for (int pass = 0; pass < 4; ++pass) {
    const bool f64 = pass & 1, random_operands = pass >= 2;
    // fp32 32x32x2: 16 passes = 64 cycles, 4096 flop; fp64 16x16x4: 64 cycles too, 2048 flop
    const double flop = f64 ? 2048.0 : 4096.0, cyc = 64.0;
    for (int iters : {20000, 400000}) {
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {     // the last repetition is the one reported (clocks settled)
        CHECK(hipEventRecord(e0, 0));
        if (f64) hipLaunchKernelGGL(spin_f64, dim3(grid), dim3(256), 0, 0, (double *)sink, iters, random_operands ? 1.0 : 0.0, 1e-30);
        else hipLaunchKernelGGL(spin_f32, dim3(grid), dim3(256), 0, 0, sink, iters, random_operands ? 1.0f : 0.f, 1e-30f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double mfma_per_simd = (double)iters * 8 * 2;           // 2 waves per SIMD, 8 MFMAs per iteration
      const double tf = mfma_per_simd * 4 * cus * flop / (ms * 1e-3) / 1e12;
      const double ghz = mfma_per_simd * cyc / (ms * 1e-3) / 1e9;
      printf("%s %s %7d iterations: %8.3f ms  %6.1f TFLOP/s  -> %.3f GHz if the pipe never idles\n", f64 ? "fp64 16x16x4" : "fp32 32x32x2",
             random_operands ? "random operands  " : "constant operands", iters, ms, tf, ghz);
    }
  }
  return 0;
}
