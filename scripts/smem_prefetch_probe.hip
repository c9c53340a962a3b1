// Can the SCALAR memory path (s_buffer_load -> scalar cache -> L2) pull lines into L2 ahead of a wave's vector loads, beside them, without
// sharing their in-order counter (vmcnt)?  winograd43h_kernel's K loop runs at the pace "4 positions = one HBM miss" because the 36 patch
// loads of a step miss to HBM and the U loads (L2 hits) queue behind them in the wave's one vector-memory counter.  A scalar load counts in
// lgkmcnt instead.  Two questions, each per CU with every CU doing the same:
//   (1) how many bytes per clock can 1 / 4 / 8 waves of a CU pull through the scalar path from HBM (lines never touched before), issuing
//       s_buffer_load_dwordx16 back to back with no wait (results discarded)?   needed by the kernel: ~8 B / clk / CU
//   (2) does a vector load of a line hit L2 once a scalar load has touched it?  dependent single-lane vector loads of fresh lines, with and
//       without a scalar touch of the same lines ~20 us earlier: average latency in shader clocks.
//   hipcc --offload-arch=gfx950 -O3 -o smem_prefetch_probe scripts/smem_prefetch_probe.hip && ./smem_prefetch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));

// the destination is a fixed block of high scalar registers nothing else uses (the data may land long after the instruction issued)
#define SMEM_TOUCH(rsrc, off) asm volatile("s_buffer_load_dwordx16 s[84:99], %0, %1" :: "s"(rsrc), "s"(off) \
  : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "memory")

__device__ __forceinline__ int4v make_rsrc(const void *p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  int4v r;
  r.x = (int)(uint32_t)a; r.y = (int)(uint32_t)(a >> 32) & 0xFFFF; r.z = (int)bytes; r.w = 0x00020000;
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
  r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  return r;
}

// (1) streaming through the scalar path: wave w of block b touches `lines` consecutive 64-byte chunks of its own region
__global__ void __launch_bounds__(512) smem_stream_kernel(const char *buf, uint32_t region_bytes, int lines, int active_waves, uint64_t *clk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  if (wave < active_waves) {
    const char *mine = buf + ((size_t)blockIdx.x * 8 + wave) * region_bytes;
    const int4v rs = make_rsrc(mine, region_bytes);
    for (int i = 0; i < lines; i += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int off = __builtin_amdgcn_readfirstlane((i + k) * 64);
        SMEM_TOUCH(rs, off);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// (2) latency of dependent vector loads of fresh lines (one lane), optionally touched by scalar loads `lead` iterations ahead
__global__ void __launch_bounds__(64) latency_kernel(const uint32_t *buf, uint32_t region_bytes, int n, int touch, uint64_t *clk, uint32_t *sink) {
  const char *mine = reinterpret_cast<const char *>(buf) + (size_t)blockIdx.x * region_bytes;
  const int4v rs = make_rsrc(mine, region_bytes);
  if (touch) {
    for (int i = 0; i < n; ++i) {
      const int off = __builtin_amdgcn_readfirstlane(i * 4096);     // one line per 4 KB: no two probes share a DRAM page burst or a 128-B line
      SMEM_TOUCH(rs, off);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    const uint32_t v = *reinterpret_cast<const volatile uint32_t *>(mine + (size_t)i * 4096 + (acc & 4));   // address depends on the previous value
    acc += v;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc; }
}

int main() {
  const size_t total = (size_t)3 << 30;                 // 3 GiB: far beyond the 256 MiB memory-side cache
  char *buf;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMemset(buf, 0, total));
  uint64_t *clk; uint32_t *sink;
  CHECK(hipMalloc(&clk, 4096 * 8)); CHECK(hipMalloc(&sink, 4096 * 4));
  std::vector<uint64_t> h(4096);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int G = 256;
  size_t cursor = 0;                                     // every measurement reads memory nothing has touched since the memset
  auto fresh = [&](size_t bytes) { if (cursor + bytes > total) cursor = 0; const size_t at = cursor; cursor += bytes; return at; };
  for (int waves : {1, 4, 8}) {
    for (int lines : {256, 2048}) {
      const uint32_t region = (uint32_t)lines * 64;
      const size_t at = fresh((size_t)G * 8 * region);
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(smem_stream_kernel, dim3(G), dim3(512), 0, 0, buf + at, region, lines, waves, clk);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h.data(), clk, G * 8, hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < G; ++i) sum += (double)h[i];
      const double per_cu = (double)waves * lines * 64 / (sum / G);
      printf("scalar stream: %d wave(s)/CU x %5d lines of 64 B: %8.0f clocks per CU = %6.2f B/clk/CU, chip %7.1f GB/s (event time %.1f us)\n", waves, lines,
             sum / G, per_cu, (double)G * waves * lines * 64 / (ms * 1e-3) / 1e9, ms * 1e3);
    }
  }
  for (int touch : {0, 1, 0, 1}) {
    const int n = 64;
    const uint32_t region = n * 4096;
    const size_t at = fresh((size_t)G * region);
    hipLaunchKernelGGL(latency_kernel, dim3(G), dim3(64), 0, 0, reinterpret_cast<const uint32_t *>(buf + at), region, n, touch, clk, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), clk, G * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (int i = 0; i < G; ++i) sum += (double)h[i];
    printf("dependent vector loads of fresh lines, %s: %7.0f clocks per load\n", touch ? "scalar-touched ~1 us earlier" : "untouched               ", sum / G / n);
  }
  return 0;
}
