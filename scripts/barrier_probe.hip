// Price of a grid-wide dependency on MI355X for SMALL grids (24 .. 256 workgroups), three ways:
//   (1) a kernel boundary: N dependent launches of a kernel that reads a 32x32 fp64 partial per workgroup (written by the previous
//       launch), sums all partials (every workgroup, as panel_q / panel_v / trailing_tz do) and writes its own partial;
//   (2) the same work inside ONE persistent launch with a fence-free counter barrier: partials stored write-through (sc1
//       atomics), every storing wave drains, one lane adds to an agent-scope counter, one lane polls it with sc1 loads;
//   (3) the bare barrier (no payload).
// Decides whether the band reduction's 7 launches per panel are worth fusing into a persistent kernel.
//   hipcc --offload-arch=gfx950 -O3 -o barrier_probe scripts/barrier_probe.hip && ./barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int E = 1024;    // doubles per partial

__device__ __forceinline__ double work(const double *in, int G, int tid) {
  // every workgroup sums all G partials: 256 threads x 4 consecutive elements, as reduce_partials
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const double2 *src = reinterpret_cast<const double2 *>(in + tid * 4);
#pragma unroll 8
  for (int c = 0; c < G; ++c) {
    const double2 lo = src[(size_t)c * (E / 2)], hi = src[(size_t)c * (E / 2) + 1];
    s0 += lo.x; s1 += lo.y; s2 += hi.x; s3 += hi.y;
  }
  return s0 + s1 + s2 + s3;
}

__global__ void __launch_bounds__(256) step_kernel(const double *in, double *out, int G, int r) {
  const int tid = threadIdx.x;
  const double s = work(in, G, tid);
  double *o = out + (size_t)blockIdx.x * E + tid * 4;
  o[0] = s * 1e-3 + 1.0; o[1] = s * 1e-3; o[2] = 1.0 + (r % 7); o[3] = 0.5 * (r % 3);   // round-dependent: a stale line shows
}

__device__ __forceinline__ bool grid_barrier(gu32 *counter, unsigned target, int tid, unsigned *tmo) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its sc1 stores
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(counter, 1u, RLX_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, RLX_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) { *tmo = 1; break; }
    }
  }
  __syncthreads();
  return true;
}

// persistent: ping-pong buffers; all exchanged data through sc1 (relaxed agent atomics, 8 bytes)
__global__ void __launch_bounds__(256) persistent_kernel(double *bufA, double *bufB, int G, int rounds, gu32 *counter, unsigned *tmo,
                                                          int payload) {
  const int tid = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    double *in = (r & 1) ? bufB : bufA, *out = (r & 1) ? bufA : bufB;
    if (payload) {
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      gu64 *src = (gu64 *)(in + tid * 4);
#pragma unroll 4
      for (int c = 0; c < G; ++c) {
        gu64 *p = src + (size_t)c * E;
        s0 += __longlong_as_double(__hip_atomic_load(p, RLX_AGENT));
        s1 += __longlong_as_double(__hip_atomic_load(p + 1, RLX_AGENT));
        s2 += __longlong_as_double(__hip_atomic_load(p + 2, RLX_AGENT));
        s3 += __longlong_as_double(__hip_atomic_load(p + 3, RLX_AGENT));
      }
      const double s = s0 + s1 + s2 + s3;
      gu64 *o = (gu64 *)(out + (size_t)blockIdx.x * E + tid * 4);
      __hip_atomic_store(o, (unsigned long long)__double_as_longlong(s * 1e-3 + 1.0), RLX_AGENT);
      __hip_atomic_store(o + 1, (unsigned long long)__double_as_longlong(s * 1e-3), RLX_AGENT);
      __hip_atomic_store(o + 2, (unsigned long long)__double_as_longlong(1.0 + (r % 7)), RLX_AGENT);
      __hip_atomic_store(o + 3, (unsigned long long)__double_as_longlong(0.5 * (r % 3)), RLX_AGENT);
    }
    grid_barrier(counter, (unsigned)G * (r + 1), tid, tmo);
  }
}

int main() {
  const int rounds = 2000;
  for (int G : {24, 48, 96, 192, 256}) {
    double *a, *b; unsigned *cnt;
    CHECK(hipMalloc(&a, (size_t)G * E * 8)); CHECK(hipMalloc(&b, (size_t)G * E * 8)); CHECK(hipMalloc(&cnt, 64));
    std::vector<double> h((size_t)G * E, 1.0);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_launch, ms_pers, ms_bare;
    // (1) launches
    CHECK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (int r = 0; r < 20; ++r) step_kernel<<<G, 256>>>((r & 1) ? b : a, (r & 1) ? a : b, G, r);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < rounds; ++r) step_kernel<<<G, 256>>>((r & 1) ? b : a, (r & 1) ? a : b, G, r);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms_launch, e0, e1));
    std::vector<double> ref((size_t)G * E);
    CHECK(hipMemcpy(ref.data(), a, ref.size() * 8, hipMemcpyDeviceToHost));
    // (2) persistent with payload
    for (int payload : {1, 0}) {
      CHECK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      CHECK(hipMemset(cnt, 0, 64));
      persistent_kernel<<<G, 256>>>(a, b, G, 4, (gu32 *)cnt, cnt + 8, payload);       // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      CHECK(hipMemset(cnt, 0, 64));
      CHECK(hipEventRecord(e0));
      persistent_kernel<<<G, 256>>>(a, b, G, rounds, (gu32 *)cnt, cnt + 8, payload);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      CHECK(hipEventElapsedTime(payload ? &ms_pers : &ms_bare, e0, e1));
      unsigned hc[16]; CHECK(hipMemcpy(hc, cnt, 64, hipMemcpyDeviceToHost));
      if (hc[8]) printf("  TIMEOUT in persistent kernel (G=%d)\n", G);
      if (payload) {
        std::vector<double> got((size_t)G * E);
        CHECK(hipMemcpy(got.data(), a, got.size() * 8, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs(got[i] - ref[i]));
        if (md > 1e-9) printf("  MISMATCH persistent vs launches: %g (G=%d)\n", md, G);
      }
    }
    printf("G=%3d workgroups: launch chain %.2f us/step, persistent sc1 + counter barrier %.2f us/step, bare barrier %.2f us\n", G,
           ms_launch * 1e3 / rounds, ms_pers * 1e3 / rounds, ms_bare * 1e3 / rounds);
    fflush(stdout);
    hipFree(a); hipFree(b); hipFree(cnt);
  }
  return 0;
}
