// What does rocprofv3's FETCH_SIZE report for the access shapes of winograd43h_kernel?  MI355X_MICROARCH.md calibrates it for 16-byte-per-lane
// streaming reads only (reported = 1/2 of the bytes) and says: calibrate your own pattern before trusting an absolute.  Each kernel below
// reads a buffer of known size exactly once (256 MiB, far beyond L2; fresh memory per launch):
//   stream16   : 16 B per lane, 1 KB contiguous per wave-instruction            (the U requests; the guide's calibrated case)
//   patch4     : 4 B per lane, 16 lanes = one 64-byte chunk, the four chunks of a wave-instruction 512 B apart  (the patch requests:
//                (tile, channel) lanes, pixels Cin * 4 = 512 bytes apart at Cin = 128; the next instruction takes the next pixel)
//   patch4_pair: the same, but two kernels' worth of lanes touch each 64-byte chunk's 128-byte line in TWO instructions far apart
//                (channel steps s and s + 1 of the K loop share a 128-byte line)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib scripts/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t BYTES = (size_t)256 << 20;

__global__ void __launch_bounds__(256) stream16(const float4 *__restrict__ x, float *sink, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.f) sink[0] = acc;
}

// the buffer as [pixels][128 channels] fp32 (512 B per pixel); a wave takes 4 pixels x 16 channels per instruction; `step` selects the
// 16-channel (64-byte) chunk, a launch reads chunks step0, step0 + stride, ... of every pixel
__global__ void __launch_bounds__(256) patch4(const float *__restrict__ x, float *sink, size_t pixels, int step0, int nsteps, int stride) {
  float acc = 0.f;
  const int lane = threadIdx.x & 63, ch = lane & 15, t = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t p0 = wave * 4; p0 < pixels; p0 += nwaves * 4)
    for (int s = 0; s < nsteps; ++s) acc += x[(p0 + t) * 128 + (step0 + s * stride) * 16 + ch];
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  float *buf, *sink;
  CHECK(hipMalloc(&buf, 4 * BYTES));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, 4 * BYTES));
  CHECK(hipDeviceSynchronize());
  const size_t pixels = BYTES / 512;
  // 1: 16-byte streaming read of region 0
  hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const float4 *>(buf), sink, BYTES / 16);
  // 2: patch4, every 64-byte chunk of region 1 once (8 steps per pixel, consecutive): whole 128-byte lines by two adjacent instructions
  hipLaunchKernelGGL(patch4, dim3(2048), dim3(256), 0, 0, buf + BYTES / 4, sink, pixels, 0, 8, 1);
  // 3: patch4, only the EVEN chunks of region 2 (half of each 128-byte line is never asked for): 128 MiB requested
  hipLaunchKernelGGL(patch4, dim3(2048), dim3(256), 0, 0, buf + 2 * (BYTES / 4), sink, pixels, 0, 4, 2);
  // 4: patch4, one chunk per pixel of region 3 (one 64-byte chunk of every 512): 32 MiB requested
  hipLaunchKernelGGL(patch4, dim3(2048), dim3(256), 0, 0, buf + 3 * (BYTES / 4), sink, pixels, 3, 1, 1);
  CHECK(hipDeviceSynchronize());
  printf("launched: stream16 256 MiB; patch4 all chunks 256 MiB; patch4 even chunks 128 MiB requested; patch4 one chunk 32 MiB requested\n");
  return 0;
}
