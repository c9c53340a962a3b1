// Shared host/device helpers for libidiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/idiff_hip.h"

#define IDIFF_API extern "C" __attribute__((visibility("default")))

namespace idiff {

void set_error(const char *fmt, ...);

inline int fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return IDIFF_EINVAL;
}

// Every launcher ends with this: reports a launch-time error without synchronising.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// ---- process-level debug switches (A/B experiments, parity tests).  Read from the environment ONCE, when the library
// is loaded (no getenv on any launch path); idiff_set_option() flips them afterwards.
enum Option { OPT_NO_WINOGRAD, OPT_NO_COLSTATS, OPT_NO_PIPE, OPT_SCALAR_EPILOGUE, OPT_DBUF_ONLY, OPT_TRIDIAG_ONESTAGE,
              OPT_UFD_ROWS, OPT_CHASE_WAVEFRONT, OPT_WINO_NGROUP, OPT_GRAM_SMALL_TILES, OPT_CHASE_SPIN_LIMIT, OPT_FAKE_CU_COUNT,
              OPT_SBR_SYNC, OPT_SBR_FULL, OPT_NO_SPLIT, OPT_WINO_SPLIT, OPT_SBR_LOOKAHEAD, OPT_NO_WINO43, OPT_NO_WINO43H, OPT_NO_PAIRS, OPT_PAIRS_MIN_TILES, OPT_NO_FUSED_ATTN, OPT_NO_WINO1D, OPT_COUNT };
bool option(Option o);
int option_value(Option o);   // the integer behind a switch (IDIFF_WINO_NGROUP: output-channel tiles per scheduling group)

// hipFuncSetAttribute is per DEVICE: one bit per device ordinal, so a process that drives several GPUs sets the
// attribute on each of them (a process-wide `static bool` would leave the second device at the 64 KB default).
struct AttrGuard { unsigned long long done_mask = 0; };
int set_dynamic_lds_once(AttrGuard &g, const void *const *fns, int n_fns, int bytes, const char *what);

// conv_narrow.hip: 3x3 / stride 1 / pad 1 convolution to <= 4 output channels on the vector ALUs (the image heads)
bool conv3x3_narrow_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_lo, int pad_hi,
                       const idiff_epilogue *ep);
int conv3x3_narrow(const float *x, const float *wt, float *out, int B, int H, int W, int Cin, int Cout,
                   const idiff_epilogue *ep, hipStream_t stream);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound grids: enough workgroups to fill 256 CUs x 8 without paying for a huge grid.
static inline int streaming_grid(int64_t work_items, int block) {
  int64_t g = ceil_div64(work_items, block);
  const int64_t cap = 256 * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    // v_exp_f32 / v_rcp_f32 (1-2 ulp each) instead of libm's expf and an IEEE division: 4 instructions, not ~25, which
    // is what kept the GroupNorm-apply pass below the HBM rate; |error| <= 3e-7 |silu(v)|, far inside the 2e-5 parity bar
    case IDIFF_ACT_SILU: return v * __frcp_rn(1.0f + __expf(-v));
    case IDIFF_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.0f);  // exp(x)-1 as ATen's elu does
    case IDIFF_ACT_RELU: return v > 0.f ? v : 0.f;
    case IDIFF_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    default: return v;
  }
}

}  // namespace idiff
