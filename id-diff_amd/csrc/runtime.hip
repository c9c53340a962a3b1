// ABI version + thread-local error string.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

namespace idiff {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
const char *const kOptionNames[OPT_COUNT] = {"IDIFF_NO_WINOGRAD", "IDIFF_NO_COLSTATS", "IDIFF_NO_PIPE", "IDIFF_SCALAR_EPILOGUE",
                                             "IDIFF_DBUF_ONLY", "IDIFF_TRIDIAG_ONESTAGE", "IDIFF_UFD_ROWS", "IDIFF_CHASE_WAVEFRONT",
                                             "IDIFF_WINO_NGROUP", "IDIFF_GRAM_SMALL_TILES", "IDIFF_CHASE_SPIN_LIMIT",
                                             "IDIFF_FAKE_CU_COUNT", "IDIFF_SBR_SYNC", "IDIFF_SBR_FULL", "IDIFF_NO_SPLIT", "IDIFF_WINO_SPLIT", "IDIFF_SBR_LOOKAHEAD",
                                             "IDIFF_NO_WINO43", "IDIFF_NO_WINO43H", "IDIFF_NO_PAIRS", "IDIFF_PAIRS_MIN_TILES", "IDIFF_NO_FUSED_ATTN", "IDIFF_NO_WINO1D"};
struct OptionTable {
  int v[OPT_COUNT];
  OptionTable() {
    for (int i = 0; i < OPT_COUNT; ++i) { const char *e = getenv(kOptionNames[i]); v[i] = (e && *e) ? atoi(e) : 0; }
  }
};
OptionTable g_options;     // constructed when the shared object is loaded
// Per-host-thread overrides (idiff_set_thread_option): every launcher reads its switches on the CALLING thread, so an override
// set here acts on this thread's launches only -- the fail-soft re-solve selects a slower eigensolver form for its own
// launch without changing what any other host thread's launches do.  kUnset = fall through to the process-wide table.
constexpr int kUnset = INT_MIN;
struct ThreadOptions {
  int v[OPT_COUNT];
  ThreadOptions() { for (int i = 0; i < OPT_COUNT; ++i) v[i] = kUnset; }
};
thread_local ThreadOptions t_options;
}  // namespace

int option_value(Option o) {
  const int t = t_options.v[o];
  return t != kUnset ? t : __atomic_load_n(&g_options.v[o], __ATOMIC_RELAXED);
}
bool option(Option o) { return option_value(o) != 0; }

int set_dynamic_lds_once(AttrGuard &g, const void *const *fns, int n_fns, int bytes, const char *what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("%s: hipGetDevice: %s", what, hipGetErrorString(e)); return (int)e; }
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&g.done_mask, __ATOMIC_ACQUIRE) & bit) return 0;
  for (int i = 0; i < n_fns; ++i) {
    e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return (int)e; }
  }
  __atomic_fetch_or(&g.done_mask, bit, __ATOMIC_RELEASE);   // setting it twice in a race is harmless
  return 0;
}
}  // namespace idiff

#ifndef IDIFF_SOURCE_STAMP
#define IDIFF_SOURCE_STAMP "unstamped"
#endif
IDIFF_API int idiff_abi_version(void) { return IDIFF_ABI_VERSION; }
// sha256 (first 16 hex digits) of csrc/*.hip, csrc/*.h and include/idiff_hip.h at build time (csrc/build.sh)
IDIFF_API const char *idiff_source_stamp(void) { return IDIFF_SOURCE_STAMP; }
#ifndef IDIFF_VARIANT_FLAGS
#define IDIFF_VARIANT_FLAGS ""
#endif
IDIFF_API const char *idiff_variant_flags(void) { return IDIFF_VARIANT_FLAGS; }
IDIFF_API const char *idiff_last_error(void) { return idiff::g_err; }

// Debug switch by its environment-variable name ("IDIFF_NO_WINOGRAD", ...): returns the previous value, -1 if unknown.
IDIFF_API int idiff_set_option(const char *name, int value) {
  using namespace idiff;
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptionNames[i]) == 0) return __atomic_exchange_n(&g_options.v[i], value, __ATOMIC_RELAXED);
  return -1;
}

// The same switch for launches made from the CALLING host thread only; `set` = 0 removes the override again.  Returns 0, or -1
// for an unknown name.
IDIFF_API int idiff_set_thread_option(const char *name, int value, int set) {
  using namespace idiff;
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptionNames[i]) == 0) { t_options.v[i] = set ? value : kUnset; return 0; }
  return -1;
}
