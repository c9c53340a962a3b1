// ABI version + thread-local error string.
#include "common.h"
#include <stdlib.h>
#include <string.h>

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
                                             "IDIFF_FAKE_CU_COUNT", "IDIFF_SBR_SYNC", "IDIFF_SBR_FULL", "IDIFF_NO_SPLIT", "IDIFF_WINO_SPLIT", "IDIFF_SBR_LOOKAHEAD"};
struct OptionTable {
  int v[OPT_COUNT];
  OptionTable() {
    for (int i = 0; i < OPT_COUNT; ++i) { const char *e = getenv(kOptionNames[i]); v[i] = (e && *e) ? atoi(e) : 0; }
  }
};
OptionTable g_options;     // constructed when the shared object is loaded
}  // namespace

bool option(Option o) { return __atomic_load_n(&g_options.v[o], __ATOMIC_RELAXED) != 0; }
int option_value(Option o) { return __atomic_load_n(&g_options.v[o], __ATOMIC_RELAXED); }

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
IDIFF_API const char *idiff_last_error(void) { return idiff::g_err; }

// Debug switch by its environment-variable name ("IDIFF_NO_WINOGRAD", ...): returns the previous value, -1 if unknown.
IDIFF_API int idiff_set_option(const char *name, int value) {
  using namespace idiff;
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptionNames[i]) == 0) return __atomic_exchange_n(&g_options.v[i], value, __ATOMIC_RELAXED);
  return -1;
}
