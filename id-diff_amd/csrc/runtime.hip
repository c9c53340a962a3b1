// ABI version + thread-local error string.
#include "common.h"
#include <string.h>

namespace idiff {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace idiff

IDIFF_API int idiff_abi_version(void) { return IDIFF_ABI_VERSION; }
IDIFF_API const char *idiff_last_error(void) { return idiff::g_err; }
