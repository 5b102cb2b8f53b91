#include <stdarg.h>

#include "common.hpp"

namespace occ4d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace occ4d

extern "C" int occ4d_abi_version(void) { return OCC4D_ABI_VERSION; }
extern "C" int occ4d_is_cpu_twin(void) { return 0; }
extern "C" const char* occ4d_last_error(void) { return occ4d::g_err; }
