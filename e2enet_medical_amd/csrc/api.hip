// Error reporting and ABI version of libe2e_hip.so.
#include "e2e_common.h"
#include <cstring>

namespace e2e {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static thread_local char g_kernel[160] = "";
void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
}  // namespace e2e

extern "C" const char* e2e_last_error(void) { return e2e::g_err; }
extern "C" const char* e2e_last_kernel(void) { return e2e::g_kernel; }
extern "C" int e2e_abi_version(void) { return 14; }
