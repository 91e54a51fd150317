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
extern "C" int e2e_abi_version(void) { return 19; }

extern "C" int e2e_diag_kernel_clock(int family, double* mhz, double* busy_ms, int reset) {
  E2E_REQUIRE(mhz != nullptr && (family == 0 || family == 1), "diag_kernel_clock: family 0 (conv133_mm) or 1 (conv133_wgrad v5)");
  if (hipDeviceSynchronize() != hipSuccess) return e2e::check_launch("diag_kernel_clock");
  unsigned long long v[2] = {0, 0};
  if (family == 0) e2e::mm_clock_read(v, reset != 0); else e2e::wgrad_clock_read(v, reset != 0);
  *mhz = v[1] ? 100.0 * (double)v[0] / (double)v[1] : 0.0;     // s_memrealtime: 100 MHz
  if (busy_ms) *busy_ms = (double)v[1] / 1e5;
  return e2e::check_launch("diag_kernel_clock");
}
