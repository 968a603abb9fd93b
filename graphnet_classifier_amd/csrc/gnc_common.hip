// Diagnostics entry points of the C ABI (include/gnc_hip.h).
#include <stdarg.h>
#include <string.h>

#include "gnc_common.h"

namespace gnc {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int num_cu() {
  static int cached[64] = {0};  // 0 = not asked yet; benign race: every thread stores the same value
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kNumCUMax;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = kNumCUMax;
    cached[dev] = n > kNumCUMax ? kNumCUMax : n;
  }
  return cached[dev];
}
}  // namespace gnc

extern "C" {
int gnc_abi_version(void) { return GNC_ABI_VERSION; }
const char* gnc_last_error_string(void) { return gnc::g_err; }
const char* gnc_target_arch(void) { return "gfx950"; }

}
