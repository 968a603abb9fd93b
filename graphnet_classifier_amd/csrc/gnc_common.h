// Internal helpers shared by the HIP translation units of libgnc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gnc_hip.h"

namespace gnc {

void set_error(const char* fmt, ...);  // thread-local message behind gnc_last_error_string()

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return GNC_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return GNC_ERR_HIP;
}

// Launch check: kernels are enqueued asynchronously, so this only catches launch-time
// failures (bad configuration, missing code object), never synchronises.
inline int check_launch(const char* kernel) { return check_hip(hipGetLastError(), kernel); }

constexpr int kWave = 64;        // gfx950 wavefront
constexpr int kNumCUMax = 256;   // MI355X in SPX mode; a partitioned device (CPX/DPX) shows fewer
// Compute units of the CURRENT device, queried once per device (persistent grids are sized from it).
int num_cu();
constexpr int kBlock = 256;      // default workgroup: 4 waves, one per SIMD

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace gnc

#define GNC_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      gnc::set_error(__VA_ARGS__);             \
      return GNC_ERR_INVALID_ARGUMENT;         \
    }                                          \
  } while (0)
