// Shared host-side helpers for libtorchpq_amd.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/torchpq_amd.h"

namespace tpq {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TPQ_ERR_HIP;
  }
  return TPQ_OK;
}

#define TPQ_REQUIRE(cond, ...)         \
  do {                                 \
    if (!(cond)) {                     \
      ::tpq::set_error(__VA_ARGS__);   \
      return TPQ_ERR_INVALID_ARGUMENT; \
    }                                  \
  } while (0)

#define TPQ_LAUNCH_CHECK(name)                                   \
  do {                                                           \
    hipError_t _e = hipGetLastError();                           \
    if (_e != hipSuccess) return ::tpq::check_hip(_e, name);     \
  } while (0)

// A/B switches read from the environment exist only in experiment builds (tools/build_variant.sh passes
// -DTPQ_AB_SWITCHES); the product library reads no environment variable and carries none of their names.
#ifdef TPQ_AB_SWITCHES
#include <stdlib.h>
#define TPQ_AB_ENV(name) getenv(name)
#else
#define TPQ_AB_ENV(name) (static_cast<const char*>(nullptr))
#endif

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;  // CDNA wavefront

}  // namespace tpq
