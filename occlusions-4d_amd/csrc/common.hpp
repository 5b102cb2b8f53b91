// Shared helpers for libocc4d.so (gfx950 only; no CUDA dual path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "occ4d.h"

namespace occ4d {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return OCC4D_ELAUNCH;
  }
  return OCC4D_OK;
}

#define OCC4D_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      occ4d::set_error(__VA_ARGS__);    \
      return OCC4D_EINVAL;              \
    }                                   \
  } while (0)

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace occ4d
