// Error reporting shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vp_hip.h"

namespace vp {
extern thread_local char g_err[512];
void set_err(const char* fmt, ...);
}  // namespace vp

#define VP_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      vp::set_err("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));       \
      return VP_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)
