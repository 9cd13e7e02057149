// Shared host-side helpers: status codes and the thread-local error message behind avmoe_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace avmoe {

enum Status : int {
  OK = 0,
  ERR_BAD_ARG = -1,        // null pointer / inconsistent descriptor
  ERR_UNSUPPORTED = -2,    // valid request the library does not implement
  ERR_ALIGNMENT = -3,      // pointer / stride alignment contract violated
  ERR_WORKSPACE = -4,      // workspace too small
  ERR_LAUNCH = -5,         // HIP launch failure
};

void set_last_error(const char* fmt, ...);
const char* last_error();

#define AVMOE_CHECK_LAUNCH(what)                                                         \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess) {                                                             \
      ::avmoe::set_last_error("%s: %s", what, hipGetErrorString(e__));                   \
      return ::avmoe::ERR_LAUNCH;                                                        \
    }                                                                                    \
  } while (0)

#define AVMOE_TRY(expr)                 \
  do {                                  \
    int s__ = (expr);                   \
    if (s__ != 0) return s__;           \
  } while (0)

// Development switches (A/B toggles, sweep overrides: scripts/README.md) exist only in builds made with -DAVMOE_DEV
// (AVMOE_DEV_BUILD=1 python -m avmoe_amd.build); the product library never reads them.  Four environment variables are part of
// the product and read with plain getenv: AVMOE_PROF_SHAPES (profiler families per launch shape, prof.cpp), AVMOE_NO_SIDE /
// AVMOE_SIDE_MIN (helper streams inside a call: off / smallest site in token elements that forks, side.cpp, moe_run.h) and
// AVMOE_NXN_CHUNK (test hook: frames per chunk of the AVVP N x N block, moe_plan.cpp).
#ifdef AVMOE_DEV
static inline const char* dev_env(const char* name) { return getenv(name); }
#else
static inline const char* dev_env(const char*) { return nullptr; }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel function: one flag per (kernel instantiation,
// device), so a second GPU in the same process gets its attribute too (a process-wide `static bool` set it on the first one only).
struct LdsAttrOnce {
  bool done[64] = {};
  int ensure(const void* fn, int bytes, const char* what);      // OK / ERR_LAUNCH ; no-op for <= 64 KiB and after the first call per device
};

// compute units of the current device (cached per device; <= 0 on a failed query)
int cu_count();

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace avmoe
