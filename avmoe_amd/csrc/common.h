// Shared host-side helpers: status codes and the thread-local error message behind avmoe_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

// Cache policy of the streaming kernels' direct global -> LDS loads: the `aux` argument of __builtin_amdgcn_global_load_lds (gfx950: 1 = sc0,
// 2 = nt, 16 = sc1).  2, the non-temporal hint: a tile that is read once is not kept in the L2 / Infinity Cache in place of lines somebody
// will come back to.  Measured on the tile loop of kk_hop1_yk rebuilt without its kernel (scripts/lds_stream_probe.hip, MI355X): with the
// loop's one small store per wave and tile (32-byte runs: the partly written lines now survive until their other parts arrive) 4.58 -> 5.08
// TB/s, without the store no difference (5.9 TB/s); sc0 alone: none.  Applied per kernel where the two-stream STEP gains (hop1_stream.hip: the
// Y streams, -1.1 %); the X-side kernels (tok_pair2 / dpost_pair / dx_stream3) measured neutral to +1 % with it and keep 0.
#ifndef AVMOE_LDS_AUX
#define AVMOE_LDS_AUX 2
#endif

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
// the product and read with plain getenv, ONCE per process: AVMOE_PROF_SHAPES (profiler families per launch shape, prof.cpp),
// AVMOE_NO_SIDE / AVMOE_SIDE_MIN (helper streams inside a call: off / smallest site in token elements that forks, side.cpp,
// moe_run.h).  The test hooks (size thresholds of the streaming kernels lifted, frames per chunk of the AVVP N x N block) are
// process state set through avmoe_test_hooks (include/avmoe.h); the environment variables of the same names seed them when the
// library is first asked -- no kernel choice depends on the environment at call time.
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

// test hooks (include/avmoe.h: avmoe_test_hooks); seeded once from AVMOE_TOKPAIR2_FORCE / AVMOE_DPAIR_FORCE / AVMOE_HOP1S_FORCE / AVMOE_NXN_CHUNK
enum { HOOK_TOKPAIR2_FORCE = 1, HOOK_DPAIR_FORCE = 2, HOOK_HOP1S_FORCE = 4, HOOK_KFS_FORCE = 8, HOOK_KFS_OFF = 16 };
unsigned test_hook_mask();
int test_hook_nxn_chunk();                   // 0 = the library's own choice
void set_test_hooks(unsigned mask, int nxn_chunk);

// compute units of the current device (cached per device; <= 0 on a failed query)
int cu_count();

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace avmoe
