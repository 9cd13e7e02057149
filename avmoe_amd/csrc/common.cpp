#include "common.h"
#include <algorithm>
#include <atomic>

namespace avmoe {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_err; }

// test hooks: two process-wide words, seeded from the environment the first time they are asked for
namespace {
struct Hooks {
  std::atomic<unsigned> mask;
  std::atomic<int> nxn_chunk;
  Hooks() {
    unsigned m = 0;
    if (getenv("AVMOE_TOKPAIR2_FORCE")) m |= HOOK_TOKPAIR2_FORCE;
    if (getenv("AVMOE_DPAIR_FORCE")) m |= HOOK_DPAIR_FORCE;
    if (getenv("AVMOE_HOP1S_FORCE")) m |= HOOK_HOP1S_FORCE;
    if (getenv("AVMOE_KFS_FORCE")) m |= HOOK_KFS_FORCE;
    if (getenv("AVMOE_KFS_OFF")) m |= HOOK_KFS_OFF;
    const char* c = getenv("AVMOE_NXN_CHUNK");
    mask.store(m);
    nxn_chunk.store(c ? std::max(0, atoi(c)) : 0);
  }
};
Hooks& hooks() { static Hooks h; return h; }
}  // namespace
unsigned test_hook_mask() { return hooks().mask.load(std::memory_order_relaxed); }
int test_hook_nxn_chunk() { return hooks().nxn_chunk.load(std::memory_order_relaxed); }
void set_test_hooks(unsigned mask, int nxn_chunk) { hooks().mask.store(mask); hooks().nxn_chunk.store(std::max(0, nxn_chunk)); }

int cu_count() {
  static int cus[64] = {};      // (written once per device with the same value: a race between two host threads is benign)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cus[dev] > 0) return cus[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev < 64) cus[dev] = prop.multiProcessorCount;
  return prop.multiProcessorCount;
}

int LdsAttrOnce::ensure(const void* fn, int bytes, const char* what) {
  if (bytes <= 65536) return OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;      // (slot 63: set on every call)
  if (done[dev] && dev != 63) return OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_last_error("%s: hipFuncSetAttribute(%d B of dynamic LDS): %s", what, bytes, hipGetErrorString(e));
    return ERR_LAUNCH;
  }
  done[dev] = true;
  return OK;
}

}  // namespace avmoe
