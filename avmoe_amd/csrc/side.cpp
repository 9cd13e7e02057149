#include "side.h"
#include "common.h"
#include "prof.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace avmoe {

namespace {
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, Side*> g_sides;   // (device, caller stream) -> helper; lives as long as the process
bool side_disabled() {
  static const bool off = [] { const char* e = getenv("AVMOE_NO_SIDE"); return e && *e && *e != '0'; }();
  return off;
}
}  // namespace

int side_mask() {
  static const int m = [] { const char* e = dev_env("AVMOE_SIDE_MASK"); return e && *e ? atoi(e) : 7; }();
  return m;
}

Side* side_acquire(hipStream_t st) {
  if (side_disabled() || prof_enabled()) return nullptr;      // (per-launch timing brackets launches on the caller's stream: a forked branch would be timed with its queueing)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(dev, st);
  auto it = g_sides.find(key);
  if (it != g_sides.end()) return it->second;
  Side* sd = new Side{nullptr, nullptr, nullptr};
  if (hipStreamCreateWithFlags(&sd->s, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&sd->fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&sd->join_ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    delete sd;
    sd = nullptr;
  }
  g_sides[key] = sd;
  return sd;
}

int side_fork(Side* sd, hipStream_t st) {
  hipError_t e = hipEventRecord(sd->fork_ev, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(sd->s, sd->fork_ev, 0);
  if (e != hipSuccess) { set_last_error("side stream fork: %s", hipGetErrorString(e)); return ERR_LAUNCH; }
  return OK;
}

int side_join(Side* sd, hipStream_t st) {
  hipError_t e = hipEventRecord(sd->join_ev, sd->s);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, sd->join_ev, 0);
  if (e != hipSuccess) { set_last_error("side stream join: %s", hipGetErrorString(e)); return ERR_LAUNCH; }
  return OK;
}

}  // namespace avmoe
