#include "prof.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace avmoe {

namespace {
struct Rec { std::string name; double bytes, flops; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
Rec g_cur;
bool g_open = false;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool prof_enabled() { return g_on; }
void prof_enable(bool on) { std::lock_guard<std::mutex> l(g_mu); g_on = on; }
void prof_reset() {
  std::lock_guard<std::mutex> l(g_mu);
  for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_recs.clear();
  g_open = false;
}
void prof_begin(const char* name, double bytes, double flops, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_open) return;                       // nested scopes: the outer one wins
  g_cur = Rec{name, bytes, flops, get_event(), get_event()};
  (void)hipEventRecord(g_cur.e0, st);
  g_open = true;
}
bool prof_shapes() {
  static const bool on = getenv("AVMOE_PROF_SHAPES") != nullptr;
  return on;
}
void prof_begin_tagged(const char* name, long tag, double bytes, double flops, hipStream_t st) {
  if (!prof_shapes()) { prof_begin(name, bytes, flops, st); return; }
  char buf[96];
  snprintf(buf, sizeof(buf), "%s NT%ld", name, tag);
  prof_begin(buf, bytes, flops, st);
}
void prof_end(hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_open) return;
  (void)hipEventRecord(g_cur.e1, st);
  g_recs.push_back(g_cur);
  g_open = false;
}
size_t prof_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> l(g_mu);
  struct Agg { long calls = 0; double ms = 0, bytes = 0, flops = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    Agg& a = agg[r.name];
    a.calls++; a.ms += ms; a.bytes += r.bytes; a.flops += r.flops;
  }
  std::string s = "[";
  bool first = true;
  for (auto& kv : agg) {
    char line[512];
    snprintf(line, sizeof(line), "%s{\"name\":\"%s\",\"calls\":%ld,\"total_ms\":%.6f,\"alg_bytes\":%.0f,\"flops\":%.0f}",
             first ? "" : ",", kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.bytes, kv.second.flops);
    s += line;
    first = false;
  }
  s += "]";
  if (buf && cap) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
  return s.size();
}

}  // namespace avmoe
