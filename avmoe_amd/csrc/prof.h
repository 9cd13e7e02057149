// Optional per-launch timing (HIP events on the launch stream), off by default.  bench.py switches it on for
// a separate profiling pass to find the dominant kernel and its average duration / algorithmic bytes; the
// numbers must agree with `rocprofv3 --kernel-trace --stats` of the same command (profiles/).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace avmoe {

bool prof_enabled();
void prof_enable(bool on);
void prof_reset();
// call right before / after a launch; `name` must be a string literal or otherwise outlive the report
void prof_begin(const char* name, double alg_bytes, double flops, hipStream_t st);
void prof_end(hipStream_t st);
// JSON array of {"name","calls","total_ms","alg_bytes","flops"}; returns bytes written (excluding NUL)
size_t prof_report(char* buf, size_t cap);

// AVMOE_PROF_SHAPES=1: one family per distinct launch shape (the GEMM engine appends M/N/K, the bottleneck-space scopes
// their token count) instead of one per kernel name
bool prof_shapes();
void prof_begin_tagged(const char* name, long tag, double alg_bytes, double flops, hipStream_t st);

struct ProfScope {
  hipStream_t st; bool on;
  ProfScope(const char* name, double bytes, double flops, hipStream_t s) : st(s), on(prof_enabled()) {
    if (on) prof_begin(name, bytes, flops, s);
  }
  ProfScope(const char* name, long tokens, double bytes, double flops, hipStream_t s) : st(s), on(prof_enabled()) {
    if (on) prof_begin_tagged(name, tokens, bytes, flops, s);
  }
  ~ProfScope() { if (on) prof_end(st); }
};

}  // namespace avmoe
