// A helper stream per caller stream for the independent branches inside one forward / backward call (fork: the helper waits for
// everything enqueued on the caller's stream so far; join: the caller's stream waits for the helper).  After the join all work is
// ordered into the caller's stream again, so the calls keep their stream-ordered contract (and stay capturable: fork / join through
// events is the capture-legal pattern).  AVMOE_NO_SIDE=1 switches the helper streams off (everything on the caller's stream); so does the per-launch timing of prof.h while it is on.
#pragma once
#include <hip/hip_runtime.h>

namespace avmoe {

struct Side { hipStream_t s; hipEvent_t fork_ev, join_ev; };

Side* side_acquire(hipStream_t st);        // nullptr: disabled or not available -- the caller then stays on its own stream
int side_mask();                           // dev: AVMOE_SIDE_MASK selects the forks (1 forward, 2 backward section 1, 4 backward section 2; default all)
int side_fork(Side* sd, hipStream_t st);   // status codes of common.h
int side_join(Side* sd, hipStream_t st);

// A fork that is joined on EVERY way out of its scope: an error return between fork and join must not leave the helper stream
// writing into the caller's workspaces after the call has returned (the caller may free or reuse them on its own stream).
struct SideScope {
  Side* sd; hipStream_t st; bool open = false;
  SideScope(Side* sd_, hipStream_t st_) : sd(sd_), st(st_) {}
  SideScope(const SideScope&) = delete;
  SideScope& operator=(const SideScope&) = delete;
  int fork() { const int r = side_fork(sd, st); open = (r == 0); return r; }
  int join() { open = false; return side_join(sd, st); }
  ~SideScope() { if (open) (void)side_join(sd, st); }
};

}  // namespace avmoe
