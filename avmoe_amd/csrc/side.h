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

}  // namespace avmoe
