// ssal_prof.h -- optional per-kernel timing with HIP events recorded on the launch stream.
// Used by bench.py's roofline leg (never inside the timed throughput region).  Not thread-safe:
// enable it from one host thread only.
#pragma once
#include <hip/hip_runtime.h>

namespace ssal {
bool prof_enabled();
// flops / bytes: ALGORITHMIC work of this launch (tensor reads + writes once, 2*MAC)
void prof_begin(const char *kernel, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char *kernel, double flops, double bytes, hipStream_t st) : s(st), on(prof_enabled())
    {
        if (on) prof_begin(kernel, flops, bytes, s);
    }
    ~ProfScope()
    {
        if (on) prof_end(s);
    }
};
}  // namespace ssal
