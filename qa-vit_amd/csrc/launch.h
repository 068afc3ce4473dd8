// Host-side error plumbing shared by every entry point.
#pragma once
#include <hip/hip_runtime.h>

namespace qv {
int set_error(int code, const char* msg);          // records msg (thread-local) and returns code
int check_launch(const char* what);                // hipGetLastError() -> QAVIT_ELAUNCH
// zero an fp32 scratch buffer with a KERNEL node (a hipMemsetAsync captured into a hipGraph was observed to race with
// its neighbours on replay: BatchNorm statistics accumulated onto stale scratch about once in 30 steps)
void zero_f32(float* p, size_t n, hipStream_t st);
}  // namespace qv
