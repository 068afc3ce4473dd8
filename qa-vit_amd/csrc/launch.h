// Host-side error plumbing shared by every entry point.
#pragma once
#include <hip/hip_runtime.h>

namespace qv {
int set_error(int code, const char* msg);          // records msg (thread-local) and returns code
int check_launch(const char* what);                // hipGetLastError() -> QAVIT_ELAUNCH
}  // namespace qv
