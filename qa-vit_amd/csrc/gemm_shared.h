// Internal: cross-file entry points of the GEMM kernels.
#pragma once
#include "../../include/qavit.h"
#include <hip/hip_runtime.h>

namespace qv {
// bf16 K-loop kernel for problems whose weight is too large for a resident LDS slice (gemm_big.hip).
// 1 = launched, 0 = not applicable, < 0 = error code.
int gemm_nt_big_try(const qavit_gemm_args& g, hipStream_t st);
}  // namespace qv
