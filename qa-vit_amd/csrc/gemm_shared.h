// Internal: cross-file entry points of the GEMM kernels.
#pragma once
#include "../../include/qavit.h"
#include <hip/hip_runtime.h>

namespace qv {
// bf16 K-loop kernel for problems whose weight is too large for a resident LDS slice (gemm_big.hip).
// 1 = launched, 0 = not applicable, < 0 = error code.
int gemm_nt_big_try(const qavit_gemm_args& g, hipStream_t st);
// the same kernel with the LayerNorm-backward epilogue (qavit_gemm_args.e_x): 1 = launched, < 0 = error code
int gemm_nt_big_lnbwd(const qavit_gemm_args& g, hipStream_t st);
bool gemm_nt_lnbwd_shape_ok(int dtype, int M, int N, int K, int a_mode);
int gemm_nt_lnbwd_parts(int M, int N, int K);
// bf16 weight-gradient GEMMs with wide output tiles, grouped by tile class (gemm_tn_wide.hip).  Problems must be
// validated bf16 problems; returns QAVIT_OK or an error code.
// `ws`: NULL or gemm_tn_wide_ws_bytes() of device memory private to this call (a class with more problems than a launch carries by
// value then takes ONE launch through a device-side table).
int gemm_tn_wide(const qavit_gemm_tn_args* a, int n, hipStream_t st, void* ws);   // launches the problems with gemm_tn_wide_ok()
size_t gemm_tn_wide_ws_bytes();
bool gemm_tn_wide_ok(const qavit_gemm_tn_args& g);
}  // namespace qv
