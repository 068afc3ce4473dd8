// One-wavefront 16x16 MFMA tiles whose operands are read element-wise (any strides) from LDS or global.
// Used by the per-(image, head) kernels -- attention core, Linformer, bank statistics, TokenLearner,
// TokenUpMix -- whose matrices are 10..256 rows: the whole problem lives in one wave's LDS slice.
//   BF = false : v_mfma_f32_16x16x4_f32   (exact fp32; the parity path)
//   BF = true  : v_mfma_f32_16x16x32_bf16 (operands rounded to bf16 on load, fp32 accumulate)
#pragma once
#include "common.cuh"

namespace qv {

// acc += A(16 x K) . B(K x 16);  A(i,k) = a[i*ars + k*aks] for i < arows;  B(k,j) = b[k*bks + j*bcs] for j < bcols
template <bool BF, typename TA, typename TB>
__device__ __forceinline__ f32x4 mma_tile(const TA* a, int ars, int aks, int arows,
                                          const TB* b, int bks, int bcs, int bcols, int K, f32x4 acc) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const bool aok = r < arows, bok = r < bcols;
  if constexpr (!BF) {
    for (int k0 = 0; k0 < K; k0 += 4) {
      const int k = k0 + q;
      const float av = (aok && k < K) ? to_f<TA>(a[r * ars + k * aks]) : 0.f;
      const float bv = (bok && k < K) ? to_f<TB>(b[k * bks + r * bcs]) : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      bf16x8 av, bv;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = k0 + 8 * q + jj;
        av[jj] = (bf16)((aok && k < K) ? to_f<TA>(a[r * ars + k * aks]) : 0.f);
        bv[jj] = (bf16)((bok && k < K) ? to_f<TB>(b[k * bks + r * bcs]) : 0.f);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
    }
  }
  return acc;
}

// C/D layout of the 16x16 MFMAs: acc[reg] is element (row = 4*(lane>>4) + reg, col = lane & 15)
__device__ __forceinline__ int tile_row(int reg) { return 4 * ((threadIdx.x & 63) >> 4) + reg; }
__device__ __forceinline__ int tile_col() { return threadIdx.x & 15; }

template <bool ADD>
__device__ __forceinline__ void tile_to_f32(float* c, int crs, int ccs, int rows, int cols, const f32x4& acc, float scale = 1.f) {
  const int col = tile_col();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = tile_row(reg);
    if (row < rows && col < cols) {
      if (ADD) c[row * crs + col * ccs] += acc[reg] * scale;
      else c[row * crs + col * ccs] = acc[reg] * scale;
    }
  }
}

template <typename T>
__device__ __forceinline__ void tile_to_global(T* c, int64_t crs, int rows, int cols, const f32x4& acc) {
  const int col = tile_col();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = tile_row(reg);
    if (row < rows && col < cols) c[row * crs + col] = from_f<T>(acc[reg]);
  }
}

}  // namespace qv
