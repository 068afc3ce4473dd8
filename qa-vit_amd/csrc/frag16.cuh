// bf16 LDS-tile fragments for v_mfma_f32_16x16x16_bf16: the building blocks of the per-image small-matrix kernels
// (attention, bank statistics, TokenLearner / TokenUpMix).  A row-major bf16 tile in LDS serves BOTH orientations:
//   rowfrag : operand whose reduction axis runs along the tile's columns  (one ds_read_b64 per lane)
//   trfrag  : operand whose reduction axis runs along the tile's rows     (one ds_read_b64_tr_b16: hardware transpose)
// Row strides must be multiples of 4 elements (8-byte aligned reads); EXEC must be all ones at a trfrag.
#pragma once
#include "common.cuh"

namespace qv {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ s16x4 as_s16(bf16x4 v) { return __builtin_bit_cast(s16x4, v); }

// operand fragment, reduction axis along the tile's columns: element (idx = lane&15, k = k0 + 4*(lane>>4) + j)
__device__ __forceinline__ s16x4 rowfrag(const bf16* tile, int ld, int r0, int k0) {
  const int lane = threadIdx.x & 63;
  return as_s16(*reinterpret_cast<const bf16x4*>(tile + (r0 + (lane & 15)) * ld + k0 + 4 * (lane >> 4)));
}
// operand fragment, reduction axis along the tile's rows: element (k = k0 + 4*(lane>>4) + j, idx = c0 + (lane&15))
__device__ __forceinline__ s16x4 trfrag(const bf16* tile, int ld, int k0, int c0) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  return as_s16(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(tile + (k0 + 4 * g + q) * ld + c0 + 4 * p)));
}
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// accumulator tile (row = 4*(lane>>4)+r, col = lane&15) -> bf16 LDS tile
__device__ __forceinline__ void acc_to_lds(bf16* tile, int ld, int r0, int c0, const f32x4& acc, float scale = 1.f) {
  const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[(r0 + 4 * q + r) * ld + c0 + col] = (bf16)(acc[r] * scale);
}

template <int W> __device__ __forceinline__ float grp_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int W> __device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// W = 16 = one DPP row: four VALU instructions (quad permutes + half-row / row mirrors) instead of four dependent ds_bpermute round
// trips through the LDS pipe.  In-kernel stamps of the up-mix backward: the LayerNorm-backward section of a 16 x 192 tile -- three
// such sums per row -- took 2/3 of an image's time on the bpermute form.
template <> __device__ __forceinline__ float grp_sum<16>(float v) { return row16_sum(v); }
template <> __device__ __forceinline__ float grp_max<16>(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return v;
}


}  // namespace qv
