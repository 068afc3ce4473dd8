// Shared by the fused branch kernels (branch_fwd.hip, branch_bwd.hip): tile geometry, the weight-chunk ring (global_load_lds into
// fragment order, counted vmcnt + raw barrier) and small register helpers.  See branch_fwd.hip for the design.
#pragma once
#include "common.cuh"
#include "frag16.cuh"

namespace qv {

namespace {

constexpr int BT = 16, BC = 192, BD = 48, BH = 4;          // tokens per image, channels, head dim, heads
constexpr int NI = 4;                                      // images per workgroup tile
constexpr int NW = 8;                                      // waves per workgroup
constexpr int NIW = 2;                                     // images per wave in the QKV / attention phase
constexpr int KST = BC / 32;                               // k-steps of 32
constexpr int CT = BC / 16;                                // 16-row MFMA tiles of a 192-row weight block: fragments per chunk
constexpr int CHUNK_BYTES = CT * 1024;                     // 12288
constexpr int RING = 5, AHEAD = 4;                         // ring slots; chunks in flight incl. the one being consumed (the slot chunk c + AHEAD
                                                           // lands in was consumed in iteration c - 1, behind this iteration's barrier)
constexpr int GLDS_PER_CHUNK = 2;                          // LDS-DMA instructions per wave per chunk (12 fragments + 4 repeats over 8 waves)
constexpr int LDB = BC + 8, LDO = BC + 8;                  // shared bank tiles [16][LDB]; per-image O / output tiles [16][LDO]
constexpr int SM_BANK = RING * CHUNK_BYTES, SM_BIAS = SM_BANK + 2 * 16 * LDB * 2, SM_OUT = SM_BIAS + 4 * BC * 4,
              OUT_BYTES = 16 * LDO * 2, SM_X = SM_OUT + NI * OUT_BYTES, SM_P = SM_X + NI * OUT_BYTES;
// 61440 ring + 12800 bank + 3072 biases + 25600 O / output tiles + 25600 token tiles (+ 25600 landmark tiles (MSDA) / q staging tiles (cross)) = 128512 (154112) bytes
constexpr int sm_total(int kind, bool save) { return (kind == 1 || save) ? SM_P + NI * OUT_BYTES : SM_P; }   // the region: MSDA's landmark tiles; SWA's v / cross's q staging tiles when they save for backward

// Token rows of a 64-row tile.  TT = 16 (the CIFAR configuration's 16 learned tokens): the tile is 4 images of 16 tokens, sub-image
// `sub` = image tile * 4 + sub.  TT = 64 (Tiny-ImageNet's 64 learned tokens, QA-ViT at 32 px without TokenLearner): the tile is ONE
// image; its four 16-row sub-images are the four 4x4 windows of the 8x8 token grid for SWA (window_partition with window 4,
// HQAViT_IN_Tiny.py:756-769: window (wy, wx), position (ty, tx) -> token (4 wy + ty) * 8 + 4 wx + tx) and the four query tiles
// 16 sub .. 16 sub + 15 for MSDA / cross-attention.
template <int TT, bool WINDOWS>
__device__ __forceinline__ int sub_token(int sub, int r) {
  if (TT == 64 && WINDOWS) return ((sub >> 1) * 4 + (r >> 2)) * 8 + (sub & 1) * 4 + (r & 3);
  return (TT == 64 ? 16 * sub : 0) + r;
}
// global row (in a [B * TT, *] matrix) of row r of sub-image `sub` of tile `tile`; rows of images past the batch are clamped to the last image
template <int TT, bool WINDOWS>
__device__ __forceinline__ int64_t tile_row(int tile, int sub, int r, int B) {
  if (TT == 64) return (int64_t)tile * 64 + sub_token<TT, WINDOWS>(sub, r);
  const int img = tile * NI + sub;
  return (int64_t)(img < B ? img : B - 1) * BT + r;
}
template <int TT>
__device__ __forceinline__ bool sub_valid(int tile, int sub, int B) { return TT == 64 ? true : tile * NI + sub < B; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ bf16x4 cvt4(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}
__device__ __forceinline__ bool nan4(const f32x4& v) { return (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]); }
__device__ __forceinline__ f32x4 mma16b(bf16x4 a, bf16x4 b, f32x4 c) { return mma16(as_s16(a), as_s16(b), c); }

// chunk = the 12 fragments {(tile t0 + j, k-step s)} of a fragment-packed [*, 192] weight (fragment (t, s) sits at (t * KST + s)
// KB).  Wave w fetches j = w and j = w + 8 (waves 4..7: fragment 11 again, same bytes), so every wave issues exactly
// GLDS_PER_CHUNK instructions and a counted `s_waitcnt vmcnt(2 k)` means "all but the k newest chunks have landed" for all.
__device__ __forceinline__ void issue_chunk(const char* wpacked, int t0, int s, char* slot, int wave, int lane) {
#pragma unroll
  for (int f = 0; f < GLDS_PER_CHUNK; ++f) {
    int j = wave + 8 * f;
    j = j < CT ? j : CT - 1;
    __builtin_amdgcn_global_load_lds((glb_void_t*)(wpacked + (size_t)((t0 + j) * KST + s) * 1024 + lane * 16), (lds_void_t*)(slot + j * 1024), 16, 0, 0);
  }
}
// `newer` = chunks issued after the one about to be consumed (0 .. AHEAD-1); `extra` = counted global STORES this wave has issued after
// that chunk's LDS-DMA (vmcnt counts loads, stores and LDS-DMA together, in issue order: "all but the N youngest are done").  This wave's
// share of the chunk has landed once at most 2 newer + extra operations are outstanding; the barrier makes it true for all waves.
// Stores the caller does NOT count only make the wait stricter than needed (the pipeline drains), never wrong; a counted store that
// was not issued would be wrong -- counted bursts are unconditional.
__device__ __forceinline__ void ring_wait(int newer, int extra = 0) {     // called with unrolled-loop constants: the chain folds
  const int n = 2 * (newer > AHEAD - 1 ? AHEAD - 1 : newer) + extra;
  if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

}  // namespace

}  // namespace qv
