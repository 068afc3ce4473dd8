// Fused attention BRANCH, forward, for the 16-learned-token problems of the CIFAR configuration (every HQA-ViT C100 block) and the
// 64-token problems of Tiny-ImageNet's 64 learned tokens / QA-ViT at 32 px (TT = 64, see "64 TOKENS" below):
//   out = dropout( proj( SDPA( q(x), [Linformer(k(x'), v(x')) ; bank rows], dropout_p ) ) )
// in ONE launch -- the chain HQAViT_CIFAR100.py:441-469 (SWA: one 4x4 window = the 16 tokens), :496-532 (MSDA: keys from
// the pooled dilated landmarks x'), :613-626 (cross: keys = projections of the bank) runs as  qkv GEMM -> Linformer ->
// bank concat -> softmax (+ attention dropout) -> P.V -> proj GEMM (+ bias, dropout).  bf16 operands, fp32 accumulation.
//
// DECOMPOSITION.  A workgroup of 8 waves (two per SIMD: the chain is full of single-wave latencies and of VALU work -- softmax,
// dropout hashes, conversions -- that one wave per SIMD issues at half rate) owns a tile of 4 images (64 token rows).
//   QKV + attention phase: WAVE = (HEAD, IMAGE PAIR).  A wave computes q, k, v of its head for its 2 images and runs their 2
//     attention cores -- every weight fragment it reads from LDS feeds 2 MFMAs, and the cores are independent chains the
//     scheduler interleaves.  (Wave = image, the first design, read every fragment once per MFMA: 1.1 MB of LDS reads per
//     tile, which is what bounded it.)
//   proj phase: WAVE = (IMAGE, COLUMN HALF).  The heads' outputs meet in one LDS tile per image; a wave multiplies its image's
//     16 x 192 attention output with 96 rows of the proj weight.
// Weights stream through an LDS ring in K-STEP CHUNKS: chunk (part, s) = the 12 MFMA tiles (192 rows: 4 heads x 48) of q, k or
// v -- or of proj -- at k-step s, 12 fragments of 1 KB (qavit_pack_desc.pad = 1: fragment order, so a fragment is one
// global_load_lds wave-instruction, 16 B per lane, no registers, and one conflict-free ds_read_b128).  5 ring slots, 4
// chunks in flight incl. the one consumed, counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads would drain them).
//
// REGISTER-RESIDENT CHAIN.  Between the QKV GEMM and the attention output nothing goes through LDS: each product is formed
// in the orientation whose ACCUMULATOR layout is the next product's OPERAND layout (16x16 MFMA: an accumulator register quad
// holds 4 consecutive rows of one column; a 16x16x16 operand quad holds 4 consecutive k of one row / column):
//   q_h   = (W_q x^T)^T    "transposed" GEMM   acc = q[query = lane%16][4 consecutive d]          -> B operand of S^T
//   k_h   = x W_k^T        plain GEMM          acc = k[4 consecutive tokens][d = lane%16]         -> A operand of Kf^T
//   Kf^T  = k^T E_k                            acc = Kf[key = lane%16][4 consecutive d]           -> A operand of S^T
//   S^T   = Kf Q^T (+ bank rows from LDS)      acc = S[query = lane%16][4 consecutive keys]       -> softmax on registers
//   P                                          the same registers, bf16                           -> B operand of O^T
//   v_h   = x W_v^T        plain GEMM          acc = v[4 consecutive tokens][d = lane%16]         -> B operand of Vf
//   Vf    = E_v^T v                            acc = Vf[4 consecutive keys][d = lane%16]          -> A operand of O^T
//   O^T   = Vf^T P^T (+ bank rows from LDS)    acc = O[query = lane%16][4 consecutive d]          -> 8-byte segments of the O tile
//   out   = (W_p O^T)^T    "transposed" GEMM on the O tile read back as 16x16x32 operands         -> 8-byte row segments
//
// 64 TOKENS (TT = 64; HQAViT_IN_Tiny.py:771-800, :826-862, :943-959, QAViT.py:588-636).  The 64-row tile is ONE image instead of four.
//   SWA:   the image's four 4x4 windows ARE four independent 16-token problems (window_partition with window 4 on the 8x8 grid):
//          the same kernel with the tile's rows gathered through the window map (sub_token) -- keys of a window = its own 16 tokens.
//   cross: keys are the bank projections only, so the four 16-query tiles of the image are independent problems as well.
//   MSDA:  the key side is ONE per (image, head): up to 48 landmarks pooled over all 64 tokens (40 for dilations (1, 2), stride 2),
//          k / v of the landmark tiles, K_f = E_k^T k and V_f = E_v^T v contracted over the landmark tiles.  Both waves of a head
//          compute it (3 landmark tiles instead of 2 images per weight fragment: +18 MFMAs per wave and part) and each runs the
//          attention cores of its two query tiles against it.
// The dropout masks keep the unfused kernels' contracts: problem id = (image * windows + window) * H + head for SWA, image * H + head
// with query index 0..63 otherwise.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"
#include "branch_shared.h"

namespace qv {

#ifdef QAVIT_BRANCH_STAMPS
__device__ unsigned long long qv_branch_stamps[4096 * 16];
#endif

namespace {

// KIND 0 = SWA, 1 = MSDA, 2 = cross.  SAVE: also write q / k / v (and MSDA's pooled landmarks) for the backward pass.  k and v leave
// the plain GEMM in operand layout (4 tokens of one column per lane -- 2-byte scattered stores); with SAVE the same weight and
// token fragments run a second, transposed MFMA whose accumulator quads are 8-byte row segments: +72 MFMAs per wave (no extra
// LDS reads) instead of a recompute GEMM launch per branch in backward.
template <int KIND, bool SAVE, int TT>
__global__ __launch_bounds__(512) void branch_fwd_kernel(qavit_branch_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr bool MODE0 = (KIND != 2);                      // Linformer + bank keys (SWA / MSDA) vs bank-projection keys only (cross)
  constexpr bool WIN = (KIND == 0);                        // TT = 64: the tile's sub-images are the 4x4 windows
  constexpr bool KSH = (TT == 64 && KIND == 1);            // ONE key side per image (MSDA on 64 tokens), shared by the wave's two query tiles
  constexpr int LT = KSH ? 3 : 1;                          // 16-row landmark tiles feeding the Linformer product
  constexpr int NKS = KSH ? LT : NIW;                      // key-source row tiles a wave runs through the k / v GEMMs
  constexpr int KI = KSH ? 1 : NIW;                        // K_f / V_f register sets per wave
  constexpr int KT0 = MODE0 ? 2 : 0, NKT = KT0 + 1, DT = 3, NKo = KT0 * 16;
  constexpr int NPART = MODE0 ? 3 : 1, NQKV = NPART * KST, NCH = NQKV + KST;      // chunks per tile: 24 (cross: 12)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  bf16* sbk = reinterpret_cast<bf16*>(smraw + SM_BANK);
  bf16* sbv = sbk + 16 * LDB;
  float* sbias = reinterpret_cast<float*>(smraw + SM_BIAS);
  bf16* so_all = reinterpret_cast<bf16*>(smraw + SM_OUT);
  bf16* sx_all = reinterpret_cast<bf16*>(smraw + SM_X);                  // the 4 images' token tiles [16][LDO] (MSDA: + landmark tiles)
  bf16* sp_all = reinterpret_cast<bf16*>(smraw + (KIND == 1 ? SM_P : SM_X));
  bf16* sp_all_q = reinterpret_cast<bf16*>(smraw + SM_P);
  const int S = a.S, NK = NKo + S;
  const float scale = rsqrtf((float)BD);
  const bf16* xg = reinterpret_cast<const bf16*>(a.x);
  bf16* og = reinterpret_cast<bf16*>(a.out);
  bf16* osv = reinterpret_cast<bf16*>(a.o_save);
  const char* wqkv = reinterpret_cast<const char*>(a.wqkv_frag);
  const char* wproj = reinterpret_cast<const char*>(a.wproj_frag);
  bool bad = false;
#ifdef QAVIT_BRANCH_STAMPS   // diagnostic build only (tools/branch_stamps.py): s_memtime at the phase boundaries, 16 words per workgroup into a buffer of their own
  unsigned long long* stamps = qv_branch_stamps + (size_t)(blockIdx.x & 4095) * 16;
#define STAMP(k) do { if (tid == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
  STAMP(0);

  // chunk c of the tile's schedule: q k-steps, k k-steps, v k-steps (cross: q only), then proj k-steps
  auto issue = [&](int c) {
    char* slot = smraw + (c % RING) * CHUNK_BYTES;
    if (c < NQKV) issue_chunk(wqkv, (c / KST) * CT, c % KST, slot, wave, lane);
    else issue_chunk(wproj, 0, c - NQKV, slot, wave, lane);
  };
  // ONE tile per workgroup (grid = ceil(B / 4)): its first chunks are on their way before anything else happens
  const int tile = blockIdx.x;
#pragma unroll
  for (int c = 0; c < AHEAD; ++c) issue(c);

  const bool adrop = a.attn_drop_p > 0.f && a.rng != nullptr;       // uniform
  AttnDrop drop;
  drop.on = adrop;
  drop.p = adrop ? a.attn_drop_p : 0.f;
  drop.inv_keep = adrop ? 1.f / (1.f - a.attn_drop_p) : 1.f;
  drop.key = adrop ? rng_key(a.rng, a.attn_drop_site) : 0u;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;

  // ---------------- global loads of the prologue, all issued before any is consumed ----------------
  // token tiles: wave w stages rows 8 (w & 1) .. + 8 of sub-image w >> 1 (16-byte pieces, 24 per row)
  bf16x8 xr[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + (size_t)tile_row<TT, WIN>(tile, wave >> 1, row, a.B) * a.ldx + 8 * c8);
  }
  // Linformer matrices as MFMA operand quads: lane holds E[l = 16 lt + 4 q4 + i][j = 16 jt + col], rows l >= L read as zero (the
  // reference's zero padding is algebraic; MSDA has 10 landmarks in the 16-row tile, 40 in three tiles on 64 tokens)
  float ek[LT][KT0 > 0 ? KT0 : 1][4], ev[LT][KT0 > 0 ? KT0 : 1][4];
  if (MODE0) {
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int l = 16 * lt + 4 * q4 + i, lc = l < a.L ? l : 0;
          ek[lt][jt][i] = a.E_k[(size_t)lc * a.KC + jt * 16 + col];
          ev[lt][jt][i] = a.E_v[(size_t)lc * a.KC + jt * 16 + col];
        }
  }
  // biases [3C qkv | C proj] and the shared key / value rows of every head (the bank, or its projections for cross)
  constexpr int NB4 = (MODE0 ? 3 * BC : BC) / 4;
  f32x4 bq = {0.f, 0.f, 0.f, 0.f}, bp = {0.f, 0.f, 0.f, 0.f};
  if (tid < NB4) bq = *reinterpret_cast<const f32x4*>(a.bqkv + 4 * tid);
  if (tid < BC / 4) bp = *reinterpret_cast<const f32x4*>(a.bproj + 4 * tid);
  f32x4 kk[2], vv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {                         // 16 rows x 48 chunks of 4 = 768 = 1.5 x 512
    const int e0 = tid + 512 * it, e = e0 < 768 ? e0 : 0, sr = e / (BC >> 2), ch = e - sr * (BC >> 2);
    kk[it] = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)sr * BC + 4 * ch);
    vv[it] = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)sr * BC + 4 * ch);
  }
  // ---------------- ... and their consumers ----------------
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
    *reinterpret_cast<bf16x8*>(sx_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = xr[it];
  }
  bf16x4 ekf[LT][KT0 > 0 ? KT0 : 1], evf[LT][KT0 > 0 ? KT0 : 1];
  if (MODE0) {
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = 16 * lt + 4 * q4 + i < a.L;
          ekf[lt][jt][i] = (bf16)(ok ? ek[lt][jt][i] : 0.f);
          evf[lt][jt][i] = (bf16)(ok ? ev[lt][jt][i] : 0.f);
        }
  }
  if (tid < NB4) *reinterpret_cast<f32x4*>(sbias + 4 * tid) = bq;
  if (tid < BC / 4) *reinterpret_cast<f32x4*>(sbias + 3 * BC + 4 * tid) = bp;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + 512 * it;
    if (e < 768) {
      const int sr = e / (BC >> 2), ch = e - sr * (BC >> 2);
      bf16x4 kb, vb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { bad |= (kk[it][j] != kk[it][j]) | (vv[it][j] != vv[it][j]); kb[j] = (bf16)kk[it][j]; vb[j] = (bf16)vv[it][j]; }
      *reinterpret_cast<bf16x4*>(sbk + sr * LDB + 4 * ch) = kb;
      *reinterpret_cast<bf16x4*>(sbv + sr * LDB + 4 * ch) = vb;
    }
  }
  STAMP(1);
  if (KIND == 1) {
    // MSDA landmarks: pooled[j] = mean_s x[idx[j * stride + s]] (HQAViT_CIFAR100.py:499-501), j < L; fp32 mean, one rounding; rows
    // j >= L are zero.  From the staged token tile(s) (two waves staged each: barrier).
    //   TT = 16: wave w: image w >> 1, k-steps 3 (w & 1) .. + 3.
    //   TT = 64: the 18 (landmark tile, k-step) units of the image's 48-row landmark tile dealt over the 8 waves; idx = token 0 .. 63 = tile row.
    __syncthreads();
    const float inv = 1.f / (float)a.pool_stride;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      int s2, lrow, src_tile, dst_tile;
      bool act = true;
      if (KSH) {
        const int u = wave + NW * s3;
        act = u < LT * KST;
        const int uc = act ? u : 0;
        dst_tile = uc / KST; s2 = uc - dst_tile * KST; lrow = 16 * dst_tile + col; src_tile = 0;
      } else {
        s2 = 3 * (wave & 1) + s3; lrow = col; src_tile = dst_tile = wave >> 1;
      }
      const bf16* sx = sx_all + src_tile * (16 * LDO);
      bf16* sp = sp_all + dst_tile * (16 * LDO);
      const int j = lrow < a.L ? lrow : 0;
      float sum[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sum[e] = 0.f;
      for (int t = 0; t < a.pool_stride; ++t) {
        const int src = a.pool_idx[j * a.pool_stride + t];
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(sx + src * LDO + 32 * s2 + 8 * q4);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += (float)v[e];
      }
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (bf16)(lrow < a.L ? sum[e] * inv : 0.f);
      if (act) *reinterpret_cast<bf16x8*>(sp + col * LDO + 32 * s2 + 8 * q4) = o8;
    }
  }
  STAMP(2);
  __syncthreads();              // tiles and constants staged; every ordinary load above (and the ring's first chunks) drained: vmcnt is 0 here
  STAMP(3);

  const int h = wave & 3;                                  // this wave's head and its two sub-images in the QKV / attention phase
  const int i0 = NIW * (wave >> 2);
  bool vimg[NIW];
  int64_t qrow[NIW];                                       // global row of this lane's query (= token col of the sub-image)
  int prob[NIW];                                           // attention-dropout problem (= image, or (image, window)) of each sub-image
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    vimg[i] = sub_valid<TT>(tile, i0 + i, a.B);
    qrow[i] = tile_row<TT, WIN>(tile, i0 + i, col, a.B);
    const int ir = tile * NI + i0 + i;
    prob[i] = (TT == 64 && !WIN) ? tile : (TT == 64 ? ir : (ir < a.B ? ir : a.B - 1));
  }
  const int qoff = (TT == 64 && !WIN) ? 16 * i0 : 0;       // query index of the sub-image's first row inside its dropout problem (+ 16 i)
  bf16* qsv = reinterpret_cast<bf16*>(a.q_save);
  bf16* kvsv = reinterpret_cast<bf16*>(a.kv_save);
  const int kv_rows = (KIND == 1) ? a.L : BT;              // key-token rows per image in kv_save (SWA on 64 tokens: the token rows themselves)
  const bf16* sxw = sx_all + i0 * (16 * LDO) + col * LDO + 8 * q4;       // this lane's fragment base in its sub-images' token tiles
  const bf16* spw = sp_all + (KSH ? 0 : i0) * (16 * LDO) + col * LDO + 8 * q4;       // ... and landmark tiles (MSDA; else the token tiles)

  // Ring step for chunk c of the tile's schedule: wait until it has landed (chunks c+1 .. c+3 may stay in flight), then refill
  // the slot chunk c - 1 used.  A k-step chunk of a QKV part = 3 fragments (this head's tiles) x 2 images = 6 MFMAs per wave.
  // SAVE: q / k / v rows for the backward pass leave through LDS staging tiles as whole rows in 16-byte pieces (a "burst" = exactly 3
  // store instructions per wave, unconditional: sub-images past the batch are replicas of the last image and rewrite its rows with the
  // same values), issued right behind the barrier of ring step B1 (q rows), B2 (k rows, SWA) and B3 (v rows, SWA).  Straight from the
  // accumulator quads a store instruction wrote 32 contiguous bytes into each of 16 rows, and the uncounted stores made every following
  // counted wait drain the whole ring.
  // With SAVE the attention output O (operand of backward's dW_proj; o_save is required then) leaves the same way from the O tiles behind
  // the first proj step's barrier (step NQKV) instead of from registers at the kernel's tail, where nothing is left to overlap it.
  constexpr int B1 = KST, B2 = (SAVE && KIND == 0) ? 2 * KST : -100, B3 = NQKV;
  constexpr int N1 = (KIND == 2) ? 0 : 3, N2 = 3, N3 = (KIND == 0) ? 6 : (KIND == 2 ? 6 : 3);      // stores per burst (cross: q and O rows both at step NQKV = KST)
  auto burst_extra = [](int c, int b, int n) { return (c > b && c <= b + AHEAD) ? n : 0; };
  // a.drain_waits (diagnostic, uniform): every ring step waits for ALL of this wave's outstanding memory operations first -- the counted
  // waits below then hold trivially.  tests/test_gpu_ops.py compares the two forms bit for bit: a miscounted burst would show there.
  const bool drain_waits = a.drain_waits != 0;
#define QV_RING_STEP(c)                                                                         \
  do {                                                                                          \
    if (drain_waits) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
    ring_wait((NCH - 1 - (c)) < (AHEAD - 1) ? (NCH - 1 - (c)) : (AHEAD - 1),                    \
              SAVE ? burst_extra((c), B1, N1) + burst_extra((c), B2, N2) + burst_extra((c), B3, N3) : 0);   \
    if ((c) + AHEAD < NCH) issue((c) + AHEAD);                                                  \
  } while (0)
  // the burst: wave w stores rows 8 (w & 1) .. + 8 of sub-image w >> 1 from its staging tile
  // The wait arithmetic counts these stores by POSITION: exactly three dwordx4 stores per burst, issued after the ring step's LDS-DMA
  // (issue(c + AHEAD)) and before the next ring step.  Nothing else enforces that, so the burst is fenced for the compiler on both sides:
  // a scheduling barrier (no instruction moves across it) and an empty asm with a memory clobber (no memory operation is hoisted, sunk,
  // split or merged across it) -- a store that slid above the DMA or below the next counted wait would let a wave read a ring slot that
  // has not landed: wrong numbers, no fault.
  auto store_rows = [&](const bf16* stage, bf16* gdst, int64_t ld) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
      const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(stage + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8);
      *reinterpret_cast<bf16x8*>(gdst + (size_t)tile_row<TT, WIN>(tile, wave >> 1, row, a.B) * ld + 8 * c8) = v8;
      if (KSH) asm volatile("" ::: "memory");              // one piece at a time: this variant has no registers for three pieces in flight
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  bf16* qstage = (KIND == 2) ? sp_all_q : so_all;           // cross: the O quads follow the q phase directly, its q rows stage in a region of their own

  bf16x4 qf[NIW][DT];
  {
    f32x4 acc[NIW][DT];
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
      for (int t = 0; t < DT; ++t) acc[i][t] = *reinterpret_cast<const f32x4*>(sbias + h * BD + t * 16 + 4 * q4);
    // Fragment reads run ONE k-step ahead of the MFMAs (here and in the k / v / proj loops): the ring step of chunk s + 1 (wait, barrier,
    // refill) and the LDS reads of its fragments are issued before the MFMAs of chunk s, so the reads' latency and the barrier's skew
    // hide behind matrix work instead of standing between two chunks' MFMAs (a chunk was ~600 cycles for 96 of MFMA per wave).
    bf16x8 wfn[DT], xfn[NIW];
    auto lds_q = [&](int s) {
      const char* slot = smraw + (s % RING) * CHUNK_BYTES;
#pragma unroll
      for (int t = 0; t < DT; ++t) wfn[t] = *reinterpret_cast<const bf16x8*>(slot + ((3 * h + t) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < NIW; ++i) xfn[i] = *reinterpret_cast<const bf16x8*>(sxw + i * (16 * LDO) + 32 * s);
    };
    QV_RING_STEP(0);
    lds_q(0);
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      bf16x8 wf[DT], xf[NIW];
#pragma unroll
      for (int t = 0; t < DT; ++t) wf[t] = wfn[t];
#pragma unroll
      for (int i = 0; i < NIW; ++i) xf[i] = xfn[i];
      if (s + 1 < KST) { QV_RING_STEP(s + 1); lds_q(s + 1); }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < NIW; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[i], acc[i][t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        qf[i][t] = cvt4(acc[i][t]);
        if (SAVE) *reinterpret_cast<bf16x4*>(qstage + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = qf[i][t];
      }
  }
  STAMP(4);
  // global row of this lane's key-source row (token / landmark col of key-source tile ks) in kv_save
  // MSDA: the landmark rows (< L per image; on 64 tokens three 16-row tiles per image) of a staging region out as whole rows.  NOT a counted
  // burst (waves whose rows are all past L skip instructions): the waits that follow drain the ring as they always did for this branch.
  auto store_landmark_rows = [&](const bf16* stage, bf16* gdst, int64_t ld) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24, st = wave >> 1;
      const int l = KSH ? 16 * st + row : row;
      const bool ok = l < a.L && (KSH ? st < LT : sub_valid<TT>(tile, st, a.B));
      const int64_t grow = KSH ? (int64_t)tile * a.L + l : (int64_t)(tile * NI + st) * a.L + l;
      if (ok) *reinterpret_cast<bf16x8*>(gdst + (size_t)grow * ld + 8 * c8) = *reinterpret_cast<const bf16x8*>(stage + st * (16 * LDO) + row * LDO + 8 * c8);
    }
  };
  bf16x4 kff[KI][KT0 > 0 ? KT0 : 1][DT];
  if (MODE0) {
    // ---- k: plain GEMM (acc quad = 4 consecutive tokens of column d = 16 t + col), then Kf^T = k^T E_k ----
    constexpr bool SAVET = SAVE && KIND == 0;              // SWA: k rows through a second, transposed MFMA (8-byte row segments into the staging tile)
    f32x4 acc[NKS][DT], accT[SAVET ? NKS : 1][DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const float b = sbias[BC + h * BD + t * 16 + col];
#pragma unroll
      for (int i = 0; i < NKS; ++i) acc[i][t] = f32x4{b, b, b, b};
      if (SAVET) {
#pragma unroll
        for (int i = 0; i < NKS; ++i) accT[i][t] = *reinterpret_cast<const f32x4*>(sbias + BC + h * BD + t * 16 + 4 * q4);
      }
    }
    bf16x8 wfn[DT], xfn[NKS];
    auto lds_k = [&](int s) {
      const char* slot = smraw + ((KST + s) % RING) * CHUNK_BYTES;
#pragma unroll
      for (int t = 0; t < DT; ++t) wfn[t] = *reinterpret_cast<const bf16x8*>(slot + ((3 * h + t) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < NKS; ++i) xfn[i] = *reinterpret_cast<const bf16x8*>(spw + i * (16 * LDO) + 32 * s);
    };
    QV_RING_STEP(KST);
    if (SAVE) store_rows(qstage, qsv, a.ldq_save);         // every head's q quads are in the staging tiles (this step's barrier)
    lds_k(0);
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      bf16x8 wf[DT], xf[NKS];
#pragma unroll
      for (int t = 0; t < DT; ++t) wf[t] = wfn[t];
#pragma unroll
      for (int i = 0; i < NKS; ++i) xf[i] = xfn[i];
      if (s + 1 < KST) { QV_RING_STEP(KST + s + 1); lds_k(s + 1); }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[t], acc[i][t], 0, 0, 0);
          if (SAVET) accT[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[i], accT[i][t], 0, 0, 0);
        }
    }
    if (SAVE && KIND == 0) {                               // k rows for backward: into the (free) O tiles, out as whole rows behind the next barrier
#pragma unroll
      for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t)
          *reinterpret_cast<bf16x4*>(so_all + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(accT[i][t]);
    } else if (SAVE) {
      // MSDA: the landmark rows' k from the plain GEMM's accumulators (4 consecutive rows of one column per lane: 2-byte LDS stores) into
      // the O tiles (free until the cores' output), out as whole rows behind the next barrier -- no transposed MFMA, no registers for it
#pragma unroll
      for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) so_all[((KSH ? i : i0 + i) * 16 + 4 * q4 + r) * LDO + h * BD + t * 16 + col] = (bf16)acc[i][t][r];
    }
    if (KSH) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int lt = 0; lt < LT; ++lt) c = mma16b(cvt4(acc[lt][t]), ekf[lt][jt], c);     // contraction over the landmark tiles
          kff[0][jt][t] = cvt4(c);
        }
    } else {
#pragma unroll
      for (int i = 0; i < NIW; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bf16x4 ktf = cvt4(acc[i < NKS ? i : 0][t]);
#pragma unroll
          for (int jt = 0; jt < KT0; ++jt) kff[i < KI ? i : 0][jt][t] = cvt4(mma16b(ktf, ekf[0][jt], f32x4{0.f, 0.f, 0.f, 0.f}));   // Kf[key = 16 jt + col][16 t + 4 q4 ..]
        }
    }
  }
  STAMP(5);
  // ---- S^T[key][query] = K_full Q^T, softmax over keys on registers, attention dropout: independent images ----
  bf16x4 pfr[NIW][NKT];
  {
    s16x4 bkf[DT];                                         // the shared key rows of this head, read in place from the bank tile
#pragma unroll
    for (int t = 0; t < DT; ++t) bkf[t] = rowfrag(sbk, LDB, 0, h * BD + t * 16);
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      f32x4 sc[NKT];
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < DT; ++t) acc2 = mma16(nt < KT0 ? as_s16(kff[KSH ? 0 : i][nt < KT0 ? nt : 0][t]) : bkf[t], as_s16(qf[i][t]), acc2);
        sc[nt] = acc2;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = nt * 16 + 4 * q4 + r < NK;
          sc[nt][r] = ok ? sc[nt][r] * scale : -INFINITY;
          mx = fmaxf(mx, sc[nt][r]);
        }
      mx = rows4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(sc[nt][r] - mx); sc[nt][r] = e; sum += e; }
      sum = rows4_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[nt][r] *= inv;
      if (adrop) {                                         // two keys per hash (drop_factor pairs 2m, 2m+1)
        const uint32_t pkey = attn_drop_pkey(drop, prob[i] * BH + h);
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[nt][r] *= attn_drop_factor(drop, pkey, qoff + 16 * i * (TT == 64 && !WIN ? 1 : 0) + col, nt * 16 + 4 * q4 + r);
      }
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) pfr[i][nt] = cvt4(sc[nt]);
    }
  }
  STAMP(6);
  bf16x4 vff[KI][KT0 > 0 ? KT0 : 1][DT];
  if (MODE0) {
    // ---- v: plain GEMM, then Vf = E_v^T v (acc quad = 4 consecutive keys of column d) ----
    constexpr bool SAVET = SAVE && KIND == 0;
    f32x4 acc[NKS][DT], accT[SAVET ? NKS : 1][DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const float b = sbias[2 * BC + h * BD + t * 16 + col];
#pragma unroll
      for (int i = 0; i < NKS; ++i) acc[i][t] = f32x4{b, b, b, b};
      if (SAVET) {
#pragma unroll
        for (int i = 0; i < NKS; ++i) accT[i][t] = *reinterpret_cast<const f32x4*>(sbias + 2 * BC + h * BD + t * 16 + 4 * q4);
      }
    }
    bf16x8 wfn[DT], xfn[NKS];
    auto lds_v = [&](int s) {
      const char* slot = smraw + ((2 * KST + s) % RING) * CHUNK_BYTES;
#pragma unroll
      for (int t = 0; t < DT; ++t) wfn[t] = *reinterpret_cast<const bf16x8*>(slot + ((3 * h + t) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < NKS; ++i) xfn[i] = *reinterpret_cast<const bf16x8*>(spw + i * (16 * LDO) + 32 * s);
    };
    QV_RING_STEP(2 * KST);
    if (SAVE && KIND == 0) store_rows(so_all, kvsv, a.ldkv_save);
    if (SAVE && KIND == 1) store_landmark_rows(so_all, kvsv, a.ldkv_save);
    lds_v(0);
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      bf16x8 wf[DT], xf[NKS];
#pragma unroll
      for (int t = 0; t < DT; ++t) wf[t] = wfn[t];
#pragma unroll
      for (int i = 0; i < NKS; ++i) xf[i] = xfn[i];
      if (s + 1 < KST) { QV_RING_STEP(2 * KST + s + 1); lds_v(s + 1); }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[t], acc[i][t], 0, 0, 0);
          if (SAVET) accT[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[i], accT[i][t], 0, 0, 0);
        }
    }
    if (SAVE && KIND == 0) {
      // v rows: the O tiles take the attention output before the next barrier and the token tiles are still being read by slower
      // waves, so these stage in a region of their own; out as whole rows behind the first proj step's barrier
#pragma unroll
      for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t)
          *reinterpret_cast<bf16x4*>(sp_all_q + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(accT[i][t]);
    } else if (SAVE) {                                     // MSDA: the landmark rows' v into the token tiles (dead since the q phase: k / v read the landmark tiles)
#pragma unroll
      for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) sx_all[((KSH ? i : i0 + i) * 16 + 4 * q4 + r) * LDO + h * BD + t * 16 + col] = (bf16)acc[i][t][r];
    }
    if (KSH) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int lt = 0; lt < LT; ++lt) c = mma16b(evf[lt][jt], cvt4(acc[lt][t]), c);
          vff[0][jt][t] = cvt4(c);
        }
    } else {
#pragma unroll
      for (int i = 0; i < NIW; ++i)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bf16x4 vtf = cvt4(acc[i < NKS ? i : 0][t]);
#pragma unroll
          for (int jt = 0; jt < KT0; ++jt) vff[i < KI ? i : 0][jt][t] = cvt4(mma16b(evf[0][jt], vtf, f32x4{0.f, 0.f, 0.f, 0.f}));   // Vf[key = 16 jt + 4 q4 ..][16 t + col]
        }
    }
  }
  STAMP(7);
  // ---- O^T[d][query] = V_full^T P^T -> columns 48 h .. of each image's O tile.  A NaN anywhere in q / k / v reaches O (the
  // softmax and both products propagate it), so this is where efficient_attention's NaN rule is checked. ----
  {
    s16x4 bvf[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) bvf[t] = trfrag(sbv, LDB, 0, h * BD + t * 16);
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) acc2 = mma16(nt < KT0 ? as_s16(vff[KSH ? 0 : i][nt < KT0 ? nt : 0][t]) : bvf[t], as_s16(pfr[i][nt]), acc2);
        bad |= nan4(acc2);
        *reinterpret_cast<bf16x4*>(so_all + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(acc2);
      }
  }
  STAMP(8);
  // ================= proj: out = dropout(O . Wproj^T + b): WAVE = (IMAGE, COLUMN HALF), 6 output tiles, 6 k-step chunks =================
  {
    const int pi = wave >> 1, half = wave & 1;
    const bool valid = sub_valid<TT>(tile, pi, a.B);
    const bf16* so = so_all + pi * (16 * LDO);
    bf16* sout = sx_all + pi * (16 * LDO);                 // the token tiles are dead once every wave is past its v phase (= past the
    bf16x8 of8[KST];                                       // barrier of the first proj chunk): they collect the output rows
    f32x4 acc[CT / 2];
#pragma unroll
    for (int jj = 0; jj < CT / 2; ++jj) acc[jj] = *reinterpret_cast<const f32x4*>(sbias + 3 * BC + (6 * half + jj) * 16 + 4 * q4);
    bf16x8 wfn[CT / 2];
    auto lds_p = [&](int s) {
      const char* slot = smraw + ((NQKV + s) % RING) * CHUNK_BYTES;
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) wfn[jj] = *reinterpret_cast<const bf16x8*>(slot + ((6 * half + jj) * 64 + lane) * 16);
    };
    QV_RING_STEP(NQKV);                                    // this barrier also publishes the 4 heads' O quads
    if (SAVE && KIND == 0) store_rows(sp_all_q, kvsv + BC, a.ldkv_save);    // ... and the staged v rows
    if (SAVE && KIND == 2) store_rows(qstage, qsv, a.ldq_save);             // cross: the q rows (its q phase is directly followed by the cores)
    if (SAVE && KIND == 1) store_landmark_rows(sx_all, kvsv + BC, a.ldkv_save);
    if (SAVE) store_rows(so_all, osv, a.ldo);                               // the attention output rows
#pragma unroll
    for (int s2 = 0; s2 < KST; ++s2) of8[s2] = *reinterpret_cast<const bf16x8*>(so + col * LDO + 32 * s2 + 8 * q4);
    lds_p(0);
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      bf16x8 wf[CT / 2];
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) wf[jj] = wfn[jj];
      if (s + 1 < KST) { QV_RING_STEP(NQKV + s + 1); lds_p(s + 1); }
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jj], of8[s], acc[jj], 0, 0, 0);
    }
    STAMP(9);
#pragma unroll
    for (int jj = 0; jj < CT / 2; ++jj) {
      if (pdrop) {
        const uint32_t base = (uint32_t)tile_row<TT, WIN>(tile, pi, col, a.B) * (uint32_t)BC + (uint32_t)((6 * half + jj) * 16 + 4 * q4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jj][r] *= drop_factor(pkey_proj, base + r, pp, pinv);
      }
      *reinterpret_cast<bf16x4*>(sout + col * LDO + (6 * half + jj) * 16 + 4 * q4) = cvt4(acc[jj]);
    }
    // ---------------- output rows: LDS tile -> global, 16-byte pieces (this wave's 96 columns: 12 per row) ----------------
    wave_sync();
    if (!SAVE && osv && valid) {                           // attention output rows without the other saves: whole rows from the O tile (all heads' quads are in since the proj phase's first barrier)
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int p = lane + 64 * it, row = p / 12, c8 = 12 * half + p % 12;
        *reinterpret_cast<bf16x8*>(osv + (size_t)tile_row<TT, WIN>(tile, pi, row, a.B) * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(so + row * LDO + 8 * c8);
      }
    }
    if (valid) {
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int p = lane + 64 * it, row = p / 12, c8 = 12 * half + p % 12;
        *reinterpret_cast<bf16x8*>(og + (size_t)tile_row<TT, WIN>(tile, pi, row, a.B) * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(sout + row * LDO + 8 * c8);
        if (SAVE && KIND == 1 && a.pooled_save) {                 // the landmark rows (operand of backward's dW_kv), from their LDS tile
          const int lrow = KSH ? 16 * pi + row : row;             // TT = 64: landmark tile pi (< LT) of the image
          const int64_t prow = KSH ? (int64_t)tile * a.L + lrow : (int64_t)(tile * NI + pi) * a.L + lrow;
          if (lrow < a.L && (!KSH || pi < LT))
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(a.pooled_save) + (size_t)prow * BC + 8 * c8) =
                *reinterpret_cast<const bf16x8*>(sp_all + pi * (16 * LDO) + row * LDO + 8 * c8);
        }
      }
    }
  }
  STAMP(10);
#undef QV_RING_STEP
  STAMP(11);
  if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
#undef STAMP
}

// efficient_attention's NaN rule (HQAViT_CIFAR100.py:356-357, :394-395) behind the fused branch: a NaN anywhere in q / k / v
// or in the attention output zeroes the WHOLE attention output, so the branch returns dropout(proj(0)) = dropout(bias) rows.
// The last workgroup to have read the flag resets it (flag[1] = arrival ticket), as qavit_nan_guard does.
// For the backward pass: `trip` (optional) receives 1 / 0 = rule applied / not, and a tripped call zeroes the saved attention output
// (o_save [rows, ldos], Co columns) -- the reference's zeros_like(q) has no history, so dW_proj = dz^T 0 = 0 and nothing flows back
// through the attention core (qavit_branch_bwd / qavit_cga_bwd read `trip`).
__global__ __launch_bounds__(256) void branch_nan_fix_kernel(bf16* out, int64_t ldo, int rows, int C, const float* bias, float p, int site,
                                                             const int64_t* rng, int* flag, int* trip, bf16* o_save, int64_t ldos, int Co) {
  __shared__ int f_s;
  if (threadIdx.x == 0) {
    f_s = *reinterpret_cast<volatile int*>(flag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the read has returned before the arrival is counted (no agent-scope fence needed)
    if (atomicAdd(flag + 1, 1) == (int)gridDim.x - 1) { flag[0] = 0; flag[1] = 0; }
  }
  __syncthreads();
  if (trip && blockIdx.x == 0 && threadIdx.x == 0) *trip = f_s != 0 ? 1 : 0;
  if (f_s == 0) return;
  const bool on = p > 0.f && rng != nullptr;
  const uint32_t key = on ? rng_key(rng, site) : 0u;
  const float inv = on ? 1.f / (1.f - p) : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * C; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
    float v = bias[c];
    if (on) v *= drop_factor(key, (uint32_t)r * (uint32_t)C + (uint32_t)c, p, inv);
    out[(size_t)r * ldo + c] = (bf16)v;
  }
  if (o_save)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * Co; i += (int64_t)gridDim.x * blockDim.x) {
      const int r = (int)(i / Co), c = (int)(i - (int64_t)r * Co);
      o_save[(size_t)r * ldos + c] = (bf16)0.f;
    }
}

int branch_validate(const qavit_branch_args* a) {
  if (!a) return set_error(QAVIT_EINVAL, "branch: null args");
  if (a->kind < 0 || a->kind > 2) return set_error(QAVIT_EINVAL, "branch: kind must be 0 (SWA), 1 (MSDA) or 2 (cross)");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "branch: the fused branch kernels are bf16 only (fp32 runs the unfused chain)");
  if ((a->T != 16 && a->T != 64) || a->C != BC || a->H != BH || a->D != BD || a->S != 16)
    return set_error(QAVIT_EINVAL, "branch: built for 16 or 64 tokens x 192 channels, 4 heads of 48, 16 bank rows");
  if (a->B <= 0 || !a->x || !a->out || !a->wqkv_frag || !a->wproj_frag || !a->bqkv || !a->bproj || !a->sh_k || !a->sh_v)
    return set_error(QAVIT_EINVAL, "branch: null operand");
  const int lmax = (a->kind == 1 && a->T == 64) ? 48 : 16;
  if (a->kind != 2 && (a->KC != 32 || a->L <= 0 || a->L > lmax || !a->E_k || !a->E_v))
    return set_error(QAVIT_EINVAL, "branch: SWA / MSDA need Linformer matrices with KC = 32 and 1 <= L <= 16 (MSDA on 64 tokens: <= 48)");
  if (a->kind == 0 && a->L != 16) return set_error(QAVIT_EINVAL, "branch: SWA works on 4x4 windows (L = 16)");
  if (a->kind == 1 && (!a->pool_idx || a->pool_stride <= 0)) return set_error(QAVIT_EINVAL, "branch: MSDA needs the landmark index table");
  if (a->q_save) {
    if (!a->o_save) return set_error(QAVIT_EINVAL, "branch: q_save comes with o_save (the backward pass needs both)");
    if ((reinterpret_cast<uintptr_t>(a->q_save) & 15) || a->ldq_save % 8) return set_error(QAVIT_EINVAL, "branch: q_save needs 16-byte alignment and ld % 8 == 0");
    if (a->kind != 2 && (!a->kv_save || (reinterpret_cast<uintptr_t>(a->kv_save) & 15) || a->ldkv_save % 8))
      return set_error(QAVIT_EINVAL, "branch: kv_save (with q_save) needs 16-byte alignment and ld % 8 == 0");
    if (a->pooled_save && (reinterpret_cast<uintptr_t>(a->pooled_save) & 15)) return set_error(QAVIT_EINVAL, "branch: pooled_save needs 16-byte alignment");
  }
  if (a->nan_trip && !a->nan_flag) return set_error(QAVIT_EINVAL, "branch: nan_trip is written by the NaN-rule launch, which needs nan_flag");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al16(a->x) || !al16(a->out) || !al16(a->wqkv_frag) || !al16(a->wproj_frag) || !al16(a->bqkv) || !al16(a->bproj) || !al16(a->sh_k) || !al16(a->sh_v) ||
      (a->o_save && !al16(a->o_save)) || a->ldx % 8 || a->ldo % 8)
    return set_error(QAVIT_EINVAL, "branch: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements");
  return QAVIT_OK;
}

}  // namespace

// the NaN -> zeros rule's rewrite launch, shared with the fused channel-group branch (cga.hip)
void branch_nan_fix_launch(void* out, int64_t ldo, int rows, int C, const float* bias, float p, int site, const int64_t* rng, int* flag,
                           int* trip, void* o_save, int64_t ldos, int Co, hipStream_t st) {
#ifdef QAVIT_NANFIX_EXPERIMENT     // diagnostic build only: what the 32 NaN-rule launches of a step cost (QAVIT_SKIP_NANFIX=1 drops them: trip stays unset)
  { static int skip = -1; if (skip < 0) { const char* e = getenv("QAVIT_SKIP_NANFIX"); skip = e ? atoi(e) : 0; } if (skip) return; }
#endif
  const int64_t n = (int64_t)rows * C;
  int nb = (int)((n + 2047) / 2048);
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(branch_nan_fix_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<bf16*>(out), ldo, rows, C, bias, p, site, rng, flag,
                     trip, reinterpret_cast<bf16*>(o_save), ldos, Co);
}

}  // namespace qv

using namespace qv;

extern "C" int qavit_branch_supported(int kind, int T, int C, int H, int D, int KC, int S, int L) {
  if (kind < 0 || kind > 2) return 0;
  if ((T != 16 && T != 64) || C != BC || H != BH || D != BD || S != 16) return 0;
  const int lmax = (kind == 1 && T == 64) ? 48 : 16;
  if (kind != 2 && (KC != 32 || L <= 0 || L > lmax)) return 0;
  if (kind == 0 && L != 16) return 0;
  return 1;
}

#ifdef QAVIT_BRANCH_STAMPS
extern "C" int qavit_branch_stamps(void* host_dst, int nwg) {   // diagnostic build: the last launch's stamps, 16 words per workgroup
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(qv_branch_stamps), (size_t)nwg * 16 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int qavit_branch_fwd(const qavit_branch_args* a, void* stream) {
  int rc = branch_validate(a);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = a->T == 64;
  const int grid = wide ? a->B : (a->B + NI - 1) / NI;     // one 64-row tile per workgroup: 4 images of 16 tokens, or one image of 64
  static bool attr_done[3][2][2] = {};
  const bool save = a->q_save != nullptr;
#define QV_BRANCH_LAUNCH(K, S, TT)                                                                                                      \
  do {                                                                                                                                   \
    if (!attr_done[K][S][TT == 64]) {                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(branch_fwd_kernel<K, S, TT>), hipFuncAttributeMaxDynamicSharedMemorySize, sm_total(K, S)); \
      attr_done[K][S][TT == 64] = true;                                                                                                  \
    }                                                                                                                                    \
    hipLaunchKernelGGL((branch_fwd_kernel<K, S, TT>), dim3(grid), dim3(512), sm_total(K, S), st, *a);                                   \
  } while (0)
#define QV_BRANCH_KIND(K)                                                                                                                \
  do {                                                                                                                                   \
    if (wide) { if (save) QV_BRANCH_LAUNCH(K, true, 64); else QV_BRANCH_LAUNCH(K, false, 64); }                                          \
    else { if (save) QV_BRANCH_LAUNCH(K, true, 16); else QV_BRANCH_LAUNCH(K, false, 16); }                                               \
  } while (0)
  if (a->kind == 0) QV_BRANCH_KIND(0);
  else if (a->kind == 1) QV_BRANCH_KIND(1);
  else QV_BRANCH_KIND(2);
#undef QV_BRANCH_KIND
#undef QV_BRANCH_LAUNCH
  if (a->nan_flag && !a->nan_defer)
    branch_nan_fix_launch(a->out, a->ldo, a->B * a->T, BC, a->bproj, a->proj_drop_p, a->proj_drop_site, a->rng, a->nan_flag, a->nan_trip,
                          a->o_save, a->ldo, BC, st);
  return check_launch("branch_fwd");
}
