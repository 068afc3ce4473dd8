// Fused attention BRANCH, backward, first half: the proj input-gradient GEMM and the whole attention-core backward of a
// 16-learned-token branch (SWA / MSDA / cross, see branch_fwd.hip) in ONE launch:
//   dO   = (dout * proj-dropout mask) . Wproj                         (also writes the masked dout: operand of dW_proj)
//   P    = softmax(Q K_full^T / sqrt(D)) recomputed from the saved q / k / v, dropout mask regenerated
//   dV_f = (P m)^T dO,  dP = dO V_full^T,  dS = P * (dP m - rowsum(dO * O)) / sqrt(D),  dQ = dS K_full,  dK_f = dS^T Q
//   dk   = E_k dK_f[:KC],  dv = E_v dV_f[:KC],  dE_k += k^T-sums,  dE_v likewise,  d(shared rows) += dK_f[KC:], dV_f[KC:]
// It replaces three launches (proj dX GEMM, attn3 backward, its partial-sum reduce); dq / dk / dv go to global memory once (the
// weight-gradient GEMMs need them anyway) and the qkv input-gradient GEMM reads them from there.
//
// Same decomposition as the forward kernel: 8 waves own 4 images; the proj GEMM runs as wave = (image, column half) on the
// weight-chunk ring, the attention phase as wave = (head, image pair).  Every product is formed from MFMA operand registers
// without an LDS round trip for intermediates.  A 16x16 accumulator X (acc[r] = X[4 q4 + r][col]) is a valid B operand for a
// contraction over X's ROWS and a valid A operand (as X^T) for the same contraction, but never for a contraction over its
// columns -- and backward contracts S-shaped matrices over queries (dV_f, dK_f) AND over keys (dQ).  So S, dP and therefore P,
// dS are computed TWICE, once per orientation, from the same operand registers with A and B swapped (9 small MFMAs each),
// the softmax statistics of the second orientation arriving by lane permutes from the first; K_f / V_f are recomputed in both
// operand layouts the same way.  q, dO, k, v sit in LDS tiles (k, v reuse the weight ring once the proj GEMM is done) and are
// read as row fragments or, for contractions over tokens, through the transposing LDS read.
// Sums over images (dE_k, dE_v, shared-row gradients) stay in accumulator registers for the wave's two images, meet across
// waves in LDS and leave as one row of per-workgroup partial sums (plain stores; qavit_ln_param_reduce folds the rows).
//
// 64 TOKENS (TT = 64, see branch_fwd.hip): the tile is one image.  SWA: its four windows are four independent problems, rows through
// the window map.  Cross: four independent query tiles.  MSDA: ONE key side per (image, head) from up to 48 landmark rows -- K_f / V_f
// are formed once per wave from the three landmark tiles; dK_f / dV_f are sums over all 64 queries, i.e. over the wave's two query
// tiles (registers) AND over the head's two waves: each wave parks the half it does not finish (wave < 4 finishes the key side,
// wave >= 4 the value side) in LDS, the owner adds, rounds the sum ONCE to a bf16 [32 keys][48] tile and reads it back in the two
// operand layouts dk = E_k dK_f and dE_k = k^T dK_f need.  A forward that tripped the NaN rule (nan_trip) has zero gradient through the
// attention core (the reference returns zeros_like(q), HQAViT_CIFAR100.py:356-357, :394-395): dq = dk = dv = 0 and all partial sums 0.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"
#include "branch_shared.h"

namespace qv {

namespace {

constexpr int BW_SM_BANK = RING * CHUNK_BYTES;                       // ring (later: k / v token tiles) | bank k, v | g -> dO tiles | q tiles
constexpr int BW_SM_G = BW_SM_BANK + 2 * 16 * LDB * 2;
constexpr int BW_TILE = 16 * LDO * 2;                                // one image's [16][LDO] bf16 tile
constexpr int BW_SM_Q = BW_SM_G + NI * BW_TILE;
constexpr int BW_SM_TOTAL = BW_SM_Q + NI * BW_TILE;                  // 61440 + 12800 + 25600 + 25600 = 125440 bytes
constexpr int part_e(int TT) { return (TT == 64 ? 48 : 16) * 32; }  // dE_k / dE_v: [L <= 16 (48 on 64 tokens)][KC = 32]
constexpr int PART_SH = 16 * BC;                                     // shared-row gradients [16][192]
constexpr int part_floats(int TT) { return 2 * part_e(TT) + 2 * PART_SH; }   // 7168 (9216) floats per workgroup
constexpr int LDF = BD + 8;                                          // dK_f / dV_f bf16 tiles [32 keys][LDF] (TT = 64 MSDA)
constexpr int LDE = 32 + 8;                                          // ... and its Linformer matrices as bf16 tiles [48 landmarks][LDE] (rows >= L zero)
constexpr int BW_SM_E = BW_SM_TOTAL;                                 // E_k tile | E_v tile, behind everything else
constexpr int BW_SM_TOTAL_KSH = BW_SM_TOTAL + 2 * 48 * LDE * 2;      // + 7680 bytes

__device__ __forceinline__ s16x4 cvt4s(const f32x4& acc) { return as_s16(cvt4(acc)); }

// a forward that tripped the NaN rule: dz = masked dout (db_proj still flows: proj(0) = bias), everything behind the attention core zero
template <int KIND, int TT>
__device__ __forceinline__ void branch_bwd_tripped(const qavit_branch_bwd_args& a, int tile) {
  constexpr bool WIN = (KIND == 0), KSH = (TT == 64 && KIND == 1);
  const int tid = threadIdx.x;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;
  const bf16* gg = reinterpret_cast<const bf16*>(a.dout);
  bf16* dzg = reinterpret_cast<bf16*>(a.dz);
  bf16* dqg = reinterpret_cast<bf16*>(a.dq);
  bf16x8 z8;
#pragma unroll
  for (int e = 0; e < 8; ++e) z8[e] = (bf16)0.f;
  for (int p = tid; p < 64 * 24; p += 512) {               // the tile's 64 rows x 24 pieces of 8
    const int r64 = p / 24, c8 = p - r64 * 24, sub = r64 >> 4, r = r64 & 15;
    if (!sub_valid<TT>(tile, sub, a.B)) continue;
    const int64_t row = tile_row<TT, WIN>(tile, sub, r, a.B);
    if (pdrop && dzg) {
      bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gg + (size_t)row * a.lddout + 8 * c8);
      const uint32_t base = (uint32_t)row * (uint32_t)BC + (uint32_t)(8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
      *reinterpret_cast<bf16x8*>(dzg + (size_t)row * a.lddz + 8 * c8) = g8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<bf16x4*>(dqg + (size_t)row * a.lddq + 8 * c8 + 4 * j) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (KIND != 2 && !KSH) {
      int64_t krow = row;
      bool ok = true;
      if (KIND == 1) { ok = r < a.kv_rows; krow = (int64_t)(tile * NI + sub) * a.kv_rows + r; }
      if (ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dk_tok) + (size_t)krow * a.lddkv + 8 * c8 + 4 * j) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dv_tok) + (size_t)krow * a.lddkv + 8 * c8 + 4 * j) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
      }
    }
    if (KSH && r64 < a.kv_rows) {                          // the image's landmark rows
      const int64_t krow = (int64_t)tile * a.kv_rows + r64;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dk_tok) + (size_t)krow * a.lddkv + 8 * c8 + 4 * j) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.dv_tok) + (size_t)krow * a.lddkv + 8 * c8 + 4 * j) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      }
    }
  }
  float* out = a.parts + (size_t)blockIdx.x * a.parts_stride;
  for (int e = tid; e < part_floats(TT); e += 512) out[e] = 0.f;
}

template <int KIND, int TT>
__global__ __launch_bounds__(512) void branch_bwd_kernel(qavit_branch_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr bool MODE0 = (KIND != 2);
  constexpr bool WIN = (KIND == 0);                        // TT = 64: the tile's sub-images are the 4x4 windows
  constexpr bool KSH = (TT == 64 && KIND == 1);            // ONE key side per image (MSDA on 64 tokens)
  constexpr int LT = KSH ? 3 : 1;                          // 16-row landmark tiles
  constexpr int PART_E = part_e(TT), PART_FLOATS = part_floats(TT);
  constexpr int KT0 = MODE0 ? 2 : 0, NKT = KT0 + 1, DT = 3, NKo = KT0 * 16;
  constexpr int KT0a = KT0 > 0 ? KT0 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  bf16* sbk = reinterpret_cast<bf16*>(smraw + BW_SM_BANK);
  bf16* sbv = sbk + 16 * LDB;
  bf16* sg_all = reinterpret_cast<bf16*>(smraw + BW_SM_G);           // dout * mask, then dO
  bf16* sq_all = reinterpret_cast<bf16*>(smraw + BW_SM_Q);
  bf16* sk_all = reinterpret_cast<bf16*>(smraw);                     // over the ring, after the proj GEMM
  bf16* sv_all = sk_all + NI * 16 * LDO;
  const int S = a.S, NK = NKo + S;
  const float scale = rsqrtf((float)BD);
  const int tile = blockIdx.x;
  const char* wpt = reinterpret_cast<const char*>(a.wprojT_frag);
  if (a.nan_trip && *a.nan_trip != 0) {                    // uniform over the grid
    branch_bwd_tripped<KIND, TT>(a, tile);
    return;
  }

  auto issue = [&](int c) { issue_chunk(wpt, 0, c, smraw + (c % RING) * CHUNK_BYTES, wave, lane); };
#pragma unroll
  for (int c = 0; c < AHEAD; ++c) issue(c);

  const bool adrop = a.attn_drop_p > 0.f && a.rng != nullptr;
  AttnDrop drop;
  drop.on = adrop;
  drop.p = adrop ? a.attn_drop_p : 0.f;
  drop.inv_keep = adrop ? 1.f / (1.f - a.attn_drop_p) : 1.f;
  drop.key = adrop ? rng_key(a.rng, a.attn_drop_site) : 0u;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;

  // ---------------- prologue loads (all issued before any is consumed) ----------------
  // wave w stages rows 8 (w & 1) .. + 8 of sub-image w >> 1: dout and q rows (24 16-byte pieces per row), k / v rows (kept in registers
  // until the ring is free; MSDA on 64 tokens: the image's landmark tile w >> 1 < 3)
  const int ssub = wave >> 1;
  const bool svalid = sub_valid<TT>(tile, ssub, a.B);
  const bf16* gg = reinterpret_cast<const bf16*>(a.dout);
  const bf16* qg = reinterpret_cast<const bf16*>(a.q);
  bf16x8 gr[3], qr[3], kr[MODE0 ? 3 : 1], vr[MODE0 ? 3 : 1];
  int64_t srow[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
    srow[it] = tile_row<TT, WIN>(tile, ssub, row, a.B);
    gr[it] = *reinterpret_cast<const bf16x8*>(gg + (size_t)srow[it] * a.lddout + 8 * c8);
    qr[it] = *reinterpret_cast<const bf16x8*>(qg + (size_t)srow[it] * a.ldq + 8 * c8);
    if (MODE0) {
      int64_t krow;
      if (KSH) { const int l = 16 * ssub + row; krow = (int64_t)tile * a.kv_rows + (l < a.kv_rows ? l : a.kv_rows - 1); }
      else if (KIND == 1) { const int kvr = row < a.kv_rows ? row : a.kv_rows - 1; krow = (srow[it] / BT) * a.kv_rows + kvr; }   // rows >= L are zero in the tiles (not in memory)
      else krow = srow[it];
      kr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.k_tok) + (size_t)krow * a.ldkv + 8 * c8);
      vr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.v_tok) + (size_t)krow * a.ldkv + 8 * c8);
    }
  }
  // Linformer matrices in both operand layouts: ekf lane holds E[l = 4 q4 + i][j = 16 jt + col], ekt lane holds E[l = col][j = 16 jt + 4 q4 + i].
  // MSDA on 64 tokens (48 landmark rows): bf16 tiles in LDS instead, fragments read where they are used (24 quads would not fit the registers)
  s16x4 ekf[KT0a], evf[KT0a], ekt[KT0a], evt[KT0a];
  bf16* sEk = reinterpret_cast<bf16*>(smraw + BW_SM_E);
  bf16* sEv = sEk + 48 * LDE;
  f32x4 er[KSH ? 2 : 1];
  if (KSH) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {                       // 2 x 48 rows x 8 quads = 768 = 1.5 x 512
      const int e0 = tid + 512 * it, e = e0 < 768 ? e0 : 0, which = e / 384, rem = e - which * 384, l = rem >> 3, c4 = rem & 7;
      const float* src = which ? a.E_v : a.E_k;
      er[it] = *reinterpret_cast<const f32x4*>(src + (size_t)(l < a.L ? l : 0) * a.KC + 4 * c4);
    }
  } else if (MODE0) {
#pragma unroll
    for (int jt = 0; jt < KT0; ++jt) {
      bf16x4 a1, a2, b1, b2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int l = 4 * q4 + i, lc = l < a.L ? l : 0;
        const float e1 = a.E_k[(size_t)lc * a.KC + jt * 16 + col], e2 = a.E_v[(size_t)lc * a.KC + jt * 16 + col];
        a1[i] = (bf16)(l < a.L ? e1 : 0.f);
        a2[i] = (bf16)(l < a.L ? e2 : 0.f);
        const int l2 = col < a.L ? col : 0;
        const float f1 = a.E_k[(size_t)l2 * a.KC + jt * 16 + 4 * q4 + i], f2 = a.E_v[(size_t)l2 * a.KC + jt * 16 + 4 * q4 + i];
        b1[i] = (bf16)(col < a.L ? f1 : 0.f);
        b2[i] = (bf16)(col < a.L ? f2 : 0.f);
      }
      ekf[jt] = as_s16(a1); evf[jt] = as_s16(a2); ekt[jt] = as_s16(b1); evt[jt] = as_s16(b2);
    }
  }
  f32x4 kk[2], vv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {                         // shared key / value rows: 16 x 48 chunks of 4 = 768 = 1.5 x 512
    const int e0 = tid + 512 * it, e = e0 < 768 ? e0 : 0, sr = e / (BC >> 2), ch = e - sr * (BC >> 2);
    kk[it] = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)sr * BC + 4 * ch);
    vv[it] = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)sr * BC + 4 * ch);
  }
  // ---------------- consumers ----------------
  bf16* dzg = reinterpret_cast<bf16*>(a.dz);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
    bf16x8 g8 = gr[it];
    if (pdrop) {
      const uint32_t base = (uint32_t)srow[it] * (uint32_t)BC + (uint32_t)(8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
      if (dzg && svalid) *reinterpret_cast<bf16x8*>(dzg + (size_t)srow[it] * a.lddz + 8 * c8) = g8;
    }
    *reinterpret_cast<bf16x8*>(sg_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = g8;
    *reinterpret_cast<bf16x8*>(sq_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = qr[it];
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + 512 * it;
    if (e < 768) {
      const int sr = e / (BC >> 2), ch = e - sr * (BC >> 2);
      bf16x4 kb, vb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { kb[j] = (bf16)kk[it][j]; vb[j] = (bf16)vv[it][j]; }
      *reinterpret_cast<bf16x4*>(sbk + sr * LDB + 4 * ch) = kb;
      *reinterpret_cast<bf16x4*>(sbv + sr * LDB + 4 * ch) = vb;
    }
  }
  if (KSH) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + 512 * it;
      if (e < 768) {
        const int which = e / 384, rem = e - which * 384, l = rem >> 3, c4 = rem & 7;
        bf16x4 v4;
#pragma unroll
        for (int j = 0; j < 4; ++j) v4[j] = (bf16)(l < a.L ? er[it][j] : 0.f);
        *reinterpret_cast<bf16x4*>((which ? sEv : sEk) + l * LDE + 4 * c4) = v4;
      }
    }
  }
  __syncthreads();              // tiles staged; the ordinary loads above and the ring's first chunks are drained: vmcnt is 0 here

  // ================= dO = gm . Wproj: WAVE = (IMAGE, COLUMN HALF), the forward proj phase with Wproj^T =================
  {
    const int pi = wave >> 1, half = wave & 1;
    bf16* sg = sg_all + pi * (16 * LDO);
    bf16x8 of8[KST];
    f32x4 acc[CT / 2];
#pragma unroll
    for (int jj = 0; jj < CT / 2; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      ring_wait((KST - 1 - s) < (AHEAD - 1) ? (KST - 1 - s) : (AHEAD - 1));
      if (s + AHEAD < KST) issue(s + AHEAD);
      if (s == 0) {
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) of8[s2] = *reinterpret_cast<const bf16x8*>(sg + col * LDO + 32 * s2 + 8 * q4);
      }
      const char* slot = smraw + (s % RING) * CHUNK_BYTES;
      bf16x8 wf[CT / 2];
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) wf[jj] = *reinterpret_cast<const bf16x8*>(slot + ((6 * half + jj) * 64 + lane) * 16);
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jj], of8[s], acc[jj], 0, 0, 0);
    }
    // both column halves of this image read their of8 rows before the barrier of k-step 1: the tile can take dO now
#pragma unroll
    for (int jj = 0; jj < CT / 2; ++jj) *reinterpret_cast<bf16x4*>(sg + col * LDO + (6 * half + jj) * 16 + 4 * q4) = cvt4(acc[jj]);
  }
  __syncthreads();              // dO tiles complete; every wave is done with the ring
  if (MODE0) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
      bf16x8 k8 = kr[it], v8 = vr[it];
      if ((KSH ? 16 * ssub + row : row) >= a.kv_rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k8[e] = (bf16)0.f; v8[e] = (bf16)0.f; }
      }
      *reinterpret_cast<bf16x8*>(sk_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = k8;
      *reinterpret_cast<bf16x8*>(sv_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = v8;
    }
    __syncthreads();
  }

  // ================= attention backward: WAVE = (HEAD, SUB-IMAGE PAIR) =================
  const int h = wave & 3, i0 = NIW * (wave >> 2);
  const bf16* og = reinterpret_cast<const bf16*>(a.o);
  bf16* dqg = reinterpret_cast<bf16*>(a.dq);
  bf16* dkg = reinterpret_cast<bf16*>(a.dk_tok);
  bf16* dvg = reinterpret_cast<bf16*>(a.dv_tok);
  // bank rows of this head in both operand layouts (MSDA on 64 tokens: re-read from the LDS tile per query tile -- the registers are
  // needed for the image's K_f / V_f and the running dK_f / dV_f sums)
  s16x4 bkA[DT], bkT[DT], bvA[DT], bvT[DT];
  auto load_bank = [&]() {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      bkA[t] = rowfrag(sbk, LDB, 0, h * BD + t * 16);        // lane: key s = col, 4 consecutive d
      bkT[t] = trfrag(sbk, LDB, 0, h * BD + t * 16);         // lane: d = col, 4 consecutive keys
      bvA[t] = rowfrag(sbv, LDB, 0, h * BD + t * 16);
      bvT[t] = trfrag(sbv, LDB, 0, h * BD + t * 16);
    }
  };
  if (!KSH) load_bank();
  f32x4 dshk[DT], dshv[DT], dEk[KT0a], dEv[KT0a];           // (MSDA on 64 tokens: dE lives in the tail only, dEt below)
#pragma unroll
  for (int t = 0; t < DT; ++t) { dshk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dshv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int jt = 0; jt < KT0a; ++jt) { dEk[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dEv[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  s16x4 idq;                                               // the 16 x 16 identity as a B operand: element (k = 4 q4 + j, column col)
  {
    bf16x4 t1;
#pragma unroll
    for (int j = 0; j < 4; ++j) t1[j] = (bf16)((4 * q4 + j == col) ? 1.f : 0.f);
    idq = as_s16(t1);
  }
  // MSDA on 64 tokens: K_f, V_f of the image's landmark rows, ONCE per wave, in both layouts; dK_f / dV_f summed over its query tiles
  s16x4 kfa[KT0a][DT], kfb[KT0a][DT], vfa[KT0a][DT], vfb[KT0a][DT];
  f32x4 dkfS[KSH ? KT0a : 1][KSH ? DT : 1], dvfS[KSH ? KT0a : 1][KSH ? DT : 1];
  if (KSH) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt) {
        f32x4 c1 = zero4, c2 = zero4, c3 = zero4, c4 = zero4;
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) {
          const s16x4 kT = trfrag(sk_all + lt * (16 * LDO), LDO, 0, h * BD + t * 16);      // lane: d = col, 4 consecutive landmarks
          const s16x4 vT = trfrag(sv_all + lt * (16 * LDO), LDO, 0, h * BD + t * 16);
          const s16x4 ekq = trfrag(sEk, LDE, 16 * lt, 16 * jt), evq = trfrag(sEv, LDE, 16 * lt, 16 * jt);   // lane: j = col, 4 consecutive landmarks
          c1 = mma16(kT, ekq, c1);
          c2 = mma16(ekq, kT, c2);
          c3 = mma16(vT, evq, c3);
          c4 = mma16(evq, vT, c4);
        }
        kfa[jt][t] = cvt4s(c1); kfb[jt][t] = cvt4s(c2); vfa[jt][t] = cvt4s(c3); vfb[jt][t] = cvt4s(c4);
        dkfS[jt][t] = zero4; dvfS[jt][t] = zero4;
      }
  }

  for (int i = 0; i < NIW; ++i) {
    if (!sub_valid<TT>(tile, i0 + i, a.B)) break;          // uniform per wave; no barrier below
    const int64_t qrow = tile_row<TT, WIN>(tile, i0 + i, col, a.B);   // global row of this lane's query
    const int prob = (TT == 64 && !WIN) ? tile : tile * NI + i0 + i;  // attention-dropout problem
    const int qoff = (TT == 64 && !WIN) ? 16 * (i0 + i) : 0;          // ... and the sub-image's first query index inside it
    if (KSH) { asm volatile("" ::: "memory"); load_bank(); }
    const bf16* sq = sq_all + (i0 + i) * (16 * LDO);
    const bf16* sdo = sg_all + (i0 + i) * (16 * LDO);
    const bf16* sk = sk_all + (i0 + i) * (16 * LDO);
    const bf16* sv = sv_all + (i0 + i) * (16 * LDO);
    s16x4 qB[DT], qT[DT], doB[DT], doT[DT];
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      qB[t] = rowfrag(sq, LDO, 0, h * BD + t * 16);        // lane: query = col, 4 consecutive d
      qT[t] = trfrag(sq, LDO, 0, h * BD + t * 16);         // lane: d = col, 4 consecutive queries
      doB[t] = rowfrag(sdo, LDO, 0, h * BD + t * 16);
      doT[t] = trfrag(sdo, LDO, 0, h * BD + t * 16);
      const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(og + (size_t)qrow * a.ldo + h * BD + t * 16 + 4 * q4);
      const bf16x4 d4 = __builtin_bit_cast(bf16x4, doB[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) dpart += (float)o4[r] * (float)d4[r];
    }
    dpart = rows4_sum(dpart);                    // D[query = col] = sum_d dO O over this head
    // K_f, V_f of the token / landmark rows in both layouts
    s16x4 kR[DT], vR[DT];
    if (MODE0 && !KSH) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const s16x4 kT = trfrag(sk, LDO, 0, h * BD + t * 16);      // lane: d = col, 4 consecutive tokens l
        const s16x4 vT = trfrag(sv, LDO, 0, h * BD + t * 16);
        kR[t] = rowfrag(sk, LDO, 0, h * BD + t * 16);              // lane: token l = col, 4 consecutive d
        vR[t] = rowfrag(sv, LDO, 0, h * BD + t * 16);
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          kfa[jt][t] = cvt4s(mma16(kT, ekf[jt], zero4));            // acc[r] = Kf[key = col][d = 4 q4 + r]: lane key, regs d
          kfb[jt][t] = cvt4s(mma16(ekf[jt], kT, zero4));            // acc[r] = Kf[key = 4 q4 + r][d = col]: lane d, regs key
          vfa[jt][t] = cvt4s(mma16(vT, evf[jt], zero4));
          vfb[jt][t] = cvt4s(mma16(evf[jt], vT, zero4));
        }
      }
    }
    // ---- S^T and dP^T (lane = query, registers = keys) ----
    f32x4 sT[NKT], dpT[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 c1 = zero4, c3 = zero4;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const s16x4 ka = nt < KT0 ? kfa[nt < KT0 ? nt : 0][t] : bkA[t];
        const s16x4 va = nt < KT0 ? vfa[nt < KT0 ? nt : 0][t] : bvA[t];
        c1 = mma16(ka, qB[t], c1);                          // S^T[key = 4 q4 + r][query = col]
        c3 = mma16(va, doB[t], c3);                         // dP^T
      }
      sT[nt] = c1; dpT[nt] = c3;
    }
    // softmax statistics per query
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = nt * 16 + 4 * q4 + r < NK;
        sT[nt][r] = ok ? sT[nt][r] * scale : -INFINITY;
        mx = fmaxf(mx, sT[nt][r]);
      }
    mx = rows4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(sT[nt][r] - mx); sT[nt][r] = e; sum += e; }
    sum = rows4_sum(sum);
    const float inv = 1.f / sum;
    const uint32_t pkey = adrop ? attn_drop_pkey(drop, prob * BH + h) : 0u;
    // (P m)^T, dS^T as bf16 quads; the other orientation (lane = key, registers = queries), which the contractions over QUERIES need,
    // by ONE MFMA against the identity each: a quad tile in accumulator layout read as an A operand is its own transpose, so the
    // product with I is exactly the transposed tile (bf16 values x 1.0) -- S and dP are not formed a second time, no second round of
    // exponentials, dropout hashes and statistic permutes.
    s16x4 dsT[NKT], ds2[NKT], pd2[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 d, pm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = sT[nt][r] * inv;
        const float m = adrop ? attn_drop_factor(drop, pkey, qoff + col, nt * 16 + 4 * q4 + r) : 1.f;
        pm[r] = p * m;
        d[r] = p * (dpT[nt][r] * m - dpart) * scale;
      }
      dsT[nt] = cvt4s(d);
      ds2[nt] = cvt4s(mma16(dsT[nt], idq, zero4));           // dS  [query = 4 q4 + r][key = col]
      pd2[nt] = cvt4s(mma16(cvt4s(pm), idq, zero4));         // P m [query = 4 q4 + r][key = col]
    }
    // ---- dQ^T[d][query] = sum_key Kf[key][d] dS^T[key][query]: 8-byte row segments of dq ----
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      f32x4 c = zero4;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) c = mma16(nt < KT0 ? kfb[nt < KT0 ? nt : 0][t] : bkT[t], dsT[nt], c);
      // into this head's columns of the sub-image's q tile (its q fragments are in registers; no other wave reads these columns): the
      // rows leave whole, in 16-byte pieces, once every head is done -- 8-byte segments straight from the quads touch 16 rows per store
      *reinterpret_cast<bf16x4*>(sq_all + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(c);
    }
    // ---- dKf, dVf: contraction over queries ----
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      dshk[t] = mma16(ds2[NKT - 1], qT[t], dshk[t]);        // shared rows: acc[r] = dKf[key s = 4 q4 + r][d = col], summed over images
      dshv[t] = mma16(pd2[NKT - 1], doT[t], dshv[t]);
    }
    if (KSH) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          dkfS[jt][t] = mma16(ds2[jt], qT[t], dkfS[jt][t]);  // dKf[key = 16 jt + 4 q4 + r][d = col], summed over the wave's query tiles
          dvfS[jt][t] = mma16(pd2[jt], doT[t], dvfS[jt][t]);
        }
    } else if (MODE0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f32x4 ck = zero4, cv = zero4;
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          const s16x4 dkf = cvt4s(mma16(ds2[jt], qT[t], zero4));       // dKf[key = 4 q4 + r][d = col]  -> A operand (lane d, regs key)
          const s16x4 dvf = cvt4s(mma16(pd2[jt], doT[t], zero4));
          ck = mma16(dkf, ekt[jt], ck);                              // dk^T[d][l] = sum_j dKf[j][d] E_k[l][j]: acc[r] = dk[l = col][d = 4 q4 + r]
          cv = mma16(dvf, evt[jt], cv);
          const s16x4 dkfT = cvt4s(mma16(qT[t], ds2[jt], zero4));       // dKf^T[d = 4 q4 + r][key = col] -> B operand (lane key, regs d)
          const s16x4 dvfT = cvt4s(mma16(doT[t], pd2[jt], zero4));
          dEk[jt] = mma16(kR[t], dkfT, dEk[jt]);                  // dE_k[l = 4 q4 + r][j = col] += sum_d k[l][d] dKf[j][d]
          dEv[jt] = mma16(vR[t], dvfT, dEv[jt]);
        }
        // dk / dv rows: into this head's columns of the sub-image's k / v tiles, out as whole rows below
        *reinterpret_cast<bf16x4*>(sk_all + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(ck);
        *reinterpret_cast<bf16x4*>(sv_all + (i0 + i) * (16 * LDO) + col * LDO + h * BD + t * 16 + 4 * q4) = cvt4(cv);
      }
    }
  }

  // dq (SWA / MSDA on 16 tokens: also dk, dv) rows out of the tiles: wave w stores rows 8 (w & 1) .. + 8 of sub-image w >> 1
  auto store_rows_out = [&]() {
    if (!sub_valid<TT>(tile, wave >> 1, a.B)) return;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24, st = wave >> 1;
      const int64_t grow = tile_row<TT, WIN>(tile, st, row, a.B);
      *reinterpret_cast<bf16x8*>(dqg + (size_t)grow * a.lddq + 8 * c8) = *reinterpret_cast<const bf16x8*>(sq_all + st * (16 * LDO) + row * LDO + 8 * c8);
      if (KSH) asm volatile("" ::: "memory");              // one piece at a time: this variant runs at the register limit
      if (MODE0 && !KSH) {
        const int64_t krow = KIND == 1 ? (grow / BT) * a.kv_rows + row : grow;          // MSDA: landmark rows < kv_rows of the image; SWA: the token rows
        if (KIND == 0 || row < a.kv_rows) {
          *reinterpret_cast<bf16x8*>(dkg + (size_t)krow * a.lddkv + 8 * c8) = *reinterpret_cast<const bf16x8*>(sk_all + st * (16 * LDO) + row * LDO + 8 * c8);
          *reinterpret_cast<bf16x8*>(dvg + (size_t)krow * a.lddkv + 8 * c8) = *reinterpret_cast<const bf16x8*>(sv_all + st * (16 * LDO) + row * LDO + 8 * c8);
        }
      }
    }
  };
  f32x4 dEt[LT][KT0a];                                     // MSDA on 64 tokens: dE_k (key-side wave) or dE_v (value-side wave) of this head
  if (KSH) {
    // ---- dK_f / dV_f of the image: this wave's partial + the partial of the head's other wave.  Wave h (< 4) finishes the KEY side of
    // head h, wave h + 4 the VALUE side; each parks the partial of the side it does not finish (fp32, [jt][t] accumulator quads). ----
    __syncthreads();                                       // the q / dO tiles are dead and every head's dq quads are in the q tiles
    store_rows_out();                                      // dq rows out before the region takes the parked partials
    __syncthreads();
    float* park = reinterpret_cast<float*>(smraw + BW_SM_G);           // [head][side] x 6 quads x 64 lanes x 4 floats = 6 KB each (48 KB <= 51.2 KB)
    const int mine = wave >> 2;                            // 0: key side, 1: value side
    {
      float* dst = park + ((h * 2 + (1 - mine)) * (KT0a * DT) * 256);
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int t = 0; t < DT; ++t) *reinterpret_cast<f32x4*>(dst + ((jt * DT + t) * 64 + lane) * 4) = mine ? dkfS[jt][t] : dvfS[jt][t];
    }
    __syncthreads();
    float* own = park + ((h * 2 + mine) * (KT0a * DT) * 256);
    bf16* ft = reinterpret_cast<bf16*>(own);               // ... then this wave's [32 keys][LDF] bf16 tile of the summed dK_f (dV_f): 3.5 KB over its own 6 KB
    f32x4 tot[KT0a][DT];
#pragma unroll
    for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(own + ((jt * DT + t) * 64 + lane) * 4);
        const f32x4 m = mine ? dvfS[jt][t] : dkfS[jt][t];
        tot[jt][t] = f32x4{o[0] + m[0], o[1] + m[1], o[2] + m[2], o[3] + m[3]};
      }
    wave_sync();                                           // every lane has read its parked quads before the tile overwrites them
#pragma unroll
    for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
      for (int t = 0; t < DT; ++t) acc_to_lds(ft, LDF, 16 * jt, 16 * t, tot[jt][t]);          // ft[key = 16 jt + 4 q4 + r][d = 16 t + col]
    wave_sync();
    const bf16* stok = mine ? sv_all : sk_all;             // the landmark rows of the side this wave finishes
    const bf16* sE = mine ? sEv : sEk;
    bf16* dtok = mine ? dvg : dkg;
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt) dEt[lt][jt] = zero4;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      s16x4 fA[KT0a], fB[KT0a];
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt) {
        fA[jt] = trfrag(ft, LDF, 16 * jt, 16 * t);         // A operand: lane d = col, 4 consecutive keys
        fB[jt] = rowfrag(ft, LDF, 16 * jt, 16 * t);        // B operand: lane key = col, 4 consecutive d
      }
#pragma unroll
      for (int lt = 0; lt < LT; ++lt) {
        f32x4 c = zero4;
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) c = mma16(fA[jt], rowfrag(sE, LDE, 16 * lt, 16 * jt), c);   // E[l = 16 lt + col][4 consecutive j]: dk[l = 16 lt + col][d = 4 q4 + r]
        if (16 * lt + col < a.kv_rows)
          *reinterpret_cast<bf16x4*>(dtok + ((size_t)tile * a.kv_rows + 16 * lt + col) * a.lddkv + h * BD + t * 16 + 4 * q4) = cvt4(c);
        const s16x4 rR = rowfrag(stok + lt * (16 * LDO), LDO, 0, h * BD + t * 16);               // lane: landmark l = col, 4 consecutive d
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) dEt[lt][jt] = mma16(rR, fB[jt], dEt[lt][jt]);             // dE[l = 16 lt + 4 q4 + r][j = 16 jt + col]
      }
    }
  }

  if (!KSH) {
    __syncthreads();                                       // every head's quads are in the tiles
    store_rows_out();
  }
  // ================= sums over the tile's images and heads -> one row of partial sums =================
  __syncthreads();                                         // every tile is dead: the ring region takes the fp32 scratch
  float* red = reinterpret_cast<float*>(smraw);            // [dE 2 * PART_E | dsh 2 * 3072] then the dE staging: image pair 1 parks, image pair 0 adds
  float* out = a.parts + (size_t)blockIdx.x * a.parts_stride;
  if (wave >= 4) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[2 * PART_E + (4 * q4 + r) * BC + h * BD + t * 16 + col] = dshk[t][r];
        red[2 * PART_E + PART_SH + (4 * q4 + r) * BC + h * BD + t * 16 + col] = dshv[t][r];
      }
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (4 * q4 + r) * BC + h * BD + t * 16 + col;
        out[2 * PART_E + o] = dshk[t][r] + red[2 * PART_E + o];
        out[2 * PART_E + PART_SH + o] = dshv[t][r] + red[2 * PART_E + PART_SH + o];
      }
  }
  if (MODE0) {
    float* re = red + PART_FLOATS;
    if (KSH) {
      // dE: the key-side wave of each head holds dE_k, the value-side wave dE_v -> LDS [4 heads][2][48][32], then 3072 sums of 4
      int q4t = q4;
      asm volatile("" : "+v"(q4t));                        // opaque here: hipcc otherwise forms these 24 row indices at kernel entry and spills them
#pragma unroll
      for (int lt = 0; lt < LT; ++lt)
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            re[(h * 2 + (wave >> 2)) * PART_E + (16 * lt + 4 * q4t + r) * 32 + jt * 16 + col] = dEt[lt][jt][r];
      __syncthreads();
      for (int e = tid; e < 2 * PART_E; e += 512) {
        const int side = e / PART_E, o = e - side * PART_E;
        float s = 0.f;
#pragma unroll
        for (int hh = 0; hh < BH; ++hh) s += re[(hh * 2 + side) * PART_E + o];
        out[e] = s;
      }
    } else {
      // dE: 8 waves -> LDS [8][2][16][32], then 1024 sums of 8 (64 tokens, SWA: the [48][32] slots' rows past the 16 window positions are zero)
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          re[(wave * 2 + 0) * 512 + (4 * q4 + r) * 32 + jt * 16 + col] = dEk[jt][r];
          re[(wave * 2 + 1) * 512 + (4 * q4 + r) * 32 + jt * 16 + col] = dEv[jt][r];
        }
      __syncthreads();
      for (int e = tid; e < 2 * PART_E; e += 512) {
        const int side = e / PART_E, o = e - side * PART_E;
        float s = 0.f;
        if (o < 512) {
#pragma unroll
          for (int w = 0; w < NW; ++w) s += re[(w * 2 + side) * 512 + o];
        }
        out[e] = s;
      }
    }
  }
}

int branch_bwd_validate(const qavit_branch_bwd_args* a) {
  if (!a) return set_error(QAVIT_EINVAL, "branch_bwd: null args");
  if (a->kind < 0 || a->kind > 2) return set_error(QAVIT_EINVAL, "branch_bwd: kind must be 0 (SWA), 1 (MSDA) or 2 (cross)");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "branch_bwd: bf16 only");
  if ((a->T != 16 && a->T != 64) || a->C != BC || a->H != BH || a->D != BD || a->S != 16)
    return set_error(QAVIT_EINVAL, "branch_bwd: built for 16 or 64 tokens x 192 channels, 4 heads of 48, 16 bank rows");
  if (a->B <= 0 || !a->dout || !a->wprojT_frag || !a->q || !a->o || !a->sh_k || !a->sh_v || !a->dq || !a->parts)
    return set_error(QAVIT_EINVAL, "branch_bwd: null operand");
  if (a->kind != 2) {
    const int lmax = (a->kind == 1 && a->T == 64) ? 48 : 16;
    if (a->KC != 32 || a->L <= 0 || a->L > lmax || !a->E_k || !a->E_v || !a->k_tok || !a->v_tok || !a->dk_tok || !a->dv_tok)
      return set_error(QAVIT_EINVAL, "branch_bwd: SWA / MSDA need Linformer matrices (KC = 32, 1 <= L <= 16; MSDA on 64 tokens <= 48) and the saved k / v rows");
    if (a->kv_rows <= 0 || a->kv_rows > lmax || a->kv_rows < a->L) return set_error(QAVIT_EINVAL, "branch_bwd: kv_rows must cover the L Linformer rows (<= 16; MSDA on 64 tokens <= 48)");
    if (a->kind == 0 && (a->L != 16 || a->kv_rows != 16)) return set_error(QAVIT_EINVAL, "branch_bwd: SWA works on 4x4 windows (L = kv_rows = 16)");
    if (a->ldkv % 8 || a->lddkv % 4 || (reinterpret_cast<uintptr_t>(a->k_tok) & 15) || (reinterpret_cast<uintptr_t>(a->v_tok) & 15) ||
        (reinterpret_cast<uintptr_t>(a->dk_tok) & 15) || (reinterpret_cast<uintptr_t>(a->dv_tok) & 15) || a->lddkv % 8)
      return set_error(QAVIT_EINVAL, "branch_bwd: k / v rows and dk / dv need 16-byte alignment (ld % 8)");
  }
  if (a->proj_drop_p > 0.f && a->rng && !a->dz) return set_error(QAVIT_EINVAL, "branch_bwd: proj dropout needs dz (the masked gradient is the operand of dW_proj)");
  if (a->parts_stride < part_floats(a->T) || (reinterpret_cast<uintptr_t>(a->parts) & 15) || a->parts_stride % 4)
    return set_error(QAVIT_EINVAL, "branch_bwd: parts rows hold QAVIT_BRANCH_PARTS_FLOATS (T = 64: QAVIT_BRANCH_PARTS_FLOATS_64) floats, 16-byte aligned");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al16(a->dout) || !al16(a->wprojT_frag) || !al16(a->q) || !al16(a->sh_k) || !al16(a->sh_v) || (a->dz && !al16(a->dz)) ||
      a->lddout % 8 || a->ldq % 8 || (a->dz && a->lddz % 8) || (reinterpret_cast<uintptr_t>(a->o) & 7) || a->ldo % 4 ||
      (reinterpret_cast<uintptr_t>(a->dq) & 15) || a->lddq % 8)
    return set_error(QAVIT_EINVAL, "branch_bwd: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements (o: 8 bytes / 4 elements)");
  return QAVIT_OK;
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_branch_bwd_parts(int B, int T) { return B > 0 ? (T == 64 ? B : (B + NI - 1) / NI) : 0; }

extern "C" int qavit_branch_bwd(const qavit_branch_bwd_args* a, void* stream) {
  int rc = branch_bwd_validate(a);
  if (rc) return rc;
  static_assert(part_floats(16) == QAVIT_BRANCH_PARTS_FLOATS && part_floats(64) == QAVIT_BRANCH_PARTS_FLOATS_64, "header constants out of date");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = a->T == 64;
  const int grid = wide ? a->B : (a->B + NI - 1) / NI;
  static bool attr_done[3][2] = {};
#define QV_BWD_LAUNCH(K, TT)                                                                                                              \
  do {                                                                                                                                   \
    if (!attr_done[K][TT == 64]) {                                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(branch_bwd_kernel<K, TT>), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                (K == 1 && TT == 64) ? BW_SM_TOTAL_KSH : BW_SM_TOTAL);                                                   \
      attr_done[K][TT == 64] = true;                                                                                                     \
    }                                                                                                                                    \
    hipLaunchKernelGGL((branch_bwd_kernel<K, TT>), dim3(grid), dim3(512), (K == 1 && TT == 64) ? BW_SM_TOTAL_KSH : BW_SM_TOTAL, st, *a); \
  } while (0)
  if (a->kind == 0) { if (wide) QV_BWD_LAUNCH(0, 64); else QV_BWD_LAUNCH(0, 16); }
  else if (a->kind == 1) { if (wide) QV_BWD_LAUNCH(1, 64); else QV_BWD_LAUNCH(1, 16); }
  else { if (wide) QV_BWD_LAUNCH(2, 64); else QV_BWD_LAUNCH(2, 16); }
#undef QV_BWD_LAUNCH
  return check_launch("branch_bwd");
}
