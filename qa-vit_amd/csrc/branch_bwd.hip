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
constexpr int PART_E = 16 * 32;                                      // dE_k / dE_v: [L <= 16][KC = 32]
constexpr int PART_SH = 16 * BC;                                     // shared-row gradients [16][192]
constexpr int PART_FLOATS = 2 * PART_E + 2 * PART_SH;                // 7168 floats per workgroup

__device__ __forceinline__ s16x4 cvt4s(const f32x4& acc) { return as_s16(cvt4(acc)); }

template <int KIND>
__global__ __launch_bounds__(512) void branch_bwd_kernel(qavit_branch_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr bool MODE0 = (KIND != 2);
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
  // wave w stages rows 8 (w & 1) .. + 8 of image w >> 1: dout and q rows (24 16-byte pieces per row), k / v rows (kept in registers
  // until the ring is free)
  const int simg_raw = tile * NI + (wave >> 1);
  const bool svalid = simg_raw < a.B;
  const int simg = svalid ? simg_raw : a.B - 1;
  const bf16* gg = reinterpret_cast<const bf16*>(a.dout);
  const bf16* qg = reinterpret_cast<const bf16*>(a.q);
  bf16x8 gr[3], qr[3], kr[MODE0 ? 3 : 1], vr[MODE0 ? 3 : 1];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * (wave & 1) + p / 24, c8 = p % 24;
    gr[it] = *reinterpret_cast<const bf16x8*>(gg + ((size_t)simg * BT + row) * a.lddout + 8 * c8);
    qr[it] = *reinterpret_cast<const bf16x8*>(qg + ((size_t)simg * BT + row) * a.ldq + 8 * c8);
    if (MODE0) {
      const int kvr = row < a.kv_rows ? row : a.kv_rows - 1;         // MSDA: rows >= L are zero in the tiles (not in memory)
      kr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.k_tok) + ((size_t)simg * a.kv_rows + kvr) * a.ldkv + 8 * c8);
      vr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.v_tok) + ((size_t)simg * a.kv_rows + kvr) * a.ldkv + 8 * c8);
    }
  }
  // Linformer matrices in both operand layouts: ekf lane holds E[l = 4 q4 + i][j = 16 jt + col], ekt lane holds E[l = col][j = 16 jt + 4 q4 + i]
  s16x4 ekf[KT0a], evf[KT0a], ekt[KT0a], evt[KT0a];
  if (MODE0) {
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
      const uint32_t base = (uint32_t)(simg * BT + row) * (uint32_t)BC + (uint32_t)(8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
      if (dzg && svalid) *reinterpret_cast<bf16x8*>(dzg + ((size_t)simg * BT + row) * a.lddz + 8 * c8) = g8;
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
      if (row >= a.kv_rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k8[e] = (bf16)0.f; v8[e] = (bf16)0.f; }
      }
      *reinterpret_cast<bf16x8*>(sk_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = k8;
      *reinterpret_cast<bf16x8*>(sv_all + (wave >> 1) * (16 * LDO) + row * LDO + 8 * c8) = v8;
    }
    __syncthreads();
  }

  // ================= attention backward: WAVE = (HEAD, IMAGE PAIR) =================
  const int h = wave & 3, i0 = NIW * (wave >> 2);
  const bf16* og = reinterpret_cast<const bf16*>(a.o);
  bf16* dqg = reinterpret_cast<bf16*>(a.dq);
  bf16* dkg = reinterpret_cast<bf16*>(a.dk_tok);
  bf16* dvg = reinterpret_cast<bf16*>(a.dv_tok);
  // bank rows of this head in both operand layouts
  s16x4 bkA[DT], bkT[DT], bvA[DT], bvT[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    bkA[t] = rowfrag(sbk, LDB, 0, h * BD + t * 16);        // lane: key s = col, 4 consecutive d
    bkT[t] = trfrag(sbk, LDB, 0, h * BD + t * 16);         // lane: d = col, 4 consecutive keys
    bvA[t] = rowfrag(sbv, LDB, 0, h * BD + t * 16);
    bvT[t] = trfrag(sbv, LDB, 0, h * BD + t * 16);
  }
  f32x4 dshk[DT], dshv[DT], dEk[KT0a], dEv[KT0a];
#pragma unroll
  for (int t = 0; t < DT; ++t) { dshk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dshv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int jt = 0; jt < KT0a; ++jt) { dEk[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dEv[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int i = 0; i < NIW; ++i) {
    const int img = tile * NI + i0 + i;
    if (img >= a.B) break;                                 // uniform per wave; no barrier below
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
      const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(og + ((size_t)img * BT + col) * a.ldo + h * BD + t * 16 + 4 * q4);
      const bf16x4 d4 = __builtin_bit_cast(bf16x4, doB[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) dpart += (float)o4[r] * (float)d4[r];
    }
    dpart = rows4_sum(dpart);                    // D[query = col] = sum_d dO O over this head
    // K_f, V_f of the token / landmark rows in both layouts
    s16x4 kfa[KT0a][DT], kfb[KT0a][DT], vfa[KT0a][DT], vfb[KT0a][DT], kR[DT], vR[DT];
    if (MODE0) {
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
    // ---- S and dP in both orientations ----
    f32x4 sT[NKT], s2[NKT], dpT[NKT], dp2[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 c1 = zero4, c2 = zero4, c3 = zero4, c4 = zero4;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const s16x4 ka = nt < KT0 ? kfa[nt < KT0 ? nt : 0][t] : bkA[t];
        const s16x4 va = nt < KT0 ? vfa[nt < KT0 ? nt : 0][t] : bvA[t];
        c1 = mma16(ka, qB[t], c1);                          // S^T[key = 4 q4 + r][query = col]
        c2 = mma16(qB[t], ka, c2);                          // S  [query = 4 q4 + r][key = col]
        c3 = mma16(va, doB[t], c3);                         // dP^T
        c4 = mma16(doB[t], va, c4);                         // dP
      }
      sT[nt] = c1; s2[nt] = c2; dpT[nt] = c3; dp2[nt] = c4;
    }
    // softmax statistics per query from the first orientation
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
    const uint32_t pkey = adrop ? attn_drop_pkey(drop, img * BH + h) : 0u;
    // first orientation: P^T, dS^T (lane = query)
    s16x4 dsT[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 d;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = sT[nt][r] * inv;
        const float m = adrop ? attn_drop_factor(drop, pkey, col, nt * 16 + 4 * q4 + r) : 1.f;
        d[r] = p * (dpT[nt][r] * m - dpart) * scale;
      }
      dsT[nt] = cvt4s(d);
    }
    // second orientation (lane = key, registers = queries 4 q4 + r): statistics by lane permute
    float mxq[4], invq[4], dq_[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mxq[r] = __shfl(mx, 4 * q4 + r, 64);
      invq[r] = __shfl(inv, 4 * q4 + r, 64);
      dq_[r] = __shfl(dpart, 4 * q4 + r, 64);
    }
    s16x4 ds2[NKT], pd2[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 d, pm;
      const bool ok = nt * 16 + col < NK;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = ok ? __expf(s2[nt][r] * scale - mxq[r]) * invq[r] : 0.f;
        const float m = adrop ? attn_drop_factor(drop, pkey, 4 * q4 + r, nt * 16 + col) : 1.f;
        pm[r] = p * m;
        d[r] = p * (dp2[nt][r] * m - dq_[r]) * scale;
      }
      ds2[nt] = cvt4s(d);
      pd2[nt] = cvt4s(pm);
    }
    // ---- dQ^T[d][query] = sum_key Kf[key][d] dS^T[key][query]: 8-byte row segments of dq ----
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      f32x4 c = zero4;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) c = mma16(nt < KT0 ? kfb[nt < KT0 ? nt : 0][t] : bkT[t], dsT[nt], c);
      *reinterpret_cast<bf16x4*>(dqg + ((size_t)img * BT + col) * a.lddq + h * BD + t * 16 + 4 * q4) = cvt4(c);
    }
    // ---- dKf, dVf: contraction over queries ----
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      dshk[t] = mma16(ds2[NKT - 1], qT[t], dshk[t]);        // shared rows: acc[r] = dKf[key s = 4 q4 + r][d = col], summed over images
      dshv[t] = mma16(pd2[NKT - 1], doT[t], dshv[t]);
    }
    if (MODE0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f32x4 ck = zero4, cv = zero4;
#pragma unroll
        for (int jt = 0; jt < KT0; ++jt) {
          const s16x4 dkf = cvt4s(mma16(ds2[jt], qT[t], zero4));       // dKf[key = 4 q4 + r][d = col]  -> A operand (lane d, regs key)
          const s16x4 dvf = cvt4s(mma16(pd2[jt], doT[t], zero4));
          ck = mma16(dkf, ekt[jt], ck);                                 // dk^T[d][l] = sum_j dKf[j][d] E_k[l][j]: acc[r] = dk[l = col][d = 4 q4 + r]
          cv = mma16(dvf, evt[jt], cv);
          const s16x4 dkfT = cvt4s(mma16(qT[t], ds2[jt], zero4));       // dKf^T[d = 4 q4 + r][key = col] -> B operand (lane key, regs d)
          const s16x4 dvfT = cvt4s(mma16(doT[t], pd2[jt], zero4));
          dEk[jt] = mma16(kR[t], dkfT, dEk[jt]);                        // dE_k[l = 4 q4 + r][j = col] += sum_d k[l][d] dKf[j][d]
          dEv[jt] = mma16(vR[t], dvfT, dEv[jt]);
        }
        if (col < a.kv_rows) {
          *reinterpret_cast<bf16x4*>(dkg + ((size_t)img * a.kv_rows + col) * a.lddkv + h * BD + t * 16 + 4 * q4) = cvt4(ck);
          *reinterpret_cast<bf16x4*>(dvg + ((size_t)img * a.kv_rows + col) * a.lddkv + h * BD + t * 16 + 4 * q4) = cvt4(cv);
        }
      }
    }
  }

  // ================= sums over the tile's images and heads -> one row of partial sums =================
  __syncthreads();                                         // every tile is dead: the ring region takes the fp32 scratch
  float* red = reinterpret_cast<float*>(smraw);            // [2][dE 2 * 512 | dsh 2 * 3072]: image pair 1 parks, image pair 0 adds
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
    // dE: 8 waves -> LDS [8][2][16][32], then 1024 sums of 8
    float* re = red + PART_FLOATS;
#pragma unroll
    for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        re[(wave * 2 + 0) * PART_E + (4 * q4 + r) * 32 + jt * 16 + col] = dEk[jt][r];
        re[(wave * 2 + 1) * PART_E + (4 * q4 + r) * 32 + jt * 16 + col] = dEv[jt][r];
      }
    __syncthreads();
    for (int e = tid; e < 2 * PART_E; e += 512) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += re[w * 2 * PART_E + e];
      out[e] = s;
    }
  }
}

int branch_bwd_validate(const qavit_branch_bwd_args* a) {
  if (!a) return set_error(QAVIT_EINVAL, "branch_bwd: null args");
  if (a->kind < 0 || a->kind > 2) return set_error(QAVIT_EINVAL, "branch_bwd: kind must be 0 (SWA), 1 (MSDA) or 2 (cross)");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "branch_bwd: bf16 only");
  if (a->T != BT || a->C != BC || a->H != BH || a->D != BD || a->S != 16)
    return set_error(QAVIT_EINVAL, "branch_bwd: built for 16 tokens x 192 channels, 4 heads of 48, 16 bank rows");
  if (a->B <= 0 || !a->dout || !a->wprojT_frag || !a->q || !a->o || !a->sh_k || !a->sh_v || !a->dq || !a->parts)
    return set_error(QAVIT_EINVAL, "branch_bwd: null operand");
  if (a->kind != 2) {
    if (a->KC != 32 || a->L <= 0 || a->L > 16 || !a->E_k || !a->E_v || !a->k_tok || !a->v_tok || !a->dk_tok || !a->dv_tok)
      return set_error(QAVIT_EINVAL, "branch_bwd: SWA / MSDA need Linformer matrices (KC = 32, 1 <= L <= 16) and the saved k / v rows");
    if (a->kv_rows <= 0 || a->kv_rows > 16 || a->kv_rows < a->L) return set_error(QAVIT_EINVAL, "branch_bwd: kv_rows must cover the L Linformer rows (<= 16)");
    if (a->ldkv % 8 || a->lddkv % 4 || (reinterpret_cast<uintptr_t>(a->k_tok) & 15) || (reinterpret_cast<uintptr_t>(a->v_tok) & 15) ||
        (reinterpret_cast<uintptr_t>(a->dk_tok) & 7) || (reinterpret_cast<uintptr_t>(a->dv_tok) & 7))
      return set_error(QAVIT_EINVAL, "branch_bwd: k / v rows need 16-byte alignment (ld % 8), dk / dv 8-byte (ld % 4)");
  }
  if (a->proj_drop_p > 0.f && a->rng && !a->dz) return set_error(QAVIT_EINVAL, "branch_bwd: proj dropout needs dz (the masked gradient is the operand of dW_proj)");
  if (a->parts_stride < PART_FLOATS || (reinterpret_cast<uintptr_t>(a->parts) & 15) || a->parts_stride % 4)
    return set_error(QAVIT_EINVAL, "branch_bwd: parts rows hold QAVIT_BRANCH_PARTS_FLOATS floats, 16-byte aligned");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al16(a->dout) || !al16(a->wprojT_frag) || !al16(a->q) || !al16(a->sh_k) || !al16(a->sh_v) || (a->dz && !al16(a->dz)) ||
      a->lddout % 8 || a->ldq % 8 || (a->dz && a->lddz % 8) || (reinterpret_cast<uintptr_t>(a->o) & 7) || a->ldo % 4 ||
      (reinterpret_cast<uintptr_t>(a->dq) & 7) || a->lddq % 4)
    return set_error(QAVIT_EINVAL, "branch_bwd: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements (o, dq: 8 bytes / 4 elements)");
  return QAVIT_OK;
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_branch_bwd_parts(int B) { return B > 0 ? (B + NI - 1) / NI : 0; }

extern "C" int qavit_branch_bwd(const qavit_branch_bwd_args* a, void* stream) {
  int rc = branch_bwd_validate(a);
  if (rc) return rc;
  static_assert(PART_FLOATS == QAVIT_BRANCH_PARTS_FLOATS, "header constant out of date");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = (a->B + NI - 1) / NI;
  static bool attr_done[3] = {false, false, false};
#define QV_BWD_LAUNCH(K)                                                                                                                  \
  do {                                                                                                                                   \
    if (!attr_done[K]) {                                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(branch_bwd_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, BW_SM_TOTAL); \
      attr_done[K] = true;                                                                                                               \
    }                                                                                                                                    \
    hipLaunchKernelGGL((branch_bwd_kernel<K>), dim3(grid), dim3(512), BW_SM_TOTAL, st, *a);                                             \
  } while (0)
  if (a->kind == 0) QV_BWD_LAUNCH(0);
  else if (a->kind == 1) QV_BWD_LAUNCH(1);
  else QV_BWD_LAUNCH(2);
#undef QV_BWD_LAUNCH
  return check_launch("branch_bwd");
}
