// Fused attention BRANCH, forward, for the 16-learned-token problems of the CIFAR configuration (every HQA-ViT C100 block):
//   out = dropout( proj( SDPA( q(x), [Linformer(k(x'), v(x')) ; bank rows], dropout_p ) ) )
// in ONE launch -- the chain HQAViT_CIFAR100.py:441-469 (SWA: one 4x4 window = the 16 tokens), :496-532 (MSDA: keys from
// the pooled dilated landmarks x'), :613-626 (cross: keys = projections of the bank) runs as  qkv GEMM -> Linformer ->
// bank concat -> softmax (+ attention dropout) -> P.V -> proj GEMM (+ bias, dropout)  with Q / K / V / P / O never leaving
// LDS.  bf16 operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16 for the two GEMMs, 16x16x16 for the 16 x 48 x 48 core).
//
// Decomposition: a workgroup of 4 waves owns 4 images, ONE IMAGE PER WAVE (its 16 x 192 token tile is 6 MFMA operand
// fragments held in registers for the whole QKV GEMM).  Weights are what the images share: they stream through LDS in
// 48-column chunks (3 MFMA tiles x 6 k-steps = 18 KB) -- q_h, k_h, v_h of head h, then the head's attention core, ... then
// the 4 chunks of proj -- double buffered: chunk i+1 lands (global_load_lds, 16 B per lane, no registers) while chunk i's
// 18 MFMAs per wave run; one barrier per chunk.  The packed weight image (qavit_pack_desc.pad = 1) is already in MFMA
// fragment order, so a chunk is a linear 18 KB copy and every fragment read is one conflict-free ds_read_b128.
// Products are formed TRANSPOSED (A = weights, B = tokens): a lane's 4 accumulator values are 4 consecutive columns of one
// token row = one 8-byte store into the row-major LDS tiles the attention core (the attn3_bf16.hip schedule) reads.
// Per-CU floor at B = 1024: every CU must take in all 288 KB of the branch's weights for its 64 rows, ~3 us at L2 rates,
// about the MFMA time of those rows (390 MFMAs x 16 cycles per wave) -- the two overlap.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

namespace {

constexpr int BT = 16, BC = 192, BD = 48, BH = 4;          // tokens per image, channels, head dim, heads
constexpr int KST = BC / 32, TPC = 3;                      // k-steps of 32, 16-column MFMA tiles per chunk (= one head's q, k or v)
constexpr int CHUNK_BYTES = TPC * KST * 1024;              // 18432
constexpr int LDD = BD + 4, LDK = 48 + 4, LDE = 32 + 4, LDO = BC + 8;
// per-wave LDS tiles (bf16 elements)
constexpr int W_Q = 0, W_KT = 16 * LDD, W_VT = 2 * 16 * LDD, W_KF = 3 * 16 * LDD, W_VF = W_KF + 48 * LDD, W_P = W_VF + 48 * LDD,
              W_O = W_P + 16 * LDK, W_TOTAL = W_O + 16 * LDO;
constexpr int WAVE_BYTES = W_TOTAL * 2;                    // 23040
constexpr int LDB = BC + 8;                                // shared bank tiles [16][LDB] (all heads), read in place by the key / value fragments
constexpr int SM_E = 2 * CHUNK_BYTES, SM_BANK = SM_E + 2 * 16 * LDE * 2, SM_WAVE = SM_BANK + 2 * 16 * LDB * 2,
              SM_TOTAL = SM_WAVE + 4 * WAVE_BYTES;         // 144128 bytes: one workgroup per CU

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void row4_lds(bf16* dst, const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  *reinterpret_cast<bf16x4*>(dst) = v;
}
__device__ __forceinline__ bool nan4(const f32x4& v) { return (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]); }

// one 18 KB weight chunk global -> LDS: 18 fragments of 1 KB, dealt to the 4 waves; lane-linear image (base + lane * 16)
__device__ __forceinline__ void issue_chunk(const char* gsrc, char* ldst, int wave, int lane) {
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    const int fr = wave + 4 * f;
    if (fr < TPC * KST)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc + fr * 1024 + lane * 16), (lds_void_t*)(ldst + fr * 1024), 16, 0, 0);
  }
}

// acc[t] (+)= W_chunk[t] . X^T : 18 MFMAs, weight fragments from the LDS chunk, token fragments from registers
__device__ __forceinline__ void chunk_gemm(const char* wbuf, const bf16x8 (&xf)[KST], f32x4 (&acc)[TPC], int lane) {
#pragma unroll
  for (int s = 0; s < KST; ++s)
#pragma unroll
    for (int t = 0; t < TPC; ++t) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wbuf + ((t * KST + s) * 64 + lane) * 16);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s], acc[t], 0, 0, 0);
    }
}

// KIND 0 = SWA, 1 = MSDA, 2 = cross
template <int KIND>
__global__ __launch_bounds__(256) void branch_fwd_kernel(qavit_branch_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr int MODE0 = (KIND != 2);                       // Linformer + bank keys (SWA / MSDA) vs bank-projection keys only (cross)
  constexpr int KT0 = MODE0 ? 2 : 0, NKT = KT0 + 1, DT = 3, NKo = KT0 * 16;
  constexpr int NQKV = MODE0 ? 12 : 4, NCH = NQKV + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  bf16* ws = reinterpret_cast<bf16*>(smraw + SM_WAVE + wave * WAVE_BYTES);
  bf16* sek = reinterpret_cast<bf16*>(smraw + SM_E);
  bf16* sev = sek + 16 * LDE;
  bf16* sbk = reinterpret_cast<bf16*>(smraw + SM_BANK);
  bf16* sbv = sbk + 16 * LDB;
  const int S = a.S, NK = NKo + S;
  const float scale = rsqrtf((float)BD);
  const bf16* xg = reinterpret_cast<const bf16*>(a.x);
  bf16* og = reinterpret_cast<bf16*>(a.out);
  const char* wqkv = reinterpret_cast<const char*>(a.wqkv_frag);
  const char* wproj = reinterpret_cast<const char*>(a.wproj_frag);
  bool bad = false;

  AttnDrop drop;
  drop.on = a.attn_drop_p > 0.f && a.rng != nullptr;
  drop.p = a.attn_drop_p;
  drop.inv_keep = drop.on ? 1.f / (1.f - a.attn_drop_p) : 1.f;
  drop.key = drop.on ? rng_key(a.rng, a.attn_drop_site) : 0u;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;

  for (int i = lane; i < W_TOTAL; i += 64) ws[i] = (bf16)0.f;
  for (int i = tid; i < 2 * 16 * LDE; i += 256) sek[i] = (bf16)0.f;
  __syncthreads();
  if (MODE0) {
    const int EC = a.KC >> 2;                              // Linformer matrices: first L rows (zero padding of the rest is algebraic)
    for (int i = tid; i < a.L * EC; i += 256) {
      const int l = i / EC, ch = i - l * EC;
      const f32x4 ek = *reinterpret_cast<const f32x4*>(a.E_k + (size_t)l * a.KC + 4 * ch);
      const f32x4 ev = *reinterpret_cast<const f32x4*>(a.E_v + (size_t)l * a.KC + 4 * ch);
      bf16x4 kb, vb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { kb[j] = (bf16)ek[j]; vb[j] = (bf16)ev[j]; }
      *reinterpret_cast<bf16x4*>(sek + l * LDE + 4 * ch) = kb;
      *reinterpret_cast<bf16x4*>(sev + l * LDE + 4 * ch) = vb;
    }
  }
  // shared key / value rows of every head: the bank (SWA / MSDA) or its projections (cross), fp32 -> bf16 once per workgroup
  for (int e = tid; e < S * (BC >> 2); e += 256) {
    const int sr = e / (BC >> 2), ch = e - sr * (BC >> 2);
    const f32x4 k = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)sr * BC + 4 * ch);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)sr * BC + 4 * ch);
    bf16x4 kb, vb;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bad |= (k[j] != k[j]) | (v[j] != v[j]); kb[j] = (bf16)k[j]; vb[j] = (bf16)v[j]; }
    *reinterpret_cast<bf16x4*>(sbk + sr * LDB + 4 * ch) = kb;
    *reinterpret_cast<bf16x4*>(sbv + sr * LDB + 4 * ch) = vb;
  }

  const int ntiles = (a.B + 3) >> 2;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int img_raw = tile * 4 + wave;
    const bool valid = img_raw < a.B;
    const int img = valid ? img_raw : a.B - 1;
    __syncthreads();                                       // nobody still reads a weight buffer of the previous tile; E / bank are staged
    issue_chunk(wqkv, smraw, wave, lane);                  // chunk 0 = q of head 0: lands while the token tile is fetched
    // ---------------- this image's token tile as MFMA operand fragments: token = col, k = 32 s + 8 q4 .. + 8 ----------------
    bf16x8 xf[KST], pf[KST];
    {
      const bf16* xr = xg + ((size_t)img * BT + col) * a.ldx + 8 * q4;
#pragma unroll
      for (int s = 0; s < KST; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xr + 32 * s);
      if (KIND == 1) {
        // MSDA landmarks: pooled[j] = mean_s x[idx[j * stride + s]] (HQAViT_CIFAR100.py:499-501), j < L; fp32 mean, one rounding
        const int j = col < a.L ? col : 0;
        const float inv = 1.f / (float)a.pool_stride;
#pragma unroll
        for (int s = 0; s < KST; ++s) {
          float sum[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) sum[e] = 0.f;
          for (int t = 0; t < a.pool_stride; ++t) {
            const int src = a.pool_idx[j * a.pool_stride + t];
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xg + ((size_t)img * BT + src) * a.ldx + 32 * s + 8 * q4);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[e] += (float)v[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[s][e] = (bf16)(col < a.L ? sum[e] * inv : 0.f);
        }
      }
    }
    bf16x8 of[KST];

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      __syncthreads();                                     // chunk i has landed (the barrier's fence drains vmcnt); buffer (i+1)&1 is free
      if (i + 1 < NCH) {
        const int n = i + 1;
        const char* src;
        if (n < NQKV) {
          const int hh = MODE0 ? n / 3 : n, part = MODE0 ? n % 3 : 0;
          src = wqkv + (size_t)(part * 4 + hh) * CHUNK_BYTES;      // packed rows: [q heads 0..3 | k heads | v heads], 48 rows per chunk
        } else {
          src = wproj + (size_t)(n - NQKV) * CHUNK_BYTES;
        }
        issue_chunk(src, smraw + (n & 1) * CHUNK_BYTES, wave, lane);
      }
      const char* wbuf = smraw + (i & 1) * CHUNK_BYTES;
      if (i < NQKV) {
        const int h = MODE0 ? i / 3 : i, part = MODE0 ? i % 3 : 0;
        f32x4 acc[TPC];
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = *reinterpret_cast<const f32x4*>(a.bqkv + part * BC + h * BD + t * 16 + 4 * q4);
        if (KIND == 1 && part > 0) chunk_gemm(wbuf, pf, acc, lane);
        else chunk_gemm(wbuf, xf, acc, lane);
        bf16* dst = ws + (part == 0 ? W_Q : (part == 1 ? W_KT : W_VT));
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
          bad |= nan4(acc[t]);
          row4_lds(dst + col * LDD + t * 16 + 4 * q4, acc[t]);
        }
        if (part == (MODE0 ? 2 : 0)) {
          // ================= attention core of head h (schedule of attn3_bf16.hip, operands already in LDS) =================
          wave_sync();
          if (MODE0) {
            // Kf^T[d][j] = sum_l kt[l][d] E_k[l][j]   (Linformer, HQAViT_CIFAR100.py:332-352)
#pragma unroll
            for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
              for (int dt = 0; dt < DT; ++dt) {
                f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
                ak = mma16(trfrag(ws + W_KT, LDD, 0, dt * 16), trfrag(sek, LDE, 0, jt * 16), ak);
                av = mma16(trfrag(ws + W_VT, LDD, 0, dt * 16), trfrag(sev, LDE, 0, jt * 16), av);
                row4_lds(ws + W_KF + (jt * 16 + col) * LDD + dt * 16 + 4 * q4, ak);
                row4_lds(ws + W_VF + (jt * 16 + col) * LDD + dt * 16 + 4 * q4, av);
              }
            wave_sync();
          }
          // S^T[key][query], softmax over keys on registers
          f32x4 sc[NKT];
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt) {
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
              acc2 = mma16(nt < KT0 ? rowfrag(ws + W_KF, LDD, nt * 16, dt * 16) : rowfrag(sbk, LDB, 0, h * BD + dt * 16),
                           rowfrag(ws + W_Q, LDD, 0, dt * 16), acc2);
            sc[nt] = acc2;
          }
          {
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool ok = nt * 16 + 4 * q4 + r < NK;
                sc[nt][r] = ok ? sc[nt][r] * scale : -INFINITY;
                mx = fmaxf(mx, sc[nt][r]);
              }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float e = __expf(sc[nt][r] - mx); sc[nt][r] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            const uint32_t pkey = drop.on ? attn_drop_pkey(drop, img * BH + h) : 0u;
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float pv = sc[nt][r] * inv;
                if (drop.on) pv *= attn_drop_factor(drop, pkey, col, nt * 16 + 4 * q4 + r);
                sc[nt][r] = pv;
              }
              row4_lds(ws + W_P + col * LDK + nt * 16 + 4 * q4, sc[nt]);
            }
          }
          wave_sync();
          // O^T[d][query] = sum_key Vf[key][d] P[query][key]  ->  columns h*48 .. of the O tile
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt)
              acc2 = mma16(nt < KT0 ? trfrag(ws + W_VF, LDD, nt * 16, dt * 16) : trfrag(sbv, LDB, 0, h * BD + dt * 16),
                           rowfrag(ws + W_P, LDK, 0, nt * 16), acc2);
            bad |= nan4(acc2);
            row4_lds(ws + W_O + col * LDO + h * BD + dt * 16 + 4 * q4, acc2);
          }
          wave_sync();
        }
      } else {
        // ================= proj: out = dropout(O . Wproj^T + b), 48 output columns per chunk =================
        const int pc = i - NQKV;
        if (pc == 0) {
#pragma unroll
          for (int s = 0; s < KST; ++s) of[s] = *reinterpret_cast<const bf16x8*>(ws + W_O + col * LDO + 32 * s + 8 * q4);
          if (a.o_save && valid) {                         // attention output rows (operand of backward's dW_proj): 16-byte stores
            bf16* osv = reinterpret_cast<bf16*>(a.o_save);
#pragma unroll
            for (int s = 0; s < KST; ++s) *reinterpret_cast<bf16x8*>(osv + ((size_t)img * BT + col) * a.ldo + 32 * s + 8 * q4) = of[s];
          }
          wave_sync();                                     // the O tile is in registers: its LDS image now collects the output rows
        }
        f32x4 acc[TPC];
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = *reinterpret_cast<const f32x4*>(a.bproj + pc * 48 + t * 16 + 4 * q4);
        chunk_gemm(wbuf, of, acc, lane);
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
          if (pdrop) {
            const uint32_t base = (uint32_t)(img * BT + col) * (uint32_t)BC + (uint32_t)(pc * 48 + t * 16 + 4 * q4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] *= drop_factor(pkey_proj, base + r, a.proj_drop_p, pinv);
          }
          row4_lds(ws + W_O + col * LDO + pc * 48 + t * 16 + 4 * q4, acc[t]);
        }
      }
    }
    // ---------------- output rows: LDS tile -> global, 16-byte pieces (24 per 384-byte row) ----------------
    wave_sync();
    if (valid) {
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int p = lane + 64 * it, row = p / 24, c8 = p - row * 24;
        *reinterpret_cast<bf16x8*>(og + ((size_t)img * BT + row) * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(ws + W_O + row * LDO + 8 * c8);
      }
    }
    wave_sync();
  }
  if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
}

// efficient_attention's NaN rule (HQAViT_CIFAR100.py:356-357, :394-395) behind the fused branch: a NaN anywhere in q / k / v
// or in the attention output zeroes the WHOLE attention output, so the branch returns dropout(proj(0)) = dropout(bias) rows.
// The last workgroup to have read the flag resets it (flag[1] = arrival ticket), as qavit_nan_guard does.
__global__ __launch_bounds__(256) void branch_nan_fix_kernel(bf16* out, int64_t ldo, int rows, int C, const float* bias, float p, int site,
                                                             const int64_t* rng, int* flag) {
  __shared__ int f_s;
  if (threadIdx.x == 0) {
    f_s = *reinterpret_cast<volatile int*>(flag);
    __threadfence();
    if (atomicAdd(flag + 1, 1) == (int)gridDim.x - 1) { flag[0] = 0; flag[1] = 0; }
  }
  __syncthreads();
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
}

int branch_validate(const qavit_branch_args* a) {
  if (!a) return set_error(QAVIT_EINVAL, "branch: null args");
  if (a->kind < 0 || a->kind > 2) return set_error(QAVIT_EINVAL, "branch: kind must be 0 (SWA), 1 (MSDA) or 2 (cross)");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "branch: the fused branch kernels are bf16 only (fp32 runs the unfused chain)");
  if (a->T != BT || a->C != BC || a->H != BH || a->D != BD || a->S != 16)
    return set_error(QAVIT_EINVAL, "branch: built for 16 tokens x 192 channels, 4 heads of 48, 16 bank rows");
  if (a->B <= 0 || !a->x || !a->out || !a->wqkv_frag || !a->wproj_frag || !a->bqkv || !a->bproj || !a->sh_k || !a->sh_v)
    return set_error(QAVIT_EINVAL, "branch: null operand");
  if (a->kind != 2 && (a->KC != 32 || a->L <= 0 || a->L > 16 || !a->E_k || !a->E_v))
    return set_error(QAVIT_EINVAL, "branch: SWA / MSDA need Linformer matrices with KC = 32 and 1 <= L <= 16");
  if (a->kind == 1 && (!a->pool_idx || a->pool_stride <= 0)) return set_error(QAVIT_EINVAL, "branch: MSDA needs the landmark index table");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al16(a->x) || !al16(a->out) || !al16(a->wqkv_frag) || !al16(a->wproj_frag) || !al16(a->bqkv) || !al16(a->bproj) || !al16(a->sh_k) || !al16(a->sh_v) ||
      (a->o_save && !al16(a->o_save)) || a->ldx % 8 || a->ldo % 8)
    return set_error(QAVIT_EINVAL, "branch: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements");
  if (a->kind != 2 && (!al16(a->E_k) || !al16(a->E_v))) return set_error(QAVIT_EINVAL, "branch: Linformer matrices must be 16-byte aligned");
  return QAVIT_OK;
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_branch_supported(int kind, int T, int C, int H, int D, int KC, int S, int L) {
  if (kind < 0 || kind > 2) return 0;
  if (T != BT || C != BC || H != BH || D != BD || S != 16) return 0;
  if (kind != 2 && (KC != 32 || L <= 0 || L > 16)) return 0;
  return 1;
}

extern "C" int qavit_branch_fwd(const qavit_branch_args* a, void* stream) {
  int rc = branch_validate(a);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ntiles = (a->B + 3) / 4;
  const int grid = ntiles < 1024 ? ntiles : 1024;
  static bool attr_done[3] = {false, false, false};
#define QV_BRANCH_LAUNCH(K)                                                                                                             \
  do {                                                                                                                                   \
    if (!attr_done[K]) {                                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(branch_fwd_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_TOTAL); \
      attr_done[K] = true;                                                                                                               \
    }                                                                                                                                    \
    hipLaunchKernelGGL((branch_fwd_kernel<K>), dim3(grid), dim3(256), SM_TOTAL, st, *a);                                                \
  } while (0)
  if (a->kind == 0) QV_BRANCH_LAUNCH(0);
  else if (a->kind == 1) QV_BRANCH_LAUNCH(1);
  else QV_BRANCH_LAUNCH(2);
#undef QV_BRANCH_LAUNCH
  if (a->nan_flag) {
    const int64_t n = (int64_t)a->B * BT * BC;
    int nb = (int)((n + 2047) / 2048);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(branch_nan_fix_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<bf16*>(a->out), a->ldo, a->B * BT, BC, a->bproj,
                       a->proj_drop_p, a->proj_drop_site, a->rng, a->nan_flag);
  }
  return check_launch("branch_fwd");
}
