// MFMA GEMMs for the QA-ViT hot path on gfx950.
//
//   qavit_gemm_nt : C[M,N] = epilogue( prologue(A)[M,K] . B[N,K]^T + bias )     (forward linears, dX)
//   qavit_gemm_tn : C[N,K] += A[M,N]^T . B[M,K]  (+ column sums of A)            (dW, db)
//
// Shapes on this path are "tall and skinny": M = batch*tokens (16k..262k rows), K,N <= 1024 and mostly
// 192.  gemm_nt therefore makes a workgroup own BM full rows of A: the rows are staged ONCE into LDS
// (where the fused LayerNorm prologue / backward-epilogue transform runs on them) and reused for every
// 64-column tile of the output, while the weight tile streams through LDS in K-chunks.
// bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "gemm_shared.h"
#include <stdlib.h>

namespace qv {

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int FK = 32;  // k covered by one 16-byte fragment per lane
  typedef bf16x8 frag;
  static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int FK = 16;
  typedef f32x4 frag;
  // lane (r = l&15, q = l>>4) holds k = 4q+s, s = 0..3, of a 16-wide k block; step s multiplies the k's
  // {s, 4+s, 8+s, 12+s}: A and B use the same permutation, so the sum over s covers all 16.
  static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  }
};

constexpr int GEMM_THREADS = 256;
constexpr int BN = 64;   // output columns per tile
constexpr int CS_LD = 68;  // fp32 epilogue scratch row stride

template <typename T> __host__ __device__ constexpr int kc_elems() { return sizeof(T) == 2 ? 128 : 64; }

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

template <typename T>
__device__ __forceinline__ void store_vec_zero(T* p) {
  typename Vec<T>::type z;
#pragma unroll
  for (int i = 0; i < Vec<T>::N; ++i) z[i] = from_f<T>(0.f);
  *reinterpret_cast<typename Vec<T>::type*>(p) = z;
}

// ------------------------------------------------------------------------------------------------
// gemm_nt
//
// A workgroup owns a BN-column slice of the output: its weight slice B[n0:n0+BN, 0:K] is staged into LDS ONCE
// and stays resident while the workgroup walks row tiles of BM = 64 rows (blockIdx.y strides the tiles).  The
// activation rows stream through LDS in K-chunks of <= KC elements with a register prefetch of the next chunk
// issued before the MFMAs of the current one, so L2/HBM latency hides under the matrix work.  4 waves form a
// 2 (M) x 2 (N) grid; each wave accumulates 2 x (BN/32) 16x16 tiles.  The fused LayerNorm prologue runs on the
// staged rows in LDS (K <= KC: whole rows resident), the backward transform on the prefetched registers.
// ------------------------------------------------------------------------------------------------
constexpr int BM = 64;

template <typename T> __host__ __device__ constexpr int kca_elems() { return sizeof(T) == 2 ? 256 : 128; }

// AMODE: 0 plain rows, 1 fused LayerNorm, 2 backward transform.  EPI: 0 = bias only, 1 = full epilogue.
template <typename T, int BNT, int AMODE, int EPI, int KC>
__device__ __forceinline__ void gemm_nt_body(const qavit_gemm_args& g, int n_tiles_m) {
  using M_ = Mma<T>;
  constexpr int VN = Vec<T>::N;
  constexpr int FK = M_::FK;
  constexpr int NT = BNT / 32;                       // 16-col tiles per wave
  constexpr int PRE = BM * KC / VN / GEMM_THREADS;   // prefetch vectors per thread
  constexpr int CLD = BNT + 4;                       // epilogue scratch row stride (floats)
  typedef typename Vec<T>::type vec_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Kp = (g.K + FK - 1) / FK * FK;
  const int ldb_s = Kp + VN;
  const int lda_s = ((Kp < KC) ? Kp : KC) + VN;
  T* Bs = reinterpret_cast<T*>(smem);
  T* As = reinterpret_cast<T*>(smem + (size_t)BNT * ldb_s * sizeof(T));
  float* Cs = reinterpret_cast<float*>(smem + (size_t)BNT * ldb_s * sizeof(T) + (size_t)BM * lda_s * sizeof(T));
  float* Gs = Cs + BM * (BNT + 4);                   // [2][Kp] LayerNorm gamma / beta (AMODE 1 only)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * BNT;
  const bool first_slice = blockIdx.x == 0;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  T* C = reinterpret_cast<T*>(g.C);
  const T* Zin = reinterpret_cast<const T*>(g.a_Z);
  T* Aout = reinterpret_cast<T*>(g.a_out);
  const T* Rr = reinterpret_cast<const T*>(g.R);
  T* Zo = reinterpret_cast<T*>(g.Z);
  constexpr bool bwd = AMODE == 2;
  const bool use_z = bwd && Zin && g.a_act;

  uint32_t key_adrop = 0, key_adp = 0, key_drop = 0, key_dp = 0;
  if ((AMODE == 2 || EPI == 1) && g.rng) {
    key_adrop = rng_key(g.rng, g.a_drop_site);
    key_adp = rng_key(g.rng, g.a_dp_site);
    key_drop = rng_key(g.rng, g.drop_site);
    key_dp = rng_key(g.rng, g.dp_site);
  }
  const float a_inv_keep = g.a_drop_p > 0.f ? 1.f / (1.f - g.a_drop_p) : 1.f;
  const float a_dp_inv = g.a_dp_p > 0.f ? 1.f / (1.f - g.a_dp_p) : 1.f;
  const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const float dp_inv = g.dp_p > 0.f ? 1.f / (1.f - g.dp_p) : 1.f;

  const bool a_vec = (g.K % VN == 0) && (g.lda % VN == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                     (!use_z || (g.a_ldz % VN == 0 && (reinterpret_cast<uintptr_t>(Zin) & 15) == 0)) &&
                     (!(bwd && Aout) || (g.a_ldo % VN == 0 && (reinterpret_cast<uintptr_t>(Aout) & 15) == 0));

  if (AMODE == 1) {
    for (int k = tid; k < Kp; k += GEMM_THREADS) {
      Gs[k] = (k < g.K) ? g.ln_gamma[k] : 0.f;
      Gs[Kp + k] = (k < g.K) ? g.ln_beta[k] : 0.f;
    }
    __syncthreads();
  }
  const int n_chunks = (Kp + KC - 1) / KC;
  vec_t pre[PRE], prez[PRE];

  // issue the global loads of (tile, chunk) into registers
  auto prefetch = [&](int tile, int chunk) {
    const int k0 = chunk * KC;
    const int kc = (Kp - k0 < KC) ? (Kp - k0) : KC;
    const int kv = kc / VN;
    const int m0 = tile * BM;
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      const int idx = tid + i * GEMM_THREADS;
      if (idx >= BM * kv) break;
      const int r = idx / kv, v = idx - r * kv;
      const int m = m0 + r, k = k0 + v * VN;
      vec_t x, z;
#pragma unroll
      for (int j = 0; j < VN; ++j) { x[j] = from_f<T>(0.f); z[j] = from_f<T>(0.f); }
      if (m < g.M && k < g.K) {
        if (a_vec) {
          x = *reinterpret_cast<const vec_t*>(A + (size_t)m * g.lda + k);
          if (use_z) z = *reinterpret_cast<const vec_t*>(Zin + (size_t)m * g.a_ldz + k);
        } else {
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            if (k + j < g.K) {
              x[j] = A[(size_t)m * g.lda + k + j];
              if (use_z) z[j] = Zin[(size_t)m * g.a_ldz + k + j];
            }
          }
        }
      }
      pre[i] = x;
      prez[i] = z;
    }
  };

  // registers -> LDS (applying the backward transform; n-slice 0 also writes dZ back)
  auto commit = [&](int tile, int chunk) {
    const int k0 = chunk * KC;
    const int kc = (Kp - k0 < KC) ? (Kp - k0) : KC;
    const int kv = kc / VN;
    const int m0 = tile * BM;
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      const int idx = tid + i * GEMM_THREADS;
      if (idx >= BM * kv) break;
      const int r = idx / kv, v = idx - r * kv;
      const int m = m0 + r, k = k0 + v * VN;
      vec_t o = pre[i];
      if (AMODE == 1 && m < g.M && k < g.K) {
        // LayerNorm prologue: statistics come from qavit_row_stats; normalise while the row goes to LDS
        const float mu = g.ln_mean[m], rs = g.ln_rstd[m];
#pragma unroll
        for (int j = 0; j < VN; j += 4) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(Gs + k + j);
          const f32x4 be = *reinterpret_cast<const f32x4*>(Gs + Kp + k + j);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) o[j + jj] = from_f<T>((to_f<T>(o[j + jj]) - mu) * rs * ga[jj] + be[jj]);
        }
      }
      if (bwd && m < g.M && k < g.K) {
        float rowf = g.a_scale;
        if (g.a_dp_p > 0.f) rowf *= drop_factor(key_adp, (uint32_t)(m / g.a_dp_rows), g.a_dp_p, a_dp_inv);
        float f[VN];
#pragma unroll
        for (int j = 0; j < VN; ++j) f[j] = to_f<T>(o[j]) * rowf;
        if (g.a_drop_p > 0.f) {
#pragma unroll
          for (int j = 0; j < VN; ++j) f[j] *= drop_factor(key_adrop, (uint32_t)m * (uint32_t)g.K + (uint32_t)(k + j), g.a_drop_p, a_inv_keep);
        }
        if (use_z) {
#pragma unroll
          for (int j = 0; j < VN; ++j) f[j] *= gelu_grad_f(to_f<T>(prez[i][j]));
        }
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = from_f<T>(f[j]);
        if (Aout && first_slice) {
          if (a_vec) *reinterpret_cast<vec_t*>(Aout + (size_t)m * g.a_ldo + k) = o;
          else {
#pragma unroll
            for (int j = 0; j < VN; ++j) if (k + j < g.K) Aout[(size_t)m * g.a_ldo + k + j] = o[j];
          }
        }
      }
      *reinterpret_cast<vec_t*>(As + (size_t)r * lda_s + v * VN) = o;
    }
  };

  f32x4 acc[2][NT];
  int tile = blockIdx.y, chunk = 0;
  if (tile < n_tiles_m) prefetch(tile, 0);           // first row tile's loads fly while the weight slice is staged
  // ---------------- resident weight slice ----------------
  {
    const int kv = Kp / VN;
    const bool b_vec = (g.K % VN == 0) && (g.ldb % VN == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    for (int idx = tid; idx < BNT * kv; idx += GEMM_THREADS) {
      const int r = idx / kv, v = idx - r * kv;
      const int n = n0 + r, k = v * VN;
      T* dst = Bs + (size_t)r * ldb_s + k;
      if (n >= g.N || k >= g.K) { store_vec_zero<T>(dst); continue; }
      if (b_vec) {
        *reinterpret_cast<vec_t*>(dst) = *reinterpret_cast<const vec_t*>(B + (size_t)n * g.ldb + k);
      } else {
        vec_t o;
#pragma unroll
        for (int i = 0; i < VN; ++i) o[i] = (k + i < g.K) ? B[(size_t)n * g.ldb + k + i] : from_f<T>(0.f);
        *reinterpret_cast<vec_t*>(dst) = o;
      }
    }
  }


  while (tile < n_tiles_m) {
    __syncthreads();                                   // previous MFMAs done with As, previous epilogue done with Cs
    commit(tile, chunk);
    __syncthreads();
    const int k0 = chunk * KC;
    const int kc = (Kp - k0 < KC) ? (Kp - k0) : KC;
    const int m0 = tile * BM;
    // next (tile, chunk) in the stream: its loads fly while this chunk's MFMAs run
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == n_chunks) { nchunk = 0; ntile = tile + gridDim.y; }
    if (ntile < n_tiles_m) prefetch(ntile, nchunk);

    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nf = kc / FK;
    for (int kf = 0; kf < nf; ++kf) {
      typename M_::frag af[2], bfr[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const typename M_::frag*>(As + (size_t)((wm * 2 + i) * 16 + fr) * lda_s + kf * FK + fq * VN);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bfr[j] = *reinterpret_cast<const typename M_::frag*>(Bs + (size_t)((wn * NT + j) * 16 + fr) * ldb_s + k0 + kf * FK + fq * VN);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) M_::mma(af[i], bfr[j], acc[i][j]);
    }

    if (chunk == n_chunks - 1) {
      // accumulators -> Cs (col = lane&15, row = 4*(lane>>4)+reg), then row-contiguous epilogue
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[((wm * 2 + i) * 16 + fq * 4 + r) * CLD + (wn * NT + j) * 16 + fr] = acc[i][j][r];
      __syncthreads();
      constexpr int CG = BNT / 16;                     // 16-column groups per row
      for (int idx = tid; idx < BM * CG; idx += GEMM_THREADS) {
        const int r = idx / CG, cg = idx - r * CG;
        const int m = m0 + r;
        const int n = n0 + cg * 16;
        if (m >= g.M || n >= g.N) continue;
        const int nv = (g.N - n < 16) ? (g.N - n) : 16;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(Cs + r * CLD + cg * 16 + i);
          v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
        const bool full = (nv == 16) && (n % VN == 0);
        if (g.bias) {
          if (nv == 16 && ((reinterpret_cast<uintptr_t>(g.bias + n) & 15) == 0)) {
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(g.bias + n + i);
              v[i] += t[0]; v[i + 1] += t[1]; v[i + 2] += t[2]; v[i + 3] += t[3];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i < nv) v[i] += g.bias[n + i];
          }
        }
        float zv[16];
        if (EPI == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) zv[i] = v[i];
          if (g.act == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = gelu_f(v[i]);
          }
          if (g.drop_p > 0.f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= drop_factor(key_drop, (uint32_t)m * (uint32_t)g.N + (uint32_t)(n + i), g.drop_p, inv_keep);
          }
          float rowf = g.scale;
          if (g.dp_p > 0.f) rowf *= drop_factor(key_dp, (uint32_t)(m / g.dp_rows), g.dp_p, dp_inv);
          if (rowf != 1.f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= rowf;
          }
          if (Rr) {
            if (full && (g.ldr % VN == 0) && ((reinterpret_cast<uintptr_t>(Rr) & 15) == 0)) {
#pragma unroll
              for (int i = 0; i < 16; i += VN) {
                const vec_t t = *reinterpret_cast<const vec_t*>(Rr + (size_t)m * g.ldr + n + i);
#pragma unroll
                for (int j = 0; j < VN; ++j) v[i + j] += to_f<T>(t[j]);
              }
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) if (i < nv) v[i] += to_f<T>(Rr[(size_t)m * g.ldr + n + i]);
            }
          }
        }
        T* crow = C + (size_t)m * g.ldc + n;
        if (full && (g.ldc % VN == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0)) {
#pragma unroll
          for (int i = 0; i < 16; i += VN) {
            vec_t o;
#pragma unroll
            for (int j = 0; j < VN; ++j) o[j] = from_f<T>(v[i + j]);
            *reinterpret_cast<vec_t*>(crow + i) = o;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) if (i < nv) crow[i] = from_f<T>(v[i]);
        }
        if (EPI == 1 && Zo) {
          T* zrow = Zo + (size_t)m * g.ldz + n;
          if (full && (g.ldz % VN == 0) && ((reinterpret_cast<uintptr_t>(Zo) & 15) == 0)) {
#pragma unroll
            for (int i = 0; i < 16; i += VN) {
              vec_t o;
#pragma unroll
              for (int j = 0; j < VN; ++j) o[j] = from_f<T>(zv[i + j]);
              *reinterpret_cast<vec_t*>(zrow + i) = o;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i < nv) zrow[i] = from_f<T>(zv[i]);
          }
        }
      }
    }
    tile = ntile;
    chunk = nchunk;
  }
}

template <typename T, int BNT, int AMODE, int EPI, int KC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(qavit_gemm_args g, int n_tiles_m) {
  gemm_nt_body<T, BNT, AMODE, EPI, KC>(g, n_tiles_m);
}

// up to 4 problems of ONE shape and mode in a grid (blockIdx.z = problem): the four compress GEMMs of a block and their
// input-gradient GEMMs are independent and each fills the chip for only a few microseconds
constexpr int NT_GROUP = 4;
struct GemmGroup { qavit_gemm_args p[NT_GROUP]; };
template <typename T, int BNT, int AMODE, int EPI, int KC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_group_kernel(GemmGroup G, int n_tiles_m) {
  gemm_nt_body<T, BNT, AMODE, EPI, KC>(G.p[blockIdx.z], n_tiles_m);
}

template <typename T, int BNT, int AMODE, int EPI, int KC>
static int launch_gemm_nt3(const qavit_gemm_args& g, hipStream_t st, const qavit_gemm_args* grp = nullptr, int ng = 1) {
  constexpr int VN = Vec<T>::N;
  constexpr int FK = Mma<T>::FK;
  const int Kp = round_up(g.K, FK);
  const size_t b_bytes = (size_t)BNT * (Kp + VN) * sizeof(T);
  const size_t a_bytes = (size_t)BM * ((Kp < KC ? Kp : KC) + VN) * sizeof(T);
  const size_t c_bytes = (size_t)BM * (BNT + 4) * sizeof(float) + (AMODE == 1 ? (size_t)2 * Kp * sizeof(float) : 0);
  const size_t smem = b_bytes + a_bytes + c_bytes;
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "gemm_nt: K too large for the resident weight slice");
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, BNT, AMODE, EPI, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int n_slices = (g.N + BNT - 1) / BNT;
  const int n_tiles_m = (g.M + BM - 1) / BM;
  static int wg_target = -1;
  if (wg_target < 0) { const char* e = getenv("QAVIT_GEMM_WGS"); wg_target = e ? atoi(e) : 768; }
  int gy = (wg_target + n_slices - 1) / n_slices;    // ~3 workgroups per CU across the chip
  if (gy > n_tiles_m) gy = n_tiles_m;
  if (gy < 1) gy = 1;
  if (grp) {
    static bool attr_g = false;
    if (!attr_g) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_group_kernel<T, BNT, AMODE, EPI, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_g = true;
    }
    GemmGroup G;
    for (int i = 0; i < ng; ++i) G.p[i] = grp[i];
    int gyg = (wg_target + n_slices * ng - 1) / (n_slices * ng);
    if (gyg > n_tiles_m) gyg = n_tiles_m;
    if (gyg < 1) gyg = 1;
    hipLaunchKernelGGL((gemm_nt_group_kernel<T, BNT, AMODE, EPI, KC>), dim3(n_slices, gyg, ng), dim3(GEMM_THREADS), smem, st, G, n_tiles_m);
    return check_launch("gemm_nt(grouped)");
  }
  hipLaunchKernelGGL((gemm_nt_kernel<T, BNT, AMODE, EPI, KC>), dim3(n_slices, gy), dim3(GEMM_THREADS), smem, st, g, n_tiles_m);
  return check_launch("gemm_nt");
}

// fp32 with a long K: the resident fp32 weight slice is large, so the row tile streams in 64-wide chunks
template <typename T, int BNT, int AMODE, int EPI>
static int launch_gemm_nt2(const qavit_gemm_args& g, hipStream_t st, const qavit_gemm_args* grp = nullptr, int ng = 1) {
  if (sizeof(T) == 4 && g.K > 256) return launch_gemm_nt3<T, BNT, AMODE, EPI, 64>(g, st, grp, ng);
  return launch_gemm_nt3<T, BNT, AMODE, EPI, 256>(g, st, grp, ng);
}

template <typename T, int BNT>
static int launch_gemm_nt(const qavit_gemm_args& g, hipStream_t st, const qavit_gemm_args* grp = nullptr, int ng = 1) {
  const bool full = g.Z || g.act || g.drop_p > 0.f || g.dp_p > 0.f || g.R || g.scale != 1.f;
  if (g.a_mode == 1) return full ? launch_gemm_nt2<T, BNT, 1, 1>(g, st, grp, ng) : launch_gemm_nt2<T, BNT, 1, 0>(g, st, grp, ng);
  if (g.a_mode == 2) return full ? launch_gemm_nt2<T, BNT, 2, 1>(g, st, grp, ng) : launch_gemm_nt2<T, BNT, 2, 0>(g, st, grp, ng);
  return full ? launch_gemm_nt2<T, BNT, 0, 1>(g, st, grp, ng) : launch_gemm_nt2<T, BNT, 0, 0>(g, st, grp, ng);
}

template <typename T>
static int dispatch_gemm_nt(const qavit_gemm_args& g, hipStream_t st, const qavit_gemm_args* grp = nullptr, int ng = 1) {
  constexpr int VN = Vec<T>::N;
  constexpr int FK = Mma<T>::FK;
  const int Kp = round_up(g.K, FK);
  const size_t row_bytes = (size_t)(Kp + VN) * sizeof(T);
  // slice width: weights <= ~28 KB so that two workgroups share a CU (slice + row tile + epilogue scratch < 80 KB)
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("QAVIT_GEMM_BN"); forced = e ? atoi(e) : 0; }
  int bn = forced > 0 ? forced : 64;
  const size_t cap = (forced > 0 ? 100 : (sizeof(T) == 4 ? 52 : 28)) * 1024;
  while (bn > 32 && (size_t)bn * row_bytes > cap) bn >>= 1;
  if (bn > 32 && g.N <= bn / 2) bn = (g.N <= 32) ? 32 : 64;
  if ((size_t)bn * row_bytes > 132 * 1024) return set_error(QAVIT_EINVAL, "gemm_nt: K too large");
  switch (bn) {
    case 128: return launch_gemm_nt<T, 128>(g, st, grp, ng);
    case 64: return launch_gemm_nt<T, 64>(g, st, grp, ng);
    default: return launch_gemm_nt<T, 32>(g, st, grp, ng);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt, M <= 16 (the bank projections: Linear applied to the 16 global-token rows): one wave per 16 output
// columns, operands straight from global memory into MFMA fragments -- every load of the tile is issued before the
// first MFMA, so the whole GEMM is ONE memory round trip (the resident-slice kernel stages a 49 KB fp32 weight slice
// through LDS first: 12 us for 1.2 MFLOP).  Plain epilogue (bias) only.
// ------------------------------------------------------------------------------------------------
template <typename T, int KB>   // KB = K / FK fragment blocks (compile-time so the loads unroll)
__device__ __forceinline__ void gemm_nt_skinny_body(const qavit_gemm_args& g, int bx) {
  using M_ = Mma<T>;
  constexpr int FK = M_::FK, VN = Vec<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = (bx * 4 + wave) * 16;
  if (n0 >= g.N) return;                                 // whole wave
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  T* C = reinterpret_cast<T*>(g.C);
  const T* R = reinterpret_cast<const T*>(g.R);
  const int am = fr < g.M ? fr : g.M - 1;                // clamped rows: loads stay in bounds, extra rows/cols are not stored
  const int bn = n0 + fr < g.N ? n0 + fr : g.N - 1;
  typename M_::frag af[KB], bf_[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    af[kb] = *reinterpret_cast<const typename M_::frag*>(A + (size_t)am * g.lda + kb * FK + fq * VN);
    bf_[kb] = *reinterpret_cast<const typename M_::frag*>(B + (size_t)bn * g.ldb + kb * FK + fq * VN);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) M_::mma(af[kb], bf_[kb], acc);
  const int n = n0 + fr;
  if (n < g.N) {
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = fq * 4 + r;
      if (m < g.M) {
        float v = acc[r] + bv;
        if (R) v += to_f<T>(R[(size_t)m * g.ldr + n]);   // residual (may alias C: same element, same thread)
        C[(size_t)m * g.ldc + n] = from_f<T>(v);
      }
    }
  }
}

template <typename T, int KB>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(qavit_gemm_args g) { gemm_nt_skinny_body<T, KB>(g, blockIdx.x); }

// up to 4 same-shape skinny problems in one grid (blockIdx.y = problem): the K and V projections of the bank, forward and backward
struct SkinnyGroup { qavit_gemm_args p[4]; };
template <typename T, int KB>
__global__ __launch_bounds__(256) void gemm_nt_skinny_group_kernel(SkinnyGroup G) { gemm_nt_skinny_body<T, KB>(G.p[blockIdx.y], blockIdx.x); }

template <typename T>
static bool skinny_ok(const qavit_gemm_args& g) {
  constexpr int FK = Mma<T>::FK, VN = Vec<T>::N;
  if (g.M > 16 || g.a_mode != 0 || g.Z || g.act || g.drop_p > 0.f || g.dp_p > 0.f || g.scale != 1.f) return false;
  if (g.K % FK || g.lda % VN || g.ldb % VN || ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15)) return false;
  const int kb = g.K / FK;
  return kb == 1 || kb == 2 || kb == 3 || kb == 4 || kb == 6 || kb == 8 || kb == 12;
}

template <typename T>
static int skinny_try(const qavit_gemm_args& g, hipStream_t st, const qavit_gemm_args* grp = nullptr, int ng = 0) {
  constexpr int FK = Mma<T>::FK;
  static int on = -1;
  if (on < 0) { const char* e = getenv("QAVIT_GEMM_SKINNY"); on = e ? atoi(e) : 1; }
  if (!on) return 0;
  if (!skinny_ok<T>(g)) return 0;
  SkinnyGroup G;
  if (grp) {
    if (ng > 4) return 0;
    for (int i = 0; i < ng; ++i) {
      if (!skinny_ok<T>(grp[i]) || grp[i].K != g.K || grp[i].N != g.N) return 0;
      G.p[i] = grp[i];
    }
  }
  const int kb = g.K / FK;
  const int grid = (g.N + 63) / 64;
#define SKINNY(KB_)                                                                                                   \
  do {                                                                                                                \
    if (grp) hipLaunchKernelGGL((gemm_nt_skinny_group_kernel<T, KB_>), dim3(grid, ng), dim3(256), 0, st, G);          \
    else hipLaunchKernelGGL((gemm_nt_skinny_kernel<T, KB_>), dim3(grid), dim3(256), 0, st, g);                        \
  } while (0)
  switch (kb) {
    case 1: SKINNY(1); break;
    case 2: SKINNY(2); break;
    case 3: SKINNY(3); break;
    case 4: SKINNY(4); break;
    case 6: SKINNY(6); break;
    case 8: SKINNY(8); break;
    case 12: SKINNY(12); break;
    default: return 0;
  }
#undef SKINNY
  const int rc = check_launch("gemm_nt(skinny)");
  return rc == QAVIT_OK ? 1 : rc;
}

// ------------------------------------------------------------------------------------------------
// gemm_tn : C[N,K] += A^T B over a slice of M, one 64x64 output tile per workgroup
// ------------------------------------------------------------------------------------------------
constexpr int TN_MC = 64;   // rows of M per staged chunk

template <typename T>
__device__ __forceinline__ void gemm_tn_body(const qavit_gemm_tn_args& g, T* At, T* Bt, int bx, int by, int bz, int rows_per_split) {
  using M_ = Mma<T>;
  constexpr int VN = Vec<T>::N;
  constexpr int FK = M_::FK;
  constexpr int LDT = TN_MC + VN;                     // LDS row stride (elements); m is the contiguous axis

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = bx * 64, k0 = by * 64;
  const int mbeg = bz * rows_per_split;
  const int mend = (mbeg + rows_per_split < g.M) ? mbeg + rows_per_split : g.M;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  const bool ln = g.ln_mean != nullptr;

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                                    // thread tid < 64: column n0+tid of A
  const int fr = lane & 15, fq = lane >> 4;

  // transposing stage: thread (c = tid & 63, rg = tid >> 6) loads 4 consecutive m for column c and writes
  // them as one packed LDS store -> rows of At/Bt are m-contiguous (what the MFMA fragment wants)
  const int c = tid & 63, rg = tid >> 6;
  for (int mc = mbeg; mc < mend; mc += TN_MC) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < TN_MC / 16; ++it) {
      const int ml = it * 16 + rg * 4;                 // local m of the 4-pack
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mc + ml + i;
        const bool ok = m < mend;
        av[i] = (ok && n0 + c < g.N) ? to_f<T>(A[(size_t)m * g.lda + n0 + c]) : 0.f;
        float b = (ok && k0 + c < g.K) ? to_f<T>(B[(size_t)m * g.ldb + k0 + c]) : 0.f;
        if (ln && ok && k0 + c < g.K) b = (b - g.ln_mean[m]) * g.ln_rstd[m] * g.ln_gamma[k0 + c] + g.ln_beta[k0 + c];
        bv[i] = b;
      }
      if (sizeof(T) == 2) {
        bf16x4 pa, pb;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pa[i] = (bf16)av[i]; pb[i] = (bf16)bv[i]; }
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(At) + c * LDT + ml) = pa;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(Bt) + c * LDT + ml) = pb;
      } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(At) + c * LDT + ml) = f32x4{av[0], av[1], av[2], av[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Bt) + c * LDT + ml) = f32x4{bv[0], bv[1], bv[2], bv[3]};
      }
    }
    __syncthreads();
    if (g.colsum && by == 0 && tid < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int m = 0; m < TN_MC; ++m) s += to_f<T>(At[tid * LDT + m]);
      csum += s;
    }
#pragma unroll
    for (int kf = 0; kf < TN_MC / FK; ++kf) {
      const typename M_::frag af = *reinterpret_cast<const typename M_::frag*>(At + (wave * 16 + fr) * LDT + kf * FK + fq * VN);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const typename M_::frag bfr = *reinterpret_cast<const typename M_::frag*>(Bt + (j * 16 + fr) * LDT + kf * FK + fq * VN);
        M_::mma(af, bfr, acc[j]);
      }
    }
  }
  // acc[j][r]: row n = n0 + wave*16 + 4*fq + r, col k = k0 + j*16 + fr
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave * 16 + fq * 4 + r, k = k0 + j * 16 + fr;
      if (n < g.N && k < g.K) atomic_add_f(g.C + (size_t)n * g.ldc + k, acc[j][r]);
    }
  if (g.colsum && by == 0 && tid < 64 && n0 + tid < g.N) atomic_add_f(g.colsum + n0 + tid, csum);
}

template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(qavit_gemm_tn_args g, int rows_per_split) {
  constexpr int LDT = TN_MC + Vec<T>::N;
  __shared__ __attribute__((aligned(16))) T At[64 * LDT];   // [n][m]
  __shared__ __attribute__((aligned(16))) T Bt[64 * LDT];   // [k][m]
  gemm_tn_body<T>(g, At, Bt, blockIdx.x, blockIdx.y, blockIdx.z, rows_per_split);
}

// ------------------------------------------------------------------------------------------------
// bf16 gemm_tn with hardware-transposed LDS reads.  Both operands are staged ROW-MAJOR ([m][n], [m][k]: 16-byte
// coalesced loads, register-prefetched one chunk ahead) and the m-contiguous MFMA fragments are produced by
// ds_read_b64_tr_b16: per 16-lane group it reads a 4-row x 16-column block and hands lane i column i of the
// four rows -- exactly the k-contiguous fragment of an operand stored with its reduction axis along rows.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int ld, int m0, int c0) {
  // fragment for lane l: rows m0 + 8*(l>>4) + j (j = 0..7), column c0 + (l & 15)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const bf16* a0 = tile + (size_t)(m0 + 8 * g + q) * ld + c0 + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * ld));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

constexpr int TN_LD = 64 + 8;                          // elements; 144-byte rows keep every tr read 8-byte aligned

__device__ __forceinline__ void gemm_tn_bf16_body(const qavit_gemm_tn_args& g, bf16* At, bf16* Bt, int bx, int by, int bz, int rows_per_split) {
  constexpr int LD = TN_LD;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int wa = wave & 1, wb = wave >> 1;
  const int n0 = bx * 64, k0 = by * 64;
  const int mbeg = bz * rows_per_split;
  const int mend = (mbeg + rows_per_split < g.M) ? mbeg + rows_per_split : g.M;
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const bool ln = g.ln_mean != nullptr;
  const bool a_vec = (g.lda % 8 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (n0 + 64 <= g.N);
  const bool b_vec = (g.ldb % 8 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (k0 + 64 <= g.K);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;

  // thread -> two 16-byte vectors per operand: rows r0 and r0 + 32, columns 8*vc .. 8*vc+7
  const int r0 = tid >> 3, vc = tid & 7;
  bf16x8 pa[2], pb[2];
  auto prefetch = [&](int mc) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = mc + r0 + 32 * h;
      bf16x8 va, vb;
#pragma unroll
      for (int j = 0; j < 8; ++j) { va[j] = (bf16)0.f; vb[j] = (bf16)0.f; }
      if (m < mend) {
        if (a_vec) va = *reinterpret_cast<const bf16x8*>(A + (size_t)m * g.lda + n0 + 8 * vc);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n0 + 8 * vc + j < g.N) va[j] = A[(size_t)m * g.lda + n0 + 8 * vc + j];
        }
        if (b_vec) vb = *reinterpret_cast<const bf16x8*>(B + (size_t)m * g.ldb + k0 + 8 * vc);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (k0 + 8 * vc + j < g.K) vb[j] = B[(size_t)m * g.ldb + k0 + 8 * vc + j];
        }
        if (ln) {
          const float mu = g.ln_mean[m], rs = g.ln_rstd[m];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int kk = k0 + 8 * vc + j;
            if (kk < g.K) vb[j] = (bf16)(((float)vb[j] - mu) * rs * g.ln_gamma[kk] + g.ln_beta[kk]);
          }
        }
      }
      pa[h] = va;
      pb[h] = vb;
    }
  };

  if (mbeg < mend) prefetch(mbeg);
  for (int mc = mbeg; mc < mend; mc += TN_MC) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<bf16x8*>(At + (r0 + 32 * h) * LD + 8 * vc) = pa[h];
      *reinterpret_cast<bf16x8*>(Bt + (r0 + 32 * h) * LD + 8 * vc) = pb[h];
    }
    __syncthreads();
    if (mc + TN_MC < mend) prefetch(mc + TN_MC);
    if (g.colsum && by == 0 && tid < 64) {
      float s_ = 0.f;
#pragma unroll 8
      for (int m = 0; m < TN_MC; ++m) s_ += (float)At[m * LD + tid];
      csum += s_;
    }
#pragma unroll
    for (int kf = 0; kf < TN_MC / 32; ++kf) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = tr_frag(At, LD, kf * 32, (wa * 2 + i) * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = tr_frag(Bt, LD, kf * 32, (wb * 2 + j) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  const int lane = tid & 63, fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wa * 2 + i) * 16 + fq * 4 + r, k = k0 + (wb * 2 + j) * 16 + fr;
        if (n < g.N && k < g.K) atomic_add_f(g.C + (size_t)n * g.ldc + k, acc[i][j][r]);
      }
  if (g.colsum && by == 0 && tid < 64 && n0 + tid < g.N) atomic_add_f(g.colsum + n0 + tid, csum);
}

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_bf16_kernel(qavit_gemm_tn_args g, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) bf16 At[TN_MC * TN_LD];   // [m][n]
  __shared__ __attribute__((aligned(16))) bf16 Bt[TN_MC * TN_LD];   // [m][k]
  gemm_tn_bf16_body(g, At, Bt, blockIdx.x, blockIdx.y, blockIdx.z, rows_per_split);
}

// Grouped launch: up to TN_GROUP independent weight-gradient GEMMs in ONE grid.  The dW GEMMs of a backward pass
// are off the critical path and individually too small to fill 256 CUs (latency-bound ~25-50 us each); batching
// a dozen of them keeps every CU busy and removes the per-launch gaps.
constexpr int TN_GROUP = 12;
struct TnGroup {
  int n;
  int wg_start[TN_GROUP + 1];
  int tn[TN_GROUP], tk[TN_GROUP], rows[TN_GROUP];
  qavit_gemm_tn_args p[TN_GROUP];
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_bf16_grouped_kernel(TnGroup G) {
  __shared__ __attribute__((aligned(16))) bf16 At[TN_MC * TN_LD];
  __shared__ __attribute__((aligned(16))) bf16 Bt[TN_MC * TN_LD];
  const int bid = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < TN_GROUP; ++j) if (j < G.n && bid >= G.wg_start[j]) i = j;
  const int local = bid - G.wg_start[i];
  const int tiles = G.tn[i] * G.tk[i];
  const int bz = local / tiles, t = local - bz * tiles;
  const int by = t / G.tn[i], bx = t - by * G.tn[i];
  gemm_tn_bf16_body(G.p[i], At, Bt, bx, by, bz, G.rows[i]);
}

// the fp32 problems of a flush (the M = 16 bank-projection weight gradients of the cross / channel-group branches: 32 per step,
// each a 5 us launch of its own before) share a grid the same way
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_f32_grouped_kernel(TnGroup G) {
  constexpr int LDT = TN_MC + Vec<float>::N;
  __shared__ __attribute__((aligned(16))) float At[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Bt[64 * LDT];
  const int bid = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < TN_GROUP; ++j) if (j < G.n && bid >= G.wg_start[j]) i = j;
  const int local = bid - G.wg_start[i];
  const int tiles = G.tn[i] * G.tk[i];
  const int bz = local / tiles, t = local - bz * tiles;
  const int by = t / G.tn[i], bx = t - by * G.tn[i];
  gemm_tn_body<float>(G.p[i], At, Bt, bx, by, bz, G.rows[i]);
}

static void tn_split_plan(const qavit_gemm_tn_args& g, int budget, int& tn, int& tk, int& splits, int& rows) {
  tn = (g.N + 63) / 64; tk = (g.K + 63) / 64;
  splits = g.splits;
  if (splits <= 0) { splits = budget / (tn * tk); if (splits < 1) splits = 1; }
  int max_splits = (g.M + 127) / 128;
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  rows = (g.M + splits - 1) / splits;
  rows = (rows + TN_MC - 1) / TN_MC * TN_MC;
  splits = (g.M + rows - 1) / rows;
}

template <typename T>
static int launch_gemm_tn(const qavit_gemm_tn_args& g, hipStream_t st) {
  const int tn = (g.N + 63) / 64, tk = (g.K + 63) / 64;
  int splits = g.splits;
  if (splits <= 0) {
    splits = 1024 / (tn * tk);                        // short per-workgroup chains: the kernel is latency-, not atomic-bound
    if (splits < 1) splits = 1;
  }
  int max_splits = (g.M + 127) / 128;                 // at least 128 rows (two staged chunks) per split
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  int rows = (g.M + splits - 1) / splits;
  rows = (rows + TN_MC - 1) / TN_MC * TN_MC;
  splits = (g.M + rows - 1) / rows;
  if (sizeof(T) == 2) hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(tn, tk, splits), dim3(GEMM_THREADS), 0, st, g, rows);
  else hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(tn, tk, splits), dim3(GEMM_THREADS), 0, st, g, rows);
  return check_launch("gemm_tn");
}

}  // namespace qv

extern "C" int qavit_gemm_nt(const qavit_gemm_args* a, void* stream) {
  using namespace qv;
  if (!a || !a->A || !a->B || !a->C) return set_error(QAVIT_EINVAL, "gemm_nt: null operand");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return set_error(QAVIT_EINVAL, "gemm_nt: non-positive dimension");
  if (a->e_x) {
    // LayerNorm-backward epilogue: only the K-loop kernel has it -- every other route is refused, loudly
    if (a->lda < a->K || a->ldb < a->K || a->ldc < a->N) return set_error(QAVIT_EINVAL, "gemm_nt: leading dimension too small");
    if ((a->a_drop_p > 0.f || a->a_dp_p > 0.f) && !a->rng) return set_error(QAVIT_EINVAL, "gemm_nt: dropout requested without rng state");
    if (a->a_dp_p > 0.f && a->a_dp_rows <= 0) return set_error(QAVIT_EINVAL, "gemm_nt: drop-path needs rows-per-sample");
    const int took = gemm_nt_big_lnbwd(*a, reinterpret_cast<hipStream_t>(stream));
    return took == 1 ? QAVIT_OK : (took < 0 ? took : set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue not applicable"));
  }
  if (a->A2) {
    // two-source A: only the K-loop kernel stages A by 64-wide chunks with a per-chunk base -- every other route is refused, loudly
    if (!qavit_gemm_nt_a2_supported(a->dtype, a->M, a->N, a->K, a->a2_k0) || a->a_mode != 0)
      return set_error(QAVIT_EINVAL, "gemm_nt: two-source A needs bf16, a_mode 0, a2_k0 % 64 == 0 and a shape the K-loop kernel takes (qavit_gemm_nt_a2_supported)");
    if (a->lda < a->a2_k0 || a->lda2 < a->K - a->a2_k0 || a->ldb < a->K || a->ldc < a->N) return set_error(QAVIT_EINVAL, "gemm_nt: leading dimension too small");
    if ((a->drop_p > 0.f || a->dp_p > 0.f) && !a->rng) return set_error(QAVIT_EINVAL, "gemm_nt: dropout requested without rng state");
    const int took = gemm_nt_big_try(*a, reinterpret_cast<hipStream_t>(stream));
    if (took < 0) return took;
    if (took == 1) return QAVIT_OK;
    return set_error(QAVIT_EINVAL, "gemm_nt: two-source A: operand alignment / strides the K-loop kernel does not take");
  }
  if (a->lda < a->K || a->ldb < a->K || a->ldc < a->N) return set_error(QAVIT_EINVAL, "gemm_nt: leading dimension too small");
  if (a->a_mode == 1 && (!a->ln_gamma || !a->ln_beta || !a->ln_mean || !a->ln_rstd))
    return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm prologue needs gamma/beta and the row statistics (qavit_row_stats)");
  if ((a->drop_p > 0.f || a->dp_p > 0.f || a->a_drop_p > 0.f || a->a_dp_p > 0.f) && !a->rng)
    return set_error(QAVIT_EINVAL, "gemm_nt: dropout requested without rng state");
  if ((a->dp_p > 0.f && a->dp_rows <= 0) || (a->a_dp_p > 0.f && a->a_dp_rows <= 0))
    return set_error(QAVIT_EINVAL, "gemm_nt: drop-path needs rows-per-sample");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->a_mode == 3) {
    // LayerNorm prologue whose row statistics are OUTPUTS of this call: the K-loop kernel computes them in its prologue;
    // every other route runs the row_stats kernel first and continues as a_mode 1
    if (!a->ln_gamma || !a->ln_beta || !a->ln_mean || !a->ln_rstd) return set_error(QAVIT_EINVAL, "gemm_nt: a_mode 3 needs gamma/beta and statistics buffers");
    if (a->dtype == QAVIT_BF16) {
      const int took = gemm_nt_big_try(*a, st);
      if (took < 0) return took;
      if (took == 1) return QAVIT_OK;
    }
    if (a->lda != a->K) return set_error(QAVIT_EINVAL, "gemm_nt: a_mode 3 needs contiguous A rows on this route");
    const int rc = qavit_row_stats(a->dtype, a->A, a->ln_eps, a->M, a->K, a->ln_mean, a->ln_rstd, stream);
    if (rc) return rc;
    qavit_gemm_args b = *a;
    b.a_mode = 1;
    return qavit_gemm_nt(&b, stream);
  }
  if (a->dtype == QAVIT_F32 || a->dtype == QAVIT_BF16) {
    const int took = a->dtype == QAVIT_F32 ? skinny_try<float>(*a, st) : skinny_try<bf16>(*a, st);
    if (took < 0) return took;
    if (took == 1) return QAVIT_OK;
  }
  if (a->dtype == QAVIT_F32) return dispatch_gemm_nt<float>(*a, st);
  if (a->dtype == QAVIT_BF16) {
    const int took = gemm_nt_big_try(*a, st);
    if (took < 0) return took;
    if (took == 1) return QAVIT_OK;
    return dispatch_gemm_nt<bf16>(*a, st);
  }
  return set_error(QAVIT_EINVAL, "gemm_nt: unknown dtype");
}

extern "C" int qavit_gemm_nt_lnbwd_supported(int dtype, int M, int N, int K, int a_mode) { return qv::gemm_nt_lnbwd_shape_ok(dtype, M, N, K, a_mode) ? 1 : 0; }
extern "C" int qavit_gemm_nt_lnbwd_parts(int M, int N, int K) { return (M > 0 && (N == 128 || N == 192 || N == 256)) ? qv::gemm_nt_lnbwd_parts(M, N, K) : 0; }

extern "C" int qavit_gemm_nt_a2_supported(int dtype, int M, int N, int K, int a2_k0) {
  // the conditions of gemm_nt_big_try (csrc/gemm_big.hip) that do not depend on pointers, plus the chunk-aligned split
  return dtype == QAVIT_BF16 && M >= 1024 && N >= 64 && K >= 96 && (long)N * K >= 64L * 128L && K % 32 == 0 && a2_k0 > 0 && a2_k0 < K && a2_k0 % 64 == 0;
}

extern "C" int qavit_gemm_nt_grouped(const qavit_gemm_args* a, int n, void* stream) {
  using namespace qv;
  if (!a || n <= 0) return set_error(QAVIT_EINVAL, "gemm_nt_grouped: empty group");
  for (int i = 0; i < n; ++i)
    if (a[i].A2) { if (n == 1) return qavit_gemm_nt(a, stream); return set_error(QAVIT_EINVAL, "gemm_nt_grouped: two-source A problems are launched singly"); }
  // one grid when the problems share shape, dtype, prologue and epilogue kind and take the resident-slice kernel;
  // otherwise one launch each
  auto kind = [](const qavit_gemm_args& g) { return (g.Z || g.act || g.drop_p > 0.f || g.dp_p > 0.f || g.R || g.scale != 1.f) ? 1 : 0; };
  bool same = n <= NT_GROUP;
  for (int i = 1; i < n && same; ++i)
    same = a[i].dtype == a[0].dtype && a[i].M == a[0].M && a[i].N == a[0].N && a[i].K == a[0].K && a[i].a_mode == a[0].a_mode && kind(a[i]) == kind(a[0]);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n >= 2 && n <= 4 && a[0].M <= 16 && (a[0].dtype == QAVIT_F32 || a[0].dtype == QAVIT_BF16)) {
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = a[i].A && a[i].B && a[i].C && a[i].M > 0 && a[i].dtype == a[0].dtype && a[i].lda >= a[i].K && a[i].ldb >= a[i].K && a[i].ldc >= a[i].N;
    if (ok) {
      const int took = a[0].dtype == QAVIT_F32 ? skinny_try<float>(a[0], st, a, n) : skinny_try<bf16>(a[0], st, a, n);
      if (took < 0) return took;
      if (took == 1) return QAVIT_OK;
    }
  }
  const bool big = a[0].dtype == QAVIT_BF16 && a[0].N >= 64 && a[0].K >= 96 && a[0].M >= 1024;     // K-loop kernel territory
  if (!same || n == 1 || big || a[0].M <= 16) {
    for (int i = 0; i < n; ++i) { const int rc = qavit_gemm_nt(a + i, stream); if (rc) return rc; }
    return QAVIT_OK;
  }
  for (int i = 0; i < n; ++i) {
    const qavit_gemm_args& g = a[i];
    if (!g.A || !g.B || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return set_error(QAVIT_EINVAL, "gemm_nt_grouped: bad problem");
    if (g.lda < g.K || g.ldb < g.K || g.ldc < g.N) return set_error(QAVIT_EINVAL, "gemm_nt_grouped: leading dimension too small");
    if (g.a_mode == 1 && (!g.ln_gamma || !g.ln_beta || !g.ln_mean || !g.ln_rstd)) return set_error(QAVIT_EINVAL, "gemm_nt_grouped: LayerNorm prologue needs gamma/beta and row statistics");
    if ((g.drop_p > 0.f || g.dp_p > 0.f || g.a_drop_p > 0.f || g.a_dp_p > 0.f) && !g.rng) return set_error(QAVIT_EINVAL, "gemm_nt_grouped: dropout requested without rng state");
  }
  if (a[0].dtype == QAVIT_F32) return dispatch_gemm_nt<float>(a[0], st, a, n);
  if (a[0].dtype == QAVIT_BF16) return dispatch_gemm_nt<bf16>(a[0], st, a, n);
  return set_error(QAVIT_EINVAL, "gemm_nt_grouped: unknown dtype");
}

extern "C" size_t qavit_gemm_tn_ws_bytes(void) { return qv::gemm_tn_wide_ws_bytes(); }

extern "C" int qavit_gemm_tn_grouped(const qavit_gemm_tn_args* a, int n, void* stream) {
  return qavit_gemm_tn_grouped_ws(a, n, nullptr, 0, stream);
}

extern "C" int qavit_gemm_tn_grouped_ws(const qavit_gemm_tn_args* a, int n, void* ws, size_t ws_bytes, void* stream) {
  using namespace qv;
  if (!a || n <= 0) return set_error(QAVIT_EINVAL, "gemm_tn_grouped: empty group");
  if (ws && (ws_bytes < gemm_tn_wide_ws_bytes() || (reinterpret_cast<uintptr_t>(ws) & 15)))
    return set_error(QAVIT_EINVAL, "gemm_tn_grouped: workspace smaller than qavit_gemm_tn_ws_bytes() or not 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static int legacy = -1;
  if (legacy < 0) { const char* e = getenv("QAVIT_TN_LEGACY"); legacy = e ? atoi(e) : 0; }
  for (int i = 0; i < n; ++i) {
    const qavit_gemm_tn_args& g = a[i];
    if (!g.A || !g.B || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return set_error(QAVIT_EINVAL, "gemm_tn_grouped: bad problem");
    if (g.lda < g.N || g.ldb < g.K || g.ldc < g.K) return set_error(QAVIT_EINVAL, "gemm_tn_grouped: leading dimension too small");
    if (g.ln_mean && (!g.ln_rstd || !g.ln_gamma || !g.ln_beta)) return set_error(QAVIT_EINVAL, "gemm_tn_grouped: incomplete LayerNorm-on-load arguments");
    if (g.dtype != QAVIT_BF16 && g.dtype != QAVIT_F32) return set_error(QAVIT_EINVAL, "gemm_tn_grouped: unknown dtype");
  }
  if (!legacy) {
    // bf16 problems: wide-tile kernel, grouped by tile class over the WHOLE list (the problems are independent, so a
    // stray fp32 problem in the queue must not cut a class group short).  fp32 problems and odd strides / alignments:
    // the single-problem path.
    const int rc = gemm_tn_wide(a, n, st, ws);
    if (rc) return rc;
    TnGroup G;
    G.n = 0;
    int wg = 0;
    auto flush_f32 = [&]() {
      if (G.n == 0) return;
      G.wg_start[G.n] = wg;
      hipLaunchKernelGGL(gemm_tn_f32_grouped_kernel, dim3(wg), dim3(GEMM_THREADS), 0, st, G);
      G.n = 0; wg = 0;
    };
    for (int i = 0; i < n; ++i) {
      if (a[i].dtype == QAVIT_BF16 && gemm_tn_wide_ok(a[i])) continue;
      if (a[i].dtype == QAVIT_BF16) {
        const int rc2 = launch_gemm_tn<bf16>(a[i], st);
        if (rc2) return rc2;
        continue;
      }
      int tn, tk, splits, rows;
      tn_split_plan(a[i], 64, tn, tk, splits, rows);
      G.p[G.n] = a[i]; G.tn[G.n] = tn; G.tk[G.n] = tk; G.rows[G.n] = rows; G.wg_start[G.n] = wg;
      wg += tn * tk * splits;
      if (++G.n == TN_GROUP) flush_f32();
    }
    flush_f32();
    if (int rc3 = check_launch("gemm_tn_grouped(f32)")) return rc3;
    return QAVIT_OK;
  }
  int done = 0;
  while (done < n) {
    TnGroup G;
    G.n = 0;
    int wg = 0;
    while (done < n && G.n < TN_GROUP) {
      const qavit_gemm_tn_args& g = a[done];
      if (g.dtype != QAVIT_BF16) {            // fp32 problems run through the single-problem path
        int rc = launch_gemm_tn<float>(g, st);
        if (rc) return rc;
        ++done;
        continue;
      }
      int tn, tk, splits, rows;
      tn_split_plan(g, 256, tn, tk, splits, rows);
      const int i = G.n++;
      G.p[i] = g; G.tn[i] = tn; G.tk[i] = tk; G.rows[i] = rows;
      G.wg_start[i] = wg;
      wg += tn * tk * splits;
      ++done;
    }
    if (G.n > 0) {
      G.wg_start[G.n] = wg;
      hipLaunchKernelGGL(gemm_tn_bf16_grouped_kernel, dim3(wg), dim3(GEMM_THREADS), 0, st, G);
    }
  }
  return check_launch("gemm_tn_grouped");
}

extern "C" int qavit_gemm_tn(const qavit_gemm_tn_args* a, void* stream) {
  using namespace qv;
  if (!a || !a->A || !a->B || !a->C) return set_error(QAVIT_EINVAL, "gemm_tn: null operand");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return set_error(QAVIT_EINVAL, "gemm_tn: non-positive dimension");
  if (a->lda < a->N || a->ldb < a->K || a->ldc < a->K) return set_error(QAVIT_EINVAL, "gemm_tn: leading dimension too small");
  if (a->ln_mean && (!a->ln_rstd || !a->ln_gamma || !a->ln_beta)) return set_error(QAVIT_EINVAL, "gemm_tn: incomplete LayerNorm-on-load arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == QAVIT_F32) return launch_gemm_tn<float>(*a, st);
  if (a->dtype == QAVIT_BF16) return qavit_gemm_tn_grouped(a, 1, stream);
  return set_error(QAVIT_EINVAL, "gemm_tn: unknown dtype");
}
