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
// ------------------------------------------------------------------------------------------------
template <typename T, int BM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(qavit_gemm_args g) {
  using M_ = Mma<T>;
  constexpr int VN = Vec<T>::N;
  constexpr int FK = M_::FK;
  constexpr int KC = kc_elems<T>();
  constexpr int WM = (BM / 16 < 4) ? BM / 16 : 4;   // waves along M
  constexpr int WN = 4 / WM;                        // waves along N
  constexpr int MT = BM / 16 / WM;                  // 16-row tiles per wave
  constexpr int NT = 4 / WN;                        // 16-col tiles per wave
  typedef typename Vec<T>::type vec_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Kp = (g.K + FK - 1) / FK * FK;
  const int lda_s = Kp + VN;                         // LDS row stride of As (elements)
  constexpr int ldb_s = KC + VN;
  T* As = reinterpret_cast<T*>(smem);
  char* region2 = smem + (size_t)BM * lda_s * sizeof(T);
  T* Bs = reinterpret_cast<T*>(region2);
  float* Cs = reinterpret_cast<float*>(region2);     // aliases Bs (used only between barriers)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m0 = blockIdx.x * BM;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  T* C = reinterpret_cast<T*>(g.C);

  uint32_t key_adrop = 0, key_adp = 0, key_drop = 0, key_dp = 0;
  if (g.rng) {
    key_adrop = rng_key(g.rng, g.a_drop_site);
    key_adp = rng_key(g.rng, g.a_dp_site);
    key_drop = rng_key(g.rng, g.drop_site);
    key_dp = rng_key(g.rng, g.dp_site);
  }

  // ---------------- stage A rows (with the backward transform when a_mode == 2) ----------------
  {
    const int kv = Kp / VN;                          // vectors per LDS row
    const bool vec_ok = (g.K % VN == 0) && (g.lda % VN == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool bwd = g.a_mode == 2;
    const T* Zin = reinterpret_cast<const T*>(g.a_Z);
    T* Aout = reinterpret_cast<T*>(g.a_out);
    const bool vec_ok2 = vec_ok && (!bwd || ((!Zin || (g.a_ldz % VN == 0 && (reinterpret_cast<uintptr_t>(Zin) & 15) == 0)) &&
                                             (!Aout || (g.a_ldo % VN == 0 && (reinterpret_cast<uintptr_t>(Aout) & 15) == 0))));
    const float a_inv_keep = g.a_drop_p > 0.f ? 1.f / (1.f - g.a_drop_p) : 1.f;
    const float a_dp_inv = g.a_dp_p > 0.f ? 1.f / (1.f - g.a_dp_p) : 1.f;
    for (int idx = tid; idx < BM * kv; idx += GEMM_THREADS) {
      const int r = idx / kv, v = idx - r * kv;
      const int m = m0 + r, k = v * VN;
      T* dst = As + (size_t)r * lda_s + k;
      if (m >= g.M || k >= g.K) { store_vec_zero<T>(dst); continue; }
      float f[VN];
      if (vec_ok2) {
        vec_t x = *reinterpret_cast<const vec_t*>(A + (size_t)m * g.lda + k);
#pragma unroll
        for (int i = 0; i < VN; ++i) f[i] = to_f<T>(x[i]);
      } else {
#pragma unroll
        for (int i = 0; i < VN; ++i) f[i] = (k + i < g.K) ? to_f<T>(A[(size_t)m * g.lda + k + i]) : 0.f;
      }
      if (bwd) {
        float rowf = g.a_scale;
        if (g.a_dp_p > 0.f) rowf *= drop_factor(key_adp, (uint32_t)(m / g.a_dp_rows), g.a_dp_p, a_dp_inv);
        float z[VN];
        if (Zin && g.a_act) {
          if (vec_ok2) {
            vec_t zz = *reinterpret_cast<const vec_t*>(Zin + (size_t)m * g.a_ldz + k);
#pragma unroll
            for (int i = 0; i < VN; ++i) z[i] = to_f<T>(zz[i]);
          } else {
#pragma unroll
            for (int i = 0; i < VN; ++i) z[i] = (k + i < g.K) ? to_f<T>(Zin[(size_t)m * g.a_ldz + k + i]) : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          float v_ = f[i] * rowf;
          if (g.a_drop_p > 0.f) v_ *= drop_factor(key_adrop, (uint32_t)m * (uint32_t)g.K + (uint32_t)(k + i), g.a_drop_p, a_inv_keep);
          if (Zin && g.a_act) v_ *= gelu_grad_f(z[i]);
          f[i] = v_;
        }
      }
      vec_t o;
#pragma unroll
      for (int i = 0; i < VN; ++i) o[i] = from_f<T>(f[i]);
      *reinterpret_cast<vec_t*>(dst) = o;
      if (bwd && Aout) {
        if (vec_ok2) {
          *reinterpret_cast<vec_t*>(Aout + (size_t)m * g.a_ldo + k) = o;
        } else {
#pragma unroll
          for (int i = 0; i < VN; ++i) if (k + i < g.K) Aout[(size_t)m * g.a_ldo + k + i] = o[i];
        }
      }
    }
  }
  __syncthreads();

  // ---------------- fused LayerNorm over K on the resident rows ----------------
  if (g.a_mode == 1) {
    const float invK = 1.f / (float)g.K;
    for (int r = wave; r < BM; r += 4) {
      const int m = m0 + r;
      if (m >= g.M) continue;                         // wave-uniform
      T* row = As + (size_t)r * lda_s;
      float s = 0.f;
      for (int k = lane; k < g.K; k += 64) s += to_f<T>(row[k]);
      const float mean = wave_sum(s) * invK;
      float s2 = 0.f;
      for (int k = lane; k < g.K; k += 64) { const float d = to_f<T>(row[k]) - mean; s2 += d * d; }
      const float rstd = rsqrtf(wave_sum(s2) * invK + g.ln_eps);
      for (int k = lane; k < g.K; k += 64)
        row[k] = from_f<T>((to_f<T>(row[k]) - mean) * rstd * g.ln_gamma[k] + g.ln_beta[k]);
      if (lane == 0) {
        if (g.ln_mean) g.ln_mean[m] = mean;
        if (g.ln_rstd) g.ln_rstd[m] = rstd;
      }
    }
    __syncthreads();
  }

  const int wm = wave % WM, wn = wave / WM;
  const int fr = lane & 15, fq = lane >> 4;
  const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const float dp_inv = g.dp_p > 0.f ? 1.f / (1.f - g.dp_p) : 1.f;
  const T* Rr = reinterpret_cast<const T*>(g.R);
  T* Zo = reinterpret_cast<T*>(g.Z);

  for (int n0 = 0; n0 < g.N; n0 += BN) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < Kp; k0 += KC) {
      const int kc = (Kp - k0 < KC) ? (Kp - k0) : KC;   // multiple of FK
      __syncthreads();                                   // previous readers of Bs / Cs done
      {  // stage B[n0:n0+64, k0:k0+kc] (rows = output columns)
        const int kv = kc / VN;
        const bool vec_ok = (g.K % VN == 0) && (g.ldb % VN == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
        for (int idx = tid; idx < BN * kv; idx += GEMM_THREADS) {
          const int r = idx / kv, v = idx - r * kv;
          const int n = n0 + r, k = k0 + v * VN;
          T* dst = Bs + r * ldb_s + v * VN;
          if (n >= g.N || k >= g.K) { store_vec_zero<T>(dst); continue; }
          if (vec_ok) {
            *reinterpret_cast<vec_t*>(dst) = *reinterpret_cast<const vec_t*>(B + (size_t)n * g.ldb + k);
          } else {
            vec_t o;
#pragma unroll
            for (int i = 0; i < VN; ++i) o[i] = (k + i < g.K) ? B[(size_t)n * g.ldb + k + i] : from_f<T>(0.f);
            *reinterpret_cast<vec_t*>(dst) = o;
          }
        }
      }
      __syncthreads();
      const int nf = kc / FK;
      for (int kf = 0; kf < nf; ++kf) {
        typename M_::frag af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = (wm * MT + i) * 16 + fr;
          af[i] = *reinterpret_cast<const typename M_::frag*>(As + (size_t)row * lda_s + k0 + kf * FK + fq * VN);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = (wn * NT + j) * 16 + fr;
          bfr[j] = *reinterpret_cast<const typename M_::frag*>(Bs + col * ldb_s + kf * FK + fq * VN);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) M_::mma(af[i], bfr[j], acc[i][j]);
      }
    }
    __syncthreads();   // all waves finished reading Bs -> reuse as Cs
    // accumulator (col = lane&15, row = 4*(lane>>4)+reg) -> Cs[row][col]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * MT + i) * 16 + fq * 4 + r;
          const int col = (wn * NT + j) * 16 + fr;
          Cs[row * CS_LD + col] = acc[i][j][r];
        }
    __syncthreads();
    // epilogue: each thread owns 16 consecutive columns of one row
    for (int idx = tid; idx < BM * 4; idx += GEMM_THREADS) {
      const int r = idx >> 2, cg = idx & 3;
      const int m = m0 + r;
      const int n = n0 + cg * 16;
      if (m >= g.M || n >= g.N) continue;
      const int nv = (g.N - n < 16) ? (g.N - n) : 16;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + cg * 16 + i);
        v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
      }
      float rowf = g.scale;
      if (g.dp_p > 0.f) rowf *= drop_factor(key_dp, (uint32_t)(m / g.dp_rows), g.dp_p, dp_inv);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < nv) {
          float x = v[i];
          if (g.bias) x += g.bias[n + i];
          if (Zo) Zo[(size_t)m * g.ldz + n + i] = from_f<T>(x);
          if (g.act == 1) x = gelu_f(x);
          if (g.drop_p > 0.f) x *= drop_factor(key_drop, (uint32_t)m * (uint32_t)g.N + (uint32_t)(n + i), g.drop_p, inv_keep);
          x *= rowf;
          if (Rr) x += to_f<T>(Rr[(size_t)m * g.ldr + n + i]);
          v[i] = x;
        }
      }
      T* crow = C + (size_t)m * g.ldc + n;
      const bool cvec = (nv == 16) && (g.ldc % VN == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (n % VN == 0);
      if (cvec) {
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
    }
  }
}

template <typename T, int BM>
static int launch_gemm_nt(const qavit_gemm_args& g, hipStream_t st) {
  constexpr int VN = Vec<T>::N;
  constexpr int FK = Mma<T>::FK;
  constexpr int KC = kc_elems<T>();
  const int Kp = round_up(g.K, FK);
  const size_t a_bytes = (size_t)BM * (Kp + VN) * sizeof(T);
  const size_t b_bytes = (size_t)BN * (KC + VN) * sizeof(T);
  const size_t c_bytes = (size_t)BM * CS_LD * sizeof(float);
  const size_t smem = a_bytes + (b_bytes > c_bytes ? b_bytes : c_bytes);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "gemm_nt: K too large for the resident-row tile");
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int grid = (g.M + BM - 1) / BM;
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM>), dim3(grid), dim3(GEMM_THREADS), smem, st, g);
  return check_launch("gemm_nt");
}

template <typename T>
static int dispatch_gemm_nt(const qavit_gemm_args& g, hipStream_t st) {
  constexpr int VN = Vec<T>::N;
  constexpr int FK = Mma<T>::FK;
  const int Kp = round_up(g.K, FK);
  const size_t row_bytes = (size_t)(Kp + VN) * sizeof(T);
  const size_t budget = 96 * 1024;   // resident A rows; leaves room for the B / C region and 1 WG/CU
  // prefer tall tiles (weight tile reuse) but keep >= ~2 waves of workgroups on 256 CUs
  int bm = 128;
  while (bm > 16 && (bm * row_bytes > budget)) bm >>= 1;
  while (bm > 32 && (g.M + bm - 1) / bm < 384) bm >>= 1;
  if (bm * row_bytes > 140 * 1024) return set_error(QAVIT_EINVAL, "gemm_nt: K too large");
  switch (bm) {
    case 128: return launch_gemm_nt<T, 128>(g, st);
    case 64: return launch_gemm_nt<T, 64>(g, st);
    case 32: return launch_gemm_nt<T, 32>(g, st);
    default: return launch_gemm_nt<T, 16>(g, st);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn : C[N,K] += A^T B over a slice of M, one 64x64 output tile per workgroup
// ------------------------------------------------------------------------------------------------
constexpr int TN_MC = 64;   // rows of M per staged chunk

template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(qavit_gemm_tn_args g, int rows_per_split) {
  using M_ = Mma<T>;
  constexpr int VN = Vec<T>::N;
  constexpr int FK = M_::FK;
  constexpr int LDT = TN_MC + VN;                     // LDS row stride (elements); m is the contiguous axis
  __shared__ __attribute__((aligned(16))) T At[64 * LDT];   // [n][m]
  __shared__ __attribute__((aligned(16))) T Bt[64 * LDT];   // [k][m]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mbeg = blockIdx.z * rows_per_split;
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
    if (g.colsum && blockIdx.y == 0 && tid < 64) {
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
  if (g.colsum && blockIdx.y == 0 && tid < 64 && n0 + tid < g.N) atomic_add_f(g.colsum + n0 + tid, csum);
}

template <typename T>
static int launch_gemm_tn(const qavit_gemm_tn_args& g, hipStream_t st) {
  const int tn = (g.N + 63) / 64, tk = (g.K + 63) / 64;
  int splits = g.splits;
  if (splits <= 0) {
    splits = 768 / (tn * tk);
    if (splits < 1) splits = 1;
  }
  int max_splits = (g.M + 255) / 256;                 // at least 256 rows per split
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  int rows = (g.M + splits - 1) / splits;
  rows = (rows + TN_MC - 1) / TN_MC * TN_MC;
  splits = (g.M + rows - 1) / rows;
  hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(tn, tk, splits), dim3(GEMM_THREADS), 0, st, g, rows);
  return check_launch("gemm_tn");
}

}  // namespace qv

extern "C" int qavit_gemm_nt(const qavit_gemm_args* a, void* stream) {
  using namespace qv;
  if (!a || !a->A || !a->B || !a->C) return set_error(QAVIT_EINVAL, "gemm_nt: null operand");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return set_error(QAVIT_EINVAL, "gemm_nt: non-positive dimension");
  if (a->lda < a->K || a->ldb < a->K || a->ldc < a->N) return set_error(QAVIT_EINVAL, "gemm_nt: leading dimension too small");
  if (a->a_mode == 1 && (!a->ln_gamma || !a->ln_beta)) return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm prologue needs gamma/beta");
  if ((a->drop_p > 0.f || a->dp_p > 0.f || a->a_drop_p > 0.f || a->a_dp_p > 0.f) && !a->rng)
    return set_error(QAVIT_EINVAL, "gemm_nt: dropout requested without rng state");
  if ((a->dp_p > 0.f && a->dp_rows <= 0) || (a->a_dp_p > 0.f && a->a_dp_rows <= 0))
    return set_error(QAVIT_EINVAL, "gemm_nt: drop-path needs rows-per-sample");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == QAVIT_F32) return dispatch_gemm_nt<float>(*a, st);
  if (a->dtype == QAVIT_BF16) return dispatch_gemm_nt<bf16>(*a, st);
  return set_error(QAVIT_EINVAL, "gemm_nt: unknown dtype");
}

extern "C" int qavit_gemm_tn(const qavit_gemm_tn_args* a, void* stream) {
  using namespace qv;
  if (!a || !a->A || !a->B || !a->C) return set_error(QAVIT_EINVAL, "gemm_tn: null operand");
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return set_error(QAVIT_EINVAL, "gemm_tn: non-positive dimension");
  if (a->lda < a->N || a->ldb < a->K || a->ldc < a->K) return set_error(QAVIT_EINVAL, "gemm_tn: leading dimension too small");
  if (a->ln_mean && (!a->ln_rstd || !a->ln_gamma || !a->ln_beta)) return set_error(QAVIT_EINVAL, "gemm_tn: incomplete LayerNorm-on-load arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == QAVIT_F32) return launch_gemm_tn<float>(*a, st);
  if (a->dtype == QAVIT_BF16) return launch_gemm_tn<bf16>(*a, st);
  return set_error(QAVIT_EINVAL, "gemm_tn: unknown dtype");
}
