/*
 * qavit.h -- C-ABI of libqavit_hip.so: the MI355X (gfx950) kernels behind the QA-ViT / HQA-ViT
 * nn.Module surface.
 *
 * The reference (cujoramirez/QA-ViT) has no FFI: its "operator interface" for this path is the set of
 * torch.nn.functional / nn.Module calls made inside the model classes.  Each entry point below names the
 * reference call sites it replaces (file:line in /root/reference).  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions:
 *   - plain pointers and sizes only; every buffer is owned by the caller (PyTorch's caching allocator);
 *     the library never allocates device memory and keeps no pointer past the call;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no internal synchronisation, so all
 *     entry points are hipGraph-capturable; the library is stateless and re-entrant;
 *   - return 0 on success, a negative QAVIT_E* code on invalid arguments or launch failure;
 *     qavit_last_error() returns a thread-local message for the last failure;
 *   - `dtype` selects the storage type of activations / packed weights: QAVIT_F32 or QAVIT_BF16;
 *     arithmetic and accumulation are fp32, 1-D parameters (bias, LayerNorm, gates) and all gradients
 *     of parameters are fp32;
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef QAVIT_H
#define QAVIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QAVIT_F32 0
#define QAVIT_BF16 1

#define QAVIT_OK 0
#define QAVIT_EINVAL (-1)   /* bad shape / dtype / alignment */
#define QAVIT_ELAUNCH (-2)  /* hipGetLastError() after launch */

int qavit_version(void);
const char* qavit_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( prologue(A)[M,K] . B[N,K]^T + bias )          (MFMA; A rows resident in LDS)
 * Replaces nn.Linear / F.linear call sites of the block, e.g. HQAViT_CIFAR100.py:448 (qkv), :464 (proj),
 * :1074-1077 (LayerNorm + compress), :652-656 (BottleneckMLP fc1+GELU+dropout, fc2+dropout), :704-712
 * (CCF-FFN fc1/fc2), and their autograd input-gradients (dX = dZ . W with B := W^T packed).
 *
 * a_mode 0: A as stored.
 * a_mode 1: A := (A - ln_mean[m]) * ln_rstd[m] * ln_gamma + ln_beta (nn.LayerNorm fused as prologue: the row
 *           statistics are INPUTS, produced by qavit_row_stats; the normalised rows are never written to HBM).
 * a_mode 3: as a_mode 1, but ln_mean / ln_rstd are OUTPUTS: the call computes the row statistics itself (inside the K-loop
 *           kernel's prologue where that kernel applies, by a row_stats launch otherwise) and leaves them for the backward.
 * a_mode 2: A := A * droppath(a_dp) * dropout(a_drop) * gelu'(a_Z)   -- the backward of a layer's epilogue
 *           applied to its incoming gradient; the transformed tile is also written to a_out (dZ).
 * epilogue: v = acc + bias; Z := v (if Z); v = gelu(v) (act==1); v *= dropout; v *= scale; v *= droppath; v += R.
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_gemm_args {
  int dtype;
  int M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  /* A prologue */
  int a_mode;
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  float* ln_mean; float* ln_rstd;
  const void* a_Z; int64_t a_ldz; int a_act;
  float a_drop_p; int a_drop_site;
  float a_dp_p; int a_dp_site; int a_dp_rows;
  float a_scale;
  void* a_out; int64_t a_ldo;
  /* epilogue */
  void* Z; int64_t ldz;
  int act;
  float drop_p; int drop_site;
  float scale;
  float dp_p; int dp_site; int dp_rows;
  const void* R; int64_t ldr;
  const int64_t* rng; /* device int64[2]: seed, step */
  /* TWO-SOURCE A (optional, A2 != NULL): the contraction runs over the columns of [A | A2] -- columns 0 .. a2_k0 come from A (row stride
   * lda >= a2_k0), columns a2_k0 .. K from A2 (row stride lda2 >= K - a2_k0): Linear(2C -> C) on cat([T, R]) (HQAViT_CIFAR100.py:951)
   * without the 2C-wide cat buffer and without a second, accumulating launch.  bf16, a_mode 0, a2_k0 % 64 == 0, shapes the K-loop
   * kernel takes (qavit_gemm_nt_a2_supported); anything else is refused. */
  const void* A2; int64_t lda2; int a2_k0;
  /* LAYERNORM-BACKWARD EPILOGUE (optional, e_x != NULL): the product is the gradient of a LayerNorm's OUTPUT (the input-gradient GEMM of
   * the Linear behind the LayerNorm: HQAViT_CIFAR100.py:1072-1082 norm1 -> qkv, :704 norm2 -> fc1, :945 gate_norm -> gate_fc) and the
   * call writes the gradient of the LayerNorm's INPUT instead:  dy = acc (+ e_add0 + e_add1: the same output's gradients from other
   * consumers, [M, N] contiguous);  C = rstd * (dy * gamma - mean_N(dy * gamma) - xhat * mean_N(dy * gamma * xhat)) + R,  xhat =
   * (e_x - e_mean) * e_rstd -- one launch and no [M, N] round trip where a GEMM and qavit_layernorm_bwd ran.  The LayerNorm parameter
   * gradients dgamma = colsum(dy * xhat), dbeta = colsum(dy) are ADDED to e_dgamma / e_dbeta with float atomics, or, with e_parts =
   * float[qavit_gemm_nt_lnbwd_parts(M, N, K)][2][N], left as partial rows in qavit_layernorm_bwd's layout (fold with
   * qavit_ln_param_reduce).  bf16, N in {128, 192, 256} = the LayerNorm width = one column block, a_mode 0 or 2, no other epilogue term
   * than R (qavit_gemm_nt_lnbwd_supported); e_x / e_add* rows of N elements, 16-byte aligned. */
  const void* e_x; const float* e_mean; const float* e_rstd; const float* e_gamma;
  const void* e_add0; const void* e_add1;
  float* e_dgamma; float* e_dbeta; float* e_parts;
} qavit_gemm_args;

int qavit_gemm_nt(const qavit_gemm_args* a, void* stream);
int qavit_gemm_nt_a2_supported(int dtype, int M, int N, int K, int a2_k0);
int qavit_gemm_nt_lnbwd_supported(int dtype, int M, int N, int K, int a_mode);
int qavit_gemm_nt_lnbwd_parts(int M, int N, int K);
/* n independent problems (host array); up to 4 of one shape / dtype / prologue / epilogue kind that take the resident-slice
 * kernel share a grid (the four compress_* Linears of a block and their input gradients), others are launched one by one */
int qavit_gemm_nt_grouped(const qavit_gemm_args* a, int n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * C[N,K] += A[M,N]^T . B[M,K]   (fp32 atomics, split over M);   colsum[N] += sum_m A[m,n]
 * The weight / bias gradient of every nn.Linear above (autograd of F.linear).  B may be LayerNorm-ed on
 * load with saved statistics (the fused-prologue layers).
 * K need not be a multiple of 8 when B's rows are padded (ldb % 8 == 0, ldb >= K rounded up to 8): the pad columns are read and
 * their products discarded.
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_gemm_tn_args {
  int dtype;
  int M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  float* C; int64_t ldc;
  float* colsum;
  const float* ln_gamma; const float* ln_beta; const float* ln_mean; const float* ln_rstd;
  int splits;
} qavit_gemm_tn_args;

int qavit_gemm_tn(const qavit_gemm_tn_args* a, void* stream);
/* n independent problems (host array) in as few grids as possible: the weight-gradient GEMMs of a backward pass are
 * off the critical path and individually too small to fill the chip, so the autograd layer defers and batches them */
int qavit_gemm_tn_grouped(const qavit_gemm_tn_args* a, int n, void* stream);
/* The same with `ws_bytes` >= qavit_gemm_tn_ws_bytes() of 16-byte-aligned device scratch that belongs to this call until its launches
 * have run: a tile class with more problems than one launch carries by value (24) is then ONE launch over a device-side problem
 * table (filled by small writer launches on `stream`) instead of one launch per 24 -- fewer workgroups per problem, fewer fp32-atomic
 * tile flushes.  ws = NULL: as qavit_gemm_tn_grouped.  While `stream` is being captured the table is filled by copy nodes from a pinned
 * host image of the library's own (kept for the life of the process: every replay of the graph reads it) instead of writer launches. */
int qavit_gemm_tn_grouped_ws(const qavit_gemm_tn_args* a, int n, void* ws, size_t ws_bytes, void* stream);
size_t qavit_gemm_tn_ws_bytes(void);

/* ---------------------------------------------------------------------------------------------------
 * nn.LayerNorm over the last dim C (HQAViT_CIFAR100.py:1072 norm1, :1083 norm2, :1273 norm, :1029 ...).
 * fwd: y = LN(x)*gamma+beta (+ add[row % add_rows] if add != NULL: the pos_embed add of :1250);
 *      mean/rstd [rows] saved for backward.
 * bwd: dx = LN'(dy); dgamma/dbeta += (fp32 atomics); dadd[row % add_rows] += dy when dadd != NULL.
 * act = 1 fuses the exact GELU that follows the norm in LMFAdapter / SplitFusion.cat_mlp (HQAViT_CIFAR100.py:823,
 * :927-931): fwd y = gelu(LN(x)), bwd takes dy through gelu' of the recomputed LN output (needs beta).
 * ------------------------------------------------------------------------------------------------- */
int qavit_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                        float eps, int rows, int C, float* mean, float* rstd,
                        const float* add, int add_rows, int act, void* stream);
/* mean / rstd of each row only: the statistics half of a LayerNorm whose normalisation is fused into qavit_gemm_nt */
int qavit_row_stats(int dtype, const void* x, float eps, int rows, int C, float* mean, float* rstd, void* stream);
int qavit_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                        const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                        int rows, int C, float* dadd, int add_rows, const float* beta, int act, const void* dres, float* part_ws, void* stream);
/* dres (optional, same shape / dtype as dx): dx = LN_backward(dy) + dres -- the other gradient that meets this one at x (a
 * residual connection around the normalised branch), added in the same pass instead of a separate elementwise kernel.
 * part_ws (optional, float[qavit_layernorm_bwd_parts(rows, C)][2][C], 16-byte aligned; needs C % 4 == 0 and vector-aligned
 * operands): the kernel leaves its per-workgroup partial sums of (dgamma, dbeta) there with plain stores and does NOT touch
 * dgamma / dbeta; qavit_ln_param_reduce adds them later -- one launch for all the LayerNorms of a backward pass.  The per-workgroup
 * same-address float atomics this replaces were ~45 % of the kernel at 16k rows. */
int qavit_layernorm_bwd_parts(int rows, int C);
/* LayerNorm backward FUSED with the input-gradient GEMM of the narrow Linear behind the norm (TokenLearner: Linear(192, 16) on LN(x),
 * HQAViT_CIFAR100.py:985-990): dx = LN_backward(dz . W) + dres, with dz [rows, KZ] (leading dimension ldz) the gradient of the Linear's
 * output and W [KZ, C] (ldw) its weight in the compute dtype; the [rows, C] product never exists in memory.  bf16, KZ = 16, C % 4 == 0,
 * C <= 256 (qavit_layernorm_bwd_lin_supported); dgamma / dbeta / part_ws as in qavit_layernorm_bwd, same partial-row count. */
/* SplitFusion's closing pair as one launch each way (HQAViT_CIFAR100.py:953-965): mixed = s0*a + s1*(t + dropout(h)), s = softmax(fw[0:2]);
 * y = LayerNorm(mixed).  fwd writes `mixed` [rows, C] (the backward's LayerNorm input), y, mean, rstd.  bwd takes dy and writes da, dt, dh,
 * adds the blend-weight gradient to dfw[0:2] (float atomics, one pair per workgroup; may be NULL) and the LayerNorm parameter gradients
 * to dgamma / dbeta -- or, with part_ws = float[n = qavit_mix3_ln_bwd_parts(rows, C)][2][C] followed by float[n][4], leaves them as partial
 * rows in the layout of qavit_layernorm_bwd's and the blend-weight contributions as n narrow rows [dfw0, dfw1, 0, 0] behind them (dfw != NULL
 * then only says they are wanted: fold with a reduce descriptor C = 2, stride 4).  Values and rounding points are those of qavit_mix3_fwd / qavit_layernorm_fwd (resp. qavit_layernorm_bwd / qavit_mix3_bwd) run
 * one after the other.  fp32 / bf16, C % 4 == 0, C <= 256 (qavit_mix3_ln_supported), rows * C < 2^32, vector-aligned operands. */
int qavit_mix3_ln_supported(int dtype, int C);
int qavit_mix3_ln_bwd_parts(int rows, int C);
int qavit_mix3_ln_fwd(int dtype, const void* a, const void* t, const void* h, const float* fw, float drop_p, int drop_site, const int64_t* rng,
                      void* mixed, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, int rows, int C,
                      void* stream);
int qavit_mix3_ln_bwd(int dtype, const void* dy, const void* a, const void* t, const void* h, const float* fw, float drop_p, int drop_site,
                      const int64_t* rng, const void* mixed, const float* gamma, const float* mean, const float* rstd, void* da, void* dt,
                      void* dh, float* dfw, float* dgamma, float* dbeta, int rows, int C, float* part_ws, void* stream);
/* The same pair with SplitFusion's GATE in front of it (HQAViT_CIFAR100.py:945-965): a = t + sigmoid(g) * r is formed in registers (rounded
 * where qavit_gate_mix_fwd stored it) instead of read, so the forward is one launch for gate + blend + norm and `a` never exists in
 * memory.  bwd writes ONE gradient for t (the gate's pass-through s0 * dm plus the blend's s1 * dm, each rounded as the separate launches
 * round them, then summed), dr and dg (qavit_gate_mix_bwd's), dh, and the parameter gradients as above. */
int qavit_gate_mix3_ln_fwd(int dtype, const void* t, const void* r, const void* g, const void* h, const float* fw, float drop_p, int drop_site,
                           const int64_t* rng, void* mixed, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                           int rows, int C, void* stream);
int qavit_gate_mix3_ln_bwd(int dtype, const void* dy, const void* t, const void* r, const void* g, const void* h, const float* fw, float drop_p,
                           int drop_site, const int64_t* rng, const void* mixed, const float* gamma, const float* mean, const float* rstd,
                           void* dt, void* dr, void* dg, void* dh, float* dfw, float* dgamma, float* dbeta, int rows, int C, float* part_ws,
                           void* stream);
/* LayerNorm backward whose incoming gradient is the SUM of n_dy (1..5) same-shape tensors -- the k gradients of a normalised tensor
 * that feeds several consumers (norm1's output and the four attention branches, HQAViT_CIFAR100.py:1072-1078) -- summed in fp32 on load
 * instead of by a k-way sum launch.  `dy` is a HOST array of device pointers.  C % 4 == 0, C <= 256, vector-aligned operands; dres /
 * part_ws as in qavit_layernorm_bwd (same partial-row count). */
int qavit_layernorm_bwd_sum(int dtype, int n_dy, const void* const* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int C, const void* dres, float* part_ws,
                            void* stream);
int qavit_layernorm_bwd_lin_supported(int dtype, int KZ, int C);
int qavit_layernorm_bwd_lin(int dtype, const void* dz, int ldz, const void* W, int ldw, int KZ, const void* x, const float* gamma,
                            const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int C,
                            const void* dres, float* part_ws, void* stream);
/* NARROW descriptors -- C in {1, 2, 4}: scalar layer scales, blend / fusion logits -- are folded by ONE workgroup in a fixed row
 * order whatever nparts is, so those gradients are bit-reproducible from run to run (a float atomic per workgroup on one address
 * is not).  C == 1: rows of 4 floats [dgamma, dbeta, -, -]; C == 2: [dgamma0, dgamma1, dbeta0, dbeta1].  Wide descriptors split
 * their rows over workgroups of 32 and meet in float atomics: reproducible up to 32 partial rows. */
typedef struct qavit_ln_reduce_desc {
  const float* parts; int nparts; int C;     /* nparts rows of [2][C] (C % 4 == 0 or C in {1, 2}; C <= 2048) ... */
  float* dgamma; float* dbeta;               /* += the two halves of the summed row (either may be NULL) */
  int64_t stride;                            /* ... `stride` floats apart (0: 2*C, dense; else a multiple of 4) */
} qavit_ln_reduce_desc;
int qavit_ln_param_reduce(const qavit_ln_reduce_desc* d, int n, void* stream);

/* The four branch norms of a QuadAttentionBlock (norm_{swa,msda,cga,cross}, HQAViT_CIFAR100.py:1046-1049,1075-1078) act on
 * four same-shape tensors that are independent of each other: n <= 4 inputs per grid.  Host arrays of device pointers. */
int qavit_row_stats_multi(int dtype, int n, const void* const* x, float eps, int rows, int C, float* const* mean,
                          float* const* rstd, void* stream);
int qavit_layernorm_bwd_multi(int dtype, int n, const void* const* dy, const void* const* x, const float* const* gamma,
                              const float* const* mean, const float* const* rstd, void* const* dx,
                              float* const* dgamma, float* const* dbeta, int rows, int C, float* const* part_ws, void* stream);
/* part_ws: NULL, or n workspaces as in qavit_layernorm_bwd (C % 4 == 0, C <= 512, vector-aligned operands) */

/* ---------------------------------------------------------------------------------------------------
 * Attention core of the four branches: O = softmax(Q K_full^T / sqrt(D)) V_full per (group g, head h), with
 *   mode 0 (SWA / MSDA):  K_full = [ E_k[:L]^T . k_tok[g] (KC rows) ; shared_k (S rows) ]   (Linformer + bank)
 *   mode 1 (CGA / cross): K_full = [ k_tok[g] (L rows, may be 0) ; shared_k (S rows) ]
 * HQAViT_CIFAR100.py:452-461 (SWA), :514-525 (MSDA), :332-352 (LinformerCompression; zero padding to
 * seq_len is skipped algebraically: only the first L rows of E contribute), :584-587 (CGA, D=4),
 * :616-621 (cross), :355-397 (efficient_attention, default scale 1/sqrt(D), no mask).
 * One wavefront per (g,h); MFMA tiles read their operands from LDS.
 *   q      : element (g,i,h,d) at q[qrow(g,i)*ldq + h*D + d]          (activation dtype; qrow: see struct)
 *   k_tok  : element (g,l,h,d) at k_tok[krow(g,l)*ldk + h*D + d]; v_tok likewise
 *   E_k/E_v: fp32 [*, KC] row-major (mode 0), first L rows used
 *   sh_k/sh_v: fp32 [S, H*D] shared rows (the bank, or a Linear of the bank)
 *   o      : element (g,i,h,d) at o[qrow(g,i)*ldo + h*D + d]
 *   nan_flag (optional): set to 1 if any q/k/v input or output element is NaN (the reference then returns
 *   zeros for the WHOLE tensor: call qavit_nan_guard afterwards).
 * bwd: writes dq / dk_tok / dv_tok, and per-wave partial sums of dE_k, dE_v [L,KC] and dsh_k, dsh_v
 *   [S,D] (head slice) into `ws`; qavit_attn_bwd_reduce folds them into the fp32 gradient buffers (+=).
 *   qavit_attn_ws_floats() gives the workspace size.
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_attn_args {
  int dtype; int mode;
  int G, Nq, L, H, D, KC, S;
  /* row addressing: group g = b*groups_per_b + gi.  Query/output row of (g,i) is
   *   b*q_rows_per_b + (q_tbl ? q_tbl[gi*Nq+i] : gi*Nq+i); key-token row of (g,l) is
   *   b*k_rows_per_b + (k_tbl ? k_tbl[gi*L+l] : gi*L+l).  Tables (device int32) express the SWA window
   *   partition (HQAViT_CIFAR100.py:419-439) and the CGA channel-group regrouping (:562-564, :588-589)
   *   without permute+contiguous copies.  groups_per_b == 0 means "one flat group axis" (row = g*Nq+i). */
  int groups_per_b; int q_rows_per_b; int k_rows_per_b;
  const int32_t* q_tbl; const int32_t* k_tbl;
  const void* q; int64_t ldq;
  const void* k_tok; int64_t ldk;
  const void* v_tok; int64_t ldv;
  const float* E_k; const float* E_v;
  const float* sh_k; const float* sh_v;
  void* o; int64_t ldo;
  int* nan_flag;
  /* backward */
  const void* d_o; int64_t lddo;
  void* dq; int64_t lddq;
  void* dk_tok; int64_t lddk;
  void* dv_tok; int64_t lddv;
  float* ws; int64_t ws_floats;
  float* dE_k; float* dE_v; float* dsh_k; float* dsh_v;
  /* attention-probability dropout = the dropout_p the reference hands to F.scaled_dot_product_attention
   * (HQAViT_CIFAR100.py:390-392; call sites :461, :524, :587, :624 pass self.dropout.p in training):
   * O = (softmax(S) * mask / (1 - drop_p)) V.  The mask is a pure function of (rng[0] = seed, rng[1] = step,
   * drop_site, problem g*H+h, query i, key j) -- see attn_shared.h: attn_drop_factor -- so the backward call
   * (same drop_p / drop_site / rng contents) regenerates it instead of reading a stored mask.
   * drop_p == 0 or rng == NULL: no dropout. */
  float drop_p; int drop_site; const int64_t* rng;
} qavit_attn_args;

int qavit_attn_fwd(const qavit_attn_args* a, void* stream);
int qavit_attn_bwd(const qavit_attn_args* a, void* stream);      /* includes the partial-sum reduction */
int64_t qavit_attn_ws_floats(const qavit_attn_args* a);
/* zero `n` elements of x if *flag != 0, then clear the flag (efficient_attention's NaN -> zeros rule) */
/* x := 0 if flag[0] != 0, then flag[0] := 0 (flag = device int[2]: the flag and the guard's arrival ticket, both zero at rest) */
int qavit_nan_guard(int dtype, void* x, int64_t n, int* flag, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused attention BRANCH for 16-token problems (every HQA-ViT CIFAR block works on 16 learned tokens) and 64-token problems
 * (HQAViT_IN_Tiny.py's 64 learned tokens, QAViT.py at 32 px: an 8x8 token grid): ONE launch for
 *   out = dropout( proj( efficient_attention( q, K_full, V_full, dropout_p ) ) )
 * with q / k / v computed in the kernel from the branch input x (= norm1's output):
 *   kind 0, SWA   (HQAViT_CIFAR100.py:441-469): qkv(x) on the one 4x4 window; K_full = [E_k^T k ; bank_k], V likewise
 *   kind 1, MSDA  (:496-532): q = qkv(x)[:, :C]; k, v = qkv(pooled)[:, C:], pooled[j] = mean_s x[idx[j*stride+s]], j < L;
 *                 K_full = [E_k[:L]^T k ; bank_k]   (the reference's zero padding to 128 rows is algebraic)
 *   kind 2, cross (:613-626): q = q_proj(x); K_full = sh_k, V_full = sh_v (k_proj / v_proj of the bank, computed by the caller)
 * Shapes are fixed: T = 16 or 64 tokens, C = 192, H = 4 heads of D = 48, S = 16 shared rows, KC = 32 Linformer rows; bf16 only.
 * T = 64: SWA = the four 4x4 windows of the 8x8 grid (window_partition with window 4, HQAViT_IN_Tiny.py:756-800: L = 16, rows gathered
 * in the kernel, q / k / v / O saved at their token rows); MSDA = 64 queries against ONE key side per image from L <= 48 landmarks
 * (kv_save / pooled_save hold L rows per image); cross = 64 queries against the 16 projected bank rows.  Dropout problems:
 * SWA (image * 4 + window) * H + head with query 0..15, MSDA / cross image * H + head with query 0..63 -- as the unfused kernels.
 * Weights come in MFMA FRAGMENT order (qavit_pack_desc.pad = 1): wqkv_frag = packed [3C, C] (kind 2: [C, C]), wproj_frag =
 * packed [C, C].  Dropout masks follow the unfused kernels' contracts exactly (attention: attn_shared.h attn_drop_factor
 * with problem id = image * H + head; proj: the qavit_gemm_nt epilogue's drop_factor(key(site), row * C + col)), so a
 * forward through this kernel and a backward through the unfused chain see the same masks.
 * nan_flag (int[2], zero at rest): the reference's NaN -> zeros rule; when set the call rewrites `out` as
 * dropout(bias) rows (= proj of an all-zero attention output) and clears the flag.
 * o_save (optional, leading dimension ldo): the attention output O [B*T, C], operand of backward's dW_proj; q_save / kv_save /
 * pooled_save: see the struct.
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_branch_args {
  int dtype; int kind;
  int B, T, C, H, D, KC, S, L;
  const void* x; int64_t ldx;
  const void* wqkv_frag; const float* bqkv;
  const void* wproj_frag; const float* bproj;
  const float* E_k; const float* E_v;
  const float* sh_k; const float* sh_v;
  const int32_t* pool_idx; int pool_stride;
  void* out; int64_t ldo;
  void* o_save;
  float attn_drop_p; int attn_drop_site; float proj_drop_p; int proj_drop_site; const int64_t* rng;
  int* nan_flag;
  int drain_waits;   /* 0; != 0 (diagnostic): every weight-ring step first waits for ALL outstanding memory operations of the wave, so the
                      * counted s_waitcnt vmcnt(N) arithmetic of the q / k / v / O save bursts holds trivially -- same bits, slower */
  /* optional saves for the backward pass (q_save == NULL: none): q rows [B*T, ldq_save] at column 0; k at column 0 and v at
   * column C of kv_save [B*rows, ldkv_save], rows = T per image (SWA; kv_save = q_save + C elements gives the usual [B*T, 3C]
   * qkv matrix) or L landmark rows per image (MSDA); pooled_save [B*L, C] = MSDA's pooled landmarks.  kind 2 saves q only. */
  void* q_save; int64_t ldq_save; void* kv_save; int64_t ldkv_save; void* pooled_save;
  /* optional (needs nan_flag): one int written by the NaN-rule launch of this call, 1 = the rule was applied (then o_save is zeroed
   * too), 0 = not; hand it to qavit_branch_bwd, whose gradient through a tripped branch is exactly zero as the reference's is */
  int* nan_trip;
  /* != 0 (needs nan_flag): the kernel still RAISES nan_flag but the call does not launch the rule's rewrite; the caller hands the
   * rewrite to the consumer that reads `out` next -- qavit_bank_stats_nanfix with the matching qavit_nan_fix -- which does it for the
   * images it visits before it reads them (one launch less per branch on the forward critical path) */
  int nan_defer;
} qavit_branch_args;

int qavit_branch_supported(int kind, int T, int C, int H, int D, int KC, int S, int L);   /* 1 if qavit_branch_fwd covers the shape */
int qavit_branch_fwd(const qavit_branch_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * TokenLearner mixing (HQAViT_CIFAR100.py:996-1000): p = softmax over the N axis of scores[B,N,M];
 * xc[B,M,C] = p^T x.  bwd: dx (direct path) and dscores.
 * ------------------------------------------------------------------------------------------------- */
int qavit_tokmix_fwd(int dtype, const void* scores, const void* x, void* p, void* xc,
                     int B, int N, int M, int C, void* stream);
int qavit_tokmix_bwd(int dtype, const void* p, const void* x, const void* dxc, void* dx, void* dscores,
                     int B, int N, int M, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * TokenUpMix (HQAViT_CIFAR100.py:1026-1029): up[b,n,c] = sum_m W[n,m] xc[b,m,c] + bias[n]; y = LN_C(up).
 * bwd: dxc, dW[N,M], dbias[N], dgamma, dbeta (fp32 atomics).
 * ------------------------------------------------------------------------------------------------- */
int qavit_upmix_fwd(int dtype, const void* xc, const float* W, const float* bias, const float* gamma,
                    const float* beta, float eps, void* y, float* mean, float* rstd,
                    int B, int N, int M, int C, void* stream);
int qavit_upmix_bwd(int dtype, const void* dy, const void* xc, const float* W, const float* bias,
                    const float* gamma, const float* mean, const float* rstd, void* dxc, float* dW,
                    float* dbias, float* dgamma, float* dbeta, int B, int N, int M, int C, void* stream);
/* The same with the parameter gradients left as PARTIAL ROWS: `parts` = float[qavit_upmix_bwd_parts(...)][N*M + N + 2*C], 16-byte
 * aligned, one row [dW | dbias | dgamma | dbeta] per workgroup written with plain stores; dW / dbias / dgamma / dbeta are not
 * touched -- fold the rows with qavit_ln_param_reduce (stride N*M + N + 2*C + 4: each row ends in four floats [dgamma_sa, 0, 0, 0],
 * the layer scale's gradient of qavit_upmix_bwd_sa, zero otherwise).  The 256-deep same-address float atomics this replaces
 * were the larger half of the kernel.  qavit_upmix_bwd_parts() == 0: no partial-row path for this dtype / shape (pass parts = NULL). */
int qavit_upmix_bwd_parts(int dtype, int B, int N, int M, int C);
#define QAVIT_UPMIX_PART_ROW(N, M, C) ((N) * (M) + (N) + 2 * (C) + 4)
/* The up-mix backward that ALSO differentiates the scale-add in front of it, xc = x + droppath(gamma * u) (the block tail,
 * HQAViT_CIFAR100.py:1085 then :1118-1121): besides dxc (= dx) it writes du = dxc * f * gamma[0] and adds sum(dxc * f * u) to dgamma_sa
 * (one float atomic per workgroup; with `parts` the workgroup's sum goes to the last four floats of its partial row instead and dgamma_sa
 * is not touched: reduce descriptor C = 1), f = the image's drop-path factor (dp_p, dp_site, rng; samples = images).  u, du [B*M, C].  bf16,
 * N = 64, M = 16, C = 192 (qavit_upmix_bwd_sa_supported). */
int qavit_upmix_bwd_sa_supported(int dtype, int N, int M, int C);
int qavit_upmix_bwd_sa(int dtype, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma,
                       const float* mean, const float* rstd, void* dxc, float* dW, float* dbias, float* dgamma, float* dbeta,
                       int B, int N, int M, int C, float* parts, const void* u, void* du, const float* gamma_sa, float* dgamma_sa,
                       float dp_p, int dp_site, const int64_t* rng, void* stream);
/* The up-mix FORWARD with the same scale-add in front of it: forms xc = x + f * gamma_sa[0] * u (f = the image's drop-path factor; the
 * arithmetic of qavit_scale_add_fwd) while it stages each image, writes it to `xc` [B*M, C] for the backward and up-mixes it -- one launch
 * for the two.  bf16, C = 192, (N, M) = (64, 16) or (256, 64) (qavit_upmix_fwd_sa_supported); x, u, xc, y 8-byte aligned. */
int qavit_upmix_fwd_sa_supported(int dtype, int N, int M, int C);
int qavit_upmix_fwd_sa(int dtype, const void* x, const void* u, const float* gamma_sa, float dp_p, int dp_site, const int64_t* rng, void* xc,
                       const float* W, const float* bias, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                       int B, int N, int M, int C, void* stream);
int qavit_upmix_bwd_p(int dtype, const void* dy, const void* xc, const float* W, const float* bias,
                      const float* gamma, const float* mean, const float* rstd, void* dxc, float* dW,
                      float* dbias, float* dgamma, float* dbeta, int B, int N, int M, int C, float* parts, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * TokenLearner (HQAViT_CIFAR100.py:971-1002) as ONE launch each way: scores = Linear(LayerNorm(x)) [B, N, M], P = softmax over the N
 * tokens, xc = P^T x [B, M, C].  It replaces the LayerNorm-prologue qavit_gemm_nt + qavit_tokmix_fwd forward and qavit_tokmix_bwd +
 * qavit_layernorm_bwd_lin + the score Linear's deferred qavit_gemm_tn problem backward: an image's N x C token tile is read once per
 * direction.  bf16, (N, M, C) = (64, 16, 192) (qavit_tl_supported).  x 16-byte aligned; W = the [M, C] compute-dtype copy of the score
 * weight (row-major), bias fp32 [M] or NULL.
 * fwd writes p (bf16 [B, N, M], saved), xc, and the row statistics mean / rstd [B * N] (saved).
 * bwd takes dxc and writes dx = dx(LayerNorm path) + dx(mixing path) and ONE partial row per workgroup,
 *   parts = float[qavit_tl_bwd_parts(B, N, M)][M*C + M + 2*C] = [dW (M x C) | dbias (M) | dgamma (C) | dbeta (C)], 16-byte aligned,
 * to be folded into the parameter gradients by qavit_ln_param_reduce (stride M*C + M + 2*C); no gradient buffer is touched here. */
typedef struct qavit_tl_args {
  const void* x; const float* ln_g; const float* ln_b; float eps;
  const void* W; const float* bias;
  void* p; void* xc; float* mean; float* rstd;
  int B, N, M, C;
} qavit_tl_args;
typedef struct qavit_tl_bwd_args {
  const void* dxc; const void* x; const void* p; const float* mean; const float* rstd;
  const float* ln_g; const float* ln_b; const void* W;
  void* dx; float* parts;
  int B, N, M, C;
} qavit_tl_bwd_args;
int qavit_tl_supported(int dtype, int N, int M, int C);
int qavit_tl_fwd(const qavit_tl_args* a, void* stream);
int qavit_tl_bwd_parts(int B, int N, int M);
int qavit_tl_bwd(const qavit_tl_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * MSDA landmark tokens (HQAViT_CIFAR100.py:499-501): dilated gathers x[:, ::d, ::d] concatenated, then
 * AvgPool1d(stride, stride) over the token axis.  idx[NP*stride] are source-token indices (host-built,
 * device-resident): pooled[b,j,:] = mean_s x[b, idx[j*stride+s], :].
 * ------------------------------------------------------------------------------------------------- */
int qavit_gather_pool_fwd(int dtype, const void* x, const int32_t* idx, void* y, int B, int N, int NP,
                          int stride, int C, void* stream);
int qavit_gather_pool_bwd(int dtype, const void* dy, const int32_t* idx, void* dx, int B, int N, int NP,
                          int stride, int C, void* stream);
/* ... with dx rows ldx elements apart (a column slice of a wider matrix: MSDA's landmark-path dk / dv gradients scattered back to token
 * rows inside the fan node's [rows, 7C] matrix, functional.FanGroup); vector kernel only (16-byte aligned, C and ldx multiples of 8 / 4). */
int qavit_gather_pool_bwd_ld(int dtype, const void* dy, const int32_t* idx, void* dx, int ldx, int B, int N, int NP,
                             int stride, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * CCF-FFN middle (HQAViT_CIFAR100.py:706-708, :670-675): h2 = LN2( scale * dwconv3x3( LN1(h) ) (+bias) )
 * on [B, Hs*Ws, C] channel-last tokens.  flags bit0: LN1/LN2 present (v1 has none), bit1: conv bias,
 * bit2: per-channel scale.  bwd accumulates all parameter grads with fp32 atomics, or leaves per-workgroup partial rows (parts).
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_ccf_args {
  int dtype; int flags;
  int B, Hs, Ws, C;
  const void* h; void* out;
  const float* g1; const float* b1; const float* g2; const float* b2; float eps;
  const float* w;      /* [C,1,3,3] */
  const float* cbias;  /* [C] or NULL */
  const float* cscale; /* [C] or NULL */
  float* mean1; float* rstd1; float* mean2; float* rstd2;  /* [B*Hs*Ws] saved stats */
  /* backward */
  const void* d_out; void* d_h;
  float* dg1; float* db1; float* dg2; float* db2; float* dw; float* dcbias; float* dcscale;
  float* parts;        /* optional (bwd, C <= 256, >= 15 tokens, C % 8 == 0): qavit_ccf_bwd_parts(B) rows of 15*C floats
                        * [dg1 | db1 | dg2 | db2 | dcbias | dcscale | dw (9C)], one per workgroup, written with plain stores INSTEAD of the
                        * atomics into the seven gradient buffers; fold them with qavit_ln_param_reduce (stride 15*C) */
} qavit_ccf_args;
int qavit_ccf_bwd_parts(int B);

int qavit_ccf_mid_fwd(const qavit_ccf_args* a, void* stream);
int qavit_ccf_mid_bwd(const qavit_ccf_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused branch BACKWARD, first half (csrc/branch_bwd.hip): the proj input-gradient GEMM and the attention-core backward of a
 * branch that went forward through qavit_branch_fwd with q_save / kv_save / o_save, in one launch (it replaces qavit_gemm_nt on
 * Wproj^T, qavit_attn_bwd and its reduction).  Same fixed shapes as qavit_branch_args (T = 16 or 64); bf16 only.
 *   in : dout [B*T, C] (gradient of the branch output), wprojT_frag = the proj weight TRANSPOSED in fragment order
 *        (qavit_pack_desc.pad = 3), q / k_tok / v_tok / o as saved by the forward (k_tok, v_tok: kv_rows rows per image, NULL for
 *        kind 2), E_k / E_v / sh_k / sh_v with their FORWARD-TIME values, the two dropout (p, site) pairs and rng of the forward.
 *   out: dz = dout * proj-dropout mask (operand of dW_proj; may be NULL when proj_drop_p == 0: dz == dout), dq [B*T, *],
 *        dk_tok / dv_tok [B*kv_rows, *] (the caller runs the qkv input-gradient GEMM and the weight-gradient GEMMs on them),
 *        parts: qavit_branch_bwd_parts(B) rows of parts_stride >= QAVIT_BRANCH_PARTS_FLOATS floats, one per workgroup:
 *        [ dE_k 16x32 | dE_v 16x32 | d sh_k 16x192 | d sh_v 16x192 ] partial sums (plain stores; fold the rows with
 *        qavit_ln_param_reduce, stride = parts_stride).  Kind 2 leaves the dE part unwritten.
 * ------------------------------------------------------------------------------------------------- */
#define QAVIT_BRANCH_PARTS_FLOATS 7168
#define QAVIT_BRANCH_PARTS_FLOATS_64 9216   /* T = 64: [ dE_k 48x32 | dE_v 48x32 | d sh_k 16x192 | d sh_v 16x192 ] (SWA uses the first 16 rows of the dE slots) */
typedef struct qavit_branch_bwd_args {
  int dtype; int kind;
  int B, T, C, H, D, KC, S, L;
  const void* dout; int64_t lddout;
  const void* wprojT_frag;
  const void* q; int64_t ldq;
  const void* k_tok; const void* v_tok; int64_t ldkv; int kv_rows;
  const void* o; int64_t ldo;
  const float* E_k; const float* E_v; const float* sh_k; const float* sh_v;
  float attn_drop_p; int attn_drop_site; float proj_drop_p; int proj_drop_site; const int64_t* rng;
  void* dz; int64_t lddz;
  void* dq; int64_t lddq;
  void* dk_tok; void* dv_tok; int64_t lddkv;
  float* parts; int64_t parts_stride;
  const int* nan_trip;   /* optional: the word qavit_branch_fwd's NaN-rule launch wrote (qavit_branch_args.nan_trip); != 0: the forward returned
                          * proj(0), so dq = dk = dv = 0 and every partial sum is 0 (HQAViT_CIFAR100.py:356-357, :394-395); dz is still written */
} qavit_branch_bwd_args;
int qavit_branch_bwd_parts(int B, int T);   /* workgroups = rows of `parts`: ceil(B / 4) for T = 16, B for T = 64 */
int qavit_branch_bwd(const qavit_branch_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused CHANNEL-GROUP attention branch (EfficientChannelGroupAttention, HQAViT_CIFAR100.py:535-595) on the 16-learned-token
 * problems: per image, for each of the G = 6 channel groups (32 channels) q / k / v = Linear(32 -> 16) of the group's slice of x,
 * 4 heads of D = 4 over [16 token keys ; 16 projected bank rows], softmax (+ attention dropout), P.V, then proj(96 -> 192) + bias
 * (+ dropout) over the concatenated groups -- ONE launch instead of qkv GEMM, attention kernel, NaN guard and proj GEMM.
 * Fixed shapes: T = 16 or 64 (csrc/cga64.hip: 64 token keys + 16 bank rows per (group, head)), C = 192, G = 6, H = 4, D = 4, S = 16; bf16.  wqkv_rm = [Wq; Wk; Wv] stacked [48, 32] bf16 row-major,
 * wproj_rm = [192, 96] bf16 row-major (both from qavit_pack_weights), biases fp32, sh_k / sh_v = the bank projections [16, 16] fp32.
 * Dropout contracts as the unfused chain (attention: problem id = (image * 6 + group) * 4 + head; proj: row * C + col);
 * nan_flag as in qavit_branch_args.  o_save (optional) [B*16, 96]: the attention output, operand of backward's dW_proj.
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_cga_args {
  int dtype;
  int B, T, C, G, H, D, S;
  const void* x; int64_t ldx;
  const void* wqkv_rm; const float* bqkv;
  const void* wproj_rm; const float* bproj;
  const float* sh_k; const float* sh_v;
  void* out; int64_t ldo;
  void* o_save;
  float attn_drop_p; int attn_drop_site; float proj_drop_p; int proj_drop_site; const int64_t* rng;
  int* nan_flag;
  int* nan_trip;   /* optional, as in qavit_branch_args: 1 / 0 written by the NaN-rule launch; a tripped call also zeroes o_save */
  int nan_defer;   /* as in qavit_branch_args */
} qavit_cga_args;
int qavit_cga_supported(int T, int C, int G, int H, int S);
int qavit_cga_fwd(const qavit_cga_args* a, void* stream);
/* Backward of the same branch in one launch (it replaces the proj dX GEMM, a q/k/v recompute GEMM, qavit_attn_bwd + its reduction and
 * the q/k/v dX GEMM): everything is recomputed from x.  in: dout [B*16, 192]; x; wqkv_rm and its transpose wqkvT_rm [32, 48];
 * wprojT_rm = the proj weight transposed [96, 192] (row-major bf16, the W^T copies qavit_pack_weights makes); bqkv; sh_k / sh_v with
 * their forward-time values; the forward's dropout (p, site) pairs and rng.  out: dz = dout * proj mask (operand of dW_proj; may be
 * NULL without proj dropout), dqkv [B*16*6, 48] (rows = (image, token, group); operand of the q/k/v weight gradients), dx [B*16, 192],
 * parts: qavit_cga_bwd_parts(B) rows of QAVIT_CGA_PARTS_FLOATS = [d sh_k 16x16 | d sh_v 16x16] (fold with qavit_ln_param_reduce). */
#define QAVIT_CGA_PARTS_FLOATS 512
typedef struct qavit_cga_bwd_args {
  int dtype;
  int B, T, C, G, H, D, S;
  const void* dout; int64_t lddout;
  const void* x; int64_t ldx;
  const void* wqkv_rm; const void* wqkvT_rm; const float* bqkv;
  const void* wprojT_rm;
  const float* sh_k; const float* sh_v;
  float attn_drop_p; int attn_drop_site; float proj_drop_p; int proj_drop_site; const int64_t* rng;
  void* dz; int64_t lddz;
  void* dqkv;
  void* dx; int64_t lddx;
  float* parts;
  const int* nan_trip;   /* optional: the forward's nan_trip word; != 0: dqkv = dx = 0 and zero partial sums (dz is still written) */
} qavit_cga_bwd_args;
int qavit_cga_bwd_parts(int B, int T);   /* workgroups = rows of `parts`: ceil(B / 4) for T = 16, B for T = 64 */
int qavit_cga_bwd(const qavit_cga_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * BottleneckMLP + residual of a QuadAttentionBlock (HQAViT_CIFAR100.py:643-656, :1082-1083) in one launch each way (csrc/mlp2.hip):
 *   out = resid + drop_path( dropout2( (dropout1(GELU(y W1^T + b1))) W2^T + b2 ) )      W1 [Hd, C], W2 [C, Hd], C = 192, Hd = 96; bf16
 * It replaces two qavit_gemm_nt launches (fc1 with GELU + dropout epilogue, fc2 with dropout + drop-path + residual epilogue) and keeps
 * their contracts: dropout masks drop_factor(key(site), row * N + col) with N = Hd (site 1) / C (site 2), drop path row / dp_rows;
 * w1_rm / w2_rm are the bf16 row-major copies qavit_pack_weights makes.  z1 (pre-GELU) and h1 (post-dropout) [M, Hd] are written for
 * the backward pass when given (both or neither): h1 is the operand of dW2, z1 feeds GELU'.
 * Backward: g = d out [M, C] -> dz2 = g * drop_path * dropout2 (operand of dW2 with h1; may be NULL without either mask: dz2 == g),
 * dz1 [M, Hd] = (dz2 W2) * dropout1 * GELU'(z1) (operand of dW1 with y), dy [M, C] = dz1 W1.  d resid == g (the caller aliases it).
 * ------------------------------------------------------------------------------------------------- */
typedef struct qavit_mlp2_args {
  int dtype; int M, C, Hd;
  const void* y; int64_t ldy;
  const void* resid; int64_t ldr;
  const void* w1_rm; const float* b1;
  const void* w2_rm; const float* b2;
  float drop1_p; int drop1_site; float drop2_p; int drop2_site; float dp_p; int dp_site; int dp_rows; const int64_t* rng;
  void* out; int64_t ldo;
  void* z1; void* h1;
} qavit_mlp2_args;
typedef struct qavit_mlp2_bwd_args {
  int dtype; int M, C, Hd;
  const void* g; int64_t ldg;
  const void* z1;
  const void* w1_rm; const void* w2_rm;
  float drop1_p; int drop1_site; float drop2_p; int drop2_site; float dp_p; int dp_site; int dp_rows; const int64_t* rng;
  void* dz2; void* dz1; void* dy; int64_t lddy;
} qavit_mlp2_bwd_args;
int qavit_mlp2_supported(int C, int Hd);
int qavit_mlp2_fwd(const qavit_mlp2_args* a, void* stream);
int qavit_mlp2_bwd(const qavit_mlp2_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * QuadAttentionBlock's HybridFusion(concat_i compress_i(norm_i(branch_i))) (HQAViT_CIFAR100.py:904-925, :1075-1081), forward, in one
 * launch for 16 tokens x 192 channels, 4 branches of Linear(192 -> 48): x[i] [B*16, 192] bf16 (contiguous rows), w_rm[i] [48, 192]
 * bf16 row-major, bias[i] fp32 or NULL, fw = the 4 fusion logits.  Writes cat [B*16, 192] (unscaled concat), y = cat * softmax(fw)
 * per branch slice, and the norms' mean[i] / rstd[i] [B*16] -- what the unfused forward (qavit_row_stats_multi, the LayerNorm-prologue
 * qavit_gemm_nt_grouped, qavit_hybrid_fuse_fwd) leaves for the backward.
 * ------------------------------------------------------------------------------------------------- */
/* The parameters of a fused branch call's NaN -> zeros rule, handed to the launch that reads the branch output next (nan_defer in
 * qavit_branch_args / qavit_cga_args; qavit_bank_stats_nanfix and qavit_cfuse_args.fix below). */
typedef struct qavit_nan_fix {
  int* flag;                 /* int[2]: the branch call's nan_flag (flag, arrival ticket) */
  int* trip;                 /* optional: the branch call's nan_trip */
  const float* bias;         /* [C]: the branch's proj bias */
  float drop_p; int drop_site; const int64_t* rng;   /* the branch's proj dropout */
  void* o_save; int64_t ldos; int Co;                /* optional: the saved attention output [B*N, ldos], Co columns, zeroed */
} qavit_nan_fix;
typedef struct qavit_cfuse_args {
  int dtype;
  int B, T, C, NB, CB;
  const void* x[4];
  const float* gamma[4]; const float* beta[4];
  const void* w_rm[4]; const float* bias[4];
  const float* fw; float eps;
  void* cat; void* y;
  float* mean[4]; float* rstd[4];
  /* fix.flag != NULL: branch `fix_branch`'s call was made with nan_defer and this launch is the next reader of its output x[fix_branch]
   * (the cross branch, which writes no bank): with the flag raised every workgroup first rewrites its images' rows of that operand as
   * dropout(bias), zeroes their o_save rows, and the launch sets `trip` and resets the flag / ticket words -- what the rule's own
   * launch does (qavit_branch_nan_fix) -- then normalises the rewritten rows. */
  qavit_nan_fix fix; int fix_branch;
} qavit_cfuse_args;
int qavit_compress_fuse_supported(int T, int C, int nb, int Cb);
int qavit_compress_fuse_fwd(const qavit_cfuse_args* a, void* stream);
/* Backward of the same node in one launch (it replaces qavit_hybrid_fuse_bwd, the grouped input-gradient qavit_gemm_nt and
 * qavit_layernorm_bwd_multi): dcat = dy * softmax(fw) per slice (written: operand of the compress weight gradients, which stay deferred
 * qavit_gemm_tn problems with LayerNorm-on-load), dx[i] = LayerNorm_i'(dcat_i W_i), and one row of QAVIT_CFUSE_PARTS_FLOATS partial
 * sums per workgroup (qavit_compress_fuse_bwd_parts(B) rows): [4 x (dgamma_i 192 | dbeta_i 192) | dfw 4 | 4 zeros]. */
#define QAVIT_CFUSE_PARTS_FLOATS 1544
typedef struct qavit_cfuse_bwd_args {
  int dtype;
  int B, T, C, NB, CB;
  const void* dy; const void* cat;
  const void* x[4];
  const float* gamma[4];
  const void* w_rm[4];
  const float* mean[4]; const float* rstd[4];
  const float* fw;
  void* dcat;
  void* dx[4];
  float* parts;
} qavit_cfuse_bwd_args;
int qavit_compress_fuse_bwd_parts(int B);
int qavit_compress_fuse_bwd(const qavit_cfuse_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Depthwise k x k convolution, stride 1, pad k/2, on channel-last tokens x[B, H*W, C] (k in {3,5,7}):
 * ConvNeXtBlock.dwconv (HQAViT_CIFAR100.py:722, dw7x7), LMFAdapter.dwconv_3x3 / dwconv_5x5 (:811-812).
 * w is the nn.Conv2d weight [C,1,k,k] fp32; bias [C] or NULL.  bwd: dx, dw += , dbias += (fp32 atomics).
 * ------------------------------------------------------------------------------------------------- */
int qavit_dwconv_fwd(int dtype, const void* x, const float* w, const float* bias, void* y,
                     int B, int H, int W, int C, int ks, void* stream);
int qavit_dwconv_bwd(int dtype, const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias,
                     int B, int H, int W, int C, int ks, void* stream);
/* The same kernels on column slices of wider buffers (LMFAdapter's cat([dw3(x), dw5(x), x]) and its gradient, :830-834, without the
 * cat / slice copies): y rows are ldy elements apart; dy rows lddy apart; dadd (NULL, or rows lddadd apart; may be dx itself) is another
 * gradient that meets this one at x and is added into dx.  Strides in elements, >= C; needs H and W multiples of 8 when they differ from C. */
int qavit_dwconv_fwd_ld(int dtype, const void* x, const float* w, const float* bias, void* y, int ldy,
                        int B, int H, int W, int C, int ks, void* stream);
/* ... and x itself copied into a third column slice by the same launch (xcopy rows ldc apart; the cat's pass-through member, :834): one launch
 * fewer per LMFAdapter.  H and W multiples of 8. */
int qavit_dwconv_fwd_ld2(int dtype, const void* x, const float* w, const float* bias, void* y, int ldy, void* xcopy, int ldc,
                         int B, int H, int W, int C, int ks, void* stream);
int qavit_dwconv_bwd_ld(int dtype, const void* dy, int lddy, const void* x, const float* w, void* dx, const void* dadd, int lddadd,
                        float* dw, float* dbias, int B, int H, int W, int C, int ks, void* stream);

/* im2col / col2im for the strided 3x3 stem convolutions (HQAViT_CIFAR100.py:752, :759) so that they run on
 * qavit_gemm_nt: cols[(b,oy,ox), c*k*k+dy*k+dx] = src[b, c, oy*s+dy-p, ox*s+dx-p].  src is the fp32 NCHW image
 * (src_nchw_f32 != 0) or channel-last tokens [B,H*W,Cin] in `dtype`; col2im scatters dcols back to tokens. */
int qavit_im2col(int dtype, const void* src, int src_nchw_f32, void* cols, int B, int Cin, int H, int W,
                 int k, int stride, int pad, void* stream);
/* The same with rows of `ld` >= Cin*k*k elements (pad columns written as zeros): a K of 27 (3 channels x 3x3) becomes 32-element,
 * 64-byte rows that the GEMMs read with aligned 16-byte loads. */
int qavit_im2col_ld(int dtype, const void* src, int src_nchw_f32, void* cols, int ld, int B, int Cin, int H, int W,
                    int k, int stride, int pad, void* stream);
int qavit_col2im(int dtype, const void* dcols, void* dx, int B, int Cin, int H, int W, int k, int stride, int pad, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * GlobalTokenBank.write (HQAViT_CIFAR100.py:296-321; QAViT.py:205-224), train mode only, no gradient.
 * stats: acc[S,C] = sum_b  softmax_tokens( tn Wg^T + bg )^T tn,  tn = LN_write(LN_branch(tokens))
 *        (per-wave partials in `ws`, then one reduction; under data parallelism `acc` is what gets
 *        all-reduced before apply -- SURVEY.md section 8e exception 1)
 * apply: U = acc / B_total; upd_v = clamp(U); upd_k = clamp(U Wc^T + bc)  (exact: softmax columns sum to 1);
 *        bank += rate*upd; clamp; update_count += 1; acc := 0.   mode 0 = HQA rule (rate by count, clamps
 *        0.05/0.5), mode 1 = QAViT.py rule (rate 0.01, clamps 0.1/1.0, no counter).
 * Contract on `acc`: float[S*C + 1], ZERO before the first stats call; apply consumes it and leaves it zero again (the
 * extra word is apply's arrival ticket for the once-per-write counter increment), so a write is stats -> apply, no memset.
 * ------------------------------------------------------------------------------------------------- */
int qavit_bank_stats(int dtype, const void* tokens, const float* g_branch, const float* b_branch,
                     const float* g_write, const float* b_write, const float* Wg, const float* bg,
                     float* acc, float* ws, int64_t ws_floats, int B, int N, int C, int S, float eps, void* stream);
int64_t qavit_bank_ws_floats(int B, int N, int C, int S);
/* The NaN -> zeros rule of a fused branch call made with nan_defer, carried out by the bank write that reads the branch output next
 * (HQAViT_CIFAR100.py:356-357, :394-395 then :296-321).  With `flag` raised, every image's rows of `tokens` are first rewritten as
 * dropout(bias) (what proj(zeros) gives), its o_save rows zeroed, `trip` set and the flag / ticket words reset -- exactly what the
 * rule's own launch does -- and the statistics are taken of the rewritten rows.  bf16 [B, N, 192] tokens with N = 16 or 64 do this
 * inside the statistics kernel; every other shape runs the rule's launch first. */
/* (struct qavit_nan_fix: declared above, in front of qavit_cfuse_args) */
/* The rule's own launch for a deferred call whose next reader cannot carry it: out [rows, C] rewritten when fix->flag is raised. */
int qavit_branch_nan_fix(int dtype, void* out, int rows, int C, const qavit_nan_fix* fix, void* stream);
int qavit_bank_stats_nanfix(int dtype, void* tokens, const float* g_branch, const float* b_branch,
                            const float* g_write, const float* b_write, const float* Wg, const float* bg,
                            float* acc, float* ws, int64_t ws_floats, int B, int N, int C, int S, float eps,
                            const qavit_nan_fix* fix, void* stream);
/* acc == NULL in bank_stats: the per-workgroup partials stay in ws ([nparts = ws_floats / (S*C)][S][C]) and bank_apply folds
 * them itself when given `parts` (single-GPU write = stats -> apply, fixed summation order).  With acc, stats also reduces
 * into it (the data-parallel path all-reduces acc between the two calls) and apply is called with parts = NULL.
 * snap_k / snap_v (optional, float[S*C] each): the new rows are ALSO written there -- the copy the next attention branch's backward
 * needs (the reference's torch.cat / Linear-on-expand copies, HQAViT_CIFAR100.py:398-399, :576-577) without a copy launch. */
int qavit_bank_apply(float* acc, const float* Wc, const float* bc, float* bank_k, float* bank_v,
                     int64_t* update_count, int S, int C, float inv_batch, int mode,
                     const float* parts, int nparts, float* snap_k, float* snap_v, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * small helpers
 * ------------------------------------------------------------------------------------------------- */
/* 4x4/stride-p patch gather for the patch-embed conv as a GEMM (HQAViT_CIFAR100.py:1133,1137):
 * cols[b*NP + py*Wp + px, c*p*p + dy*p + dx] = img[b,c,py*p+dy,px*p+dx] (img fp32 NCHW) */
int qavit_patchify(int dtype, const float* img, void* cols, int B, int Cin, int H, int W, int p, void* stream);
/* y[b,:] = mean_n x[b,n,:] (HQAViT_CIFAR100.py:1274) and its backward */
int qavit_token_mean_fwd(int dtype, const void* x, void* y, int B, int N, int C, void* stream);
int qavit_token_mean_bwd(int dtype, const void* dy, void* dx, int B, int N, int C, void* stream);
/* HybridFusion (HQAViT_CIFAR100.py:637-640): y[:, i*Cb:(i+1)*Cb] = x[:, i*Cb:(i+1)*Cb] * softmax(fw)[i];
 * bwd: dx and dfw[nb] += (through the softmax)
 * NARROW PARTIAL ROWS (this entry point, qavit_mix2_bwd, qavit_mix3_bwd, qavit_scale_add_bwd): a parameter gradient of 1-4 values summed
 * over the whole tensor.  part_ws == NULL: every workgroup adds its contribution with float atomics (summation order, hence the last
 * bits, vary from run to run).  part_ws = float[QAVIT_NARROW_PARTS_MAX][row], 16-byte aligned, with nparts != NULL: workgroup b stores
 * its contribution as row b with plain stores, the parameter gradient is NOT touched and *nparts (host) receives the number of rows
 * written; fold them with qavit_ln_param_reduce (a narrow descriptor: fixed order, bit-reproducible).  Row = 4 floats, descriptor
 * C = 1 (scale_add: [dgamma, 0, 0, 0]) or C = 2 (mix2 / mix3: [dfw0, dfw1, 0, 0]); hybrid_fuse: 8 floats, C = 4 (nb <= 4; more
 * branches keep the atomics and report *nparts = 0). */
#define QAVIT_NARROW_PARTS_MAX 1024
int qavit_hybrid_fuse_fwd(int dtype, const void* x, const float* fw, void* y, int rows, int nb, int Cb, void* stream);
int qavit_hybrid_fuse_bwd(int dtype, const void* dy, const void* x, const float* fw, void* dx, float* dfw,
                          int rows, int nb, int Cb, float* part_ws, int* nparts, void* stream);
/* out = xs[0] + ... + xs[k-1] (k <= 8 same-shape tensors, host array of device pointers): the gradient of norm1's output,
 * which feeds five consumers in QuadAttentionBlock (HQAViT_CIFAR100.py:1072-1078), summed in one pass. */
int qavit_sum_k(int dtype, const void* const* xs, int k, void* out, int64_t n, void* stream);
/* perm[0..B) = a uniform random permutation drawn from the counter RNG (rng[0] seed, rng[1] step, site): the
 * torch.randperm of train_epoch (HQAViT_CIFAR100.py:1383,1395) as one capturable kernel; B <= 16384. */
int qavit_rand_perm(int64_t* perm, int B, const int64_t* rng, int site, void* stream);
/* Device-side CutMix / MixUp of an fp32 NCHW batch (train_epoch, HQAViT_CIFAR100.py:1381-1399).  plan = device float[6]:
 * mode (0 none, 1 cutmix, 2 mixup), lambda, x1, y1, x2, y2 (the box of rand_bbox :1339-1363); perm = device int64[B].
 * out[b] = x[b] with the box pasted from x[perm[b]] (cutmix) or lambda*x[b] + (1-lambda)*x[perm[b]] (mixup).  out != x. */
int qavit_mix_apply(const float* x, const int64_t* perm, const float* plan, float* out, int B, int C, int H, int W, void* stream);
/* SplitFusion gate (HQAViT_CIFAR100.py:945-949): y = t + sigmoid(g) * r (n elements, same dtype, 16-byte aligned).
 * bwd: dt = dy (the caller passes it through), dr = dy * sigmoid(g), dg = dy * r * sigmoid(g) * (1 - sigmoid(g)). */
int qavit_gate_mix_fwd(int dtype, const void* t, const void* r, const void* g, void* y, int64_t n, void* stream);
int qavit_gate_mix_bwd(int dtype, const void* dy, const void* r, const void* g, void* dr, void* dg, int64_t n, void* stream);
/* SplitFusion blend (HQAViT_CIFAR100.py:959-963): y = s0*a + s1*b with s = softmax(fw[0..1]); n = element count (a multiple
 * of the 16-byte vector, operands 16-byte aligned).  bwd: da = s0*dy, db = s1*dy, dfw[2] += (through the softmax; may be NULL) */
int qavit_mix2_fwd(int dtype, const void* a, const void* b, const float* fw, void* y, int64_t n, void* stream);
int qavit_mix2_bwd(int dtype, const void* dy, const void* a, const void* b, const float* fw, void* da, void* db, float* dfw,
                   int64_t n, float* part_ws, int* nparts, void* stream);
/* The blend with its second operand built in place (HQAViT_CIFAR100.py:953-963): y = s0*a + s1*(t + dropout(h)) -- dropout with
 * the library's counter-based mask (drop_p, drop_site, rng as qavit_dropout; element index = flat index).  bwd: da = s0*dy,
 * dt = s1*dy, dh = dt * mask, dfw[2] += (may be NULL).  Same size / alignment rules as qavit_mix2_*; n < 2^32. */
int qavit_mix3_fwd(int dtype, const void* a, const void* t, const void* h, const float* fw, void* y, int64_t n,
                   float drop_p, int drop_site, const int64_t* rng, void* stream);
int qavit_mix3_bwd(int dtype, const void* dy, const void* a, const void* t, const void* h, const float* fw, void* da, void* dt, void* dh,
                   float* dfw, int64_t n, float drop_p, int drop_site, const int64_t* rng, float* part_ws, int* nparts, void* stream);
/* y = x + droppath( gamma[0] * u )  (CCF-FFN gamma + residual, HQAViT_CIFAR100.py:712,1083); gamma may be NULL (=1) */
int qavit_scale_add_fwd(int dtype, const void* x, const void* u, const float* gamma, void* y, int rows, int C,
                        float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream);
int qavit_scale_add_bwd(int dtype, const void* dy, const void* u, const float* gamma, void* du, float* dgamma,
                        int rows, int C, float dp_p, int dp_site, int dp_rows, const int64_t* rng, float* part_ws, int* nparts, void* stream);
/* nn.BatchNorm2d (+ optional exact GELU, act = 1) of the CNN stem on channel-last rows x[M = B*H*W, C]
 * (HQAViT_CIFAR100.py:753,760,768,775).  training != 0: batch statistics (biased variance for the output, unbiased
 * for running_var), running_mean / running_var updated in place with `momentum`, save_mean / save_rstd [C] written
 * for backward; ws = float[3*C] scratch.  training == 0: normalises with the running statistics.
 * bwd: dx, and dgamma / dbeta [C] ACCUMULATE (either may be NULL); ws = float[2*C] scratch; `training` as in forward
 * (0: save_mean / save_rstd hold the running mean and rsqrt(running_var + eps), treated as constants).
 * C must be a multiple of 8 (bf16) / 4 (fp32) with 256 % (C/vec) == 0; rows 16-byte aligned.
 * Data-parallel "exact" statistics (SyncBN, SURVEY.md 8e exception 2): both directions are two passes over ws, so the
 * caller may run them apart -- phase 1 = statistics only (ws[0..2C) holds this rank's column sums), all-reduce (SUM)
 * ws[0..2C) across ranks, phase 2 = apply with M_total = the rows of all ranks; phase 0 = both passes back to back
 * (M_total <= 0 means M).  Backward's dgamma / dbeta must stay this rank's own sums (the gradient all-reduce adds the
 * other ranks'): pass the pre-all-reduce copy as ws_param (NULL = ws). */
int qavit_bn_fwd(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float momentum, float eps, int act,
                 float* save_mean, float* save_rstd, float* ws, int training, int phase, int64_t M_total, void* stream);
int qavit_bn_bwd(int dtype, const void* dy, const void* x, int M, int C, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_rstd, int act, int training, void* dx, float* dgamma, float* dbeta,
                 float* ws, int phase, int64_t M_total, const float* ws_param, void* stream);
/* nn.LayerNorm([C,H,W]) of the ConvNeXt-Tiny style stem (HQAViTv2_CIFAR100.py:766, :777, :791) on channel-last tokens
 * x [B, N=H*W, C]: each SAMPLE is normalised over its N*C elements; w / b keep the reference's [C][N] layout.  N*C must be
 * 4096, 8192 or 16384, C % 4 == 0, rows 16-byte aligned.  mean / rstd [B] are written by fwd and read by bwd; bwd writes
 * dx and ACCUMULATES dw / db ([C][N] fp32). */
int qavit_spatial_ln_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                         int B, int N, int C, float eps, void* stream);
int qavit_spatial_ln_bwd(int dtype, const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                         void* dx, float* dw, float* db, int B, int N, int C, void* stream);
/* ConvNeXt layer scale + drop path + residual on rows [rows, C]: y = x + droppath(gamma[c] * u)
 * (HQAViTv2_CIFAR100.py:744-748).  bwd: du = dy * mask * gamma[c]; dgamma [C] ACCUMULATES; dx = dy is the caller's. */
int qavit_chan_scale_add_fwd(int dtype, const void* x, const void* u, const float* gamma, void* y, int rows, int C,
                             float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream);
int qavit_chan_scale_add_bwd(int dtype, const void* dy, const void* u, const float* gamma, void* du, float* dgamma,
                             int rows, int C, float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream);
/* y = dropout(x) (pos_drop, HQAViT_CIFAR100.py:1251); bwd is the same call on dy */
int qavit_dropout(int dtype, const void* x, void* y, int64_t n, float p, int site, const int64_t* rng, void* stream);
/* packed weights: dst = cast(src) and dstT = cast(src)^T for 2-D [rows, cols] fp32 params; descriptor table on device.
 * pad = 3: as pad = 1 for the TRANSPOSE of src (fragment (t, s) = rows 16 t.. of src^T, i.e. columns of src; rows % 32 == 0,
 * cols % 16 == 0): the operand image of a GEMM that contracts over src's rows (csrc/branch_bwd.hip).
 * pad = 1: dst is written in MFMA FRAGMENT order for v_mfma_f32_16x16x32_bf16 instead of row-major (rows % 16 == 0,
 * cols % 32 == 0): the 16-row x 32-column fragment (t, s) is 1 KB at ((t * (cols/32) + s) * 512) elements, and inside it
 * lane l's 8 elements are src[16 t + l % 16][32 s + 8 (l / 16) .. + 8] -- a wave reads it with one 16-byte load per lane,
 * and a run of fragments is a linear copy into LDS (csrc/branch_fwd.hip). */
typedef struct qavit_pack_desc { const float* src; void* dst; void* dstT; int rows; int cols; int ldT; int pad; } qavit_pack_desc;
/* dst[r*cols + c] = src[r][c]; dstT[c*ldT + r] = src[r][c] (ldT >= rows lets several sources stack into one
 * transposed matrix, e.g. CGA's q/k/v projections) */
int qavit_pack_weights(int dtype, const qavit_pack_desc* descs_dev, int n_desc, int max_elems, void* stream);
/* rng[1] += 1 */
int qavit_rng_advance(int64_t* rng, void* stream);
/* diagnostic: *dst = the device's constant 100 MHz wall clock when `stream` reaches this launch (one lane; tools/chain_stamps.py
 * places these at the fork / join points of a captured step to see where its two chains run without a profiler attached) */
int qavit_stamp(uint64_t* dst, void* stream);
/* fused AdamW over a flat fp32 buffer with a per-element group mask (skip[i] != 0: parameter never receives a
 * gradient -> untouched, as torch.optim.AdamW skips grad-is-None tensors, HQAViT_CIFAR100.py:1566-1571) and
 * global-norm clipping folded in: g *= min(1, max_norm / (*gnorm + 1e-6)) (clip_grad_norm_, :1432).
 * lr and step come from device scalars so a captured graph replays with a moving schedule. */
int qavit_adamw(float* p, const float* g, float* m, float* v, const uint8_t* skip, int64_t n,
                const float* lr_dev, float beta1, float beta2, float eps, float wd,
                const float* step_dev, const float* gnorm_dev, float max_norm, void* stream);
/* per-tensor clip of nseg segments of the flat gradient buffer, seg = device int64 [nseg][2] = (offset, length):
 * g[seg] *= min(1, clip / (||g[seg]||_2 + 1e-6))   (clip_grad_norm_ per parameter, HQAViT_CIFAR100.py:1416-1418).
 * ws = float[2*nseg], zero on first use; the call leaves it zero again. */
int qavit_local_clip(float* g, const int64_t* seg, int nseg, float clip, float* ws, void* stream);
/* dst_a = src_a, dst_b = src_b (n fp32 each, n % 4 == 0, 16-byte aligned): forward-time snapshot of the bank's K / V rows
 * (the reference's torch.cat / Linear-on-expand copies, HQAViT_CIFAR100.py:398-399, :576-577) in one launch */
int qavit_copy2(const float* src_a, const float* src_b, float* dst_a, float* dst_b, int64_t n, void* stream);
/* Label-smoothed cross entropy with reduction = 'mean' (nn.CrossEntropyLoss(label_smoothing), HQAViT_CIFAR100.py:1373), loss and
 * gradient in one launch: loss[0] = mean_i ( - sum_c t_ic log softmax(logits_i)_c ), dlogits = (softmax - t) / B (NULL: loss only),
 * t_i = (1 - ls) * (lam * onehot(y_a[i]) + (1 - lam) * onehot(y_b[i])) + ls / C.  y_b == NULL: plain labels (lam = 1); lam_dev is a
 * DEVICE scalar (the MixUp / CutMix lambda of :1404-1408, decided on the device inside a captured step).
 * ws = float[1 + ceil(B / 16)], ws[0] zero on first use; the call leaves it zero again (arrival ticket of the deterministic loss sum). */
int qavit_ce_label_smooth(int dtype, const void* logits, const int64_t* y_a, const int64_t* y_b, const float* lam_dev,
                          float label_smoothing, int B, int C, float* loss, void* dlogits, float* ws, void* stream);
/* out[0] = sqrt(sum g^2) over a flat buffer (two-pass, deterministic order within a block);
 * out[1] = max(out[1], out[0]) with NaN sticky: the largest norm any call has seen (`out` is float[2], zero at start),
 * so one host read after N captured steps checks every one of them. */
int qavit_l2norm(const float* g, int64_t n, float* partial, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QAVIT_H */
