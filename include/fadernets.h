/*
 * fadernets.h - C ABI of libfadernets_hip.so: the MI355X (gfx950) kernels behind the
 * MusicAttrRegGMVAE training / inference path of Music FaderNets.
 *
 * The reference has no FFI: the path is PyTorch module code (gmm_model.py:10-259) driven by
 * trainer_gmm.py:109-303.  Each entry point below replaces the torch operator(s) the reference
 * executes at the cited lines; the Python class in music-fader-nets_amd/gmm_model.py keeps the
 * reference's class surface and calls these through ctypes.
 *
 * Conventions
 *   - every function returns int: 0 = OK, <0 = FN_E_* argument error, >0 = hipError_t;
 *   - all pointers are DEVICE pointers (fp32 unless stated, int32 for token ids), owned by the
 *     caller; nothing is allocated, freed or synchronised inside (graph-capture safe);
 *   - `stream` is a hipStream_t passed as void*; work is stream-ordered;
 *   - functions are re-entrant and thread-safe; the only process-wide state is a handful of write-once caches of immutable facts
 *     (compute-unit count and kernel attributes per device, the run-time binding of librccl), held in atomics / behind
 *     std::call_once; no environment variable is read;
 *   - per-step tensors are TIME-MAJOR: [T][B][...] so that one step is one contiguous slab;
 *     token tensors are batch-major [B][T] int32 exactly as the data loader yields them.
 */
#ifndef FADERNETS_H
#define FADERNETS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FN_OK 0
#define FN_E_NULL (-1)      /* required pointer is NULL                     */
#define FN_E_SHAPE (-2)     /* unsupported / inconsistent sizes             */
#define FN_E_ALIGN (-3)     /* pointer or leading dimension not aligned     */
#define FN_E_WORKSPACE (-4) /* workspace too small                          */
#define FN_E_COUNT (-5)     /* too many scans in one call                   */
#define FN_E_UNSUPPORTED (-6) /* valid arguments, but this single-launch path is not eligible on this device / shape: */
                            /* nothing was enqueued, the caller takes the general path                             */
#define FN_E_COMM (-7)      /* fn_comm_*: librccl could not be loaded / lacks a symbol                             */
#define FN_COMM_ERROR_BASE 2000000  /* fn_comm_*: FN_COMM_ERROR_BASE + ncclResult_t                                 */

#define FN_MAX_SCANS 8

int fn_version(void);                 /* ABI version, currently 6 (round 5: fn_gru_fwd_x6_ok, fn_gru_bwd_x6_ok, fn_comm_count / fn_comm_rank, fn_weight_images kinds 3 / 4, FnGruBwd.variant bit 14;
                                         round 6: FN_GEMM_X6_PERWAVE / _WIDE / _PERTILE, FN_GEMM_BF16X6 on the Linear-forward form, FnGruCell.variant bit 14, fn_weight_images kind 5) */
const char* fn_strerror(int code);    /* static string for FN_E_* / hipError_t */

/* ------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the f32 MFMA (v_mfma_f32_16x16x4_f32), exact-f32 fma chains.
 *   C[M,N] = alpha * opA(A) * opB(B) + beta * C + bias[N]
 *   a_kmajor=1: A is [M,K] row-major (lda)      a_kmajor=0: A is stored [K,M] row-major (lda)
 *   b_kmajor=1: B is [N,K] row-major (ldb) i.e. a torch Linear weight; 0: B is [K,N] row-major
 * Replaces nn.Linear forward (gmm_model.py:86,91,108,113,123,137: a_k=1,b_k=1), its input
 * gradient (a_k=1,b_k=0) and its weight gradient dW = dY^T X (a_k=0,b_k=0).
 * splitk>1 needs ws of fn_gemm_ws_bytes(M,N,splitk) bytes (deterministic slab reduction).
 * splitk | FN_GEMM_LEAN (weight-gradient form a_k=0,b_k=0 and fn_gru_dwhh_f32): the instance whose
 * wavefronts need <= 128 vector registers, so that one of them fits on a SIMD BESIDE a wavefront of a weight-stationary scan
 * (fn_gru_seq_*: 336-376 registers of 512) and the product fills the scan's idle MFMA cycles instead of waiting for its CUs.
 * Same results bit for bit (same k order).
 * The Linear-forward form (a_k=1,b_k=1) with whole 128 x 128 tiles (>= 256 of them), K % 16 == 0, 16-byte aligned operands and no
 * split runs as an LDS-free kernel (MFMA operands straight from memory; k summed in a fixed order that differs from the staged
 * kernel's inside every 16-k step); there FN_GEMM_LEAN selects its <= 128-register instance (four workgroups per CU; fits beside a
 * wavefront of the decoder pipeline's forward scans) - same results as the full instance.
 * ------------------------------------------------------------------------------------------ */
#define FN_GEMM_LEAN 0x10000
/* splitk | FN_GEMM_BF16X6 (weight-gradient form a_k=0,b_k=0 and fn_gru_dwhh_f32): the products run on the bf16 MFMA with every fp32
 * operand value cut EXACTLY into three bf16 pieces (x = hi + mid + lo) and six of the nine exact partial products accumulated in
 * fp32, smallest first (the three dropped ones are <= 2^-24 of |a b| and, with B's pieces rounded, zero-mean).  Against float64 the
 * result is as accurate as the fp32 MFMA chain (tests/test_gpu_parity.py::test_gemm_tn_bf16x6, ::test_bf16x6_adversarial_operands_vs_float64);
 * it is a different summation, so results differ from the fp32-MFMA kernel in the last bits.  Split-K ranges are whole 32-k blocks; the
 * rows of a K tail below 32 (unsplit products, the last range) are multiplied in a zero-padded block. */
#define FN_GEMM_BF16X6 0x20000
/* FN_GEMM_BF16X6 on the Linear-forward form (a_k=1,b_k=1; whole 128 x 128 tiles - at least 128 of them -, K % 32 == 0, 16-byte aligned operands, no
 * split): the same arithmetic through gemm_nt_x6w_kernel (gemm.hip); other shapes of that form ignore the flag and run on the fp32 MFMA. */
/* ... | FN_GEMM_X6_PERWAVE (only with FN_GEMM_BF16X6; tests / A-B measurements): the round-5 kernel in which every wavefront splits its own operands.
 * The default bf16 x 6 kernel has producer wavefronts that split every operand value once per workgroup and consumer wavefronts that only multiply
 * (gemm.hip: gemm_tn_x6w_kernel); both accumulate the same products in the same order, so on K ranges of whole 32-k blocks their results are
 * bit-identical. */
#define FN_GEMM_X6_PERWAVE 0x40000
/* ... | FN_GEMM_X6_WIDE (only with FN_GEMM_BF16X6): 128 x 256 output tiles per workgroup (gemm.hip: gemm_tn_x6v_kernel) instead of 128 x 128: a CU's
 * operand loads, which bound the 128 x 128 kernel, drop from 21 to 16 bytes per MFMA clock.  The caller wants twice the K ranges it would use with
 * 128 x 128 tiles (the same number of workgroups).  Same products in the same order per accumulator: bit-identical to the other two kernels on the
 * same K ranges. */
#define FN_GEMM_X6_WIDE 0x80000
/* ... | FN_GEMM_X6_PERTILE (only with FN_GEMM_BF16X6; tests / A-B measurements): one workgroup per output tile (Linear-forward form) or per (tile, K range)
 * item (weight-gradient form with K ranges in multiples of 8).  Default (round 6): one workgroup per CU walks its items, the next item's first blocks
 * are requested and cut while the finished one is stored.  Same arithmetic per item: bit-identical. */
#define FN_GEMM_X6_PERTILE 0x100000
size_t fn_gemm_ws_bytes(int M, int N, int splitk);
int fn_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, float alpha,
                const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
                const float* bias, int splitk, float* ws, size_t ws_bytes, void* stream);

/* Up to 12 small independent GEMMs in ONE launch, each the sum of up to 4 products over separate operands (the heads of the path are
 * chains of 256-row GEMMs: e.g. mu_r = [h_fwd | h_rev] W^T is two products into one output, gmm_model.py:85-86):
 *   C_j[M,N] = sum_i opA(A_ji) opB(B_ji) + beta_j * C_j + bias_j      (a_kmajor / b_kmajor as in fn_gemm_f32, shared by all jobs) */
typedef struct FnGemmSeg {
    const float* A;
    int32_t lda;
    const float* B;
    int32_t ldb;
    int32_t K;
} FnGemmSeg;
typedef struct FnGemmJob {
    int32_t M, N;
    FnGemmSeg seg[4];
    int32_t n_seg;
    float beta;
    float* C;
    int32_t ldc;
    const float* bias;
} FnGemmJob;
int fn_gemm_multi(int a_kmajor, int b_kmajor, const FnGemmJob* jobs, int n_jobs, void* stream);

/* dst[c*dst_ld + r] = src[r*src_ld + c]  for r < R, c < C */
int fn_transpose_f32(const float* src, int R, int C, int src_ld, float* dst, int dst_ld, void* stream);
/* Column sums of up to FN_COLSUM_MAX_JOBS small matrices in ONE launch: out_j[n] = beta_j*out_j[n] + sum_m X_j[m*ld_j + n].
 * The bias gradients of a step (autograd of every `+ b` on the path, trainer_gmm.py:249) are ~40 column sums over <= 256-row
 * matrices; as separate launch pairs they were 76 launches of a few microseconds each on the critical tail of the step.
 * One workgroup per (job, 64 columns): rows are summed in a fixed order (4 row phases x 4 accumulators, then a fixed
 * tree) - deterministic.  Meant for M <= a few thousand rows; taller matrices belong to fn_colsum_f32. */
#define FN_COLSUM_MAX_JOBS 64
typedef struct FnColsumJob {
    const float* X;
    int32_t M, N, ld;
    float beta;
    float* out;
} FnColsumJob;
int fn_colsum_multi(const FnColsumJob* jobs, int n_jobs, void* stream);

/* out[n] = beta*out[n] + sum_m X[m*ld + n]; ws >= fn_colsum_ws_bytes(M,N) */
size_t fn_colsum_ws_bytes(int M, int N);
int fn_colsum_f32(const float* X, int M, int N, int ld, float beta, float* out, float* ws, size_t ws_bytes,
                  void* stream);
/* y[i] += alpha * x[i] */
int fn_axpy_f32(int64_t n, float alpha, const float* x, float* y, void* stream);
/* out[0] = sum x[0..n) (single workgroup, deterministic order) */
int fn_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * GRU sequence scans (replace nn.GRU / nn.GRUCell loops: gmm_model.py:84,89,109,114,131-136).
 *
 * One call runs up to FN_MAX_SCANS independent scans concurrently: every time step is ONE
 * launch whose grid covers all scans (fused  h W_hh^T  MFMA GEMM + gate epilogue).
 * Gate order r,z,n (torch).  Input-side pre-activation of step p for batch row b:
 *    gx[b,:] = b_ih + gx_dense[p][b,:] + gx_table[tok][:] + gx_rowbias[b,:]     (each optional)
 *    tok = (tau < 0) ? start_token : idx[b*idx_ld + tau],  tau = (reverse ? T-1-p : p) + idx_shift
 * Storage is in PROCESSING order p = 0..T-1 (for reverse scans p=0 is the last time step).
 * H must be a multiple of 32.  The recurrent weights are passed in the FRAGMENT-MAJOR image produced by
 * fn_frag_pack (one contiguous 1 KB run per wave load instruction); frag_ws is caller-owned scratch of
 * 2 * fn_frag_floats(B, H) floats in which the scan ping-pongs the state in the same layout.
 * ------------------------------------------------------------------------------------------ */
typedef struct FnGruFwd {
    int32_t B, T, H;
    int32_t reverse;          /* 1: consume tokens from the end (the *_reverse direction)      */
    const float* w_hh_frag;   /* fn_frag_pack(W_hh [3H][H])                                     */
    const float* b_hh;        /* [3H]                                                          */
    const float* b_ih;        /* [3H] or NULL                                                  */
    const float* h0;          /* [B][H] or NULL (= zeros)                                      */
    const float* gx_dense;    /* [T][B][3H] or NULL                                            */
    const float* gx_table;    /* [V][3H] (= W_ih[:, :V]^T) or NULL                             */
    const int32_t* idx;       /* [B][idx_ld] token ids, required with gx_table                 */
    int32_t idx_ld;
    int32_t idx_shift;        /* -1 for the global decoder (input of step i is token i-1)      */
    int32_t start_token;      /* used when tau < 0                                             */
    const float* gx_rowbias;  /* [B][3H] or NULL (per-sequence constant part, W_ih[:,V:] z)    */
    float* h_all;             /* [T][B][H] state after each step                               */
    float* gates;             /* [T][fn_gru_gates_floats(B,H)] saved r,z,n,(W_hn h + b_hn) in a    */
                              /* private blocked layout (opaque to the caller); NULL = inference */
    float* frag_ws;           /* scratch, 2 * fn_frag_floats(B, H) floats, 16-byte aligned         */
    void* sync_ws;            /* fn_gru_sync_ws_bytes() bytes, zero-filled ONCE by the caller, shared by  */
                              /* the scans of one call (scans[0]'s is used); NULL = per-step launches only */
    int32_t cu_budget;        /* compute units the single launch may occupy (scans[0]'s is used; 0 = all). */
                              /* Launches that can overlap on different streams must share the chip:       */
                              /* the sum of their budgets must not exceed the CU count.                    */
    const float* h0_frag;     /* optional: h0 already in the fragment-major operand layout (fn_frag_floats(B,H)  */
                              /* floats) - saves the packing launch when a scan continues a previous call (time  */
                              /* chunks, step-by-step decoding); h0 is still needed (row-major, gate epilogue)   */
    float* h_last_frag;       /* optional: the final state, fragment-major (the next call's h0_frag)       */
    int32_t variant;          /* 0 = automatic (scans[0]'s is used).  Tuning / tests: low byte = force this many batch rows per   */
                              /* workgroup of the single launch (16/32/64/128; not eligible -> per-step kernels), bit 8 = the     */
                              /* alternative wave tiling of the 64-row configuration.  Results never depend on it.               */
                              /* bit 9: sync_ws counters are ALREADY zero (the caller zero-fills a pool of regions once and gives  */
                              /* every launch its own region: saves one memset node per launch)                                  */
                              /* bit 12 (tests): the workgroups of every row group are spread over all XCDs instead of sharing    */
                              /* one (the default placement is speed only; tests/test_gpu_parity.py runs both and compares)       */
    void* err_ws;             /* optional: sticky error word outside sync_ws (>= 4 bytes, scans[0]'s is used); NULL = the last     */
                              /* 128 bytes of sync_ws                                                                            */
} FnGruFwd;

/* fragment-major operand image: floats needed for a [rows][K] matrix, and the packing kernel
 * (src row-major with leading dimension ld, K % 32 == 0; rows are zero-padded to a multiple of 16) */
/* OPT-IN (FnGruFwd.variant bit 14 = 0x4000): the forward scan with exact split products on the bf16 MFMA ("bf16 x 6", see FN_GEMM_BF16X6):
 * w_hh_frag must then be the fn_frag3_pack image of W_hh [3H][H] (bf16 triples hi | mid | lo, hi + mid + lo == W exactly; 3/2 of
 * fn_frag_floats(3H, H) floats) and frag_ws 3 * fn_frag_floats(B, H) floats (the state is exchanged as triples too).  H = 512, every
 * scan in full row groups of 64 (or 128) rows, saved gates, T >= 2 (h0_frag / h_last_frag are then triple images of 3/2 * fn_frag_floats(B, H)
 * floats); anything else returns FN_E_UNSUPPORTED (the caller repeats the call without the bit).  Same gate arithmetic; against the default kernels the states differ by fp32 rounding only
 * (tests/test_gpu_parity.py::test_forward_scan_bf16x6). */
int fn_frag3_pack(const float* src, int rows, int K, int ld, void* dst, void* stream);
size_t fn_frag_floats(int rows, int K);
int fn_frag_pack(const float* src, int rows, int K, int ld, float* dst, void* stream);

/* All weight images of one optimiser step in ONE launch (the refresh after every Adam update: ~40 small matrices, ~56 with the bf16 triple images).
 *   kind 0: dst [cols][rows] = src^T                       (the one-hot column table W_ih[:, :V]^T; src [rows][cols], leading dim ld)
 *   kind 1: dst = fn_frag_pack image of src [rows][K = cols]  (cols % 32 == 0)
 *   kind 2: dst = fn_frag_pack image of src^T, the [cols][K = rows] matrix (rows % 32 == 0): W_hh^T for the backward scans
 *   kind 3: dst = fn_frag3_pack image (bf16 triples) of src [rows][K = cols]            (bf16 x 6 forward scans, variant bit 14)
 *   kind 4: dst = fn_frag3_pack image of src^T, the [cols][K = rows] matrix              (bf16 x 6 backward scans: W_hh^T)
 *   kind 5: dst [rows][cols] dense = src [rows][cols] (leading dim ld)                   (round 6: 16-byte aligned image of a column slice, W_ih[:, V:])
 * up to 56 jobs; dst of kinds 1 / 2 16-byte aligned with fn_frag_floats(...) floats, of kinds 3 / 4 with 3/2 of that. */
typedef struct FnWeightImage {
    const float* src;
    float* dst;
    int32_t rows, cols, ld, kind;
} FnWeightImage;
int fn_weight_images(const FnWeightImage* jobs, int n_jobs, void* stream);

/* floats per time step of the saved-gates buffer (4*H*ceil16(B)) */
size_t fn_gru_gates_floats(int B, int H);

/* When sync_ws is given, all scans share H <= 512 and sum_s ceil(B_s / rows) * H/16 fits one workgroup per CU, the
 * whole call is ONE weight-stationary launch whose workgroups exchange the state through frag_ws and meet at arrival
 * counters in sync_ws (every spin is bounded).  The last word of sync_ws is a sticky error flag: non-zero after a
 * launch whose workgroups gave up waiting (results of that and later calls are invalid until it is cleared). */
size_t fn_gru_sync_ws_bytes(void);
int fn_gru_seq_fwd(const FnGruFwd* scans, int n_scans, void* stream);
/* 1 when fn_gru_seq_fwd would take this call with variant bit 14 (bf16 x 6) on the current device, else 0 (shapes and gates != NULL only;
 * nothing is enqueued).  A scan that is cut into several calls (time chunks with h_last_frag -> h0_frag hand-over) must run ALL of its
 * calls on one arithmetic - the hand-over images differ (bf16 triples / fp32 fragments) - so the caller asks for every call of the chain first. */
int fn_gru_fwd_x6_ok(const FnGruFwd* scans, int n_scans);

/* ONE GRU cell step for a large batch (nn.GRUCell, gmm_model.py:131-136 in the eval-mode decode loop of thousands of rows):
 *    gi = x W_ih^T + b_ih + gx_table[tok] + gx_rowbias[b]      (every term optional)       gh = h_prev W_hh^T + b_hh
 *    r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h_out = (1 - z) n + z h_prev
 * as ONE MFMA launch with the gates in its epilogue (both products in its K loops: layer 2 of the decoder needs no separate W_ih
 * projection launch and no [B][3H] round trip).  Three kernels: above 512 rows (16-byte aligned operands, K1 % 16 == 0, row strides
 * % 4 == 0) loops whose lanes read their state-row operands straight from the K-contiguous rows - weights likewise, or as an
 * LDS-resident slice (one workgroup of ceil(rows / 512) x 64 rows x 16 units per CU; at K1 = H = 512 filled under the K loops, up to 2048 rows) -, otherwise an LDS-staged GEMM; they sum k in different orders (fp32 rounding apart).  Weights are the torch matrices themselves ([3H][K] row-major),
 * states row-major; h_out must not alias h_prev.  H % 32 == 0.
 * tok = idx ? idx[b * idx_ld] : start_token (point idx at the column of the previous step's tokens). */
typedef struct FnGruCell {
    int32_t B, H;
    const float* x;           /* [B][ldx] dense input, K1 valid columns, or NULL              */
    int32_t ldx, K1;
    const float* w_ih;        /* [3H][ldw_ih], K1 valid columns (required with x)             */
    int32_t ldw_ih;
    const float* gx_table;    /* [V][3H] (= W_ih[:, :V]^T) or NULL                            */
    const int32_t* idx;       /* token of row b at idx[b * idx_ld], or NULL = start_token     */
    int32_t idx_ld, start_token;
    const float* gx_rowbias;  /* [B][3H] or NULL                                              */
    const float* h_prev;      /* [B][ldh]                                                     */
    int32_t ldh;
    const float* w_hh;        /* [3H][ldw_hh]                                                 */
    int32_t ldw_hh;
    const float* b_ih;        /* [3H] or NULL                                                 */
    const float* b_hh;        /* [3H]                                                         */
    float* h_out;             /* [B][ldo]                                                     */
    int32_t ldo;
    int32_t variant;          /* 0 = automatic.  Tuning / tests: 1-3, 8 force a staged tiling (8 = its default: 64 rows x 32 units), 4-7 the LDS-free
                                 loop (4, 6: 128 rows x 32 units per workgroup, 1 / 2 k steps in flight; 5, 7: 64 rows, 4 / 2), 9-12 the loop with
                                 the weight slice in LDS (9, 11: 256 rows x 16 units, 2 steps; 10, 12: 128 rows, 4 / 8; 13, 14: 192 rows, 4 / 2), 15-18 the same with
                                 the slice fills under the K loops (K1 = H = 512; 15, 18: 128 rows, 4 / 2 steps; 16: 192 rows; 17: 256 rows) where eligible.
                                 | 0x4000 (bit 14, as in FnGruFwd): the cell on the bf16 MFMA with exact triple splits (gru.hip: gru_cell_x6_kernel) where
                                 B % 128 == 0, H % 32 == 0, K1 % 32 == 0 and the operands are 16-byte aligned; other shapes run as without the bit */
    const uint64_t* idx_best; /* NULL, or the packed argmax words of the previous token (fn_out_argmax_f32): the token of row b is
                                 best_v - 1 - (uint32_t)idx_best[b]; takes precedence over idx                                */
    int32_t best_v;           /* vocabulary size the words were packed with                                                   */
} FnGruCell;
int fn_gru_cell_f32(const FnGruCell* c, void* stream);

/* Output layer of the eval-mode decode with the argmax in its epilogue (gmm_model.py:137 + 73-80 when only the tokens are wanted; the
 * log-softmax is monotone, so argmax(log_softmax(logits)) = argmax(logits)): for every row b
 *     best[b] = max(best[b], max_v pack(h[b] . W[v] + bias[v], v)),   pack(x, v) = (uint64_t)key(x) << 32 | (uint32_t)(V - 1 - v),
 *     key(x)  = bits(x) ^ (bits(x) >> 31 ? 0xffffffff : 0x80000000)                (order-preserving; first index wins a tie)
 * by 64-bit atomic max: ZERO best[0 .. B) before the call; the logits are never written.  h [B][ldh], W [V][ldw] K-contiguous rows
 * (the torch matrix), 16-byte aligned, K % 16 == 0, ldh % 4 == ldw % 4 == 0.  One launch, LDS-free loop (as fn_gru_cell_f32's).
 * fn_best_tokens: tokens[b * tok_ld + t] = V - 1 - (uint32_t)best[t * B + b] for steps x B words. */
int fn_out_argmax_f32(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int V, int K, uint64_t* best,
                      void* stream);
int fn_best_tokens(const uint64_t* best, int steps, int B, int V, int32_t* tokens, int tok_ld, void* stream);

/* Backward of the same scans (autograd of nn.GRU / GRUCell in loss.backward(), trainer_gmm.py:249).
 *   dh_p = dh_ext[p] (+ dh_last at p = T-1) + carried gradient
 *   outputs: dgx_all [T][B][3H] = d(pre-activations r,z,n) (= d gx, also d(W_hh h + b_hh) for r,z)
 *            dghn_all[T][B][H]  = d(W_hn h + b_hn)
 *            dh0 [B][H]  gradient wrt h0 (NULL = not needed)
 *            dgx_rowsum [B][3H] += sum_p dgx_all[p], dghn_rowsum [B][H] += sum_p dghn_all[p]
 *                        (NULL = not needed; caller zero-fills; their column sums are the bias gradients)
 * w_hh_t_frag = fn_frag_pack(W_hh^T [H][3H]).  scratch: [B][H] floats; frag_ws: 2 * fn_frag_floats(B, 3H) floats. */
typedef struct FnGruBwd {
    int32_t B, T, H;
    const float* w_hh_t_frag; /* fn_frag_pack(W_hh^T [H][3H])                                  */
    const float* h0;          /* [B][H] or NULL                                                */
    const float* h_all;       /* [T][B][H]  from forward                                       */
    const float* gates;       /* [T][fn_gru_gates_floats(B,H)] from forward                    */
    const float* dh_last;     /* [B][H] or NULL                                                */
    const float* dh_ext;      /* [T][B][H] or NULL                                             */
    float* dgx_all;           /* [T][B][3H]                                                    */
    float* dghn_all;          /* [T][B][H]                                                     */
    float* dh0;               /* [B][H] or NULL                                                */
    float* dgx_rowsum;        /* [B][3H] or NULL                                               */
    float* dghn_rowsum;       /* [B][H] or NULL                                                */
    float* scratch;           /* [B][H]                                                        */
    float* frag_ws;           /* 2 * fn_frag_floats(B, 3H) floats, 16-byte aligned             */
    void* sync_ws;            /* as in FnGruFwd (NULL = per-step launches; then scratch is required) */
    int32_t cu_budget;        /* as in FnGruFwd                                                */
    int32_t variant;          /* as in FnGruFwd                                                */
    void* err_ws;             /* as in FnGruFwd                                                */
} FnGruBwd;

int fn_gru_seq_bwd(const FnGruBwd* scans, int n_scans, void* stream);
/* variant bit 14 (0x4000), as in FnGruFwd: the backward scan with exact split products on the bf16 MFMA (gru_bwd_x6_kernel).  w_hh_t_frag must then be
 * the bf16 TRIPLE image of W_hh^T [H][3H] (fn_weight_images kind 4; 3/2 of fn_frag_floats(H, 3H) floats) and frag_ws 3 * fn_frag_floats(B, 3H) floats
 * (the gate gradients are exchanged as triples).  H = 512, every scan in full groups of 64 rows (with variant bit 15 also: of 32 rows - that form
 * measured slower than the fp32 kernel and is not chosen on its own), T >= 2, 9..16 row groups that are all resident at once (the shapes of the
 * register-stationary fp32 kernel); anything else returns FN_E_UNSUPPORTED.  fn_gru_bwd_x6_ok answers without
 * enqueuing anything (1 / 0).  Same element-wise gate arithmetic; gradients differ from the default kernels by fp32 rounding only. */
int fn_gru_bwd_x6_ok(const FnGruBwd* scans, int n_scans);

/* Recurrent weight gradient of one scan from its saved gate gradients (autograd of W_hh in nn.GRU / GRUCell,
 * trainer_gmm.py:249):  dW_hh[3H][H] = beta * dW_hh + [dgx[:, 0:2H] | dghn]^T hprev  over `rows` (time x batch) rows;
 * dgx [rows][3H], dghn [rows][H], hprev [rows][H] (the state BEFORE each step).  One split-K launch when 2H % 128 == 0. */
size_t fn_gru_dwhh_ws_bytes(int H, int splitk);
int fn_gru_dwhh_f32(const float* dgx, const float* dghn, const float* hprev, int64_t rows, int H, float beta, float* dW,
                    int splitk, float* ws, size_t ws_bytes, void* stream);

/* Gradient of the one-hot columns of W_ih (autograd of  onehot(x) @ W_ih[:, :V]^T, gmm_model.py:84,89,109,114,132-133):
 *   dTable[v][:] = sum over (p,b) with tok(p,b)==v of dgx_all[p][b][:]      (tok defined as in FnGruFwd)
 * Two steps, so that ONE sort of a token matrix serves every scan that consumes it (4 encoder directions + decoder layer 1):
 *   fn_token_sort: idx [B][idx_ld] int32 -> img (fn_token_sort_ints(B*T, V) int32: segment starts, piece starts, positions
 *                  sorted by token, stable); ws >= fn_token_sort_ws_bytes(B*T, V), 16-byte aligned.  Tokens are clamped to [0, V).
 *   fn_embed_grad_sorted: up to 8 jobs (scans) that read the SAME token matrix in one launch pair; per job idx_shift in {0, -1}
 *                  (-1 only with reverse = 0: step p reads token p-1, step 0 the start token).  out is [V][out_ld] (out_ld >= N3),
 *                  or with transposed != 0 the transposed table [N3][out_ld] (out_ld >= V), e.g. dW_ih[:, :V] in place.
 *                  ws >= fn_embed_grad_sorted_ws_bytes(B*T, B, V, N3, n_jobs).  Deterministic (fixed summation order).
 * fn_embed_grad_f32 = both steps for one scan (ws >= fn_embed_grad_ws_bytes(T*B, V, N3)). */
typedef struct FnEmbedGrad {
    const float* dgx_all;     /* [T][B][N3] gate gradients in processing order                    */
    float* out;               /* table, see above                                                  */
    int32_t out_ld;
    int32_t transposed;
    int32_t reverse, idx_shift, start_token;
} FnEmbedGrad;
size_t fn_token_sort_ints(int64_t rows, int V);
size_t fn_token_sort_ws_bytes(int64_t rows, int V);
int fn_token_sort(const int32_t* idx, int B, int T, int idx_ld, int V, int32_t* img, void* ws, size_t ws_bytes, void* stream);
size_t fn_embed_grad_sorted_ws_bytes(int64_t rows, int B, int V, int N3, int n_jobs);
int fn_embed_grad_sorted(const FnEmbedGrad* jobs, int n_jobs, int B, int T, int N3, int V, const int32_t* img, float* ws,
                         size_t ws_bytes, void* stream);
size_t fn_embed_grad_ws_bytes(int64_t rows, int V, int N3);
int fn_embed_grad_f32(const float* dgx_all, int B, int T, int N3, const int32_t* idx, int idx_ld, int idx_shift,
                      int start_token, int reverse, int V, float* out, float* ws, size_t ws_bytes, void* stream);

/* out[m] = sum_t X[t*M + m]   (per-sequence sums over time of the gate gradients; M % 4 == 0, 16-byte aligned) */
int fn_time_sum_f32(const float* X, int T, int64_t M, float* out, void* stream);

/* Greedy autoregressive decode of the global decoder for B <= 2048 sequences (H <= 512) as ONE launch
 * (gmm_model.py:119-149 with model.eval(): layer-1 cell, layer-2 cell (state initialised with the first layer-1 state, :134-135),
 * 512 -> V output layer, log_softmax, feedback = first-index argmax, :73-80,147-148).  Workgroup sets keep the four weight
 * matrices in LDS and hand activations over through L2 (bounded spins, sticky error word as in fn_gru_seq_fwd).  More than 32
 * sequences (the reference's evaluators decode 8 fader values x 100 samples at once, test_class.py:84-85,253) travel through the
 * role workgroups as a pipeline of 32-row blocks (64-row blocks from 353 sequences on: a role's time per block is mostly latency), two
 * replicas of the role set side by side when the device has the CUs. */
typedef struct FnDecode {
    int32_t B, steps, H, V;
    int32_t start_token;      /* input token of step 0 (V - 1)                                       */
    const float* w_hh1_frag;  /* fn_frag_pack(grucell_g.weight_hh [3H][H])                            */
    const float* b_hh1;       /* [3H]                                                                */
    const float* b_ih1;       /* [3H] or NULL                                                        */
    const float* table1;      /* [V][3H] = grucell_g.weight_ih[:, :V]^T (token rows)                  */
    const float* rowbias1;    /* [B][3H] = z @ grucell_g.weight_ih[:, V:]^T, or NULL                  */
    const float* h0;          /* [B][H] = linear_init_global(z)                                       */
    const float* w_ih2_frag;  /* fn_frag_pack(grucell_g_2.weight_ih [3H][H])                          */
    const float* b_ih2;       /* [3H] or NULL                                                        */
    const float* w_hh2_frag;  /* fn_frag_pack(grucell_g_2.weight_hh [3H][H])                          */
    const float* b_hh2;       /* [3H]                                                                */
    const float* w_out_frag;  /* fn_frag_pack(linear_out_g.weight [V][H])                             */
    const float* b_out;       /* [V]                                                                 */
    int32_t* tokens;          /* out [B][tok_ld]: token of every step                                */
    int32_t tok_ld;
    float* logp;              /* out [B][steps][V] log-probabilities, or NULL                        */
    float* ws;                /* fn_decode_ws_bytes(B, H, V) bytes, 16-byte aligned                   */
    void* sync_ws;            /* fn_decode_sync_ws_bytes() bytes, zero-filled once by the caller       */
} FnDecode;
size_t fn_decode_ws_bytes(int B, int H, int V);
size_t fn_decode_sync_ws_bytes(void);
int fn_decode_greedy(const FnDecode* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Output heads
 * ------------------------------------------------------------------------------------------ */
/* vocab-axis log_softmax (+ optional fused NLL and its gradient): gmm_model.py:137 + trainer_gmm.py:131.
 * logits [T*B][ld] time-major rows (row = t*B + b), E valid columns.
 *   logp_bt  [B][T][E] or NULL   log-probabilities in the reference's (B,T,E) layout
 *   target   [B][T] int32 or NULL
 *   nll_rows [T*B] or NULL       -logp[target] per row
 *   dlogits  [T*B][ld] or NULL   grad_scale * (softmax - onehot(target))   (may alias logits)  */
int fn_vocab_logsoftmax(const float* logits, int B, int T, int E, int ld, float* logp_bt, const int32_t* target,
                        float* nll_rows, float grad_scale, float* dlogits, void* stream);
/* The same head FUSED with its projection (gmm_model.py:137 `log_softmax(linear_out_g(hx[1]))` + trainer_gmm.py:131-132 `nll_loss` and
 * their autograd): logits = h W^T + bias never reach memory.
 *   h        [T*B][ldh] time-major rows (row = t*B + b), H valid columns (the layer-2 states)
 *   W        [V][ldw]   linear_out_g.weight, bias [V];  V <= 384 (else FN_E_UNSUPPORTED)
 *   target   [B][T] int32
 *   nll_rows [T*B] or NULL       -log_softmax(logits)[target]
 *   dlogits  [T*B][ld] or NULL   grad_scale * (softmax - onehot(target)); ld a multiple of 4, V <= ld <= 384, 16-byte aligned;
 *                                columns [V, ld) are written as zeros
 * With 16-byte aligned operands and H % 16 == 0 the projection runs as an LDS-free loop whose k order inside a 16-k step differs from
 * fn_gemm_f32's (fixed, deterministic); the row sums run in a different order than fn_vocab_logsoftmax's. */
int fn_out_head_f32(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int T, int V, int H,
                    const int32_t* target, float grad_scale, float* nll_rows, float* dlogits, int ld, void* stream);
/* generic backward of the same log_softmax: dlogits[row] = g - softmax * sum(g), g = gout_bt[b][t][:] */
int fn_vocab_logsoftmax_bwd(const float* logp_bt, const float* gout_bt, int B, int T, int E, int ld, float* dlogits,
                            void* stream);
/* greedy head for inference (gmm_model.py:73-80,137,147-148): log_softmax + argmax (first index on ties)
 * logits [B][ld] -> logp_out [B][E] (row stride logp_ld) or NULL, tok_out[b*tok_ld] */
int fn_vocab_argmax(const float* logits, int B, int E, int ld, float* logp_out, int64_t logp_ld, int32_t* tok_out,
                    int tok_ld, void* stream);

/* TIME-axis log_softmax of the sub-decoders (gmm_model.py:110,115; the reference's dim=1 quirk).
 * logits [Tr][B][Cc] time-major.  logp_bt [B][Tr][Cc].  target [B][Tr] or NULL.
 * nll_bc [B][Cc] (sum over t with target==c of -logp) ; dlogits [Tr][B][Cc] = grad_scale * dNLLsum/dlogits. */
int fn_time_logsoftmax(const float* logits, int B, int Tr, int Cc, float* logp_bt, const int32_t* target, float* nll_bc,
                       float grad_scale, float* dlogits, void* stream);
int fn_time_logsoftmax_bwd(const float* logp_bt, const float* gout_bt, int B, int Tr, int Cc, float* dlogits,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Latent block: reparameterised sample + Gaussian-mixture posterior (gmm_model.py:86,91,229-242,
 * 194-218) and the KL / class terms of trainer_gmm.py:150-194.  One wavefront per batch row.
 *   pre [B][2Z]: columns [0,Z) = mu, [Z,2Z) = v (sigma = exp(v));  eps [B][Z]
 *   mu_lk, lv_lk [K][Z] component means / log-variances
 * fwd outputs: sigma [B][Z], z [B][Z], ll [B][K], qy [B][K], y [B] int32,
 *   terms [B][4]: {sum_k qy_k KLmean_k, mean_k(qy log qy), KLmean_{label}, -log softmax(qy)[label]}
 *   (label terms only when labels != NULL).  K <= 8.
 * ------------------------------------------------------------------------------------------ */
int fn_latent_fwd(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                  const int32_t* labels, float* sigma, float* z, float* ll, float* qy, int32_t* y, float* terms,
                  void* stream);
/* Backward. Upstream gradients (any may be NULL): g_z, g_mu, g_sigma [B][Z]; g_ll, g_qy [B][K].
 * w3 = DEVICE pointer to {w_lat, w_cls, w_clf} (NULL = all zero; written by fn_step_params so that a captured graph never
 * bakes the step-dependent beta into a kernel argument).  Fused loss gradient (trainer_gmm.py:150-194):
 *    L += w_lat * sum_b sum_k qy KLmean_k  + w_cls * sum_b mean_k(qy log qy)          (unsupervised)
 *    L += w_lat * sum_b KLmean_label       + w_clf * sum_b CE(softmax(qy), label)     (labels != NULL)
 * outputs: dpre [B][2Z]; dmu_lk_rows [B][K][Z] per-row contributions to d mu_lookup (reduce with
 * fn_colsum_f32 over B). */
int fn_latent_bwd(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                  const int32_t* labels, const float* z, const float* qy, const float* g_z, const float* g_mu,
                  const float* g_sigma, const float* g_ll, const float* g_qy, const float* w3, float* dpre,
                  float* dmu_lk_rows, void* stream);

/* Pairwise latent regulariser (trainer_gmm.py:199-217): rows [row0,row0+nrows) of the global batch.
 *   attr_all is float64 like the reference's numpy densities (only the sign of a_i - a_j is used).
 *   loss_rows[i] = sum_j (tanh(z0[row0+i]-z0[j]) - sign(a[row0+i]-a[j]))^2
 *   dz0[i]       = grad_scale * 4 * sum_j (tanh(D)-S)(1-tanh(D)^2)     (the exact d/dz0 of sum_ij, by antisymmetry) */
int fn_pairwise_reg(const float* z0_all, const double* attr_all, int n_all, int row0, int nrows, float* loss_rows,
                    float grad_scale, float* dz0, void* stream);

/* Masked probability sums of the GLSR regulariser (trainer_glsr.py:121-139: approx_played_notes / approx_time_separators = the
 * softmax mass of a token range, per decoder step).  logits [rows][ld], E valid columns; two ranges [lo0,hi0), [lo1,hi1).
 *   sums [rows][2] (or NULL) = P_0, P_1;   with w [rows][2] and dlogits [rows][ld] (may alias logits):
 *   dlogits[u] = p_u (w0 [u in range 0] + w1 [u in range 1] - (w0 P_0 + w1 P_1)) = gradient of w0 P_0 + w1 P_1 wrt the logits. */
int fn_masked_prob(const float* logits, int64_t rows, int E, int ld, int lo0, int hi0, int lo1, int hi1, float* sums, const float* w,
                   float* dlogits, void* stream);

/* Adversarial heads of the Fader-Networks sibling (model_v2.py:572-575, trainer_fader.py:105-110), forward + gradient in one pass,
 * one wavefront per row.  For a in {0: rhythm, 1: note}:  pre = w_a . z[b] + b_a ;  o[b][a] = relu(pre) * mask[b][a] (mask = the
 * dropout keep-mask already scaled by 1/(1-p)) ;  loss_rows[b][a] = (o - dens[b][a])^2 ;
 * da[b][a] = lam * 2 (o - dens) * inv_global_batch * mask * [pre > 0]  (lam read from DEVICE memory, fn_step_params out[6]) ;
 * g_z[b][:] -= sum_a da[b][a] w_a   - MINUS: ReverseLayerF hands the encoder the negated gradient (model_v2.py:426-435).
 * z [B][ldz], g_z [B][ldg] (may be NULL), w_a [Z], b_a [1], mask / dens / o / loss_rows / da [B][2]. */
int fn_adv_head(const float* z, int ldz, int Z, int B, const float* w_r, const float* w_n, const float* b_r, const float* b_n,
                const float* mask, const float* dens, const float* lam_dev, float inv_global_batch, float* o, float* loss_rows, float* da,
                float* g_z, int ldg, void* stream);

/* ------------------------------------------------------------------------------------------
 * clip_grad_norm_(., max_norm) + Adam (trainer_gmm.py:250-251, torch.optim.Adam defaults) over a
 * flat fp32 buffer.  fn_sumsq writes sum(g^2) to out[0]; fn_clip_adam reads the (all-reduced)
 * total from sumsq[0] on device, so no host sync is needed.
 *
 * fn_step_params keeps the step counters ON THE DEVICE (counters[0] = training step, counters[1] = Adam t; int64) and
 * derives every step-dependent scalar from them: out[0..2] = {w_lat, w_cls, w_clf} for fn_latent_bwd with
 * beta0 = 0 if step < 1000 else min((step-10000)/10000*beta, beta)  (trainer_gmm.py:125-128), out[3] = lr/(1-beta1^t),
 * out[4] = 1/sqrt(1-beta2^t), out[5] = beta0, out[6] = min(step/2000*1e-4, 1e-4) (adversarial weight of the Fader-Networks
 * sibling, trainer_fader.py:105), out[7] = beta * inv_global_batch (the constant KL weight of trainer_singlevae.py:104); out has
 * 8 floats.  advance != 0 increments both counters (t is advanced BEFORE use, the step AFTER).  fn_clip_adam takes
 * hyper = &out[3].  The whole training step is therefore capturable in one hipGraph.
 * ------------------------------------------------------------------------------------------ */
int fn_step_params(int64_t* counters, float beta, float lr, float beta1, float beta2, int supervised, float inv_global_batch,
                   int advance, float* out, void* stream);
int fn_sumsq_f32(const float* g, int64_t n, float* out, float* ws, size_t ws_bytes, void* stream);
size_t fn_sumsq_ws_bytes(int64_t n);
int fn_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                 const float* hyper, float beta1, float beta2, float eps, void* stream);

/* one-hot (B,T,V) -> int32 indices (argmax along the last axis); used at the class boundary where callers
 * hand over convert_to_one_hot tensors (trainer_gmm.py:296-303) */
int fn_onehot_to_index(const float* oh, int64_t rows, int V, int32_t* idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data parallelism: RCCL collectives on the CALLER's stream (one process per GPU).
 * The reference has no distributed code; the step that must see the REDUCED gradient is
 * clip_grad_norm_ + optimizer.step(), trainer_gmm.py:249-251.  Rank 0 calls fn_comm_unique_id and hands the
 * FN_COMM_ID_BYTES bytes to the other ranks out of band (music-fader-nets_amd/parallel.py uses the torch.distributed
 * store); every rank then calls fn_comm_init with the device it will use CURRENT.  The collectives are ordinary
 * stream-ordered work: they may be captured into a hipGraph together with the kernels of the step, and no helper thread
 * touches the streams.  librccl is bound at run time (dlopen by SONAME: a copy already in the process is reused);
 * FN_E_COMM when it is missing.  In-place SUM all-reduce of fp32; all-gather of raw bytes (recv holds world * bytes_per_rank).
 * ------------------------------------------------------------------------------------------ */
#define FN_COMM_ID_BYTES 128
int fn_comm_unique_id(void* id_out);
int fn_comm_init(void** comm_out, int world, int rank, const void* id);
int fn_comm_destroy(void* comm);
int fn_comm_all_reduce_f32(void* comm, float* buf, size_t n, void* stream);
int fn_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank): bench.py prints them, so that a multi-GPU line
 * proves that its gradient all-reduce (trainer_gmm.py:249-251) ran over N RCCL ranks */
int fn_comm_count(void* comm, int* count_out);
int fn_comm_rank(void* comm, int* rank_out);

/* Diagnostic: `blocks` workgroups that each hold `lds_bytes` of LDS (<= 64 KB) and spin for about `cycles` clock ticks - a stand-in
 * for a foreign kernel (an RCCL channel, another process' launch) that is RESIDENT on some compute units when a weight-stationary
 * launch starts: the scan's remaining workgroups cannot become resident until it leaves, the resident ones spin at their counters
 * (bounded).  tests/test_gpu_parity.py uses it to show that such contention delays a step but never changes its results. */
int fn_occupy_cus(int blocks, int lds_bytes, long long cycles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FADERNETS_H */
