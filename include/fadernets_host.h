/*
 * fadernets_host.h - HOST twins of the hot-path entry points of fadernets.h (libfadernets_host.so).
 *
 * Every function has the signature of its device counterpart (same argument order, layouts, opaque fragment-major /
 * blocked-gate images, error codes) with HOST pointers; `stream` is ignored and the call is synchronous.  Plain loops, double
 * accumulation where the kernels accumulate in double, libm transcendentals (the kernels use the hardware exp / rcp: results
 * agree to fp32 rounding, not bit for bit).  They exist so that the arithmetic of the C ABI can be driven in a machine without a
 * GPU (tests/test_host_twins.py pushes the `small` golden fixture through them) and under AddressSanitizer - the library is
 * built with -fsanitize=address.  NOT a fallback: nothing in the product loads this library.
 */
#ifndef FADERNETS_HOST_H
#define FADERNETS_HOST_H

#include "fadernets.h"

#ifdef __cplusplus
extern "C" {
#endif

int fn_frag_pack_host(const float* src, int rows, int K, int ld, float* dst, void* stream);              /* fn_frag_pack   */
int fn_gru_seq_fwd_host(const FnGruFwd* scans, int n_scans, void* stream);                               /* fn_gru_seq_fwd */
int fn_gru_seq_bwd_host(const FnGruBwd* scans, int n_scans, void* stream);                               /* fn_gru_seq_bwd */
int fn_latent_fwd_host(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                       const int32_t* labels, float* sigma, float* z, float* ll, float* qy, int32_t* y, float* terms,
                       void* stream);                                                                    /* fn_latent_fwd  */
int fn_latent_bwd_host(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                       const int32_t* labels, const float* z, const float* qy, const float* g_z, const float* g_mu,
                       const float* g_sigma, const float* g_ll, const float* g_qy, const float* w3, float* dpre,
                       float* dmu_lk_rows, void* stream);                                                /* fn_latent_bwd  */
int fn_out_head_f32_host(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int T, int V, int H,
                         const int32_t* target, float grad_scale, float* nll_rows, float* dlogits, int ld, void* stream);   /* fn_out_head_f32 */
int fn_sumsq_f32_host(const float* g, int64_t n, float* out, float* ws, size_t ws_bytes, void* stream);  /* fn_sumsq_f32   */
int fn_clip_adam_host(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                      const float* hyper, float beta1, float beta2, float eps, void* stream);            /* fn_clip_adam   */
/* dense products */
size_t fn_gemm_ws_bytes_host(int M, int N, int splitk);                                                  /* fn_gemm_ws_bytes */
int fn_gemm_f32_host(int a_kmajor, int b_kmajor, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                     float beta, float* C, int ldc, const float* bias, int splitk, float* ws, size_t ws_bytes, void* stream);   /* fn_gemm_f32 */
size_t fn_gru_dwhh_ws_bytes_host(int H, int splitk);                                                     /* fn_gru_dwhh_ws_bytes */
int fn_gru_dwhh_f32_host(const float* dgx, const float* dghn, const float* hprev, int64_t rows, int H, float beta, float* dW,
                         int splitk, float* ws, size_t ws_bytes, void* stream);                          /* fn_gru_dwhh_f32 */
/* token sort + segment sums: the sort image has the device layout {seg [V+1], pstart [V+1], 2 pad, order [rows]}; the workspaces are the
 * twins' own (ask the *_ws_bytes_host functions) */
size_t fn_token_sort_ints_host(int64_t rows, int V);                                                     /* fn_token_sort_ints */
size_t fn_token_sort_ws_bytes_host(int64_t rows, int V);                                                 /* fn_token_sort_ws_bytes */
int fn_token_sort_host(const int32_t* idx, int B, int T, int idx_ld, int V, int32_t* img, void* ws, size_t ws_bytes, void* stream);   /* fn_token_sort */
size_t fn_embed_grad_sorted_ws_bytes_host(int64_t rows, int B, int V, int N3, int n_jobs);               /* fn_embed_grad_sorted_ws_bytes */
int fn_embed_grad_sorted_host(const FnEmbedGrad* jobs, int n_jobs, int B, int T, int N3, int V, const int32_t* img, float* ws,
                              size_t ws_bytes, void* stream);                                            /* fn_embed_grad_sorted */
/* sub-decoder heads, pairwise regulariser */
int fn_time_logsoftmax_host(const float* logits, int B, int Tr, int Cc, float* logp_bt, const int32_t* target, float* nll_bc,
                            float grad_scale, float* dlogits, void* stream);                             /* fn_time_logsoftmax */
int fn_time_logsoftmax_bwd_host(const float* logp_bt, const float* gout_bt, int B, int Tr, int Cc, float* dlogits, void* stream);   /* fn_time_logsoftmax_bwd */
int fn_pairwise_reg_host(const float* z0_all, const double* attr_all, int n_all, int row0, int nrows, float* loss_rows,
                         float grad_scale, float* dz0, void* stream);                                    /* fn_pairwise_reg */
/* eval-mode decode */
int fn_gru_cell_f32_host(const FnGruCell* c, void* stream);                                              /* fn_gru_cell_f32 */
int fn_out_argmax_f32_host(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int V, int K, uint64_t* best,
                           void* stream);                                                                /* fn_out_argmax_f32 */
int fn_best_tokens_host(const uint64_t* best, int steps, int B, int V, int32_t* tokens, int tok_ld, void* stream);   /* fn_best_tokens */
size_t fn_decode_ws_bytes_host(int B, int H, int V);                                                     /* fn_decode_ws_bytes */
size_t fn_decode_sync_ws_bytes_host(void);                                                               /* fn_decode_sync_ws_bytes */
int fn_decode_greedy_host(const FnDecode* d, void* stream);                                              /* fn_decode_greedy */
size_t fn_frag_floats_host(int rows, int K);                                                             /* fn_frag_floats */
size_t fn_gru_gates_floats_host(int B, int H);                                                           /* fn_gru_gates_floats */

#ifdef __cplusplus
}
#endif
#endif
