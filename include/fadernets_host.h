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
size_t fn_frag_floats_host(int rows, int K);                                                             /* fn_frag_floats */
size_t fn_gru_gates_floats_host(int B, int H);                                                           /* fn_gru_gates_floats */

#ifdef __cplusplus
}
#endif
#endif
