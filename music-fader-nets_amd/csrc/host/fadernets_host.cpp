// fadernets_host.cpp - host twins of the hot-path entry points (include/fadernets_host.h): plain loops over host memory, the layouts
// and error behaviour of the device functions.  Built with -fsanitize=address into libfadernets_host.so; test infrastructure for
// machines without a GPU, never loaded by the product.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/fadernets_host.h"

namespace {

// the private layouts of csrc/gru_layout.h
inline long gate_off(int b, int q, int u, int nrt) {
    return ((((long)(u >> 4) * nrt + (b >> 4)) * 4 + q) * 4 + (b & 3)) * 64 + ((b & 15) >> 2) * 16 + (u & 15);
}
inline long frag_off(int row, int k, int NC) {
    return ((((long)(row >> 4) * NC + (k >> 5)) * 2 + ((k >> 4) & 1)) * 64 + (((k & 15) >> 2) * 16 + (row & 15))) * 4 + (k & 3);
}
inline float sigmoidf(float x) { return 1.0f / (1.0f + std::exp(-x)); }

// row-major copy of a fragment-major [rows][K] image
std::vector<float> unfrag(const float* img, int rows, int K) {
    std::vector<float> out((size_t)rows * K);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) out[(size_t)r * K + k] = img[frag_off(r, k, K >> 5)];
    return out;
}

constexpr double LOG_2PI = 1.8378770664093453;
constexpr int KMAX = 8;

}  // namespace

extern "C" {

size_t fn_frag_floats_host(int rows, int K) { return (size_t)((rows + 15) / 16 * 16) * (size_t)K; }
size_t fn_gru_gates_floats_host(int B, int H) { return (size_t)4 * H * ((B + 15) / 16 * 16); }

int fn_frag_pack_host(const float* src, int rows, int K, int ld, float* dst, void*) {
    if (!src || !dst) return FN_E_NULL;
    if (rows <= 0 || K <= 0 || (K % 32) != 0 || ld < K) return FN_E_SHAPE;
    const int rp = (rows + 15) / 16 * 16;
    for (int r = 0; r < rp; ++r)
        for (int k = 0; k < K; ++k) dst[frag_off(r, k, K >> 5)] = r < rows ? src[(long)r * ld + k] : 0.0f;
    return FN_OK;
}

int fn_gru_seq_fwd_host(const FnGruFwd* scans, int n_scans, void*) {
    if (!scans) return FN_E_NULL;
    if (n_scans <= 0 || n_scans > FN_MAX_SCANS) return FN_E_COUNT;
    for (int si = 0; si < n_scans; ++si) {
        const FnGruFwd& d = scans[si];
        if (!d.w_hh_frag || !d.b_hh || !d.h_all || !d.frag_ws) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 32) != 0) return FN_E_SHAPE;
        if (d.gx_table && !d.idx) return FN_E_NULL;
        if (d.h0_frag && !d.h0) return FN_E_NULL;
    }
    for (int si = 0; si < n_scans; ++si) {
        const FnGruFwd& d = scans[si];
        const int B = d.B, T = d.T, H = d.H, nrt = (B + 15) / 16;
        const std::vector<float> W = unfrag(d.w_hh_frag, 3 * H, H);              // [3H][H]
        std::vector<float> h((size_t)B * H, 0.0f), hn((size_t)B * H), gh(3 * (size_t)H);
        if (d.h0) std::memcpy(h.data(), d.h0, sizeof(float) * B * H);
        for (int p = 0; p < T; ++p) {
            const int tau = (d.reverse ? T - 1 - p : p) + d.idx_shift;
            for (int b = 0; b < B; ++b) {
                const float* hb = &h[(size_t)b * H];
                for (int j = 0; j < 3 * H; ++j) {
                    float a = 0.0f;
                    const float* w = &W[(size_t)j * H];
                    for (int k = 0; k < H; ++k) a = std::fmaf(hb[k], w[k], a);
                    gh[j] = a + d.b_hh[j];
                }
                const int tok = d.gx_table ? (tau >= 0 ? d.idx[(long)b * d.idx_ld + tau] : d.start_token) : 0;
                for (int u = 0; u < H; ++u) {
                    float gx[3];
                    for (int q = 0; q < 3; ++q) {
                        float x = 0.0f;
                        if (d.gx_table) x = d.gx_table[(long)tok * 3 * H + q * H + u];
                        if (d.gx_dense) x += d.gx_dense[((long)p * B + b) * 3 * H + q * H + u];
                        gx[q] = ((d.b_ih ? d.b_ih[q * H + u] : 0.0f) + x) + (d.gx_rowbias ? d.gx_rowbias[(long)b * 3 * H + q * H + u] : 0.0f);
                    }
                    const float r = sigmoidf(gx[0] + gh[u]);
                    const float z = sigmoidf(gx[1] + gh[H + u]);
                    const float n = std::tanh(gx[2] + r * gh[2 * H + u]);
                    hn[(size_t)b * H + u] = (1.0f - z) * n + z * hb[u];
                    if (d.gates) {
                        float* gt = d.gates + (long)p * 4 * H * nrt * 16;
                        gt[gate_off(b, 0, u, nrt)] = r;
                        gt[gate_off(b, 1, u, nrt)] = z;
                        gt[gate_off(b, 2, u, nrt)] = n;
                        gt[gate_off(b, 3, u, nrt)] = gh[2 * H + u];
                    }
                }
            }
            h.swap(hn);
            std::memcpy(d.h_all + (long)p * B * H, h.data(), sizeof(float) * B * H);
        }
        if (d.h_last_frag) fn_frag_pack_host(h.data(), B, H, H, d.h_last_frag, nullptr);
    }
    return FN_OK;
}

int fn_gru_seq_bwd_host(const FnGruBwd* scans, int n_scans, void*) {
    if (!scans) return FN_E_NULL;
    if (n_scans <= 0 || n_scans > FN_MAX_SCANS) return FN_E_COUNT;
    for (int si = 0; si < n_scans; ++si) {
        const FnGruBwd& d = scans[si];
        if (!d.w_hh_t_frag || !d.h_all || !d.gates || !d.dgx_all || !d.dghn_all || !d.scratch || !d.frag_ws) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 32) != 0) return FN_E_SHAPE;
    }
    for (int si = 0; si < n_scans; ++si) {
        const FnGruBwd& d = scans[si];
        const int B = d.B, T = d.T, H = d.H, nrt = (B + 15) / 16;
        const std::vector<float> Wt = unfrag(d.w_hh_t_frag, H, 3 * H);           // W_hh^T [H][3H]
        std::vector<float> carry((size_t)B * H, 0.0f), df(3 * (size_t)H);
        if (d.dh_last) std::memcpy(carry.data(), d.dh_last, sizeof(float) * B * H);
        for (int q = T - 1; q >= 0; --q) {
            const float* gt = d.gates + (long)q * 4 * H * nrt * 16;
            for (int b = 0; b < B; ++b) {
                float* cb = &carry[(size_t)b * H];
                for (int u = 0; u < H; ++u) {
                    const float dh = cb[u] + (d.dh_ext ? d.dh_ext[((long)q * B + b) * H + u] : 0.0f);
                    const float r = gt[gate_off(b, 0, u, nrt)], z = gt[gate_off(b, 1, u, nrt)], n = gt[gate_off(b, 2, u, nrt)],
                                hn = gt[gate_off(b, 3, u, nrt)];
                    const float hp = q > 0 ? d.h_all[((long)(q - 1) * B + b) * H + u] : (d.h0 ? d.h0[(long)b * H + u] : 0.0f);
                    const float dn = dh * (1.0f - z), dz = dh * (hp - n), dnp = dn * (1.0f - n * n), dr = dnp * hn;
                    const float drp = dr * r * (1.0f - r), dzp = dz * z * (1.0f - z), dg = dnp * r;
                    float* dgx = d.dgx_all + ((long)q * B + b) * 3 * H;
                    dgx[u] = drp; dgx[H + u] = dzp; dgx[2 * H + u] = dnp;
                    d.dghn_all[((long)q * B + b) * H + u] = dg;
                    if (d.dgx_rowsum) {
                        float* rs = d.dgx_rowsum + (long)b * 3 * H;
                        rs[u] += drp; rs[H + u] += dzp; rs[2 * H + u] += dnp;
                    }
                    if (d.dghn_rowsum) d.dghn_rowsum[(long)b * H + u] += dg;
                    df[u] = drp; df[H + u] = dzp; df[2 * H + u] = dg;
                    cb[u] = dh * z;
                }
                for (int u = 0; u < H; ++u) {                                    // + df W_hh  (column u of W_hh = row u of W_hh^T)
                    float a = 0.0f;
                    const float* w = &Wt[(size_t)u * 3 * H];
                    for (int k = 0; k < 3 * H; ++k) a = std::fmaf(df[k], w[k], a);
                    cb[u] += a;
                }
            }
        }
        if (d.dh0) std::memcpy(d.dh0, carry.data(), sizeof(float) * B * H);
    }
    return FN_OK;
}

int fn_latent_fwd_host(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K, const int32_t* labels,
                       float* sigma, float* z, float* ll, float* qy, int32_t* y, float* terms, void*) {
    if (!pre || !eps || !mu_lk || !lv_lk || !sigma || !z || !ll || !qy || !y || !terms) return FN_E_NULL;
    if (B <= 0 || Z <= 0 || K <= 0 || K > KMAX) return FN_E_SHAPE;
    for (int b = 0; b < B; ++b) {
        double l[KMAX], klm[KMAX];
        for (int k = 0; k < K; ++k) l[k] = klm[k] = 0.0;
        for (int d = 0; d < Z; ++d) {
            const float mu = pre[(long)b * 2 * Z + d], s = std::exp(pre[(long)b * 2 * Z + Z + d]);
            const float zz = mu + s * eps[(long)b * Z + d];
            sigma[(long)b * Z + d] = s;
            z[(long)b * Z + d] = zz;
            for (int k = 0; k < K; ++k) {
                const double mk = mu_lk[k * Z + d], lv = lv_lk[k * Z + d];
                const double t = (double)zz - mk;
                l[k] += -0.5 * (t * t / std::exp(lv) + lv + LOG_2PI);             // likelihood: exp(logvar) as VARIANCE (gmm_model.py:194-218)
                const double sp = std::exp(lv), vr = ((double)s / sp) * ((double)s / sp), t1 = (((double)mu - mk) / sp) * (((double)mu - mk) / sp);
                klm[k] += 0.5 * (vr + t1 - 1.0 - std::log(vr));                   // KL: exp(logvar) as STD (trainer_gmm.py:150-178)
            }
        }
        double mx = -1e300;
        for (int k = 0; k < K; ++k) { l[k] += std::log(1.0 / K); klm[k] /= Z; mx = std::fmax(mx, l[k]); }
        double den = 0.0;
        for (int k = 0; k < K; ++k) den += std::exp(l[k] - mx);
        double q[KMAX], t0 = 0.0, t1s = 0.0;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            q[k] = std::exp(l[k] - mx) / den;
            ll[(long)b * K + k] = (float)l[k];
            qy[(long)b * K + k] = (float)q[k];
            if (q[k] > q[arg]) arg = k;
            t0 += q[k] * klm[k];
            t1s += q[k] * (l[k] - mx - std::log(den));                            // q log_softmax(ll)
        }
        y[b] = arg;
        float* tr = terms + (long)b * 4;
        tr[0] = (float)t0; tr[1] = (float)(t1s / K); tr[2] = tr[3] = 0.0f;
        if (labels) {
            const int lb = labels[b];
            double qm = 0.0, qd = 0.0;
            for (int k = 0; k < K; ++k) qm = std::fmax(qm, q[k]);
            for (int k = 0; k < K; ++k) qd += std::exp(q[k] - qm);
            tr[2] = (float)klm[lb];
            tr[3] = (float)-(q[lb] - qm - std::log(qd));                          // CrossEntropy applied to PROBABILITIES (trainer_gmm.py:181-194)
        }
    }
    return FN_OK;
}

int fn_latent_bwd_host(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K, const int32_t* labels,
                       const float* zin, const float* qy, const float* g_z, const float* g_mu, const float* g_sigma, const float* g_ll,
                       const float* g_qy, const float* w3, float* dpre, float* dmu_lk_rows, void*) {
    if (!pre || !eps || !mu_lk || !lv_lk || !zin || !qy || !dpre) return FN_E_NULL;
    if (B <= 0 || Z <= 0 || K <= 0 || K > KMAX) return FN_E_SHAPE;
    const float w_lat = w3 ? w3[0] : 0.f, w_cls = w3 ? w3[1] : 0.f, w_clf = w3 ? w3[2] : 0.f;
    for (int b = 0; b < B; ++b) {
        const int lb = labels ? labels[b] : -1;
        double klk[KMAX] = {0};
        if (w_lat != 0.f && !labels) {
            for (int d = 0; d < Z; ++d) {
                const double mu = pre[(long)b * 2 * Z + d], s = std::exp(pre[(long)b * 2 * Z + Z + d]);
                for (int k = 0; k < K; ++k) {
                    const double sp = std::exp((double)lv_lk[k * Z + d]), vr = (s / sp) * (s / sp), t = (mu - mu_lk[k * Z + d]) / sp;
                    klk[k] += 0.5 * (vr + t * t - 1.0 - std::log(vr));
                }
            }
            for (int k = 0; k < K; ++k) klk[k] /= Z;
        }
        double q[KMAX], dq[KMAX], dll[KMAX], qmax = -1.0, qden = 0.0, dot = 0.0;
        for (int k = 0; k < K; ++k) { q[k] = qy[(long)b * K + k]; qmax = std::fmax(qmax, q[k]); }
        if (labels && w_clf != 0.f)
            for (int k = 0; k < K; ++k) qden += std::exp(q[k] - qmax);
        for (int k = 0; k < K; ++k) {
            double g = g_qy ? g_qy[(long)b * K + k] : 0.0;
            if (!labels) {
                g += (double)w_lat * klk[k];
                g += (double)w_cls * (std::log(std::fmax(q[k], 1e-300)) + 1.0) / K;
            } else if (w_clf != 0.f) {
                g += (double)w_clf * (std::exp(q[k] - qmax) / qden - (k == lb ? 1.0 : 0.0));
            }
            dq[k] = g;
            dot += q[k] * g;
        }
        for (int k = 0; k < K; ++k) dll[k] = q[k] * (dq[k] - dot) + (g_ll ? g_ll[(long)b * K + k] : 0.0);
        for (int d = 0; d < Z; ++d) {
            const float mu = pre[(long)b * 2 * Z + d], s = std::exp(pre[(long)b * 2 * Z + Z + d]), e = eps[(long)b * Z + d], zz = zin[(long)b * Z + d];
            float dz = g_z ? g_z[(long)b * Z + d] : 0.f, dmu = g_mu ? g_mu[(long)b * Z + d] : 0.f, dsg = g_sigma ? g_sigma[(long)b * Z + d] : 0.f;
            for (int k = 0; k < K; ++k) {
                const float mk = mu_lk[k * Z + d], lv = lv_lk[k * Z + d], iv = std::exp(-lv), t = (zz - mk) * iv;
                dz += (float)dll[k] * (-t);
                float dmk = (float)dll[k] * t;
                const float wk = labels ? (k == lb ? w_lat : 0.f) : w_lat * (float)q[k];
                if (wk != 0.f) {
                    const float sp = std::exp(lv), isp2 = 1.0f / (sp * sp);
                    dmu += wk / (float)Z * (mu - mk) * isp2;
                    dsg += wk / (float)Z * (s * isp2 - 1.0f / s);
                    dmk += wk / (float)Z * (-(mu - mk)) * isp2;
                }
                if (dmu_lk_rows) dmu_lk_rows[((long)b * K + k) * Z + d] = dmk;
            }
            dmu += dz;
            dsg += dz * e;
            dpre[(long)b * 2 * Z + d] = dmu;
            dpre[(long)b * 2 * Z + Z + d] = dsg * s;
        }
    }
    return FN_OK;
}

int fn_out_head_f32_host(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int T, int V, int H, const int32_t* target,
                         float grad_scale, float* nll_rows, float* dlogits, int ld, void*) {
    if (!h || !W || !bias || !target) return FN_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || H <= 0 || ldh < H || ldw < H) return FN_E_SHAPE;
    if (V > 384) return FN_E_UNSUPPORTED;
    if (dlogits && (ld < V || ld > 384 || (ld & 3))) return FN_E_SHAPE;
    std::vector<double> lg(V);
    for (long row = 0; row < (long)T * B; ++row) {
        const int t = (int)(row / B), b = (int)(row % B);
        double mx = -1e300;
        for (int v = 0; v < V; ++v) {
            float a = 0.0f;
            for (int k = 0; k < H; ++k) a = std::fmaf(h[row * ldh + k], W[(long)v * ldw + k], a);
            lg[v] = (double)(a + bias[v]);
            mx = std::fmax(mx, lg[v]);
        }
        double den = 0.0;
        for (int v = 0; v < V; ++v) den += std::exp(lg[v] - mx);
        const int tg = target[(long)b * T + t];
        if (nll_rows) nll_rows[row] = (float)-(lg[tg] - mx - std::log(den));
        if (dlogits) {
            for (int v = 0; v < V; ++v) dlogits[row * ld + v] = grad_scale * (float)(std::exp(lg[v] - mx) / den - (v == tg ? 1.0 : 0.0));
            for (int v = V; v < ld; ++v) dlogits[row * ld + v] = 0.0f;
        }
    }
    return FN_OK;
}

int fn_sumsq_f32_host(const float* g, int64_t n, float* out, float*, size_t, void*) {
    if (!g || !out) return FN_E_NULL;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += (double)g[i] * g[i];
    out[0] = (float)s;
    return FN_OK;
}

int fn_clip_adam_host(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm, const float* hyper, float beta1,
                      float beta2, float eps, void*) {
    if (!p || !g || !m || !v || !sumsq || !hyper) return FN_E_NULL;
    if (n <= 0) return FN_E_SHAPE;
    const float coef = std::fmin(1.0f, max_norm / (std::sqrt(sumsq[0]) + 1e-6f));      // clip_grad_norm_ (trainer_gmm.py:250)
    for (int64_t i = 0; i < n; ++i) {
        const float gg = g[i] * coef;
        m[i] = beta1 * m[i] + (1.0f - beta1) * gg;
        v[i] = beta2 * v[i] + (1.0f - beta2) * gg * gg;
        p[i] -= hyper[0] * m[i] / (std::sqrt(v[i]) * hyper[1] + eps);                  // hyper = {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)}
    }
    return FN_OK;
}


// ---- dense products (gemm.hip) ------------------------------------------------------------------------------------------------------
size_t fn_gemm_ws_bytes_host(int M, int N, int splitk) { return (splitk & 0xffff) > 1 ? (size_t)(splitk & 0xffff) * M * N * sizeof(float) : 0; }

int fn_gemm_f32_host(int a_kmajor, int b_kmajor, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta,
                     float* Cm, int ldc, const float* bias, int splitk, float* ws, size_t ws_bytes, void*) {
    if (!A || !B || !Cm) return FN_E_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return FN_E_SHAPE;
    if (lda < (a_kmajor ? K : M) || ldb < (b_kmajor ? K : N)) return FN_E_SHAPE;
    const int sk = splitk & 0xffff;                        // FN_GEMM_LEAN selects a kernel instance, not an arithmetic
    if (sk > 1 && (!ws || ws_bytes < fn_gemm_ws_bytes_host(M, N, sk))) return FN_E_WORKSPACE;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float a = 0.0f;
            for (int k = 0; k < K; ++k) {
                const float x = a_kmajor ? A[(long)m * lda + k] : A[(long)k * lda + m];
                const float y = b_kmajor ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
                a = std::fmaf(x, y, a);
            }
            float c = alpha * a;
            if (beta != 0.0f) c += beta * Cm[(long)m * ldc + n];
            if (bias) c += bias[n];
            Cm[(long)m * ldc + n] = c;
        }
    return FN_OK;
}

size_t fn_gru_dwhh_ws_bytes_host(int H, int splitk) { return fn_gemm_ws_bytes_host(3 * H, H, splitk); }

int fn_gru_dwhh_f32_host(const float* dgx, const float* dghn, const float* hprev, int64_t rows, int H, float beta, float* dW, int splitk, float* ws,
                         size_t ws_bytes, void*) {
    if (!dgx || !dghn || !hprev || !dW) return FN_E_NULL;
    if (rows <= 0 || H <= 0) return FN_E_SHAPE;
    if ((splitk & 0xffff) > 1 && (!ws || ws_bytes < fn_gru_dwhh_ws_bytes_host(H, splitk))) return FN_E_WORKSPACE;
    for (int j = 0; j < 3 * H; ++j)
        for (int k = 0; k < H; ++k) {
            double a = 0.0;                                // rows reaches 65 280: the device sums 16 K ranges of fp32 chains, the twin sums in double
            for (int64_t r = 0; r < rows; ++r) {
                const float g = j < 2 * H ? dgx[r * 3 * H + j] : dghn[r * H + (j - 2 * H)];
                a += (double)g * hprev[r * H + k];
            }
            dW[(long)j * H + k] = (float)(a + (beta != 0.0f ? (double)beta * dW[(long)j * H + k] : 0.0));
        }
    return FN_OK;
}

// ---- token sort + segment sums (embed.hip) --------------------------------------------------------------------------------------------
namespace {
constexpr int EG_PIECE_H = 256;                            // rows per partial sum (EG_PIECE of embed.hip: part of the image's contract)
}
size_t fn_token_sort_ints_host(int64_t rows, int V) { return (size_t)2 * (V + 1) + 2 + (size_t)rows; }
size_t fn_token_sort_ws_bytes_host(int64_t, int V) { return (size_t)V * sizeof(int32_t) + 16; }

int fn_token_sort_host(const int32_t* idx, int B, int T, int idx_ld, int V, int32_t* img, void* ws, size_t ws_bytes, void*) {
    if (!idx || !img || !ws) return FN_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || V > 1024 || idx_ld < T) return FN_E_SHAPE;
    if (ws_bytes < fn_token_sort_ws_bytes_host((int64_t)B * T, V)) return FN_E_WORKSPACE;
    if (((uintptr_t)ws) & 15) return FN_E_ALIGN;
    const long rows = (long)B * T;
    int32_t* seg = img;                                    // [V + 1] first sorted position of every token
    int32_t* pstart = img + (V + 1);                       // [V + 1] first piece of every token (pieces of EG_PIECE rows)
    int32_t* order = img + 2 * (V + 1) + 2;                // [rows]  positions r = tau * B + b, sorted by token, stable
    auto tok = [&](long r) { const int t = idx[(r % B) * idx_ld + r / B]; return t < 0 ? 0 : (t >= V ? V - 1 : t); };
    int32_t* fill = reinterpret_cast<int32_t*>(ws);
    for (int v = 0; v < V; ++v) fill[v] = 0;
    for (long r = 0; r < rows; ++r) ++fill[tok(r)];
    seg[0] = pstart[0] = 0;
    for (int v = 0; v < V; ++v) {
        seg[v + 1] = seg[v] + fill[v];
        pstart[v + 1] = pstart[v] + (fill[v] + EG_PIECE_H - 1) / EG_PIECE_H;
        fill[v] = seg[v];
    }
    for (long r = 0; r < rows; ++r) order[fill[tok(r)]++] = (int32_t)r;
    return FN_OK;
}

size_t fn_embed_grad_sorted_ws_bytes_host(int64_t, int, int, int N3, int n_jobs) { return (size_t)n_jobs * N3 * sizeof(double) + 16; }

int fn_embed_grad_sorted_host(const FnEmbedGrad* jobs, int n_jobs, int B, int T, int N3, int V, const int32_t* img, float* ws, size_t ws_bytes, void*) {
    if (!jobs || !img || !ws) return FN_E_NULL;
    if (n_jobs <= 0 || n_jobs > 8) return FN_E_COUNT;
    if (B <= 0 || T <= 0 || N3 <= 0 || (N3 & 3) || V <= 0 || V > 1024) return FN_E_SHAPE;
    if (ws_bytes < fn_embed_grad_sorted_ws_bytes_host((int64_t)B * T, B, V, N3, n_jobs)) return FN_E_WORKSPACE;
    if (((uintptr_t)ws) & 15) return FN_E_ALIGN;
    const int32_t* seg = img;
    const int32_t* order = img + 2 * (V + 1) + 2;
    for (int j = 0; j < n_jobs; ++j) {
        const FnEmbedGrad& d = jobs[j];
        if (!d.dgx_all || !d.out) return FN_E_NULL;
        if (d.idx_shift > 0 || d.idx_shift < -1 || (d.idx_shift && d.reverse) || d.out_ld < (d.transposed ? V : N3)) return FN_E_SHAPE;
        if (d.idx_shift < 0 && (d.start_token < 0 || d.start_token >= V)) return FN_E_SHAPE;
    }
    std::vector<double> acc(N3);
    for (int j = 0; j < n_jobs; ++j) {
        const FnEmbedGrad& d = jobs[j];
        for (int v = 0; v < V; ++v) {
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int s = seg[v]; s < seg[v + 1]; ++s) {
                const long r = order[s];
                const int tau = (int)(r / B), b = (int)(r % B);
                // the step that READ the token at time tau: p = tau (forward), T - 1 - tau (reverse scan), tau + 1 (input shifted by one)
                const int p = d.idx_shift < 0 ? tau + 1 : (d.reverse ? T - 1 - tau : tau);
                if (p >= T) continue;
                const float* row = d.dgx_all + ((long)p * B + b) * N3;
                for (int c = 0; c < N3; ++c) acc[c] += row[c];
            }
            if (d.idx_shift < 0 && v == d.start_token)     // step 0 of every sequence read the start token
                for (int b = 0; b < B; ++b)
                    for (int c = 0; c < N3; ++c) acc[c] += d.dgx_all[(long)b * N3 + c];
            for (int c = 0; c < N3; ++c) {
                if (d.transposed) d.out[(long)c * d.out_ld + v] = (float)acc[c];
                else d.out[(long)v * d.out_ld + c] = (float)acc[c];
            }
        }
    }
    return FN_OK;
}

// ---- time-axis log_softmax of the sub-decoders, pairwise regulariser (loss.hip) -----------------------------------------------------------
int fn_time_logsoftmax_host(const float* logits, int B, int Tr, int Cc, float* logp_bt, const int32_t* target, float* nll_bc, float grad_scale,
                            float* dlogits, void*) {
    if (!logits) return FN_E_NULL;
    if (B <= 0 || Tr <= 0 || Cc <= 0) return FN_E_SHAPE;
    if ((nll_bc || dlogits) && !target) return FN_E_NULL;
    const long st = (long)B * Cc;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cc; ++c) {
            const float* x = logits + (long)b * Cc + c;
            float mx = -INFINITY;
            for (int t = 0; t < Tr; ++t) mx = std::fmax(mx, x[t * st]);
            double s = 0.0;
            for (int t = 0; t < Tr; ++t) s += std::exp((double)x[t * st] - mx);
            const float lse = mx + (float)std::log(s);
            float nll = 0.0f;
            int cnt = 0;
            for (int t = 0; t < Tr; ++t) {
                const float l = x[t * st] - lse;
                if (logp_bt) logp_bt[((long)b * Tr + t) * Cc + c] = l;
                if (target && target[(long)b * Tr + t] == c) { nll -= l; ++cnt; }
            }
            if (nll_bc) nll_bc[(long)b * Cc + c] = nll;
            if (dlogits)
                for (int t = 0; t < Tr; ++t) {
                    const float l = x[t * st] - lse;
                    dlogits[t * st + (long)b * Cc + c] = grad_scale * (std::exp(l) * (float)cnt - (target[(long)b * Tr + t] == c ? 1.0f : 0.0f));
                }
        }
    return FN_OK;
}

int fn_time_logsoftmax_bwd_host(const float* logp_bt, const float* gout_bt, int B, int Tr, int Cc, float* dlogits, void*) {
    if (!logp_bt || !gout_bt || !dlogits) return FN_E_NULL;
    if (B <= 0 || Tr <= 0 || Cc <= 0) return FN_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cc; ++c) {
            float s = 0.0f;
            for (int t = 0; t < Tr; ++t) s += gout_bt[((long)b * Tr + t) * Cc + c];
            for (int t = 0; t < Tr; ++t) {
                const long o = ((long)b * Tr + t) * Cc + c;
                dlogits[(long)t * B * Cc + (long)b * Cc + c] = gout_bt[o] - std::exp(logp_bt[o]) * s;
            }
        }
    return FN_OK;
}

int fn_pairwise_reg_host(const float* z0_all, const double* attr_all, int n_all, int row0, int nrows, float* loss_rows, float grad_scale, float* dz0,
                         void*) {
    if (!z0_all || !attr_all || !loss_rows) return FN_E_NULL;
    if (n_all <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > n_all) return FN_E_SHAPE;
    for (int i = 0; i < nrows; ++i) {
        const float zi = z0_all[row0 + i];
        const double ai = attr_all[row0 + i];
        double l = 0.0, g = 0.0;
        for (int j = 0; j < n_all; ++j) {
            const float th = std::tanh(zi - z0_all[j]);
            const double da = ai - attr_all[j];
            const float sg = da > 0.0 ? 1.0f : (da < 0.0 ? -1.0f : 0.0f);
            const float df = th - sg;
            l += (double)df * df;
            g += (double)df * (1.0f - th * th);
        }
        loss_rows[i] = (float)l;
        if (dz0) dz0[i] = grad_scale * 4.0f * (float)g;
    }
    return FN_OK;
}

// ---- eval-mode decode (gru.hip: one cell for a large batch; decode_persist.hip: the whole greedy decode) ---------------------------------
int fn_gru_cell_f32_host(const FnGruCell* c, void*) {
    if (!c) return FN_E_NULL;
    if (!c->h_prev || !c->w_hh || !c->b_hh || !c->h_out) return FN_E_NULL;
    if (c->B <= 0 || c->H <= 0 || (c->H % 32) != 0 || c->ldh < c->H || c->ldo < c->H || c->ldw_hh < c->H) return FN_E_SHAPE;
    if (c->x && (!c->w_ih || c->K1 <= 0 || c->ldx < c->K1 || c->ldw_ih < c->K1)) return FN_E_SHAPE;
    if (c->h_out == c->h_prev) return FN_E_SHAPE;
    if (c->idx_best && (c->best_v <= 0 || !c->gx_table)) return FN_E_SHAPE;
    const int B = c->B, H = c->H;
    std::vector<float> gi(3 * (size_t)H), gh(3 * (size_t)H);
    for (int b = 0; b < B; ++b) {
        const float* hp = c->h_prev + (long)b * c->ldh;
        const int tok = c->idx_best ? c->best_v - 1 - (int)(uint32_t)(c->idx_best[b] & 0xffffffffull) : (c->idx ? c->idx[(long)b * c->idx_ld] : c->start_token);
        for (int j = 0; j < 3 * H; ++j) {
            float a = 0.0f;
            for (int k = 0; k < H; ++k) a = std::fmaf(hp[k], c->w_hh[(long)j * c->ldw_hh + k], a);
            gh[j] = a + c->b_hh[j];
            float x = 0.0f;
            if (c->x)
                for (int k = 0; k < c->K1; ++k) x = std::fmaf(c->x[(long)b * c->ldx + k], c->w_ih[(long)j * c->ldw_ih + k], x);
            if (c->b_ih) x += c->b_ih[j];
            if (c->gx_table) x += c->gx_table[(long)tok * 3 * H + j];
            if (c->gx_rowbias) x += c->gx_rowbias[(long)b * 3 * H + j];
            gi[j] = x;
        }
        for (int u = 0; u < H; ++u) {
            const float r = sigmoidf(gi[u] + gh[u]);
            const float z = sigmoidf(gi[H + u] + gh[H + u]);
            const float n = std::tanh(gi[2 * H + u] + r * gh[2 * H + u]);
            c->h_out[(long)b * c->ldo + u] = (1.0f - z) * n + z * hp[u];
        }
    }
    return FN_OK;
}

int fn_out_argmax_f32_host(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int V, int K, uint64_t* best, void*) {
    if (!h || !W || !bias || !best) return FN_E_NULL;
    if (B <= 0 || V <= 0 || K <= 0 || (K % 16) != 0 || ldh < K || ldw < K || (ldh & 3) || (ldw & 3)) return FN_E_SHAPE;
    for (int b = 0; b < B; ++b)
        for (int v = 0; v < V; ++v) {
            float a = 0.0f;
            for (int k = 0; k < K; ++k) a = std::fmaf(h[(long)b * ldh + k], W[(long)v * ldw + k], a);
            a += bias[v];
            uint32_t bits;
            std::memcpy(&bits, &a, 4);
            const uint32_t key = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
            const uint64_t w = ((uint64_t)key << 32) | (uint32_t)(V - 1 - v);
            if (w > best[b]) best[b] = w;
        }
    return FN_OK;
}

int fn_best_tokens_host(const uint64_t* best, int steps, int B, int V, int32_t* tokens, int tok_ld, void*) {
    if (!best || !tokens) return FN_E_NULL;
    if (steps <= 0 || B <= 0 || V <= 0 || tok_ld < steps) return FN_E_SHAPE;
    for (int t = 0; t < steps; ++t)
        for (int b = 0; b < B; ++b) tokens[(long)b * tok_ld + t] = V - 1 - (int)(uint32_t)(best[(long)t * B + b] & 0xffffffffull);
    return FN_OK;
}

size_t fn_decode_ws_bytes_host(int B, int H, int) { return (size_t)4 * B * H * sizeof(float) + 16; }
size_t fn_decode_sync_ws_bytes_host(void) { return 16; }

int fn_decode_greedy_host(const FnDecode* d, void*) {
    if (!d) return FN_E_NULL;
    if (!d->w_hh1_frag || !d->b_hh1 || !d->table1 || !d->h0 || !d->w_ih2_frag || !d->w_hh2_frag || !d->b_hh2 || !d->w_out_frag || !d->b_out ||
        !d->tokens || !d->ws || !d->sync_ws)
        return FN_E_NULL;
    if (d->B <= 0 || d->B > 2048 || d->steps <= 0 || d->H <= 0 || d->H > 512 || (d->H % 32) != 0 || d->V <= 0 || d->tok_ld < d->steps) return FN_E_SHAPE;
    if (d->start_token < 0 || d->start_token >= d->V) return FN_E_SHAPE;
    const int B = d->B, H = d->H, V = d->V;
    const std::vector<float> Whh1 = unfrag(d->w_hh1_frag, 3 * H, H), Wih2 = unfrag(d->w_ih2_frag, 3 * H, H), Whh2 = unfrag(d->w_hh2_frag, 3 * H, H),
                             Wout = unfrag(d->w_out_frag, (V + 15) / 16 * 16, H);
    std::vector<float> h1((size_t)B * H), h2((size_t)B * H), n1(H), n2(H), ga(3 * (size_t)H), gb(3 * (size_t)H), lg(V);
    std::memcpy(h1.data(), d->h0, sizeof(float) * B * H);
    auto mv = [&](const std::vector<float>& W, const float* x, const float* bias, std::vector<float>& out) {
        for (int j = 0; j < 3 * H; ++j) {
            float a = 0.0f;
            for (int k = 0; k < H; ++k) a = std::fmaf(x[k], W[(size_t)j * H + k], a);
            out[j] = a + (bias ? bias[j] : 0.0f);
        }
    };
    for (int b = 0; b < B; ++b) {
        float* s1 = &h1[(size_t)b * H];
        float* s2 = &h2[(size_t)b * H];
        int tok = d->start_token;
        for (int t = 0; t < d->steps; ++t) {
            mv(Whh1, s1, d->b_hh1, gb);                                                                  // layer 1 (gmm_model.py:131-133)
            for (int u = 0; u < H; ++u) {
                float gx[3];
                for (int q = 0; q < 3; ++q)
                    gx[q] = ((d->b_ih1 ? d->b_ih1[q * H + u] : 0.0f) + d->table1[(long)tok * 3 * H + q * H + u]) +
                            (d->rowbias1 ? d->rowbias1[(long)b * 3 * H + q * H + u] : 0.0f);
                const float r = sigmoidf(gx[0] + gb[u]), z = sigmoidf(gx[1] + gb[H + u]);
                const float n = std::tanh(gx[2] + r * gb[2 * H + u]);
                n1[u] = (1.0f - z) * n + z * s1[u];
            }
            std::memcpy(s1, n1.data(), sizeof(float) * H);
            if (t == 0) std::memcpy(s2, s1, sizeof(float) * H);                                          // hx1 <- hx0 at i == 0 (:134-135)
            mv(Wih2, s1, d->b_ih2, ga);                                                                  // layer 2 (:136)
            mv(Whh2, s2, d->b_hh2, gb);
            for (int u = 0; u < H; ++u) {
                const float r = sigmoidf(ga[u] + gb[u]), z = sigmoidf(ga[H + u] + gb[H + u]);
                const float n = std::tanh(ga[2 * H + u] + r * gb[2 * H + u]);
                n2[u] = (1.0f - z) * n + z * s2[u];
            }
            std::memcpy(s2, n2.data(), sizeof(float) * H);
            float mx = -INFINITY;                                                                        // output layer + log_softmax + argmax (:137,147-148)
            int best = 0;
            for (int v = 0; v < V; ++v) {
                float a = 0.0f;
                for (int k = 0; k < H; ++k) a = std::fmaf(s2[k], Wout[(size_t)v * H + k], a);
                lg[v] = a + d->b_out[v];
                if (lg[v] > mx) { mx = lg[v]; best = v; }                                                // first index on ties
            }
            if (d->logp) {
                double den = 0.0;
                for (int v = 0; v < V; ++v) den += std::exp((double)lg[v] - mx);
                const float lse = mx + (float)std::log(den);
                for (int v = 0; v < V; ++v) d->logp[((long)b * d->steps + t) * V + v] = lg[v] - lse;
            }
            d->tokens[(long)b * d->tok_ld + t] = best;
            tok = best;
        }
    }
    return FN_OK;
}

}  // extern "C"
