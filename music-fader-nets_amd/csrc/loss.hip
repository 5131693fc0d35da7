// loss.hip - output heads and latent block of the GM-VAE path:
//   vocab log_softmax + NLL (gmm_model.py:137, trainer_gmm.py:131-132), the sub-decoders' TIME-axis
//   log_softmax (gmm_model.py:110,115), the reparameterised sample + Gaussian-mixture posterior
//   (gmm_model.py:86,91,194-218,229-242), the KL / class terms (trainer_gmm.py:150-194) and the pairwise
//   latent regulariser (trainer_gmm.py:199-217).  All reductions are wavefront-level (64 lanes).
#include "common.h"

namespace {

constexpr float LOG_2PI = 1.8378770664093453f;

// ------------------------------------------------------------------ vocab axis ---------------
// one wavefront per row (row = t*B + b), 4 rows per 256-thread block
__global__ __launch_bounds__(256) void vocab_logsoftmax_kernel(const float* __restrict__ logits, int B, int T, int E, int ld,
                                                               float* __restrict__ logp_bt, const int* __restrict__ target,
                                                               float* __restrict__ nll_rows, float grad_scale,
                                                               float* dlogits) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int t = (int)(row / B), b = (int)(row % B);
    const float* x = logits + row * ld;
    float mx = -INFINITY;
    for (int e = lane; e < E; e += 64) mx = fmaxf(mx, x[e]);
    mx = fn_wave_max(mx);
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += expf(x[e] - mx);
    s = fn_wave_sum(s);
    const float lse = mx + logf(s);
    const int tg = target ? target[(long)b * T + t] : -1;
    if (nll_rows && lane == 0) nll_rows[row] = lse - x[tg];
    float* lp = logp_bt ? logp_bt + ((long)b * T + t) * E : nullptr;
    float* dl = dlogits ? dlogits + row * ld : nullptr;
    for (int e = lane; e < E; e += 64) {
        const float l = x[e] - lse;
        if (lp) lp[e] = l;
        if (dl) dl[e] = grad_scale * (expf(l) - (e == tg ? 1.0f : 0.0f));
    }
}

__global__ __launch_bounds__(256) void vocab_logsoftmax_bwd_kernel(const float* __restrict__ logp_bt, const float* __restrict__ gout_bt,
                                                                   int B, int T, int E, int ld, float* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int t = (int)(row / B), b = (int)(row % B);
    const float* lp = logp_bt + ((long)b * T + t) * E;
    const float* g = gout_bt + ((long)b * T + t) * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += g[e];
    s = fn_wave_sum(s);
    float* dl = dlogits + row * ld;
    for (int e = lane; e < E; e += 64) dl[e] = g[e] - expf(lp[e]) * s;
}

// greedy head: log_softmax + first-index argmax (torch.max semantics, gmm_model.py:74)
__global__ __launch_bounds__(256) void vocab_argmax_kernel(const float* __restrict__ logits, int B, int E, int ld,
                                                           float* __restrict__ logp_out, long logp_ld, int* __restrict__ tok_out,
                                                           int tok_ld) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* x = logits + (long)b * ld;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int e = lane; e < E; e += 64) {
        const float v = x[e];
        if (v > mx) { mx = v; am = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
    }
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += expf(x[e] - mx);
    s = fn_wave_sum(s);
    const float lse = mx + logf(s);
    if (logp_out)
        for (int e = lane; e < E; e += 64) logp_out[(long)b * logp_ld + e] = x[e] - lse;
    if (lane == 0) tok_out[(long)b * tok_ld] = am;
}

// ------------------------------------------------------------------ masked probability sums (GLSR, trainer_glsr.py:121-139) -----
// per row of logits: P_a = sum over tokens [lo_a, hi_a) of softmax(logits) for two token ranges (played notes 2..89, time separators
// 180..277 in the reference).  sums [rows][2].  Gradient form: dlogits[u] = p_u (a_u - (w0 P_0 + w1 P_1)),  a_u = w0 [u in range 0] +
// w1 [u in range 1], i.e. the backward of  w0 P_0 + w1 P_1  through the softmax; w [rows][2].  One wavefront per row.
__global__ __launch_bounds__(256) void masked_prob_kernel(const float* __restrict__ logits, long rows, int E, int ld, int lo0, int hi0,
                                                          int lo1, int hi1, float* __restrict__ sums, const float* __restrict__ w,
                                                          float* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = logits + row * ld;
    float mx = -INFINITY;
    for (int e = lane; e < E; e += 64) mx = fmaxf(mx, x[e]);
    mx = fn_wave_max(mx);
    float s = 0.f, s0 = 0.f, s1 = 0.f;
    for (int e = lane; e < E; e += 64) {
        const float p = expf(x[e] - mx);
        s += p;
        if (e >= lo0 && e < hi0) s0 += p;
        if (e >= lo1 && e < hi1) s1 += p;
    }
    s = fn_wave_sum(s);
    s0 = fn_wave_sum(s0) / s;
    s1 = fn_wave_sum(s1) / s;
    if (sums && lane == 0) { sums[row * 2] = s0; sums[row * 2 + 1] = s1; }
    if (dlogits) {
        const float w0 = w[row * 2], w1 = w[row * 2 + 1];
        const float dot = w0 * s0 + w1 * s1;
        float* d = dlogits + row * ld;
        for (int e = lane; e < E; e += 64) {
            const float p = expf(x[e] - mx) / s;
            const float a = ((e >= lo0 && e < hi0) ? w0 : 0.f) + ((e >= lo1 && e < hi1) ? w1 : 0.f);
            d[e] = p * (a - dot);                       // may alias logits: every lane reads x[e] before it writes d[e]
        }
    }
}

// ------------------------------------------------------------------ time axis ----------------
// one thread per (b, c); loops over Tr.  logits [Tr][B][Cc]
__global__ void time_logsoftmax_kernel(const float* __restrict__ logits, int B, int Tr, int Cc, float* __restrict__ logp_bt,
                                       const int* __restrict__ target, float* __restrict__ nll_bc, float grad_scale,
                                       float* __restrict__ dlogits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Cc) return;
    const int b = i / Cc, c = i % Cc;
    const long st = (long)B * Cc;
    const float* x = logits + (long)b * Cc + c;
    float mx = -INFINITY;
    for (int t = 0; t < Tr; ++t) mx = fmaxf(mx, x[t * st]);
    float s = 0.f;
    for (int t = 0; t < Tr; ++t) s += expf(x[t * st] - mx);
    const float lse = mx + logf(s);
    float nll = 0.f;
    int cnt = 0;
    for (int t = 0; t < Tr; ++t) {
        const float l = x[t * st] - lse;
        if (logp_bt) logp_bt[((long)b * Tr + t) * Cc + c] = l;
        if (target && target[(long)b * Tr + t] == c) { nll -= l; ++cnt; }
    }
    if (nll_bc) nll_bc[i] = nll;
    if (dlogits) {
        for (int t = 0; t < Tr; ++t) {
            const float l = x[t * st] - lse;
            const float hit = (target[(long)b * Tr + t] == c) ? 1.0f : 0.0f;
            dlogits[t * st + (long)b * Cc + c] = grad_scale * (expf(l) * (float)cnt - hit);
        }
    }
}

// the same for Tr <= 64 with one WAVEFRONT per column, lane t = time step t: the exponentials, the log-probabilities and the gradients of a column are
// computed side by side instead of by one thread walking 64 steps (36 - 46 us of dependent transcendental code on the side lane between the output
// head and the first backward launch).  The two sums keep the loop kernel's ORDER (s over t = 0, 1, ...; nll over the hits in time order) by walking
// the lanes with v_readlane, the maximum is order-free: bit-identical results.
__global__ __launch_bounds__(256) void time_logsoftmax_wave_kernel(const float* __restrict__ logits, int B, int Tr, int Cc, float* __restrict__ logp_bt,
                                                                    const int* __restrict__ target, float* __restrict__ nll_bc, float grad_scale,
                                                                    float* __restrict__ dlogits) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);          // column (b, c): uniform per wavefront
    if (i >= B * Cc) return;
    const int t = threadIdx.x & 63;
    const int b = i / Cc, c = i % Cc;
    const long st = (long)B * Cc;
    const bool on = t < Tr;
    const float v = on ? logits[t * st + (long)b * Cc + c] : -INFINITY;
    const int tg = (target && on) ? target[(long)b * Tr + t] : -1;
    float mx = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = on ? expf(v - mx) : 0.f;
    float s = 0.f;
    for (int u = 0; u < Tr; ++u) s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), u));
    const float lse = mx + logf(s);
    const float l = v - lse;
    if (on && logp_bt) logp_bt[((long)b * Tr + t) * Cc + c] = l;
    const bool hit = on && tg == c;
    const unsigned long long hits = __ballot(hit);
    if (nll_bc) {
        float nll = 0.f;
        for (int u = 0; u < Tr; ++u)
            if ((hits >> u) & 1ull) nll -= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), u));
        if (t == 0) nll_bc[i] = nll;
    }
    if (on && dlogits) {
        const int cnt = __popcll(hits);
        dlogits[t * st + (long)b * Cc + c] = grad_scale * (expf(l) * (float)cnt - (hit ? 1.0f : 0.0f));
    }
}

__global__ void time_logsoftmax_bwd_kernel(const float* __restrict__ logp_bt, const float* __restrict__ gout_bt, int B, int Tr,
                                           int Cc, float* __restrict__ dlogits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Cc) return;
    const int b = i / Cc, c = i % Cc;
    float s = 0.f;
    for (int t = 0; t < Tr; ++t) s += gout_bt[((long)b * Tr + t) * Cc + c];
    for (int t = 0; t < Tr; ++t) {
        const long o = ((long)b * Tr + t) * Cc + c;
        dlogits[(long)t * B * Cc + (long)b * Cc + c] = gout_bt[o] - expf(logp_bt[o]) * s;
    }
}

// ------------------------------------------------------------------ latent block -------------
constexpr int KMAX = 8;

// KL( N(mu, s) || N(m_k, exp(lv_k)) ) per dimension; exp(logvar) is used as the STD (trainer_gmm.py:156-157)
__device__ __forceinline__ float kl_dim(float mu, float s, float mk, float lvk) {
    const float sp = expf(lvk);
    const float vr = (s / sp) * (s / sp);
    const float t1 = ((mu - mk) / sp) * ((mu - mk) / sp);
    return 0.5f * (vr + t1 - 1.0f - logf(vr));
}

__device__ __forceinline__ double kl_dim_f64(double mu, double s, double mk, double lvk) {
    const double sp = exp(lvk);
    const double vr = (s / sp) * (s / sp);
    const double t1 = ((mu - mk) / sp) * ((mu - mk) / sp);
    return 0.5 * (vr + t1 - 1.0 - log(vr));
}

__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ eps,
                                                         const float* __restrict__ mu_lk, const float* __restrict__ lv_lk, int B, int Z,
                                                         int K, const int* __restrict__ labels, float* __restrict__ sigma,
                                                         float* __restrict__ zout, float* __restrict__ ll, float* __restrict__ qy,
                                                         int* __restrict__ y, float* __restrict__ terms) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    // log-likelihoods are ~1e3 in magnitude while the posterior only depends on their DIFFERENCES: accumulate in
    // double so that q(y|x) is not limited by the float32 ulp (~1e-4) of the sums (the reference's own float32
    // sums carry that noise; we stay closer to the exact value than it does).
    double llk[KMAX], klk[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { llk[k] = 0.0; klk[k] = 0.0; }
    for (int d = lane; d < Z; d += 64) {
        const float mu = pre[(long)b * 2 * Z + d];
        const float s = expf(pre[(long)b * 2 * Z + Z + d]);
        const float zz = mu + s * eps[(long)b * Z + d];
        sigma[(long)b * Z + d] = s;
        zout[(long)b * Z + d] = zz;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float mk = mu_lk[k * Z + d], lv = lv_lk[k * Z + d];
                const double dzm = (double)zz - (double)mk;
                llk[k] += -0.5 * (dzm * dzm * exp(-(double)lv) + (double)lv + (double)LOG_2PI);
                klk[k] += kl_dim_f64(mu, s, mk, lv);
            }
        }
    }
    const double logpk = log(1.0 / (double)K);
    double mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            llk[k] = fn_wave_sum_f64(llk[k]) + logpk;
            klk[k] = fn_wave_sum_f64(klk[k]) / (double)Z;
            mx = fmax(mx, llk[k]);
        }
    }
    double den = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) den += exp(llk[k] - mx);
    const double lse = mx + log(den);
    if (lane == 0) {
        float qk[KMAX];
        float t_lat = 0.f, t_ent = 0.f, qmax = -1.f, qden = 0.f;
        int am = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float lq = (float)(llk[k] - lse);
                const float q = (float)exp(llk[k] - lse);
                qk[k] = q;
                ll[(long)b * K + k] = (float)llk[k];
                qy[(long)b * K + k] = q;
                t_lat += q * (float)klk[k];
                t_ent += q * lq;
                if (q > qmax) { qmax = q; am = k; }
            }
        }
        y[b] = am;
        float t_sup = 0.f, t_clf = 0.f;
        if (labels) {
            const int lb = labels[b];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    qden += expf(qk[k] - qmax);
                    if (k == lb) t_sup = (float)klk[k];
                }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K && k == lb) t_clf = -(qk[k] - qmax - logf(qden));
        }
        float* tr = terms + (long)b * 4;
        tr[0] = t_lat;
        tr[1] = t_ent / (float)K;
        tr[2] = t_sup;
        tr[3] = t_clf;
    }
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ eps,
                                                         const float* __restrict__ mu_lk, const float* __restrict__ lv_lk, int B, int Z,
                                                         int K, const int* __restrict__ labels, const float* __restrict__ zin,
                                                         const float* __restrict__ qy, const float* __restrict__ g_z,
                                                         const float* __restrict__ g_mu, const float* __restrict__ g_sigma,
                                                         const float* __restrict__ g_ll, const float* __restrict__ g_qy,
                                                         const float* __restrict__ w3, float* __restrict__ dpre,
                                                         float* __restrict__ dmu_lk_rows) {
    const float w_lat = w3 ? w3[0] : 0.f, w_cls = w3 ? w3[1] : 0.f, w_clf = w3 ? w3[2] : 0.f;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lb = labels ? labels[b] : -1;
    // pass 1: KLmean_k for dL/dq.  The softmax backward q_k (dq_k - sum_j q_j dq_j) only sees DIFFERENCES of the dq_k while
    // each KLmean_k is ~1e3: evaluate and accumulate them in double so that the float32 ulp of those sums (~1e-4) does
    // not leak into the gradient.
    double klk[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) klk[k] = 0.0;
    if (w_lat != 0.f && !labels) {
        for (int d = lane; d < Z; d += 64) {
            const float mu = pre[(long)b * 2 * Z + d];
            const float s = expf(pre[(long)b * 2 * Z + Z + d]);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) klk[k] += kl_dim_f64(mu, s, mu_lk[k * Z + d], lv_lk[k * Z + d]);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) klk[k] = fn_wave_sum_f64(klk[k]) / (double)Z;
    }
    // dL/dq_k then softmax backward -> dll_k (all lanes compute the same K-sized scalars, in double)
    float q[KMAX], dll[KMAX];
    double dq[KMAX];
    float qmax = -1.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        q[k] = 0.f; dq[k] = 0.0; dll[k] = 0.f;
        if (k < K) { q[k] = qy[(long)b * K + k]; qmax = fmaxf(qmax, q[k]); }
    }
    float qden = 0.f;
    if (labels && w_clf != 0.f)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) qden += expf(q[k] - qmax);
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            double g = g_qy ? (double)g_qy[(long)b * K + k] : 0.0;
            if (!labels) {
                g += (double)w_lat * klk[k];
                g += (double)w_cls * (log(fmax((double)q[k], 1e-300)) + 1.0) / (double)K;   // q==0 contributes 0 (as q*log_softmax does)
            } else if (w_clf != 0.f) {
                g += (double)(w_clf * (expf(q[k] - qmax) / qden - (k == lb ? 1.0f : 0.0f)));
            }
            dq[k] = g;
            dot += (double)q[k] * g;
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) dll[k] = (float)((double)q[k] * (dq[k] - dot)) + (g_ll ? g_ll[(long)b * K + k] : 0.f);
    // pass 2: per-dimension gradients
    for (int d = lane; d < Z; d += 64) {
        const float mu = pre[(long)b * 2 * Z + d];
        const float s = expf(pre[(long)b * 2 * Z + Z + d]);
        const float e = eps[(long)b * Z + d];
        const float zz = zin[(long)b * Z + d];
        float dz = g_z ? g_z[(long)b * Z + d] : 0.f;
        float dmu = g_mu ? g_mu[(long)b * Z + d] : 0.f;
        float dsg = g_sigma ? g_sigma[(long)b * Z + d] : 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                const float mk = mu_lk[k * Z + d], lv = lv_lk[k * Z + d];
                const float iv = expf(-lv);                       // 1/exp(logvar)  (likelihood: variance)
                const float t = (zz - mk) * iv;
                dz += dll[k] * (-t);
                float dmk = dll[k] * t;
                const float wk = labels ? (k == lb ? w_lat : 0.f) : w_lat * q[k];   // weight of KLmean_k
                if (wk != 0.f) {
                    const float sp = expf(lv);                    // KL: exp(logvar) as STD
                    const float isp2 = 1.0f / (sp * sp);
                    dmu += wk / (float)Z * (mu - mk) * isp2;
                    dsg += wk / (float)Z * (s * isp2 - 1.0f / s);
                    dmk += wk / (float)Z * (-(mu - mk)) * isp2;
                }
                if (dmu_lk_rows) dmu_lk_rows[((long)b * K + k) * Z + d] = dmk;
            }
        }
        dmu += dz;
        dsg += dz * e;
        dpre[(long)b * 2 * Z + d] = dmu;
        dpre[(long)b * 2 * Z + Z + d] = dsg * s;               // sigma = exp(v)
    }
}

// ------------------------------------------------------------------ pairwise regulariser -----
__global__ __launch_bounds__(256) void pairwise_reg_kernel(const float* __restrict__ z0, const double* __restrict__ attr, int n_all,
                                                           int row0, int nrows, float* __restrict__ loss_rows, float grad_scale,
                                                           float* __restrict__ dz0) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const float zi = z0[row0 + i];
    const double ai = attr[row0 + i];
    float l = 0.f, g = 0.f;
    for (int j = lane; j < n_all; j += 64) {
        const float th = tanhf(zi - z0[j]);
        const double da = ai - attr[j];          // float64 difference, only its sign is used (trainer_gmm.py:208-210)
        const float sg = (da > 0.0) ? 1.0f : ((da < 0.0) ? -1.0f : 0.0f);
        const float df = th - sg;
        l += df * df;
        g += df * (1.0f - th * th);
    }
    l = fn_wave_sum(l);
    g = fn_wave_sum(g);
    if (lane == 0) {
        loss_rows[i] = l;
        if (dz0) dz0[i] = grad_scale * 4.0f * g;
    }
}

// ------------------------------------------------------------------ adversarial heads of the Fader-Networks sibling -----
// model_v2.py:572-575 + trainer_fader.py:105-110: per attribute a in {rhythm, note}
//     o_a = dropout(relu(w_a . reverse(z) + b_a)),   l_adv_a = lam * mean_b (o_a - dens_a)^2
// One wavefront per row; forward and the gradient of lam * sum_a mean_b(...) in the same pass:
//     da[b][a] = lam * 2 (o - dens) / Bg * mask * [pre > 0]   (gradient wrt the pre-activation; its column sums / da^T z are
//                                                              the discriminator's bias / weight gradients)
//     g_z[b][:] -= sum_a da[b][a] w_a                          (ReverseLayerF: the encoder receives the NEGATED gradient)
__global__ __launch_bounds__(256) void adv_head_kernel(const float* __restrict__ z, int ldz, int Z, int B, const float* __restrict__ w_r,
                                                       const float* __restrict__ w_n, const float* __restrict__ b_r,
                                                       const float* __restrict__ b_n, const float* __restrict__ mask,
                                                       const float* __restrict__ dens, const float* __restrict__ lam_dev, float inv_bg,
                                                       float* __restrict__ o, float* __restrict__ loss_rows, float* __restrict__ da,
                                                       float* __restrict__ g_z, int ldg) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float sr = 0.f, sn = 0.f;
    for (int d = lane; d < Z; d += 64) {
        const float zz = z[(long)b * ldz + d];
        sr += zz * w_r[d];
        sn += zz * w_n[d];
    }
    sr = fn_wave_sum(sr) + b_r[0];
    sn = fn_wave_sum(sn) + b_n[0];
    const float lam = lam_dev ? lam_dev[0] : 0.f;
    const float pre[2] = {sr, sn};
    float dav[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float m = mask[b * 2 + a];
        const float ov = fmaxf(pre[a], 0.f) * m;
        const float df = ov - dens[b * 2 + a];
        dav[a] = lam * 2.0f * df * inv_bg * m * (pre[a] > 0.f ? 1.0f : 0.f);
        if (lane == 0) {
            o[b * 2 + a] = ov;
            loss_rows[b * 2 + a] = df * df;
            if (da) da[b * 2 + a] = dav[a];
        }
    }
    if (g_z)
        for (int d = lane; d < Z; d += 64) g_z[(long)b * ldg + d] -= dav[0] * w_r[d] + dav[1] * w_n[d];
}

__global__ void onehot_to_index_kernel(const float* __restrict__ oh, long rows, int V, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = oh + row * V;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int e = lane; e < V; e += 64) {
        const float v = x[e];
        if (v > mx) { mx = v; am = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
    }
    if (lane == 0) idx[row] = am;
}

}  // namespace

extern "C" {

int fn_masked_prob(const float* logits, int64_t rows, int E, int ld, int lo0, int hi0, int lo1, int hi1, float* sums, const float* w,
                   float* dlogits, void* stream) {
    if (!logits || (!sums && !dlogits) || (dlogits && !w)) return FN_E_NULL;
    if (rows <= 0 || E <= 0 || ld < E || lo0 < 0 || hi0 > E || lo1 < 0 || hi1 > E) return FN_E_SHAPE;
    hipLaunchKernelGGL(masked_prob_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, (long)rows, E, ld,
                       lo0, hi0, lo1, hi1, sums, w, dlogits);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_adv_head(const float* z, int ldz, int Z, int B, const float* w_r, const float* w_n, const float* b_r, const float* b_n,
                const float* mask, const float* dens, const float* lam_dev, float inv_global_batch, float* o, float* loss_rows, float* da,
                float* g_z, int ldg, void* stream) {
    if (!z || !w_r || !w_n || !b_r || !b_n || !mask || !dens || !o || !loss_rows) return FN_E_NULL;
    if (Z <= 0 || B <= 0 || ldz < Z || (g_z && ldg < Z)) return FN_E_SHAPE;
    hipLaunchKernelGGL(adv_head_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, ldz, Z, B, w_r, w_n, b_r, b_n, mask, dens,
                       lam_dev, inv_global_batch, o, loss_rows, da, g_z, ldg);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_vocab_logsoftmax(const float* logits, int B, int T, int E, int ld, float* logp_bt, const int32_t* target, float* nll_rows,
                        float grad_scale, float* dlogits, void* stream) {
    if (!logits) return FN_E_NULL;
    if (B <= 0 || T <= 0 || E <= 0 || ld < E) return FN_E_SHAPE;
    if ((nll_rows || dlogits) && !target) return FN_E_NULL;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(vocab_logsoftmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, B, T, E, ld,
                       logp_bt, target, nll_rows, grad_scale, dlogits);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_vocab_logsoftmax_bwd(const float* logp_bt, const float* gout_bt, int B, int T, int E, int ld, float* dlogits, void* stream) {
    if (!logp_bt || !gout_bt || !dlogits) return FN_E_NULL;
    if (B <= 0 || T <= 0 || E <= 0 || ld < E) return FN_E_SHAPE;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(vocab_logsoftmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logp_bt, gout_bt,
                       B, T, E, ld, dlogits);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_vocab_argmax(const float* logits, int B, int E, int ld, float* logp_out, int64_t logp_ld, int32_t* tok_out, int tok_ld,
                    void* stream) {
    if (!logits || !tok_out) return FN_E_NULL;
    if (B <= 0 || E <= 0 || ld < E) return FN_E_SHAPE;
    hipLaunchKernelGGL(vocab_argmax_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, B, E, ld, logp_out,
                       (long)logp_ld, tok_out, tok_ld);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_time_logsoftmax(const float* logits, int B, int Tr, int Cc, float* logp_bt, const int32_t* target, float* nll_bc,
                       float grad_scale, float* dlogits, void* stream) {
    if (!logits) return FN_E_NULL;
    if (B <= 0 || Tr <= 0 || Cc <= 0) return FN_E_SHAPE;
    if ((nll_bc || dlogits) && !target) return FN_E_NULL;
    if (Tr <= 64)
        hipLaunchKernelGGL(time_logsoftmax_wave_kernel, dim3((B * Cc + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, B, Tr, Cc,
                           logp_bt, target, nll_bc, grad_scale, dlogits);
    else
        hipLaunchKernelGGL(time_logsoftmax_kernel, dim3((B * Cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, B, Tr, Cc,
                           logp_bt, target, nll_bc, grad_scale, dlogits);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_time_logsoftmax_bwd(const float* logp_bt, const float* gout_bt, int B, int Tr, int Cc, float* dlogits, void* stream) {
    if (!logp_bt || !gout_bt || !dlogits) return FN_E_NULL;
    if (B <= 0 || Tr <= 0 || Cc <= 0) return FN_E_SHAPE;
    hipLaunchKernelGGL(time_logsoftmax_bwd_kernel, dim3((B * Cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, logp_bt, gout_bt, B,
                       Tr, Cc, dlogits);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_latent_fwd(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                  const int32_t* labels, float* sigma, float* z, float* ll, float* qy, int32_t* y, float* terms, void* stream) {
    if (!pre || !eps || !mu_lk || !lv_lk || !sigma || !z || !ll || !qy || !y || !terms) return FN_E_NULL;
    if (B <= 0 || Z <= 0 || K <= 0 || K > KMAX) return FN_E_SHAPE;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, pre, eps, mu_lk, lv_lk, B, Z, K, labels,
                       sigma, z, ll, qy, y, terms);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_latent_bwd(const float* pre, const float* eps, const float* mu_lk, const float* lv_lk, int B, int Z, int K,
                  const int32_t* labels, const float* z, const float* qy, const float* g_z, const float* g_mu, const float* g_sigma,
                  const float* g_ll, const float* g_qy, const float* w3, float* dpre, float* dmu_lk_rows, void* stream) {
    if (!pre || !eps || !mu_lk || !lv_lk || !z || !qy || !dpre) return FN_E_NULL;
    if (B <= 0 || Z <= 0 || K <= 0 || K > KMAX) return FN_E_SHAPE;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, pre, eps, mu_lk, lv_lk, B, Z, K, labels, z,
                       qy, g_z, g_mu, g_sigma, g_ll, g_qy, w3, dpre, dmu_lk_rows);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_pairwise_reg(const float* z0_all, const double* attr_all, int n_all, int row0, int nrows, float* loss_rows, float grad_scale,
                    float* dz0, void* stream) {
    if (!z0_all || !attr_all || !loss_rows) return FN_E_NULL;
    if (n_all <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > n_all) return FN_E_SHAPE;
    hipLaunchKernelGGL(pairwise_reg_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, z0_all, attr_all, n_all, row0,
                       nrows, loss_rows, grad_scale, dz0);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_onehot_to_index(const float* oh, int64_t rows, int V, int32_t* idx, void* stream) {
    if (!oh || !idx) return FN_E_NULL;
    if (rows <= 0 || V <= 0) return FN_E_SHAPE;
    hipLaunchKernelGGL(onehot_to_index_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, oh, (long)rows, V, idx);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
