// Per-concept algebra of the word-attention generators (model/concept_gan.py), f32, C = 16 concepts x P = 4 state
// channels -- one wave per sample with lane = (concept, state channel), or one workgroup for the whole batch where BatchNorm1d couples
// the samples.  What runs here was a chain of 10-20 framework launches per stage on [B,16,<=360] tensors:
//   xmc_gvec_*       grouped 1x1 convolution of a per-sample vector (the gamma / beta heads on cat(global condition, context),
//                    concept_gan.py:346-371,404-418; the samplers' query / value projections, 545-580)
//   xmc_reasoner_*   ConceptReasoner (632-654): adj = tanh(x We^T), x + adj x, BatchNorm1d over (batch, state) per concept, relu
//   xmc_word_ctx_*   OutConceptBlock.get_context_embs (374-394): states normalised over the CONCEPT axis, words over the state axis,
//                    cosine scores, masked_fill(-inf), softmax over the T words (wave shuffles), weighted word sum
//   xmc_word_keys_*  CondConceptSampler's keys (566-575): GroupNorm over (state, word) per concept, L2 normalisation per word
// A caption whose every word is padding gives NaN, as torch.softmax of an all -inf row does.
#include "common.h"

namespace {

constexpr int C_ = 16, P_ = 4, CP = 64, TMAX = 32;

// ------------------------------------------------------------------------------------------------------------------ grouped vector
// one wave per (b, g): lanes stride over the inputs, one shuffle reduction per output
__global__ __launch_bounds__(256) void gvec_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ xg, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B, int G, int O, int Is, int Ig) {
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= B * G) return;
    const int b = wv / G, g = wv - b * G, I = Is + Ig;
    for (int o = 0; o < O; ++o) {
        const float* __restrict__ w = W + ((size_t)g * O + o) * I;
        float s = 0.f;
        for (int i = lane; i < Is; i += 64) s += w[i] * xs[(size_t)b * Is + i];
        for (int i = lane; i < Ig; i += 64) s += w[Is + i] * xg[((size_t)b * G + g) * Ig + i];
        s = wave_sum(s);
        if (lane == 0) y[((size_t)b * G + g) * O + o] = s + (bias ? bias[g * O + o] : 0.f);
    }
}

// one thread per output element of dW | dbias | dxs | dxg (ranges of one index space)
__global__ __launch_bounds__(256) void gvec_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ xg, const float* __restrict__ W,
                                                       const float* __restrict__ dy, float* __restrict__ dxs, float* __restrict__ dxg,
                                                       float* __restrict__ dW, float* __restrict__ dbias, int B, int G, int O, int Is, int Ig) {
    const int I = Is + Ig;
    const int64_t nW = dW ? (int64_t)G * O * I : 0, nB = dbias ? (int64_t)G * O : 0, nS = dxs ? (int64_t)B * Is : 0, nG = dxg ? (int64_t)B * G * Ig : 0;
    int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id < nW) {
        const int i = (int)(id % I), go = (int)(id / I), g = go / O;
        float s = 0.f;
        if (i < Is) for (int b = 0; b < B; ++b) s += dy[(size_t)b * G * O + go] * xs[(size_t)b * Is + i];
        else for (int b = 0; b < B; ++b) s += dy[(size_t)b * G * O + go] * xg[((size_t)b * G + g) * Ig + i - Is];
        dW[id] = s;
        return;
    }
    id -= nW;
    if (id < nB) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(size_t)b * G * O + id];
        dbias[id] = s;
        return;
    }
    id -= nB;
    if (id < nS) {
        const int i = (int)(id % Is), b = (int)(id / Is);
        float s = 0.f;
        for (int go = 0; go < G * O; ++go) s += W[(size_t)go * I + i] * dy[(size_t)b * G * O + go];
        dxs[id] = s;
        return;
    }
    id -= nS;
    if (id < nG) {
        const int i = (int)(id % Ig), bg = (int)(id / Ig), g = bg % G;
        float s = 0.f;
        for (int o = 0; o < O; ++o) s += W[((size_t)g * O + o) * I + Is + i] * dy[(size_t)bg * O + o];
        dxg[id] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------ reasoner
// ONE workgroup walks the batch (BatchNorm1d's statistics couple the samples; the whole tensor is B x 64 floats): a wave per sample,
// lane = (concept c, state channel k); sums over k are quad DPP adds, sums over c four shuffles, a concept's row of the adjacency is
// 16 registers.  (A first version gave every THREAD a sample -- 3 x 64 registers of state and the We gradient through LDS atomics on
// 64 addresses: 2.2 ms per call.)
__device__ __forceinline__ float rs_k(float v) { v += xmc_xor1(v); v += xmc_xor2(v); return v; }
__device__ __forceinline__ float rs_c(float v) {
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}
// per-channel total over the workgroup of a per-lane partial (already summed over this wave's samples): [c] for every lane
__device__ __forceinline__ float block_channel_sum(float v, float* s_red /*[4][16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = rs_k(v);
    if ((lane & 3) == 0) s_red[wave * C_ + (lane >> 2)] = v;
    __syncthreads();
    const int c = lane >> 2;
    const float t = s_red[c] + s_red[C_ + c] + s_red[2 * C_ + c] + s_red[3 * C_ + c];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void reasoner_fwd_kernel(const float* __restrict__ x, const float* __restrict__ We, const float* __restrict__ bn_w,
                                                           const float* __restrict__ bn_b, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           int training, float momentum, float eps, float* __restrict__ y, float* __restrict__ pre_out,
                                                           float* __restrict__ stat, int B) {
    __shared__ float s_red[4 * C_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane & 3, c = lane >> 2;
    float we[C_];                                   // We[e][k] for this lane's k
#pragma unroll
    for (int e = 0; e < C_; ++e) we[e] = We[e * P_ + k];
    float sum = 0.f;
    for (int b = wave; b < B; b += 4) {
        const float xv = x[(size_t)b * CP + lane];
        float pre = xv;
#pragma unroll
        for (int e = 0; e < C_; ++e) {
            const float a = tanhf(rs_k(xv * we[e]));               // adj[c][e]
            pre += a * __shfl(xv, e * P_ + k, 64);
        }
        pre_out[(size_t)b * CP + lane] = pre;
        sum += pre;
    }
    float mean = 0.f, rstd = 1.f;
    if (bn_w && training) {
        const float n = (float)B * P_;
        mean = block_channel_sum(sum, s_red) / n;
        float sq = 0.f;
        for (int b = wave; b < B; b += 4) { const float dlt = pre_out[(size_t)b * CP + lane] - mean; sq += dlt * dlt; }
        const float var = block_channel_sum(sq, s_red) / n;
        rstd = rsqrtf(var + eps);
        if (wave == 0 && k == 0) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
        }
    } else if (bn_w) {
        mean = run_mean[c]; rstd = rsqrtf(run_var[c] + eps);
    }
    if (wave == 0 && k == 0 && stat) { stat[c] = mean; stat[C_ + c] = rstd; }
    if (!y) return;
    const float w = bn_w ? bn_w[c] : 1.f, bb = bn_w ? bn_b[c] : 0.f;
    for (int b = wave; b < B; b += 4)
        y[(size_t)b * CP + lane] = fmaxf((pre_out[(size_t)b * CP + lane] - mean) * rstd * w + bb, 0.f);
}

__global__ __launch_bounds__(256) void reasoner_bwd_kernel(const float* __restrict__ x, const float* __restrict__ We, const float* __restrict__ bn_w,
                                                           const float* __restrict__ bn_b, const float* __restrict__ pre, const float* __restrict__ stat,
                                                           int batch_stats, const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dWe,
                                                           float* __restrict__ dbn_w, float* __restrict__ dbn_b, int B) {
    __shared__ float s_red[4 * C_], s_dwe[4 * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane & 3, c = lane >> 2;
    float we[C_];
#pragma unroll
    for (int e = 0; e < C_; ++e) we[e] = We[e * P_ + k];
    const float wec = We[lane];                                     // We[c][k]
    const float mean = stat[c], rstd = stat[C_ + c], w = bn_w ? bn_w[c] : 1.f, bb = bn_w ? bn_b[c] : 0.f;
    // pass 1: sums of g = dy * relu' and of g * xhat per concept (BatchNorm's parameter gradients and its batch-statistics terms)
    float sg = 0.f, sgx = 0.f;
    for (int b = wave; b < B; b += 4) {
        const float xh = (pre[(size_t)b * CP + lane] - mean) * rstd;
        const float g = (xh * w + bb) > 0.f ? dy[(size_t)b * CP + lane] : 0.f;
        sg += g; sgx += g * xh;
    }
    const float tg = block_channel_sum(sg, s_red), tgx = block_channel_sum(sgx, s_red);
    if (wave == 0 && k == 0 && bn_w) { if (dbn_w) dbn_w[c] = tgx; if (dbn_b) dbn_b[c] = tg; }
    const float n = (float)B * P_;
    // pass 2: d pre, then through pre = x + tanh(x We^T) x
    float dwe = 0.f;                                                // d We[c][k], this wave's samples
    for (int b = wave; b < B; b += 4) {
        const float xv = x[(size_t)b * CP + lane];
        const float xh = (pre[(size_t)b * CP + lane] - mean) * rstd;
        const float g = (xh * w + bb) > 0.f ? dy[(size_t)b * CP + lane] : 0.f;
        float dp = g * w * rstd;
        if (batch_stats) dp -= w * rstd * (tg + xh * tgx) / n;
        float dxv = dp;
#pragma unroll
        for (int e = 0; e < C_; ++e) {
            const float xe = __shfl(xv, e * P_ + k, 64), dpe = __shfl(dp, e * P_ + k, 64);
            const float a_ce = tanhf(rs_k(xv * we[e]));                                 // adj[c][e]
            const float a_ec = tanhf(rs_k(xe * wec));                                   // adj[e][c]
            dxv += a_ec * dpe;                                                          // pre[e] = ... + adj[e][c] x[c]
            const float dz = rs_k(dp * xe) * (1.f - a_ce * a_ce);                       // d z[c][e],  z[c][e] = x[c] . We[e]
            dxv += dz * we[e];
            const float t = rs_c(dz * xv);                                              // sum_c dz[c][e] x[c][k]: d We[e][k]
            if (c == e) dwe += t;
        }
        if (dx) dx[(size_t)b * CP + lane] = dxv;
    }
    if (dWe) {
        s_dwe[wave * CP + lane] = dwe;
        __syncthreads();
        if (wave == 0) dWe[lane] = s_dwe[lane] + s_dwe[CP + lane] + s_dwe[2 * CP + lane] + s_dwe[3 * CP + lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------- word context
__device__ __forceinline__ float sum_k(float v) { v += xmc_xor1(v); v += xmc_xor2(v); return v; }                       // over the 4 state lanes of a concept
__device__ __forceinline__ float sum_c(float v) {                                                                      // over the 16 concepts of a state channel
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

__global__ __launch_bounds__(256) void word_ctx_fwd_kernel(const float* __restrict__ st, const float* __restrict__ w, const unsigned char* __restrict__ pad,
                                                           float* __restrict__ ctx, float* __restrict__ prob, int B, int T) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, k = lane & 3, c = lane >> 2;
    if (b >= B) return;
    const float s0 = st[(size_t)b * CP + lane];
    const float sn = s0 / fmaxf(sqrtf(sum_c(s0 * s0)), 1e-12f);
    float wd[TMAX], sc[TMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        wd[t] = 0.f; sc[t] = -INFINITY;
        if (t < T) {
            const float wv = w[((size_t)b * T + t) * P_ + k];
            wd[t] = wv / fmaxf(sqrtf(sum_k(wv * wv)), 1e-12f);
            const float s = sum_k(sn * wd[t]);
            sc[t] = pad[(size_t)b * T + t] ? -INFINITY : s;
            mx = fmaxf(mx, sc[t]);
        }
    }
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) if (t < T) { sc[t] = __expf(sc[t] - mx); se += sc[t]; }      // all words padded: -inf - -inf = NaN, like torch
    float o = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) {
            const float p = sc[t] / se;
            o += p * wd[t];
            if (k == 0) prob[((size_t)b * C_ + c) * T + t] = p;
        }
    ctx[(size_t)b * CP + lane] = o;
}

__global__ __launch_bounds__(256) void word_ctx_bwd_kernel(const float* __restrict__ st, const float* __restrict__ w, const float* __restrict__ prob,
                                                           const float* __restrict__ dctx, float* __restrict__ dst, float* __restrict__ dw, int B, int T) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, k = lane & 3, c = lane >> 2;
    if (b >= B) return;
    const float s0 = st[(size_t)b * CP + lane];
    const float nk = fmaxf(sqrtf(sum_c(s0 * s0)), 1e-12f), sn = s0 / nk;
    const float dc = dctx[(size_t)b * CP + lane];
    float wd[TMAX], p[TMAX], dp[TMAX], mt[TMAX];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        wd[t] = p[t] = dp[t] = 0.f; mt[t] = 1.f;
        if (t < T) {
            const float wv = w[((size_t)b * T + t) * P_ + k];
            mt[t] = fmaxf(sqrtf(sum_k(wv * wv)), 1e-12f);
            wd[t] = wv / mt[t];
            p[t] = prob[((size_t)b * C_ + c) * T + t];
            dp[t] = sum_k(dc * wd[t]);
            dot += p[t] * dp[t];
        }
    }
    float dsn = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) {
            const float ds = p[t] * (dp[t] - dot);                 // d sim[c][t]
            dsn += ds * wd[t];
            const float dwd = sum_c(p[t] * dc + ds * sn);          // d wd[t][k]
            const float proj = sum_k(wd[t] * dwd);
            if (c == 0 && dw) dw[((size_t)b * T + t) * P_ + k] = (dwd - wd[t] * proj) / mt[t];
        }
    if (dst) dst[(size_t)b * CP + lane] = (dsn - sn * sum_c(sn * dsn)) / nk;
}

// ------------------------------------------------------------------------------------------------------------------- word keys
__global__ __launch_bounds__(256) void word_keys_fwd_kernel(const float* __restrict__ kraw, const float* __restrict__ gnw, const float* __restrict__ gnb,
                                                            float eps, float* __restrict__ kh, float* __restrict__ stat, int B, int T) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, k = lane & 3, c = lane >> 2;
    if (b >= B) return;
    float v[TMAX], s = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { v[t] = t < T ? kraw[((size_t)b * T + t) * CP + lane] : 0.f; s += v[t]; }
    float mean = 0.f, rstd = 1.f;
    if (gnw) {
        const float n = 4.f * T;
        mean = sum_k(s) / n;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) if (t < T) q += (v[t] - mean) * (v[t] - mean);
        rstd = rsqrtf(sum_k(q) / n + eps);
        if (k == 0 && stat) { stat[((size_t)b * C_ + c) * 2] = mean; stat[((size_t)b * C_ + c) * 2 + 1] = rstd; }
    }
    const float gw = gnw ? gnw[lane] : 1.f, gb = gnw ? gnb[lane] : 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) {
            const float yv = (v[t] - mean) * rstd * gw + gb;
            kh[(((size_t)b * C_ + c) * T + t) * P_ + k] = yv / fmaxf(sqrtf(sum_k(yv * yv)), 1e-12f);
        }
}

__global__ __launch_bounds__(256) void word_keys_bwd_kernel(const float* __restrict__ kraw, const float* __restrict__ gnw, const float* __restrict__ gnb,
                                                            const float* __restrict__ stat, const float* __restrict__ dkh, float* __restrict__ dkraw,
                                                            float* __restrict__ dgnw, float* __restrict__ dgnb, int B, int T) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, k = lane & 3, c = lane >> 2;
    if (b >= B) return;
    const float mean = gnw ? stat[((size_t)b * C_ + c) * 2] : 0.f, rstd = gnw ? stat[((size_t)b * C_ + c) * 2 + 1] : 1.f;
    const float gw = gnw ? gnw[lane] : 1.f, gb = gnw ? gnb[lane] : 0.f;
    float xh[TMAX], dyv[TMAX], s1 = 0.f, s2 = 0.f, sw = 0.f, sb = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        xh[t] = dyv[t] = 0.f;
        if (t < T) {
            xh[t] = (kraw[((size_t)b * T + t) * CP + lane] - mean) * rstd;
            const float yv = xh[t] * gw + gb;
            const float nrm = fmaxf(sqrtf(sum_k(yv * yv)), 1e-12f), h = yv / nrm;
            const float d = dkh[(((size_t)b * C_ + c) * T + t) * P_ + k];
            dyv[t] = (d - h * sum_k(h * d)) / nrm;
            sw += dyv[t] * xh[t]; sb += dyv[t];
            s1 += dyv[t] * gw; s2 += dyv[t] * gw * xh[t];
        }
    }
    if (gnw) {
        if (dgnw) atomicAdd(&dgnw[lane], sw);
        if (dgnb) atomicAdd(&dgnb[lane], sb);
        const float n = 4.f * T, m1 = sum_k(s1) / n, m2 = sum_k(s2) / n;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) if (t < T) dkraw[((size_t)b * T + t) * CP + lane] = rstd * (dyv[t] * gw - m1 - xh[t] * m2);
    } else {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) if (t < T) dkraw[((size_t)b * T + t) * CP + lane] = dyv[t];
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int xmc_gvec_fwd(const float* xs, const float* xg, const float* W, const float* bias, float* y, int B, int G, int O, int Is, int Ig, void* s) {
    if (!W || !y || (Is > 0 && !xs) || (Ig > 0 && !xg) || B < 1 || G < 1 || O < 1 || Is < 0 || Ig < 0 || Is + Ig < 1) return XMC_EINVAL;
    hipLaunchKernelGGL(gvec_fwd_kernel, dim3((B * G + 3) / 4), dim3(256), 0, ST(s), xs, xg, W, bias, y, B, G, O, Is, Ig);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_gvec_bwd(const float* xs, const float* xg, const float* W, const float* dy, float* dxs, float* dxg, float* dW, float* dbias,
                            int B, int G, int O, int Is, int Ig, void* s) {
    if (!W || !dy || (Is > 0 && !xs) || (Ig > 0 && !xg) || B < 1 || G < 1 || O < 1 || Is < 0 || Ig < 0 || Is + Ig < 1) return XMC_EINVAL;
    if ((dxs && Is == 0) || (dxg && Ig == 0)) return XMC_EINVAL;
    const int64_t total = (dW ? (int64_t)G * O * (Is + Ig) : 0) + (dbias ? (int64_t)G * O : 0) + (dxs ? (int64_t)B * Is : 0) + (dxg ? (int64_t)B * G * Ig : 0);
    if (total == 0) return 0;
    hipLaunchKernelGGL(gvec_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(s), xs, xg, W, dy, dxs, dxg, dW, dbias, B, G, O, Is, Ig);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_reasoner_fwd(const float* x, const float* We, const float* bn_w, const float* bn_b, float* run_mean, float* run_var, int training,
                                float momentum, float eps, float* y, float* pre, float* stat, int B, void* s) {
    if (!x || !We || !pre || B < 1 || (bn_w && (!bn_b || !run_mean || !run_var))) return XMC_EINVAL;
    hipLaunchKernelGGL(reasoner_fwd_kernel, dim3(1), dim3(256), 0, ST(s), x, We, bn_w, bn_b, run_mean, run_var, training, momentum, eps, y, pre, stat, B);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_reasoner_bwd(const float* x, const float* We, const float* bn_w, const float* bn_b, const float* pre, const float* stat, int batch_stats,
                                const float* dy, float* dx, float* dWe, float* dbn_w, float* dbn_b, int B, void* s) {
    if (!x || !We || !pre || !stat || !dy || B < 1 || (bn_w && !bn_b)) return XMC_EINVAL;
    hipLaunchKernelGGL(reasoner_bwd_kernel, dim3(1), dim3(256), 0, ST(s), x, We, bn_w, bn_b, pre, stat, batch_stats, dy, dx, dWe, dbn_w, dbn_b, B);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_word_ctx_fwd(const float* st, const float* w, const unsigned char* pad, float* ctx, float* prob, int B, int T, void* s) {
    if (!st || !w || !pad || !ctx || !prob || B < 1) return XMC_EINVAL;
    if (T < 1 || T > TMAX) return XMC_ESHAPE;
    hipLaunchKernelGGL(word_ctx_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(s), st, w, pad, ctx, prob, B, T);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_word_ctx_bwd(const float* st, const float* w, const float* prob, const float* dctx, float* dst, float* dw, int B, int T, void* s) {
    if (!st || !w || !prob || !dctx || B < 1) return XMC_EINVAL;
    if (T < 1 || T > TMAX) return XMC_ESHAPE;
    hipLaunchKernelGGL(word_ctx_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(s), st, w, prob, dctx, dst, dw, B, T);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_word_keys_fwd(const float* kraw, const float* gnw, const float* gnb, float eps, float* kh, float* stat, int B, int T, void* s) {
    if (!kraw || !kh || B < 1 || (gnw && (!gnb || !stat))) return XMC_EINVAL;
    if (T < 1 || T > TMAX) return XMC_ESHAPE;
    hipLaunchKernelGGL(word_keys_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(s), kraw, gnw, gnb, eps, kh, stat, B, T);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_word_keys_bwd(const float* kraw, const float* gnw, const float* gnb, const float* stat, const float* dkh, float* dkraw, float* dgnw,
                                 float* dgnb, int B, int T, void* s) {
    if (!kraw || !dkh || !dkraw || B < 1 || (gnw && (!gnb || !stat))) return XMC_EINVAL;
    if (T < 1 || T > TMAX) return XMC_ESHAPE;
    hipLaunchKernelGGL(word_keys_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(s), kraw, gnw, gnb, stat, dkh, dkraw, dgnw, dgnb, B, T);
    XMC_LAUNCH_CHECK();
    return 0;
}
