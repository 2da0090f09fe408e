// Frozen text front end of the training step: RNN_ENCODER.forward (reference model/encoder.py:118-153) in eval mode --
// nn.Embedding lookup, then a one-layer bidirectional nn.LSTM over length-packed captions.
//
//   xmc_embedding_gather   rows of the f32 table -> [n_tokens, dim]; pure copy, 16-byte units, HBM-bound
//   (input projection x W_ih^T + b_ih + b_hh for every token and both directions: one call of the 1x1 path of the
//    implicit-GEMM kernel, [B*T, 300] x [300, 2*4H])
//   xmc_lstm_bidir         the recurrence.  Samples are independent, so there is no cross-workgroup dependency:
//                          one workgroup = (direction, NB samples), 4H = 512 threads, thread j keeps row j of W_hh
//                          (H = 128 floats) in registers for the whole sequence, h lives in LDS and is read as
//                          broadcasts, the T outputs of a sample are staged in LDS and written once as [2H, T] rows.
//                          Latency-bound (T dependent steps of a 512x128 mat-vec), ~1 GFLOP in total.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, const f32x4* __restrict__ table,
                                                          f32x4* __restrict__ out, int64_t n_tokens, int dim4, int64_t vocab) {
    // one wave per token row
    const int lane = threadIdx.x & 63;
    for (int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tok < n_tokens; tok += (int64_t)gridDim.x * 4) {
        const int64_t id = ids[tok];
        const bool ok = id >= 0 && id < vocab;          // host side validates; never read outside the table
        const f32x4* src = table + (ok ? id : 0) * dim4;
        for (int c = lane; c < dim4; c += 64) out[tok * dim4 + c] = ok ? src[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int LH = 128;            // hidden units per direction (TEXT.EMBEDDING_DIM 256 / 2 directions)
constexpr int LG = 4 * LH;         // gate rows i, f, g, o  (torch.nn.LSTM order)

template <int NB>
__global__ __launch_bounds__(LG) void lstm_bidir_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                        const int32_t* __restrict__ lens, float* __restrict__ words,
                                                        float* __restrict__ sent, int B, int T) {
    extern __shared__ float hist[];                     // [NB][T][LH] outputs of this direction
    __shared__ __attribute__((aligned(16))) float h_s[NB][LH];
    __shared__ float g_s[NB][LG];
    const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;

    float w[LH];
    {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w_hh + ((size_t)dir * LG + j) * LH);
#pragma unroll
        for (int k = 0; k < LH / 4; ++k) {
            const f32x4 v = wr[k];
            w[4 * k] = v[0], w[4 * k + 1] = v[1], w[4 * k + 2] = v[2], w[4 * k + 3] = v[3];
        }
    }
    int len[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = b0 + nb;
        len[nb] = b < B ? min(max(lens[b], 0), T) : 0;
    }
    for (int i = j; i < NB * LH; i += LG) (&h_s[0][0])[i] = 0.f;
    for (int i = j; i < NB * T * LH; i += LG) hist[i] = 0.f;
    // cell state: thread j < NB*LH owns unit (j % LH) of sample (j / LH)
    float c_state = 0.f;
    __syncthreads();

    // input projection of (sample nb, step): fetched one step ahead so its HBM/L2 latency hides behind the mat-vec
    auto xin = [&](int nb, int step) -> float {
        const int t = dir == 0 ? step : len[nb] - 1 - step;
        return step < len[nb] ? xproj[(((size_t)(b0 + nb) * T + t) * 2 + dir) * LG + j] : 0.f;
    };
    float xnext[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xnext[nb] = xin(nb, 0);

    for (int step = 0; step < T; ++step) {
        float acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = xnext[nb];
            xnext[nb] = xin(nb, step + 1);
        }
#pragma unroll
        for (int k = 0; k < LH; k += 4) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(&h_s[nb][k]);
                acc[nb] += w[k] * hv[0] + w[k + 1] * hv[1] + w[k + 2] * hv[2] + w[k + 3] * hv[3];
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) g_s[nb][j] = acc[nb];
        __syncthreads();
        if (j < NB * LH) {
            const int nb = j / LH, u = j % LH;
            if (step < len[nb]) {
                const int t = dir == 0 ? step : len[nb] - 1 - step;
                const float ig = sigmoid_f(g_s[nb][u]), fg = sigmoid_f(g_s[nb][LH + u]);
                const float gg = tanhf(g_s[nb][2 * LH + u]), og = sigmoid_f(g_s[nb][3 * LH + u]);
                c_state = fg * c_state + ig * gg;
                const float h = og * tanhf(c_state);
                h_s[nb][u] = h;
                hist[((size_t)nb * T + t) * LH + u] = h;
            }
        }
        __syncthreads();
    }
    // sentence embedding: the last hidden state of each direction (h_fwd(len-1), h_rev(0)); words: [B, 2H, T], zero at t >= len
    if (j < NB * LH) {
        const int nb = j / LH, u = j % LH, b = b0 + nb;
        if (b < B) {
            sent[(size_t)b * 2 * LH + dir * LH + u] = h_s[nb][u];
            float* row = words + ((size_t)b * 2 * LH + dir * LH + u) * T;
            for (int t = 0; t < T; ++t) row[t] = hist[((size_t)nb * T + t) * LH + u];
        }
    }
}

// nn.GRU (encoder.py:99-102: TEXT.RNN_TYPE 'GRU'), same launch geometry and data layout as the LSTM kernel with three gate rows
// (r, z, n in torch.nn.GRU's order):   r = s(x_r + W_hr h)   z = s(x_z + W_hz h)   n = tanh(x_n + r * (W_hn h + b_hn))
//                                      h' = (1 - z) * n + z * h
// xproj holds W_i* x + b_i* (+ b_h* for r and z, which commute with the sum); b_hn sits INSIDE the product with r and is a
// separate argument.
constexpr int GG = 3 * LH;

template <int NB>
__global__ __launch_bounds__(GG) void gru_bidir_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_hn, const int32_t* __restrict__ lens,
                                                       float* __restrict__ words, float* __restrict__ sent, int B, int T) {
    extern __shared__ float hist[];                     // [NB][T][LH] outputs of this direction
    __shared__ __attribute__((aligned(16))) float h_s[NB][LH];
    __shared__ float gh_s[NB][GG];                      // r, z: x + W h (pre-activation); n: W_hn h + b_hn
    __shared__ float gx_s[NB][LH];                      // x part of the candidate gate
    const int j = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * NB;

    float w[LH];
    {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w_hh + ((size_t)dir * GG + j) * LH);
#pragma unroll
        for (int k = 0; k < LH / 4; ++k) {
            const f32x4 v = wr[k];
            w[4 * k] = v[0], w[4 * k + 1] = v[1], w[4 * k + 2] = v[2], w[4 * k + 3] = v[3];
        }
    }
    const float bn = j >= 2 * LH ? b_hn[dir * LH + (j - 2 * LH)] : 0.f;
    int len[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = b0 + nb;
        len[nb] = b < B ? min(max(lens[b], 0), T) : 0;
    }
    for (int i = j; i < NB * LH; i += GG) (&h_s[0][0])[i] = 0.f;
    for (int i = j; i < NB * T * LH; i += GG) hist[i] = 0.f;
    __syncthreads();

    auto xin = [&](int nb, int step) -> float {
        const int t = dir == 0 ? step : len[nb] - 1 - step;
        return step < len[nb] ? xproj[(((size_t)(b0 + nb) * T + t) * 2 + dir) * GG + j] : 0.f;
    };
    float xnext[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xnext[nb] = xin(nb, 0);

    for (int step = 0; step < T; ++step) {
        float acc[NB], xcur[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            xcur[nb] = xnext[nb];
            acc[nb] = 0.f;
            xnext[nb] = xin(nb, step + 1);
        }
#pragma unroll
        for (int k = 0; k < LH; k += 4) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(&h_s[nb][k]);
                acc[nb] += w[k] * hv[0] + w[k + 1] * hv[1] + w[k + 2] * hv[2] + w[k + 3] * hv[3];
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (j < 2 * LH) gh_s[nb][j] = acc[nb] + xcur[nb];
            else { gh_s[nb][j] = acc[nb] + bn; gx_s[nb][j - 2 * LH] = xcur[nb]; }
        }
        __syncthreads();
        if (j < NB * LH) {
            const int nb = j / LH, u = j % LH;
            if (step < len[nb]) {
                const int t = dir == 0 ? step : len[nb] - 1 - step;
                const float rg = sigmoid_f(gh_s[nb][u]), zg = sigmoid_f(gh_s[nb][LH + u]);
                const float ng = tanhf(gx_s[nb][u] + rg * gh_s[nb][2 * LH + u]);
                const float h = (1.f - zg) * ng + zg * h_s[nb][u];
                h_s[nb][u] = h;
                hist[((size_t)nb * T + t) * LH + u] = h;
            }
        }
        __syncthreads();
    }
    if (j < NB * LH) {
        const int nb = j / LH, u = j % LH, b = b0 + nb;
        if (b < B) {
            sent[(size_t)b * 2 * LH + dir * LH + u] = h_s[nb][u];
            float* row = words + ((size_t)b * 2 * LH + dir * LH + u) * T;
            for (int t = 0; t < T; ++t) row[t] = hist[((size_t)nb * T + t) * LH + u];
        }
    }
}

}  // namespace

extern "C" int xmc_embedding_gather(const int64_t* ids, const float* table, float* out, int64_t n_tokens, int dim, int64_t vocab,
                                    void* stream) {
    if (!ids || !table || !out || n_tokens < 0 || dim <= 0 || vocab <= 0) return XMC_EINVAL;
    if (dim % 4) return XMC_EALIGN;
    if (n_tokens == 0) return 0;
    int64_t nb = (n_tokens + 3) / 4;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, ids,
                       reinterpret_cast<const f32x4*>(table), reinterpret_cast<f32x4*>(out), n_tokens, dim / 4, vocab);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_lstm_bidir(const float* xproj, const float* w_hh, const int32_t* lens, float* words, float* sent, int B, int T,
                              int H, void* stream) {
    if (!xproj || !w_hh || !lens || !words || !sent || B <= 0 || T <= 0) return XMC_EINVAL;
    if (H != LH) return XMC_EINVAL;                     // TEXT.EMBEDDING_DIM 256 (every RNN preset); other widths are not built
    // 2 samples per workgroup fills the 256 CUs at the per-GPU batch of 256; small batches use 1 to shorten the critical path
    const int NB = B >= 256 ? 2 : 1;
    const size_t lds = (size_t)NB * T * LH * sizeof(float);
    if (lds > 96 * 1024) return XMC_EINVAL;
    dim3 grid((B + NB - 1) / NB, 2), blk(LG);
    if (NB == 2) {
        XMC_ALLOW_BIG_LDS(lstm_bidir_kernel<2>);
        hipLaunchKernelGGL(lstm_bidir_kernel<2>, grid, blk, lds, (hipStream_t)stream, xproj, w_hh, lens, words, sent, B, T);
    } else {
        XMC_ALLOW_BIG_LDS(lstm_bidir_kernel<1>);
        hipLaunchKernelGGL(lstm_bidir_kernel<1>, grid, blk, lds, (hipStream_t)stream, xproj, w_hh, lens, words, sent, B, T);
    }
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_gru_bidir(const float* xproj, const float* w_hh, const float* b_hn, const int32_t* lens, float* words, float* sent,
                             int B, int T, int H, void* stream) {
    if (!xproj || !w_hh || !b_hn || !lens || !words || !sent || B <= 0 || T <= 0) return XMC_EINVAL;
    if (H != LH) return XMC_EINVAL;                     // TEXT.EMBEDDING_DIM 256, as for the LSTM
    const int NB = B >= 256 ? 2 : 1;
    const size_t lds = (size_t)NB * T * LH * sizeof(float);
    if (lds > 96 * 1024) return XMC_EINVAL;
    dim3 grid((B + NB - 1) / NB, 2), blk(GG);
    if (NB == 2) {
        XMC_ALLOW_BIG_LDS(gru_bidir_kernel<2>);
        hipLaunchKernelGGL(gru_bidir_kernel<2>, grid, blk, lds, (hipStream_t)stream, xproj, w_hh, b_hn, lens, words, sent, B, T);
    } else {
        XMC_ALLOW_BIG_LDS(gru_bidir_kernel<1>);
        hipLaunchKernelGGL(gru_bidir_kernel<1>, grid, blk, lds, (hipStream_t)stream, xproj, w_hh, b_hn, lens, words, sent, B, T);
    }
    XMC_LAUNCH_CHECK();
    return 0;
}
