// Word-region attention pooling of the word-attention generator concept_gan.InNetG (reference model/concept_gan.py
// CondConceptSampler.get_context_embs 532-555): every REGION of the map queries the caption's words.
//   qmap [N][HW][16*4]   the grouped-1x1 query projection of the map (after its GroupNorm), 16-bit or f32
//   kh   f32 [N][16][T][4]  the per-concept word keys, already L2-normalised over the 4 state channels (a [B,64,T] tensor: host side)
//   pad  u8 [N][T]       1 = padding word (masked_fill -inf, 541-543)
//   ctx  f32 [N][16][4]  mean over the regions of  sum_t softmax_t(<q^, k^_t>) k^_t    with q^ = q / max(|q|, 1e-12)
// The [N,16,HW,T] score / attention tensors of the reference (75 MB per sample at 256 px) never exist: a lane owns one (pixel, concept),
// reads its 4 query channels (a wave covers 4 whole 128-byte pixels), and runs the T <= 32 words out of LDS with a running softmax.
// Backward recomputes the attention from q and k (nothing but ctx is kept): with g = dctx / HW the same for every region,
//   sum_u a_u <g, k_u> = <g, ctx_pixel>,  ds_t = a_t (<g, k_t> - <g, ctx_pixel>),  dq^ = sum_t ds_t k_t,  dk^_t = g sum_p a_tp + sum_p ds_tp q^_p
// so one pass writes the query gradient map and leaves 5 T per-lane accumulators, merged over the 16 pixel slots by shuffles + LDS and
// added to dkh with one atomic per (image, run, concept, word, channel).  HBM-bound: 8 bytes per lane read (+ 8 written backward).
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int WP_CON = 16, WP_P = 4;          // concepts, state channels (concept_gan.py:132: state_dim = 4)
constexpr int WP_SLOTS = NT / WP_CON;         // pixels in flight per workgroup
constexpr int WP_WAVES = NT / 64;
constexpr float WP_EPS = 1e-12f;              // torch.nn.functional.normalize's eps

template <int DT> struct Q4;                  // the 4 query channels of one (pixel, concept)
template <> struct Q4<XMC_BF16> {
    typedef __attribute__((ext_vector_type(4))) xmc_h16 vt;
    __device__ static __forceinline__ void load(const void* b, size_t e, float (&v)[4]) {
        vt t = *reinterpret_cast<const vt*>(reinterpret_cast<const xmc_h16*>(b) + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
    }
    __device__ static __forceinline__ void store(void* b, size_t e, const float (&v)[4]) {
        vt t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (xmc_h16)v[k];
        *reinterpret_cast<vt*>(reinterpret_cast<xmc_h16*>(b) + e) = t;
    }
};
template <> struct Q4<XMC_F32> {
    __device__ static __forceinline__ void load(const void* b, size_t e, float (&v)[4]) {
        f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(b) + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = t[k];
    }
    __device__ static __forceinline__ void store(void* b, size_t e, const float (&v)[4]) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(b) + e) = t;
    }
};

// the image's keys and padding flags into LDS; returns through sk[c][t][k] / spad[t]
template <int TMAX>
__device__ __forceinline__ void load_keys(const float* kh, const unsigned char* pad, int n, int T, float (*sk)[TMAX][WP_P], int* spad) {
    for (int i = threadIdx.x; i < WP_CON * T * WP_P; i += NT) {
        const int c = i / (T * WP_P), r = i % (T * WP_P);
        sk[c][r / WP_P][r % WP_P] = kh[(size_t)n * WP_CON * T * WP_P + i];
    }
    for (int t = threadIdx.x; t < TMAX; t += NT) spad[t] = (t < T) ? (int)pad[(size_t)n * T + t] : 1;
    __syncthreads();
}

// one region's softmax over the words: running max M, sum L (of exp(s - M)), weighted key sum w[4] (un-normalised)
template <int TMAX>
__device__ __forceinline__ void region_softmax(const float (&qh)[4], const float (*sk)[WP_P], const int* spad, int T, float& M, float& L, float (&w)[4]) {
    M = -INFINITY; L = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = 0.f;
    for (int t = 0; t < T; ++t) {
        if (spad[t]) continue;                         // uniform over the workgroup (one image)
        const float s = qh[0] * sk[t][0] + qh[1] * sk[t][1] + qh[2] * sk[t][2] + qh[3] * sk[t][3];
        if (s > M) {
            const float r = __expf(M - s);             // exp(-inf) = 0 on the first word
            L *= r;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] *= r;
            M = s;
        }
        const float e = __expf(s - M);
        L += e;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] += e * sk[t][k];
    }
}

template <int DT, int TMAX>
__global__ __launch_bounds__(NT) void word_pool_fwd_kernel(const void* qmap, const float* kh, const unsigned char* pad, float* ctx,
                                                           int HW, int T, int ppc, float inv_hw) {
    __shared__ float sk[WP_CON][TMAX][WP_P];
    __shared__ int spad[TMAX];
    __shared__ float red[WP_WAVES][WP_CON][WP_P];
    const int n = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x & (WP_CON - 1), slot = threadIdx.x / WP_CON;
    load_keys<TMAX>(kh, pad, n, T, sk, spad);
    float acc[4] = {0, 0, 0, 0};
    const int p1 = min(HW, (chunk + 1) * ppc);
    for (int p = chunk * ppc + slot; p < p1; p += WP_SLOTS) {
        float q[4], qh[4], w[4], M, L;
        Q4<DT>::load(qmap, ((size_t)n * HW + p) * (WP_CON * WP_P) + c * WP_P, q);
        const float inv = 1.f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), WP_EPS);
#pragma unroll
        for (int k = 0; k < 4; ++k) qh[k] = q[k] * inv;
        region_softmax<TMAX>(qh, sk[c], spad, T, M, L, w);
        const float il = 1.f / L;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += w[k] * il;
    }
    // merge the pixel slots: lanes of one concept within a wave sit 16 apart, then the waves through LDS
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc[k] += __shfl_xor(acc[k], 16);
        acc[k] += __shfl_xor(acc[k], 32);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < WP_CON) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][lane][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < WP_CON * WP_P) {
        const int cc = threadIdx.x / WP_P, k = threadIdx.x % WP_P;
        float t = 0.f;
        for (int wv = 0; wv < WP_WAVES; ++wv) t += red[wv][cc][k];
        atomicAdd(&ctx[((size_t)n * WP_CON + cc) * WP_P + k], t * inv_hw);
    }
}

template <int DT, int TMAX>
__global__ __launch_bounds__(NT) void word_pool_bwd_kernel(const void* qmap, const float* kh, const unsigned char* pad, const float* dctx,
                                                           void* dq, float* dkh, int HW, int T, int ppc, float inv_hw) {
    __shared__ float sk[WP_CON][TMAX][WP_P];
    __shared__ int spad[TMAX];
    __shared__ float red[WP_WAVES][WP_CON][TMAX][WP_P];
    const int n = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x & (WP_CON - 1), slot = threadIdx.x / WP_CON;
    load_keys<TMAX>(kh, pad, n, T, sk, spad);
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = dctx[((size_t)n * WP_CON + c) * WP_P + k] * inv_hw;
    float A[TMAX], Dk[TMAX][4];                        // sum_p a_tp,  sum_p ds_tp q^_p
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        A[t] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) Dk[t][k] = 0.f;
    }
    const int p1 = min(HW, (chunk + 1) * ppc);
    for (int p = chunk * ppc + slot; p < p1; p += WP_SLOTS) {
        float q[4], qh[4], w[4], M, L;
        const size_t e = ((size_t)n * HW + p) * (WP_CON * WP_P) + c * WP_P;
        Q4<DT>::load(qmap, e, q);
        const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float inv = 1.f / fmaxf(nrm, WP_EPS);
#pragma unroll
        for (int k = 0; k < 4; ++k) qh[k] = q[k] * inv;
        region_softmax<TMAX>(qh, sk[c], spad, T, M, L, w);
        const float il = 1.f / L;
        const float gc = (g[0] * w[0] + g[1] * w[1] + g[2] * w[2] + g[3] * w[3]) * il;      // <g, ctx of this region>
        float dqh[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T && !spad[t]) {
                const float k0 = sk[c][t][0], k1 = sk[c][t][1], k2 = sk[c][t][2], k3 = sk[c][t][3];
                const float a = __expf(qh[0] * k0 + qh[1] * k1 + qh[2] * k2 + qh[3] * k3 - M) * il;
                const float ds = a * (g[0] * k0 + g[1] * k1 + g[2] * k2 + g[3] * k3 - gc);
                A[t] += a;
#pragma unroll
                for (int k = 0; k < 4; ++k) { dqh[k] += ds * sk[c][t][k]; Dk[t][k] += ds * qh[k]; }
            }
        }
        // through q^ = q / max(|q|, eps): (dq^ - q^ <q^, dq^>) / |q|; below eps the map is a plain scale
        float o[4];
        const float qd = (nrm > WP_EPS) ? qh[0] * dqh[0] + qh[1] * dqh[1] + qh[2] * dqh[2] + qh[3] * dqh[3] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (dqh[k] - qh[k] * qd) * inv;
        Q4<DT>::store(dq, e, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = Dk[t][k] + g[k] * A[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < WP_CON) red[wave][lane][t][k] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WP_CON * T * WP_P; i += NT) {
        const int cc = i / (T * WP_P), r = i % (T * WP_P), t = r / WP_P, k = r % WP_P;
        if (spad[t]) continue;                          // a padding word's key receives nothing (and dkh arrives zero)
        float v = 0.f;
        for (int wv = 0; wv < WP_WAVES; ++wv) v += red[wv][cc][t][k];
        atomicAdd(&dkh[(size_t)n * WP_CON * T * WP_P + i], v);
    }
}

// pixels per workgroup run: ~2048 workgroups over the batch, at least 64 pixels each, a multiple of the 16 slots
static inline int wp_pixels_per_chunk(int N, int HW) {
    int64_t ppc = ((int64_t)N * HW + 2047) / 2048;
    if (ppc < 64) ppc = 64;
    ppc = (ppc + WP_SLOTS - 1) / WP_SLOTS * WP_SLOTS;
    if (xmc_fixed_order()) ppc = ((int64_t)HW + WP_SLOTS - 1) / WP_SLOTS * WP_SLOTS;        // one run per image: one addition per target
    return (int)ppc;
}
}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int xmc_word_pool_fwd(const void* qmap, const float* kh, const unsigned char* pad, float* ctx, int N, int HW, int ncon, int pk,
                                 int T, int dtype, void* s) {
    if (!qmap || !kh || !pad || !ctx || N < 1 || HW < 1) return XMC_EINVAL;
    if (ncon != WP_CON || pk != WP_P || T < 1 || T > 32) return XMC_ESHAPE;
    if (xmc_zero_acc(ctx, sizeof(float) * (size_t)N * WP_CON * WP_P, ST(s)) != hipSuccess) return XMC_EINVAL;
    const int ppc = wp_pixels_per_chunk(N, HW);
    dim3 grid((HW + ppc - 1) / ppc, N);
    const float ih = 1.f / (float)HW;
#define WP_FWD(DT, TM) hipLaunchKernelGGL((word_pool_fwd_kernel<DT, TM>), grid, dim3(NT), 0, ST(s), qmap, kh, pad, ctx, HW, T, ppc, ih)
    if (dtype == XMC_BF16) { if (T <= 16) WP_FWD(XMC_BF16, 16); else if (T <= 24) WP_FWD(XMC_BF16, 24); else WP_FWD(XMC_BF16, 32); }
    else if (dtype == XMC_F32) { if (T <= 16) WP_FWD(XMC_F32, 16); else if (T <= 24) WP_FWD(XMC_F32, 24); else WP_FWD(XMC_F32, 32); }
    else return XMC_EINVAL;
#undef WP_FWD
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_word_pool_bwd(const void* qmap, const float* kh, const unsigned char* pad, const float* dctx, void* dq, float* dkh,
                                 int N, int HW, int ncon, int pk, int T, int dtype, void* s) {
    if (!qmap || !kh || !pad || !dctx || !dq || !dkh || N < 1 || HW < 1) return XMC_EINVAL;
    if (ncon != WP_CON || pk != WP_P || T < 1 || T > 32) return XMC_ESHAPE;
    if (xmc_zero_acc(dkh, sizeof(float) * (size_t)N * WP_CON * T * WP_P, ST(s)) != hipSuccess) return XMC_EINVAL;
    const int ppc = wp_pixels_per_chunk(N, HW);
    dim3 grid((HW + ppc - 1) / ppc, N);
    const float ih = 1.f / (float)HW;
#define WP_BWD(DT, TM) hipLaunchKernelGGL((word_pool_bwd_kernel<DT, TM>), grid, dim3(NT), 0, ST(s), qmap, kh, pad, dctx, dq, dkh, HW, T, ppc, ih)
    if (dtype == XMC_BF16) { if (T <= 16) WP_BWD(XMC_BF16, 16); else if (T <= 24) WP_BWD(XMC_BF16, 24); else WP_BWD(XMC_BF16, 32); }
    else if (dtype == XMC_F32) { if (T <= 16) WP_BWD(XMC_F32, 16); else if (T <= 24) WP_BWD(XMC_F32, 24); else WP_BWD(XMC_F32, 32); }
    else return XMC_EINVAL;
#undef WP_BWD
    XMC_LAUNCH_CHECK();
    return 0;
}
