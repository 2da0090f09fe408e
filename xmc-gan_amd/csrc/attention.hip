// Attention-modulation building blocks of the concept generators (reference model/df_concept_gan.py):
//   * GroupNorm over NHWC (+ optional fused LeakyReLU)            nn.GroupNorm at df_concept_gan.py:171,270-271,549-550
//   * region attention pooling: scores = <q, key>, softmax over H*W, attention-weighted sum of x
//                                                                 CondConceptSampler.forward 293-299, ConceptSampler 570-578
// All HBM/latency bound (4-8 channels per concept): coalesced 16-byte accesses, wave-shuffle + LDS reductions.
#include "common.h"

namespace {
constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0];
    for (int w = 1; w < NT / 64; ++w) t = fmaxf(t, sh[w]);
    return t;
}

// ---- per-(image, channel) sums: MODE 0: (x, x^2)   MODE 1: (dy', dy'*xhat) with dy' = dy * lrelu'(y) if slope >= 0
template <int DT, int MODE>
__global__ void gn_sums_kernel(const void* a, const void* b, const float* stats, const float* w, const float* bias,
                               float* out, int HW, int C8, int cpg, float slope, int pix_per_block) {
    const int n = blockIdx.y, groups = NT / C8;
    const int cc = threadIdx.x % C8, g = threadIdx.x / C8;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mean[8], rstd[8], wv[8], bv[8];
    const int C = C8 * 8;
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int c = cc * 8 + k, grp = c / cpg;
            mean[k] = stats[((size_t)n * (C / cpg) + grp) * 2];
            rstd[k] = stats[((size_t)n * (C / cpg) + grp) * 2 + 1];
            wv[k] = w[c]; bv[k] = bias[c];
        }
    }
    const int p_end = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    if (g < groups)
        for (int p = blockIdx.x * pix_per_block + g; p < p_end; p += groups) {
            float u[8], v[8];
            const size_t idx = ((size_t)n * HW + p) * C8 + cc;
            Vec8<DT>::load(a, idx, u);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1[k] += u[k]; s2[k] += u[k] * u[k]; }
            } else {
                Vec8<DT>::load(b, idx, v);            // a = x, b = dy
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float xh = (u[k] - mean[k]) * rstd[k];
                    float d = v[k];
                    if (slope >= 0.f) d *= (xh * wv[k] + bv[k]) > 0.f ? 1.f : slope;
                    s1[k] += d; s2[k] += d * xh;
                }
            }
        }
    __shared__ float red[NT * 8];
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = q == 0 ? s1[k] : s2[k];
        __syncthreads();
        if (threadIdx.x < C8) {
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int gg = 0; gg < groups; ++gg)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] += red[(gg * C8 + threadIdx.x) * 8 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&out[(((size_t)n * C) + threadIdx.x * 8 + k) * 2 + q], t[k]);
        }
    }
}
// sums [N][C][2] -> stats [N][G][2] = (mean, rstd)
__global__ void gn_finalize_kernel(const float* sums, float* stats, int N, int C, int cpg, int HW, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int G = C / cpg;
    if (i >= N * G) return;
    int n = i / G, g = i % G;
    float s1 = 0.f, s2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s1 += sums[((size_t)n * C + c) * 2]; s2 += sums[((size_t)n * C + c) * 2 + 1]; }
    float m = (float)HW * cpg, mean = s1 / m;
    float var = fmaxf(s2 / m - mean * mean, 0.f);
    stats[(size_t)i * 2] = mean; stats[(size_t)i * 2 + 1] = rsqrtf(var + eps);
}
template <int DT>
__global__ void gn_apply_kernel(const void* x, const float* stats, const float* w, const float* b, void* y,
                                int N, int HW, int C8, int cpg, float slope) {
    const int C = C8 * 8, G = C / cpg;
    const int64_t total = (int64_t)N * HW * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int n = (int)(i / ((int64_t)HW * C8));
        float v[8];
        Vec8<DT>::load(x, i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int c = cc * 8 + k, grp = c / cpg;
            float o = (v[k] - stats[((size_t)n * G + grp) * 2]) * stats[((size_t)n * G + grp) * 2 + 1] * w[c] + b[c];
            v[k] = slope >= 0.f ? (o > 0.f ? o : slope * o) : o;
        }
        Vec8<DT>::store(y, i, v);
    }
}
// per (image, group): A_g = sum_{c in g} w_c S1[n,c], B_g = sum_{c in g} w_c S2[n,c]   (S = sums of dy', dy'*xhat)
__global__ void gn_bwd_groupsums_kernel(const float* sums2, const float* w, float* ab, int N, int C, int cpg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int G = C / cpg;
    if (i >= N * G) return;
    int n = i / G, g = i % G;
    float A = 0.f, Bq = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        A += w[c] * sums2[((size_t)n * C + c) * 2];
        Bq += w[c] * sums2[((size_t)n * C + c) * 2 + 1];
    }
    ab[(size_t)i * 2] = A; ab[(size_t)i * 2 + 1] = Bq;
}
// dx = rstd * ( w*dy' - (A_g + xhat*B_g)/m )
template <int DT>
__global__ void gn_bwd_apply_kernel(const void* x, const void* dy, const float* stats, const float* ab, const float* w,
                                    const float* b, void* dx, int N, int HW, int C8, int cpg, float slope) {
    const int C = C8 * 8, G = C / cpg;
    const float inv_m = 1.f / ((float)HW * cpg);
    const int64_t total = (int64_t)N * HW * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int n = (int)(i / ((int64_t)HW * C8));
        float xv[8], dv[8];
        Vec8<DT>::load(x, i, xv);
        Vec8<DT>::load(dy, i, dv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int c = cc * 8 + k, grp = c / cpg;
            const size_t gi = ((size_t)n * G + grp) * 2;
            float mean = stats[gi], rstd = stats[gi + 1];
            float xh = (xv[k] - mean) * rstd;
            float d = dv[k];
            if (slope >= 0.f) d *= (xh * w[c] + b[c]) > 0.f ? 1.f : slope;
            xv[k] = rstd * (w[c] * d - (ab[gi] + xh * ab[gi + 1]) * inv_m);
        }
        Vec8<DT>::store(dx, i, xv);
    }
}
// dw[c] = sum_n S2[n][c], db[c] = sum_n S1[n][c]
__global__ void gn_param_grads_kernel(const float* sums2, float* dw, float* db, int N, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, bq = 0.f;
    for (int n = 0; n < N; ++n) { a += sums2[((size_t)n * C + c) * 2]; bq += sums2[((size_t)n * C + c) * 2 + 1]; }
    db[c] = a; dw[c] = bq;
}

// ---- region attention pooling, one workgroup per (image, concept); each thread moves whole per-concept vectors
// (pk key channels, px value channels: 8 or 16 bytes) with one vector access per pixel
template <int DT, int P> struct PVec;            // P consecutive channels of one pixel <-> floats
template <int P> struct PVec<XMC_BF16, P> {
    typedef __attribute__((ext_vector_type(P))) __bf16 vt;
    __device__ static __forceinline__ void load(const void* base, size_t elem, float (&v)[8]) {
        vt t = *reinterpret_cast<const vt*>(reinterpret_cast<const __bf16*>(base) + elem);
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = (float)t[k];
    }
    __device__ static __forceinline__ void store(void* base, size_t elem, const float (&v)[8]) {
        vt t;
#pragma unroll
        for (int k = 0; k < P; ++k) t[k] = (__bf16)v[k];
        *reinterpret_cast<vt*>(reinterpret_cast<__bf16*>(base) + elem) = t;
    }
};
template <int P> struct PVec<XMC_F32, P> {
    typedef __attribute__((ext_vector_type(P))) float vt;
    __device__ static __forceinline__ void load(const void* base, size_t elem, float (&v)[8]) {
        vt t = *reinterpret_cast<const vt*>(reinterpret_cast<const float*>(base) + elem);
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = t[k];
    }
    __device__ static __forceinline__ void store(void* base, size_t elem, const float (&v)[8]) {
        vt t;
#pragma unroll
        for (int k = 0; k < P; ++k) t[k] = v[k];
        *reinterpret_cast<vt*>(reinterpret_cast<float*>(base) + elem) = t;
    }
};

// key [N][HW][CK] (CK = ncon*PK), q f32 [N][ncon][PK], x [N][HW][CX] (CX = ncon*PX); attn f32 [N][ncon][HW]; ctx f32 [N][ncon][PX]
template <int DT, int PK, int PX>
__global__ void attn_pool_fwd_kernel(const void* key, const float* q, const void* x, float* attn, float* ctx,
                                     int HW, int ncon, float scale) {
    const int n = blockIdx.x / ncon, c = blockIdx.x % ncon;
    __shared__ float sh[NT / 64];
    __shared__ float red[NT * 8];
    float qv[8];
#pragma unroll
    for (int k = 0; k < PK; ++k) qv[k] = q[((size_t)n * ncon + c) * PK + k] * scale;
    const int CK = ncon * PK, CX = ncon * PX;
    float* arow = attn + ((size_t)n * ncon + c) * HW;
    float mx = -INFINITY;
    for (int p = threadIdx.x; p < HW; p += NT) {
        float kv[8];
        PVec<DT, PK>::load(key, ((size_t)n * HW + p) * CK + c * PK, kv);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PK; ++k) s += qv[k] * kv[k];
        arow[p] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, sh);
    float se = 0.f;
    for (int p = threadIdx.x; p < HW; p += NT) { float e = __expf(arow[p] - mx); arow[p] = e; se += e; }
    se = block_sum(se, sh);
    const float inv = 1.f / se;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = threadIdx.x; p < HW; p += NT) {
        float a = arow[p] * inv;
        arow[p] = a;
        float xv[8];
        PVec<DT, PX>::load(x, ((size_t)n * HW + p) * CX + c * PX, xv);
#pragma unroll
        for (int k = 0; k < PX; ++k) acc[k] += a * xv[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < PX) {
        float t = 0.f;
        for (int i = 0; i < NT; ++i) t += red[i * 8 + threadIdx.x];
        ctx[((size_t)n * ncon + c) * PX + threadIdx.x] = t;
    }
}
// given dctx: dq, dkey (written, every element owned by exactly one workgroup), dx (written)
template <int DT, int PK, int PX>
__global__ void attn_pool_bwd_kernel(const void* key, const float* q, const void* x, const float* attn, const float* dctx,
                                     float* dq, void* dkey, void* dx, int HW, int ncon, float scale) {
    const int n = blockIdx.x / ncon, c = blockIdx.x % ncon;
    __shared__ float sh[NT / 64];
    __shared__ float red[NT * 8];
    const int CK = ncon * PK, CX = ncon * PX;
    float qv[8], dc[8];
#pragma unroll
    for (int k = 0; k < PK; ++k) qv[k] = q[((size_t)n * ncon + c) * PK + k] * scale;
#pragma unroll
    for (int k = 0; k < PX; ++k) dc[k] = dctx[((size_t)n * ncon + c) * PX + k];
    const float* arow = attn + ((size_t)n * ncon + c) * HW;
    float dot = 0.f;                                   // sum_hw attn * dattn
    for (int p = threadIdx.x; p < HW; p += NT) {
        float xv[8];
        PVec<DT, PX>::load(x, ((size_t)n * HW + p) * CX + c * PX, xv);
        float da = 0.f;
#pragma unroll
        for (int k = 0; k < PX; ++k) da += dc[k] * xv[k];
        dot += arow[p] * da;
    }
    dot = block_sum(dot, sh);
    float dqa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = threadIdx.x; p < HW; p += NT) {
        const float a = arow[p];
        float xv[8], kv[8], g[8];
        const size_t ex = ((size_t)n * HW + p) * CX + c * PX, ek = ((size_t)n * HW + p) * CK + c * PK;
        PVec<DT, PX>::load(x, ex, xv);
        PVec<DT, PK>::load(key, ek, kv);
        float da = 0.f;
#pragma unroll
        for (int k = 0; k < PX; ++k) { da += dc[k] * xv[k]; g[k] = a * dc[k]; }
        PVec<DT, PX>::store(dx, ex, g);
        const float ds = a * (da - dot);
#pragma unroll
        for (int k = 0; k < PK; ++k) { dqa[k] += ds * kv[k]; g[k] = ds * qv[k]; }
        PVec<DT, PK>::store(dkey, ek, g);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = dqa[k];
    __syncthreads();
    if (threadIdx.x < PK) {
        float t = 0.f;
        for (int i = 0; i < NT; ++i) t += red[i * 8 + threadIdx.x];
        dq[((size_t)n * ncon + c) * PK + threadIdx.x] = t * scale;
    }
}
}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

// GroupNorm forward: y = lrelu?((x - mean)/sqrt(var+eps) * w + b); stats f32 [N][G][2] (mean, rstd) is written for backward;
// ws f32 [N][C][2] scratch (zeroed here).  slope < 0: no activation.
extern "C" int xmc_groupnorm_fwd(const void* x, const float* w, const float* b, void* y, float* stats, float* ws,
                                 int N, int HW, int C, int G, float eps, float slope, int dtype, void* s) {
    if (C % 8 || C % G || C / 8 > NT) return XMC_EALIGN;
    const int C8 = C / 8, cpg = C / G, groups = NT / C8;
    int bx = (HW + groups * 16 - 1) / (groups * 16); if (bx < 1) bx = 1;
    int ppb = (HW + bx - 1) / bx;
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)N * C * 2 * 4, ST(s));
    if (e != hipSuccess) return (int)e;
    int64_t total = (int64_t)N * HW * C8; int blocks = (int)((total + NT - 1) / NT); if (blocks > 4096) blocks = 4096;
    if (dtype == XMC_BF16) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_BF16, 0>), dim3(bx, N), dim3(NT), 0, ST(s), x, nullptr, nullptr, nullptr, nullptr, ws, HW, C8, cpg, -1.f, ppb);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((N * G + NT - 1) / NT), dim3(NT), 0, ST(s), ws, stats, N, C, cpg, HW, eps);
        hipLaunchKernelGGL((gn_apply_kernel<XMC_BF16>), dim3(blocks), dim3(NT), 0, ST(s), x, stats, w, b, y, N, HW, C8, cpg, slope);
    } else if (dtype == XMC_F32) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_F32, 0>), dim3(bx, N), dim3(NT), 0, ST(s), x, nullptr, nullptr, nullptr, nullptr, ws, HW, C8, cpg, -1.f, ppb);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((N * G + NT - 1) / NT), dim3(NT), 0, ST(s), ws, stats, N, C, cpg, HW, eps);
        hipLaunchKernelGGL((gn_apply_kernel<XMC_F32>), dim3(blocks), dim3(NT), 0, ST(s), x, stats, w, b, y, N, HW, C8, cpg, slope);
    } else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_groupnorm_bwd(const void* x, const void* dy, const float* w, const float* b, const float* stats, void* dx,
                                 float* dw, float* db, float* ws, int N, int HW, int C, int G, float slope, int dtype, void* s) {
    if (C % 8 || C % G || C / 8 > NT) return XMC_EALIGN;
    const int C8 = C / 8, cpg = C / G, groups = NT / C8;
    int bx = (HW + groups * 16 - 1) / (groups * 16); if (bx < 1) bx = 1;
    int ppb = (HW + bx - 1) / bx;
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)N * C * 2 * 4, ST(s));
    if (e != hipSuccess) return (int)e;
    float* ab = ws + (size_t)N * C * 2;              // [N][G][2], second part of the workspace
    int64_t total = (int64_t)N * HW * C8; int blocks = (int)((total + NT - 1) / NT); if (blocks > 4096) blocks = 4096;
    if (dtype == XMC_BF16) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_BF16, 1>), dim3(bx, N), dim3(NT), 0, ST(s), x, dy, stats, w, b, ws, HW, C8, cpg, slope, ppb);
        hipLaunchKernelGGL(gn_bwd_groupsums_kernel, dim3((N * G + NT - 1) / NT), dim3(NT), 0, ST(s), ws, w, ab, N, C, cpg);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<XMC_BF16>), dim3(blocks), dim3(NT), 0, ST(s), x, dy, stats, ab, w, b, dx, N, HW, C8, cpg, slope);
    } else if (dtype == XMC_F32) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_F32, 1>), dim3(bx, N), dim3(NT), 0, ST(s), x, dy, stats, w, b, ws, HW, C8, cpg, slope, ppb);
        hipLaunchKernelGGL(gn_bwd_groupsums_kernel, dim3((N * G + NT - 1) / NT), dim3(NT), 0, ST(s), ws, w, ab, N, C, cpg);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<XMC_F32>), dim3(blocks), dim3(NT), 0, ST(s), x, dy, stats, ab, w, b, dx, N, HW, C8, cpg, slope);
    } else return XMC_EINVAL;
    hipLaunchKernelGGL(gn_param_grads_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, ST(s), ws, dw, db, N, C);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_attn_pool_fwd(const void* key, const float* q, const void* x, float* attn, float* ctx, int N, int HW,
                                 int ncon, int pk, int px, float scale, int dtype, void* s) {
    if (pk != 4 || px != 8 || ncon < 1) return XMC_ESHAPE;          // state_dim 4, bottleneck width 8 (df_concept_gan.py:110,118)
    if (dtype == XMC_BF16) hipLaunchKernelGGL((attn_pool_fwd_kernel<XMC_BF16, 4, 8>), dim3(N * ncon), dim3(NT), 0, ST(s), key, q, x, attn, ctx, HW, ncon, scale);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((attn_pool_fwd_kernel<XMC_F32, 4, 8>), dim3(N * ncon), dim3(NT), 0, ST(s), key, q, x, attn, ctx, HW, ncon, scale);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_attn_pool_bwd(const void* key, const float* q, const void* x, const float* attn, const float* dctx, float* dq,
                                 void* dkey, void* dx, int N, int HW, int ncon, int pk, int px, float scale, int dtype, void* s) {
    if (pk != 4 || px != 8 || ncon < 1) return XMC_ESHAPE;
    if (dtype == XMC_BF16) hipLaunchKernelGGL((attn_pool_bwd_kernel<XMC_BF16, 4, 8>), dim3(N * ncon), dim3(NT), 0, ST(s), key, q, x, attn, dctx, dq, dkey, dx, HW, ncon, scale);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((attn_pool_bwd_kernel<XMC_F32, 4, 8>), dim3(N * ncon), dim3(NT), 0, ST(s), key, q, x, attn, dctx, dq, dkey, dx, HW, ncon, scale);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
