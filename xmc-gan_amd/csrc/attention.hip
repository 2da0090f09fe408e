// Attention-modulation building blocks of the concept generators (reference model/df_concept_gan.py):
//   * GroupNorm over NHWC (+ optional fused LeakyReLU)            nn.GroupNorm at df_concept_gan.py:171,270-271,549-550
//   * region attention pooling: scores = <q, key>, softmax over H*W, attention-weighted sum of x
//                                                                 CondConceptSampler.forward 293-299, ConceptSampler 570-578
// All HBM/latency bound (4-8 channels per concept): coalesced 16-byte accesses, wave-shuffle + LDS reductions.
#include "common.h"
#include <stdlib.h>

namespace {
constexpr int NT = 256;
constexpr int UN = 4;      // vectors per operand in flight per thread (pointwise.hip: 4.7 -> 5.6 TB/s on flat passes)
constexpr int GN_MAXG = 8 * NT;        // GroupNorm groups per image: at most one per channel, and C / 8 <= NT
// a 16/32-byte vector kept as loaded until it is used
template <int DT> struct Raw8;
template <> struct Raw8<XMC_BF16> {
    bf16x8 v;
    __device__ __forceinline__ void load(const void* p, size_t idx8) { v = reinterpret_cast<const bf16x8*>(p)[idx8]; }
    __device__ __forceinline__ void unpack(float (&o)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    }
};
template <> struct Raw8<XMC_F32> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const void* p, size_t idx8) {
        const f32x4* q = reinterpret_cast<const f32x4*>(p) + idx8 * 2;
        a = q[0]; b = q[1];
    }
    __device__ __forceinline__ void unpack(float (&o)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
    }
};

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
        for (int p0 = blockIdx.x * pix_per_block + g; p0 < p_end; p0 += groups * UN) {
            Raw8<DT> ra[UN], rb[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int p = p0 + j * groups;
                if (p < p_end) {
                    const size_t idx = ((size_t)n * HW + p) * C8 + cc;
                    ra[j].load(a, idx);
                    if (MODE == 1) rb[j].load(b, idx);          // a = x, b = dy
                }
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                if (p0 + j * groups >= p_end) continue;
                float u[8], v[8];
                ra[j].unpack(u);
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s1[k] += u[k]; s2[k] += u[k] * u[k]; }
                } else {
                    rb[j].unpack(v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float xh = (u[k] - mean[k]) * rstd[k];
                        float d = v[k];
                        if (slope >= 0.f) d *= (xh * wv[k] + bv[k]) > 0.f ? 1.f : slope;
                        s1[k] += d; s2[k] += d * xh;
                    }
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
// Everything per (image, channel) is loaded ONCE per thread: blockIdx.y = image, and the grid stride keeps a thread on its
// 8-channel unit (the first version re-read mean / rstd / w / b / A / B per element: 48 dependent loads per 16-byte vector).
// y = lrelu?((x - mean) * rstd * w + b) with (mean, rstd) formed from the per-channel sums; block x == 0 of every image also
// writes stats [N][G][2] = (mean, rstd) for the backward
template <int DT>
__global__ void gn_apply_kernel(const void* x, const float* sums, float* stats, const float* w, const float* b, void* y,
                                int HW, int C8, int cpg, float slope, float eps) {
    const int C = C8 * 8, G = C / cpg, n = blockIdx.y;
    const int stride = gridDim.x * blockDim.x;               // a multiple of C8 (host)
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, cc = i0 % C8;
    // The (mean, rstd) of the image's G groups are formed ONCE per workgroup, one thread per group, and shared through LDS.  (Every thread
    // used to sum its own groups' 2 cpg sums for each of its 8 channels: 128 dependent scalar loads in front of ~32 vector accesses of
    // real work -- the pass ran at 2.6 TB/s inside the 128 px iteration where the plain affine pass reaches 4.2.)
    __shared__ float s_mean[GN_MAXG], s_rstd[GN_MAXG];
    const float m = (float)HW * cpg;
    for (int grp = threadIdx.x; grp < G; grp += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (int q = grp * cpg; q < (grp + 1) * cpg; ++q) { s1 += sums[((size_t)n * C + q) * 2]; s2 += sums[((size_t)n * C + q) * 2 + 1]; }
        const float mean = s1 / m, rstd = rsqrtf(fmaxf(s2 / m - mean * mean, 0.f) + eps);
        s_mean[grp] = mean; s_rstd[grp] = rstd;
        if (blockIdx.x == 0) { stats[((size_t)n * G + grp) * 2] = mean; stats[((size_t)n * G + grp) * 2 + 1] = rstd; }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cc * 8 + k, grp = c / cpg;
        sc[k] = s_rstd[grp] * w[c];
        sh[k] = b[c] - s_mean[grp] * sc[k];
    }
    const int total = HW * C8;
    for (int ib = i0; ib < total; ib += stride * UN) {
        Raw8<DT> raw[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int i = ib + j * stride;
            if (i < total) raw[j].load(x, (size_t)n * total + i);
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int i = ib + j * stride;
            if (i >= total) continue;
            float v[8];
            raw[j].unpack(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float o = v[k] * sc[k] + sh[k];
                v[k] = slope >= 0.f ? (o > 0.f ? o : slope * o) : o;
            }
            Vec8<DT>::store(y, (size_t)n * total + i, v);
        }
    }
}
// dx = rstd * ( w*dy' - (A_g + xhat*B_g)/m ),  A_g = sum_{c in g} w_c S1[n,c], B_g = sum_{c in g} w_c S2[n,c]
// (S = sums of dy', dy'*xhat).  blockIdx.y == N: dw[c] = sum_n S2[n][c], db[c] = sum_n S1[n][c] (one wave per channel).
template <int DT>
__global__ void gn_bwd_apply_kernel(const void* x, const void* dy, const float* stats, const float* sums2, const float* w,
                                    const float* b, void* dx, float* dw, float* db, int N, int HW, int C8, int cpg, float slope) {
    const int C = C8 * 8, G = C / cpg, n = blockIdx.y;
    if (n == N) {
        const int lane = threadIdx.x & 63;
        for (int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c < C; c += gridDim.x * (blockDim.x >> 6)) {
            float a = 0.f, bq = 0.f;
            for (int q = lane; q < N; q += 64) { a += sums2[((size_t)q * C + c) * 2]; bq += sums2[((size_t)q * C + c) * 2 + 1]; }
            a = wave_sum(a); bq = wave_sum(bq);
            if (lane == 0) { db[c] = a; dw[c] = bq; }
        }
        return;
    }
    const int stride = gridDim.x * blockDim.x;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, cc = i0 % C8;
    const float inv_m = 1.f / ((float)HW * cpg);
    // the groups' A_g, B_g once per workgroup through LDS (as in gn_apply_kernel: it was 3 cpg loads per channel and thread)
    __shared__ float s_A[GN_MAXG], s_B[GN_MAXG];
    for (int grp = threadIdx.x; grp < G; grp += blockDim.x) {
        float a = 0.f, bq = 0.f;
        for (int q = grp * cpg; q < (grp + 1) * cpg; ++q) { a += w[q] * sums2[((size_t)n * C + q) * 2]; bq += w[q] * sums2[((size_t)n * C + q) * 2 + 1]; }
        s_A[grp] = a * inv_m; s_B[grp] = bq * inv_m;
    }
    __syncthreads();
    float mean[8], rstd[8], wv[8], bv[8], A[8], B[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cc * 8 + k, grp = c / cpg;
        mean[k] = stats[((size_t)n * G + grp) * 2];
        rstd[k] = stats[((size_t)n * G + grp) * 2 + 1];
        wv[k] = w[c]; bv[k] = b[c];
        A[k] = s_A[grp]; B[k] = s_B[grp];
    }
    const int total = HW * C8;
    for (int ib = i0; ib < total; ib += stride * UN) {
        Raw8<DT> rx[UN], rd[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int i = ib + j * stride;
            if (i < total) { rx[j].load(x, (size_t)n * total + i); rd[j].load(dy, (size_t)n * total + i); }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int i = ib + j * stride;
            if (i >= total) continue;
            float xv[8], dv[8];
            rx[j].unpack(xv);
            rd[j].unpack(dv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = (xv[k] - mean[k]) * rstd[k];
                float d = dv[k];
                if (slope >= 0.f) d *= (xh * wv[k] + bv[k]) > 0.f ? 1.f : slope;
                xv[k] = rstd[k] * (wv[k] * d - (A[k] + xh * B[k]));
            }
            Vec8<DT>::store(dx, (size_t)n * total + i, xv);
        }
    }
}

// ---- region attention pooling, one workgroup per (image, concept); each thread moves whole per-concept vectors
// (pk key channels, px value channels: 8 or 16 bytes) with one vector access per pixel
template <int DT, int P> struct PVec;            // P consecutive channels of one pixel <-> floats
template <int P> struct PVec<XMC_BF16, P> {
    typedef __attribute__((ext_vector_type(P))) xmc_h16 vt;
    __device__ static __forceinline__ void load(const void* base, size_t elem, float (&v)[8]) {
        vt t = *reinterpret_cast<const vt*>(reinterpret_cast<const xmc_h16*>(base) + elem);
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = (float)t[k];
    }
    __device__ static __forceinline__ void store(void* base, size_t elem, const float (&v)[8]) {
        vt t;
#pragma unroll
        for (int k = 0; k < P; ++k) t[k] = (xmc_h16)v[k];
        *reinterpret_cast<vt*>(reinterpret_cast<xmc_h16*>(base) + elem) = t;
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

// ---- region attention pooling (CondConceptSampler.forward 293-299 / ConceptSampler.forward 570-578), coalesced form.
// key [N][HW][16*4], x [N][HW][16*8], q f32 [N][16][4]; ctx f32 [N][16][8]; stats f32 [N][16][2] = (row max, sum of exp).
// A workgroup owns a run of pixels of ONE image for ALL 16 concepts: lane = (pixel slot, concept), so a wave reads 4 whole
// pixels -- 4 x 128 contiguous bytes of key, 4 x 256 of x -- instead of one 8/16-byte piece out of every 256-byte pixel per
// lane (which also made the 16 workgroups of an image, dealt over the 8 XCDs, each pull every line into its own L2).
// Forward: one pass with a running softmax (max, sum, weighted sum re-scaled when the max moves), the 16 pixel slots merged
// through LDS, one partial per (image, run, concept), then a tiny merge kernel.  The attention weights are never stored:
// backward recomputes exp(s - max) / sum from key and q, and the softmax-Jacobian term sum_p a_p <dctx, x_p> is simply
// <dctx, ctx> (ctx being that very sum), so backward is ONE pass too: read key and x, write dkey and dx.
constexpr int AP_CON = 16;                 // concepts (cardinality, df_concept_gan.py:110)
constexpr int AP_SLOTS = NT / AP_CON;      // pixels in flight per workgroup
constexpr int AP_PK = 4, AP_PX = 8;        // state_dim, bottleneck width (df_concept_gan.py:110,118)
constexpr int AP_REC = 2 + AP_PX;          // partial record: max, sum, weighted sum[8]

template <int DT>
__global__ __launch_bounds__(NT) void attn_pool_part_kernel(const void* key, const float* q, const void* x, float* part,
                                                            int HW, float scale, int ppc) {
    __shared__ float sm[AP_SLOTS][AP_CON][AP_REC + 1];
    const int n = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x & (AP_CON - 1), slot = threadIdx.x / AP_CON;
    float qv[AP_PK];
#pragma unroll
    for (int k = 0; k < AP_PK; ++k) qv[k] = q[((size_t)n * AP_CON + c) * AP_PK + k] * scale;
    float m = -INFINITY, l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int p1 = min(HW, (chunk + 1) * ppc);
    for (int p = chunk * ppc + slot; p < p1; p += AP_SLOTS) {
        float kv[8], xv[8];
        PVec<DT, AP_PK>::load(key, ((size_t)n * HW + p) * (AP_CON * AP_PK) + c * AP_PK, kv);
        PVec<DT, AP_PX>::load(x, ((size_t)n * HW + p) * (AP_CON * AP_PX) + c * AP_PX, xv);
        float sc = 0.f;
#pragma unroll
        for (int k = 0; k < AP_PK; ++k) sc += qv[k] * kv[k];
        const float mn = fmaxf(m, sc), f = __expf(m - mn), e = __expf(sc - mn);
        l = l * f + e;
#pragma unroll
        for (int k = 0; k < AP_PX; ++k) acc[k] = acc[k] * f + e * xv[k];
        m = mn;
    }
    sm[slot][c][0] = m; sm[slot][c][1] = l;
#pragma unroll
    for (int k = 0; k < AP_PX; ++k) sm[slot][c][2 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < AP_CON) {
        float M = -INFINITY;
        for (int sl = 0; sl < AP_SLOTS; ++sl) M = fmaxf(M, sm[sl][c][0]);
        float Lq = 0.f, A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int sl = 0; sl < AP_SLOTS; ++sl) {
            const float ms = sm[sl][c][0];
            if (ms == -INFINITY) continue;                       // slot saw no pixel
            const float f = __expf(ms - M);
            Lq += sm[sl][c][1] * f;
#pragma unroll
            for (int k = 0; k < AP_PX; ++k) A[k] += sm[sl][c][2 + k] * f;
        }
        float* o = part + (((size_t)n * gridDim.x + chunk) * AP_CON + c) * AP_REC;
        o[0] = M; o[1] = Lq;
#pragma unroll
        for (int k = 0; k < AP_PX; ++k) o[2 + k] = A[k];
    }
}

__global__ void attn_pool_merge_kernel(const float* part, float* stats, float* ctx, int N, int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (n, c)
    if (i >= N * AP_CON) return;
    const int n = i / AP_CON, c = i % AP_CON;
    float M = -INFINITY;
    for (int ch = 0; ch < chunks; ++ch) M = fmaxf(M, part[(((size_t)n * chunks + ch) * AP_CON + c) * AP_REC]);
    float Lq = 0.f, A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < chunks; ++ch) {
        const float* r = part + (((size_t)n * chunks + ch) * AP_CON + c) * AP_REC;
        const float f = __expf(r[0] - M);
        Lq += r[1] * f;
#pragma unroll
        for (int k = 0; k < AP_PX; ++k) A[k] += r[2 + k] * f;
    }
    stats[(size_t)i * 2] = M; stats[(size_t)i * 2 + 1] = Lq;
    const float inv = 1.f / Lq;
#pragma unroll
    for (int k = 0; k < AP_PX; ++k) ctx[(size_t)i * AP_PX + k] = A[k] * inv;
}

// dq must be zero on entry (partial sums of the runs are added atomically); dkey, dx: every element written exactly once.
// dx_in (may alias dx): another gradient of x that is added on the way out -- the block's `affine` consumes x as well, and its
// gradient would otherwise meet this one in a separate add pass
template <int DT>
__global__ __launch_bounds__(NT) void attn_pool_bwd_kernel(const void* key, const float* q, const void* x, const float* stats,
                                                           const float* ctx, const float* dctx, float* dq, void* dkey, void* dx,
                                                           const void* dx_in, int HW, float scale, int ppc) {
    __shared__ float sm[AP_SLOTS][AP_CON][AP_PK + 1];
    const int n = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x & (AP_CON - 1), slot = threadIdx.x / AP_CON;
    const size_t nc = (size_t)n * AP_CON + c;
    float qv[AP_PK], dc[AP_PX];
#pragma unroll
    for (int k = 0; k < AP_PK; ++k) qv[k] = q[nc * AP_PK + k] * scale;
    float dot = 0.f;                                           // sum_p a_p <dctx, x_p> = <dctx, ctx>
#pragma unroll
    for (int k = 0; k < AP_PX; ++k) { dc[k] = dctx[nc * AP_PX + k]; dot += dc[k] * ctx[nc * AP_PX + k]; }
    const float M = stats[nc * 2], invL = 1.f / stats[nc * 2 + 1];
    float dqa[AP_PK] = {0, 0, 0, 0};
    const int p1 = min(HW, (chunk + 1) * ppc);
    for (int p = chunk * ppc + slot; p < p1; p += AP_SLOTS) {
        float kv[8], xv[8], g[8];
        const size_t ek = ((size_t)n * HW + p) * (AP_CON * AP_PK) + c * AP_PK, ex = ((size_t)n * HW + p) * (AP_CON * AP_PX) + c * AP_PX;
        PVec<DT, AP_PK>::load(key, ek, kv);
        PVec<DT, AP_PX>::load(x, ex, xv);
        float sc = 0.f, da = 0.f;
#pragma unroll
        for (int k = 0; k < AP_PK; ++k) sc += qv[k] * kv[k];
        const float a = __expf(sc - M) * invL;
#pragma unroll
        for (int k = 0; k < AP_PX; ++k) { da += dc[k] * xv[k]; g[k] = a * dc[k]; }
        if (dx_in) {
            float o[8];
            PVec<DT, AP_PX>::load(dx_in, ex, o);
#pragma unroll
            for (int k = 0; k < AP_PX; ++k) g[k] += o[k];
        }
        PVec<DT, AP_PX>::store(dx, ex, g);
        const float ds = a * (da - dot);
#pragma unroll
        for (int k = 0; k < AP_PK; ++k) { dqa[k] += ds * kv[k]; g[k] = ds * qv[k]; }
        PVec<DT, AP_PK>::store(dkey, ek, g);
    }
#pragma unroll
    for (int k = 0; k < AP_PK; ++k) sm[slot][c][k] = dqa[k];
    __syncthreads();
    if (threadIdx.x < AP_CON * AP_PK) {
        const int cc = threadIdx.x / AP_PK, k = threadIdx.x % AP_PK;
        float t = 0.f;
        for (int sl = 0; sl < AP_SLOTS; ++sl) t += sm[sl][cc][k];
        atomicAdd(&dq[((size_t)n * AP_CON + cc) * AP_PK + k], t * scale);
    }
}

// pixels per workgroup run: ~2048 workgroups over the whole batch, at least 64 pixels each, a multiple of the 16 slots
static inline int ap_pixels_per_chunk(int N, int HW) {
    int64_t ppc = ((int64_t)N * HW + 2047) / 2048;
    if (ppc < 64) ppc = 64;
    ppc = (ppc + AP_SLOTS - 1) / AP_SLOTS * AP_SLOTS;
    return (int)ppc;
}
// blocks per image of the apply kernels: ~2048 workgroups over the batch, each thread >= 2 vectors, grid stride a multiple of C8
static inline int gn_apply_blocks(int N, int HW, int C8) {
    int64_t per = ((int64_t)HW * C8 + 2 * NT - 1) / (2 * NT);
    int64_t cap = (2048 + N - 1) / N;
    int b = (int)(per < cap ? per : cap);
    if (b < 1) b = 1;
    int g = NT, r = C8;                               // blocks * NT must be a multiple of C8
    while (r) { int t = g % r; g = r; r = t; }
    const int q = C8 / g;
    return (b + q - 1) / q * q;
}
}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

// GroupNorm forward: y = lrelu?((x - mean)/sqrt(var+eps) * w + b); stats f32 [N][G][2] (mean, rstd) is written for backward;
// workgroups per image of the statistics pass: 64 pixels per thread lane on the large maps (fewer atomics: see below); on the small maps of
// a 64-image batch that is ONE workgroup per image, 64 on a 256-CU chip, each walking its pixels in a serial loop (32x32x128: 7.6 us for
// 17 MB) -- there 32 or 16 pixels per lane until the launch has a workgroup per CU
static int gn_sums_blocks(int N, int HW, int groups) {
    static const bool wide = xmc_debug_off("gn_sums_64");
    int ppl = 64;
    while (!wide && ppl > 16 && (long long)N * ((HW + groups * ppl - 1) / (groups * ppl)) < 256) ppl >>= 1;
    const int bx = (HW + groups * ppl - 1) / (groups * ppl);
    return bx < 1 ? 1 : bx;
}
// ws f32 [N][C][2] scratch (zeroed here).  slope < 0: no activation.
extern "C" int xmc_groupnorm_fwd(const void* x, const float* w, const float* b, void* y, float* stats, float* ws,
                                 int N, int HW, int C, int G, float eps, float slope, int dtype, void* s) {
    if (C % 8 || C % G || C / 8 > NT) return XMC_EALIGN;
    const int C8 = C / 8, cpg = C / G, groups = NT / C8;
    // 64 pixels per thread lane of the statistics pass: with 16, a 128x128x128 map had 4096 workgroups per launch ending in 1 M
    // atomics on 16 k addresses and read at 2.0 TB/s (sums) / 3.0 (backward sums); 64: whole forward 272 -> 202 us
    int bx = gn_sums_blocks(N, HW, groups);
    if (xmc_fixed_order()) bx = 1;                   // one workgroup per image: each (image, channel) sum is formed in one order
    int ppb = (HW + bx - 1) / bx;
    hipError_t e = xmc_zero_acc(ws, (size_t)N * C * 2 * 4, ST(s));
    if (e != hipSuccess) return -(1000 + (int)e);
    const int blocks = gn_apply_blocks(N, HW, C8);
    if (dtype == XMC_BF16) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_BF16, 0>), dim3(bx, N), dim3(NT), 0, ST(s), x, nullptr, nullptr, nullptr, nullptr, ws, HW, C8, cpg, -1.f, ppb);
        hipLaunchKernelGGL((gn_apply_kernel<XMC_BF16>), dim3(blocks, N), dim3(NT), 0, ST(s), x, ws, stats, w, b, y, HW, C8, cpg, slope, eps);
    } else if (dtype == XMC_F32) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_F32, 0>), dim3(bx, N), dim3(NT), 0, ST(s), x, nullptr, nullptr, nullptr, nullptr, ws, HW, C8, cpg, -1.f, ppb);
        hipLaunchKernelGGL((gn_apply_kernel<XMC_F32>), dim3(blocks, N), dim3(NT), 0, ST(s), x, ws, stats, w, b, y, HW, C8, cpg, slope, eps);
    } else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_groupnorm_bwd(const void* x, const void* dy, const float* w, const float* b, const float* stats, void* dx,
                                 float* dw, float* db, float* ws, int N, int HW, int C, int G, float slope, int dtype, void* s) {
    if (C % 8 || C % G || C / 8 > NT) return XMC_EALIGN;
    const int C8 = C / 8, cpg = C / G, groups = NT / C8;
    // 64 pixels per thread lane of the statistics pass: with 16, a 128x128x128 map had 4096 workgroups per launch ending in 1 M
    // atomics on 16 k addresses and read at 2.0 TB/s (sums) / 3.0 (backward sums); 64: whole forward 272 -> 202 us
    int bx = gn_sums_blocks(N, HW, groups);
    if (xmc_fixed_order()) bx = 1;                   // one workgroup per image: each (image, channel) sum is formed in one order
    int ppb = (HW + bx - 1) / bx;
    hipError_t e = xmc_zero_acc(ws, (size_t)N * C * 2 * 4, ST(s));
    if (e != hipSuccess) return -(1000 + (int)e);
    const int blocks = gn_apply_blocks(N, HW, C8);   // grid.y = N images + one slice of blocks for the parameter gradients
    if (dtype == XMC_BF16) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_BF16, 1>), dim3(bx, N), dim3(NT), 0, ST(s), x, dy, stats, w, b, ws, HW, C8, cpg, slope, ppb);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<XMC_BF16>), dim3(blocks, N + 1), dim3(NT), 0, ST(s), x, dy, stats, ws, w, b, dx, dw, db, N, HW, C8, cpg, slope);
    } else if (dtype == XMC_F32) {
        hipLaunchKernelGGL((gn_sums_kernel<XMC_F32, 1>), dim3(bx, N), dim3(NT), 0, ST(s), x, dy, stats, w, b, ws, HW, C8, cpg, slope, ppb);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<XMC_F32>), dim3(blocks, N + 1), dim3(NT), 0, ST(s), x, dy, stats, ws, w, b, dx, dw, db, N, HW, C8, cpg, slope);
    } else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int64_t xmc_attn_pool_ws_floats(int N, int HW) {
    if (N < 1 || HW < 1) return 0;
    const int ppc = ap_pixels_per_chunk(N, HW), chunks = (HW + ppc - 1) / ppc;
    return (int64_t)N * chunks * AP_CON * AP_REC;
}
extern "C" int xmc_attn_pool_fwd(const void* key, const float* q, const void* x, float* stats, float* ctx, float* ws, int N, int HW,
                                 int ncon, int pk, int px, float scale, int dtype, void* s) {
    if (!key || !q || !x || !stats || !ctx || !ws || N < 1 || HW < 1) return XMC_EINVAL;
    if (pk != AP_PK || px != AP_PX || ncon != AP_CON) return XMC_ESHAPE;   // cardinality 16, state_dim 4, bottleneck width 8 (df_concept_gan.py:110,118)
    const int ppc = ap_pixels_per_chunk(N, HW), chunks = (HW + ppc - 1) / ppc;
    dim3 grid(chunks, N);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((attn_pool_part_kernel<XMC_BF16>), grid, dim3(NT), 0, ST(s), key, q, x, ws, HW, scale, ppc);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((attn_pool_part_kernel<XMC_F32>), grid, dim3(NT), 0, ST(s), key, q, x, ws, HW, scale, ppc);
    else return XMC_EINVAL;
    hipLaunchKernelGGL(attn_pool_merge_kernel, dim3((N * AP_CON + NT - 1) / NT), dim3(NT), 0, ST(s), ws, stats, ctx, N, chunks);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_attn_pool_bwd_acc(const void* key, const float* q, const void* x, const float* stats, const float* ctx,
                                     const float* dctx, float* dq, void* dkey, void* dx, const void* dx_in, int N, int HW, int ncon,
                                     int pk, int px, float scale, int dtype, void* s) {
    if (!key || !q || !x || !stats || !ctx || !dctx || !dq || !dkey || !dx || N < 1 || HW < 1) return XMC_EINVAL;
    if (pk != AP_PK || px != AP_PX || ncon != AP_CON) return XMC_ESHAPE;
    if (xmc_zero_acc(dq, sizeof(float) * (size_t)N * AP_CON * AP_PK, ST(s)) != hipSuccess) return XMC_EINVAL;
    int ppc = ap_pixels_per_chunk(N, HW);
    if (xmc_fixed_order()) ppc = (HW + AP_SLOTS - 1) / AP_SLOTS * AP_SLOTS;      // one run per image: dq gets one addition per element
    const int chunks = (HW + ppc - 1) / ppc;
    dim3 grid(chunks, N);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((attn_pool_bwd_kernel<XMC_BF16>), grid, dim3(NT), 0, ST(s), key, q, x, stats, ctx, dctx, dq, dkey, dx, dx_in, HW, scale, ppc);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((attn_pool_bwd_kernel<XMC_F32>), grid, dim3(NT), 0, ST(s), key, q, x, stats, ctx, dctx, dq, dkey, dx, dx_in, HW, scale, ppc);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_attn_pool_bwd(const void* key, const float* q, const void* x, const float* stats, const float* ctx,
                                 const float* dctx, float* dq, void* dkey, void* dx, int N, int HW, int ncon, int pk, int px,
                                 float scale, int dtype, void* s) {
    return xmc_attn_pool_bwd_acc(key, q, x, stats, ctx, dctx, dq, dkey, dx, nullptr, N, HW, ncon, pk, px, scale, dtype, s);
}
