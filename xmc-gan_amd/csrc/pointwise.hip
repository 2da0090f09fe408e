// HBM-bound pointwise / small-reduction kernels (NHWC, 8-channel vectors, 16-byte accesses).
// Each cites the reference op it replaces in include/xmc_gan_hip.h.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <cstdarg>
#include <cstdio>

namespace {

constexpr int NT = 256;
inline int nblocks(int64_t items, int per_block = NT, int cap = 256 * 16) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// ------------------------------------------------------------------ generic 8-wide maps
// Streaming shape of the flat pointwise kernels in this file (measured on 1 GB bf16 operands, two reads + one write): one 16-byte
// vector per thread per loop iteration and 2048-4096 workgroups moved 4.7 TB/s; UN = 4 vectors per operand requested before the
// first use (a workgroup walks UN*NT consecutive vectors per step) and 4 workgroups per CU move 5.6 (the same traffic through
// ATen's add: 6.0; plain device copy 4.7-5.3, fill 6.8).
constexpr int UN = 4;
constexpr int STREAM_BLOCKS = 1024;
inline int sblocks(int64_t n8) { return nblocks((n8 + UN - 1) / UN, NT, STREAM_BLOCKS); }

template <int DT, class F>
__global__ void map1_kernel(const void* x, void* y, int64_t n8, F f) {
    for (int64_t base = (int64_t)blockIdx.x * (NT * UN); base < n8; base += (int64_t)gridDim.x * (NT * UN)) {
        float v[UN][8];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) Vec8<DT>::load(x, i, v[j]);
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[j][k] = f(v[j][k]);
                Vec8<DT>::store(y, i, v[j]);
            }
        }
    }
}
template <int DT, class F>
__global__ void map2_kernel(const void* a, const void* b, void* y, int64_t n8, F f) {
    for (int64_t base = (int64_t)blockIdx.x * (NT * UN); base < n8; base += (int64_t)gridDim.x * (NT * UN)) {
        float u[UN][8], v[UN][8];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) { Vec8<DT>::load(a, i, u[j]); Vec8<DT>::load(b, i, v[j]); }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) u[j][k] = f(u[j][k], v[j][k]);
                Vec8<DT>::store(y, i, u[j]);
            }
        }
    }
}

struct FLrelu { float slope; __device__ float operator()(float v) const { return v > 0.f ? v : slope * v; } };
struct FTanh { __device__ float operator()(float v) const { return tanhf(v); } };
struct FLreluMask { float slope; __device__ float operator()(float dy, float ref) const { return ref > 0.f ? dy : slope * dy; } };
struct FTanhBwd { __device__ float operator()(float dy, float y) const { return dy * (1.f - y * y); } };
struct FAxpby { const float* alpha; __device__ float operator()(float a, float b) const { return a + (*alpha) * b; } };
struct FScale { const float* alpha; __device__ float operator()(float v) const { return (*alpha) * v; } };

template <class F>
int run_map1(const void* x, void* y, int64_t n, int dtype, hipStream_t st, F f) {
    if (n % 8) return XMC_EALIGN;
    if (n == 0) return 0;
    int64_t n8 = n / 8;
    if (dtype == XMC_BF16) hipLaunchKernelGGL((map1_kernel<XMC_BF16, F>), dim3(sblocks(n8)), dim3(NT), 0, st, x, y, n8, f);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((map1_kernel<XMC_F32, F>), dim3(sblocks(n8)), dim3(NT), 0, st, x, y, n8, f);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
// y = dy * LeakyReLU'(branch) with the branch given as its sign bits (XmcConvDesc.sign_bits: one byte per 8-channel unit, bit r =
// element r > 0): the mask pass of a discriminator block's backward without the branch tensor (17/16 of a tensor read, one written)
template <int DT>
__global__ void signmask_kernel(const void* dy, const unsigned char* __restrict__ bits, void* y, int64_t n8, float slope) {
    for (int64_t base = (int64_t)blockIdx.x * (NT * UN); base < n8; base += (int64_t)gridDim.x * (NT * UN)) {
        float u[UN][8];
        unsigned b[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) { Vec8<DT>::load(dy, i, u[j]); b[j] = bits[i]; }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) u[j][k] = ((b[j] >> k) & 1u) ? u[j][k] : slope * u[j][k];
                Vec8<DT>::store(y, i, u[j]);
            }
        }
    }
}
template <class F>
int run_map2(const void* a, const void* b, void* y, int64_t n, int dtype, hipStream_t st, F f) {
    if (n % 8) return XMC_EALIGN;
    if (n == 0) return 0;
    int64_t n8 = n / 8;
    if (dtype == XMC_BF16) hipLaunchKernelGGL((map2_kernel<XMC_BF16, F>), dim3(sblocks(n8)), dim3(NT), 0, st, a, b, y, n8, f);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((map2_kernel<XMC_F32, F>), dim3(sblocks(n8)), dim3(NT), 0, st, a, b, y, n8, f);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ cast
template <int SD, int DD>
__global__ void cast_kernel(const void* x, void* y, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        Vec8<SD>::load(x, i, v);
        Vec8<DD>::store(y, i, v);
    }
}

// ------------------------------------------------------------------ dot / colsum
template <int DT>
__global__ void dot_kernel(const void* a, const void* b, float* out, int64_t n8) {
    float s = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float u[8], v[8];
        Vec8<DT>::load(a, i, u);
        Vec8<DT>::load(b, i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += u[k] * v[k];
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        atomicAdd(out, t);
    }
}

// g = alpha * dy * LeakyReLU'(ref)  and  dot += sum(dy * ref)  in one pass: the backward of `shortcut + gamma * residual`
// (df_gan.py:284) towards the residual branch whose last op is a LeakyReLU (ref = its output), and d(gamma).
template <int DT>
__global__ void scale_mask_dot_kernel(const void* dy, const void* ref, const float* alpha, void* g, float* dot, int64_t n8) {
    const float al = *alpha;
    float s = 0.f;
    for (int64_t base = (int64_t)blockIdx.x * (NT * UN); base < n8; base += (int64_t)gridDim.x * (NT * UN)) {
        float u[UN][8], v[UN][8];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) { Vec8<DT>::load(dy, i, u[j]); Vec8<DT>::load(ref, i, v[j]); }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * NT + threadIdx.x;
            if (i < n8) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += u[j][k] * v[j][k]; o[k] = al * u[j][k] * lrelu_slope(v[j][k]); }
                Vec8<DT>::store(g, i, o);
            }
        }
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        atomicAdd(dot, t);
    }
}

// out[c] += sum_r x[r][c];  thread owns one 8-channel chunk and strides over rows; blockIdx.y walks
// over groups of NT chunks when C/8 > NT (e.g. the 4096-wide proj_noise bias)
template <int DT>
__global__ void colsum_kernel(const void* x, float* out, int64_t rows, int C8) {
    const int c8_base = blockIdx.y * NT;
    const int c8_here = min(C8 - c8_base, NT);
    const int groups = NT / c8_here;                  // row-lanes per block
    const int cl = threadIdx.x % c8_here, g = threadIdx.x / c8_here;
    const int cc = c8_base + cl;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < groups) {
        const int64_t step = (int64_t)gridDim.x * groups;
        int64_t r = (int64_t)blockIdx.x * groups + g;
        for (; r + 3 * step < rows; r += 4 * step) {       // 4 independent 16-byte loads in flight per thread
            float v0[8], v1[8], v2[8], v3[8];
            Vec8<DT>::load(x, r * C8 + cc, v0);
            Vec8<DT>::load(x, (r + step) * C8 + cc, v1);
            Vec8<DT>::load(x, (r + 2 * step) * C8 + cc, v2);
            Vec8<DT>::load(x, (r + 3 * step) * C8 + cc, v3);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (v0[k] + v1[k]) + (v2[k] + v3[k]);
        }
        for (; r < rows; r += step) {
            float v[8];
            Vec8<DT>::load(x, r * C8 + cc, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
    }
    __shared__ float red[NT * 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < c8_here) {
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int gg = 0; gg < groups; ++gg)
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += red[(gg * c8_here + threadIdx.x) * 8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&out[(c8_base + threadIdx.x) * 8 + k], t[k]);
    }
}

// ------------------------------------------------------------------ pooling / resampling
template <int DT>
__global__ void pool2_kernel(const void* x, void* y, int N, int H, int W, int C8, float scale) {
    const int OH = H / 2, OW = W / 2;
    const int64_t total = (int64_t)N * OH * OW * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int64_t p = i / C8;
        int ow = (int)(p % OW); p /= OW;
        int oh = (int)(p % OH);
        int n = (int)(p / OH);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                Vec8<DT>::load(x, (((int64_t)n * H + 2 * oh + dy) * W + 2 * ow + dx) * C8 + cc, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= scale;
        Vec8<DT>::store(y, i, acc);
    }
}
template <int DT>
__global__ void up2_kernel(const void* x, void* y, int N, int H, int W, int C8, float scale) {
    const int OH = H * 2, OW = W * 2;
    const int64_t total = (int64_t)N * OH * OW * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int64_t p = i / C8;
        int ow = (int)(p % OW); p /= OW;
        int oh = (int)(p % OH);
        int n = (int)(p / OH);
        float v[8];
        Vec8<DT>::load(x, (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * C8 + cc, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= scale;
        Vec8<DT>::store(y, i, v);
    }
}
template <int DT, int OD>
__global__ void gap_kernel(const void* x, void* y, int N, int HW, int C8) {
    const int64_t total = (int64_t)N * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int n = (int)(i / C8);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
        for (int p = 0; p < HW; ++p) {
            Vec8<DT>::load(x, ((int64_t)n * HW + p) * C8 + cc, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        const float inv = 1.f / HW;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= inv;
        Vec8<OD>::store(y, i, acc);
    }
}
// Large maps (the concept samplers pool [B,128,128,128] activations, df_concept_gan.py:557): block = (pixel run, image), thread =
// (channel chunk, pixel lane); partial means are added atomically into the zeroed f32 output.  (gap_kernel above walks all of
// H*W in ONE thread per channel chunk: fine for the discriminator's 4x4 maps it was written for, 1.2 ms per call on these.)
template <int DT>
__global__ void gap_big_kernel(const void* x, float* y, int HW, int C8, int pix_per_block) {
    const int n = blockIdx.y, groups = NT / C8;
    const int cc = threadIdx.x % C8, g = threadIdx.x / C8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
    const int p_end = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    if (g < groups)
        for (int p = blockIdx.x * pix_per_block + g; p < p_end; p += groups) {
            Vec8<DT>::load(x, ((size_t)n * HW + p) * C8 + cc, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
    __shared__ float red[NT * 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < C8) {
        const float inv = 1.f / HW;
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int gg = 0; gg < groups; ++gg)
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += red[(gg * C8 + threadIdx.x) * 8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&y[((size_t)n * C8 + threadIdx.x) * 8 + k], t[k] * inv);
    }
}
template <int DT, int ID>
__global__ void gap_bwd_kernel(const void* dy, void* dx, int N, int HW, int C8) {
    const int64_t total = (int64_t)N * HW * C8;
    const float inv = 1.f / HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(i % C8);
        int n = (int)(i / ((int64_t)HW * C8));
        float v[8];
        Vec8<ID>::load(dy, (int64_t)n * C8 + cc, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= inv;
        Vec8<DT>::store(dx, i, v);
    }
}

// ------------------------------------------------------------------ NCHW f32 <-> NHWC8
template <int DT>
__global__ void nchw_to_nhwc8_kernel(const float* src, void* dst, int N, int C, int HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i - n * HW;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < C; ++c) v[c] = src[(n * C + c) * HW + p];
        Vec8<DT>::store(dst, i, v);
    }
}
template <int DT>
__global__ void nhwc8_to_nchw_kernel(const void* src, float* dst, int N, int C, int HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i - n * HW;
        float v[8];
        Vec8<DT>::load(src, i, v);
        for (int c = 0; c < C; ++c) dst[(n * C + c) * HW + p] = v[c];
    }
}

// ------------------------------------------------------------------ fused DF-GAN affine pair
// block = (pixel range, image n); thread owns channel chunk cc = tid % C8 and strides over pixels
template <int DT>
__global__ void affine2_fwd_kernel(const void* x, const float* g0, const float* b0, const float* g1, const float* b1,
                                   void* y, int HW, int C8, int pix_per_block, float slope) {
    const int n = blockIdx.y, groups = NT / C8;
    const int cc = threadIdx.x % C8, g = threadIdx.x / C8;
    if (g >= groups) return;
    float G0[8], B0[8], G1[8], B1[8];
    const size_t pb = ((size_t)n * C8 + cc) * 8;
    const bool two = g1 != nullptr;
#pragma unroll
    for (int k = 0; k < 8; ++k) { G0[k] = g0[pb + k]; B0[k] = b0[pb + k]; G1[k] = two ? g1[pb + k] : 1.f; B1[k] = two ? b1[pb + k] : 0.f; }
    const int p_end = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    for (int p = blockIdx.x * pix_per_block + g; p < p_end; p += groups) {
        float v[8];
        const size_t idx = ((size_t)n * HW + p) * C8 + cc;
        Vec8<DT>::load(x, idx, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float u = v[k] * G0[k] + B0[k];
            u = u > 0.f ? u : slope * u;
            const float t = u * G1[k] + B1[k];
            v[k] = two ? (t > 0.f ? t : slope * t) : u;
        }
        Vec8<DT>::store(y, idx, v);
    }
}
// POOL: the loop runs over vertical pixel PAIRS (HW = (H/2) * W pairs, pair p = (row pair p / W, column p % W)): consecutive
// lanes still walk consecutive pixels of a row (as without POOL), both pixels of a pair are loaded before either is used, and the
// 2x2 sum of the ROUNDED dx -- own pair + the pair of the neighbouring column, one lane exchange -- goes to dx_pool
// [N, H/2, W/2, C]: the 2x2 sum pool of dx, i.e. the gradient of the producing generator block's half-resolution shortcut
// (df_gan.py:200-202), without another pass over dx.  (A first version gave each thread a whole 2x2 quad, one pixel after the
// other: four dependent round trips per step and half-line accesses -- 3.9 ms per iteration for what the pooling pass did in 0.5.)
// Needs C8 a power of two <= 32 (the column neighbour is lane ^ C8), even W, H.
template <int DT, bool POOL>
__global__ __launch_bounds__(NT, 2) void affine2_bwd_kernel(        // two pixels' operands live at once: 128 registers spill
                                   const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                   const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                   int HW, int C8, int pix_per_block, float slope, const void* dx_in,
                                   const float* alpha_dev, float* dot, void* dx_pool, int Wq) {
    // alpha_dev / dot (both optional): dy arrives UNSCALED from a consumer y -> alpha * f(y) (the block sum `shortcut + gamma *
    // c2(y)`, df_gan.py:200-202): dot += <dy, y> with y this node's forward output recomputed here, which is d(alpha) up to the
    // consumer's bias term, and dy is multiplied by alpha before it is used -- so the consumer's output never has to be stored
    // for d(gamma), and gamma = 0 (the reference's initial value) loses nothing.
    const float al = alpha_dev ? *alpha_dev : 1.f;
    float dacc = 0.f;
    const int n = blockIdx.y, groups = NT / C8;
    const int cc = threadIdx.x % C8, g = threadIdx.x / C8;
    float G0[8], B0[8], G1[8], B1[8];
    float sg0[8], sb0[8], sg1[8], sb1[8];
    const size_t pb = ((size_t)n * C8 + cc) * 8;
    const bool two = g1 != nullptr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        G0[k] = g0[pb + k]; B0[k] = b0[pb + k]; G1[k] = two ? g1[pb + k] : 1.f; B1[k] = two ? b1[pb + k] : 0.f;
        sg0[k] = sb0[k] = sg1[k] = sb1[k] = 0.f;
    }
    // one pixel from its loaded operands: dx is stored and (POOL) added, rounded as stored, to acc8
    auto pixel = [&](size_t idx, float (&xv)[8], const float (&dv)[8], const float (&o)[8], float* acc8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float u = xv[k] * G0[k] + B0[k];
            float a1 = u > 0.f ? u : slope * u;
            float v = a1 * G1[k] + B1[k];
            if (dot) {
                const float y2 = two ? (v > 0.f ? v : slope * v) : a1;
                dacc += dv[k] * (DT == XMC_BF16 ? (float)(xmc_h16)y2 : y2);       // the consumer read the stored (rounded) y
            }
            const float dvk = dv[k] * al;
            float dvv = two ? dvk * (v > 0.f ? 1.f : slope) : dvk;
            sg1[k] += dvv * a1; sb1[k] += dvv;
            float du = dvv * G1[k] * (u > 0.f ? 1.f : slope);
            sg0[k] += du * xv[k]; sb0[k] += du;
            xv[k] = du * G0[k];
        }
        if (dx_in) {                  // another gradient of x (the block's shortcut branch) joins here instead of in an add pass
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] += o[k];
        }
        Vec8<DT>::store(dx, idx, xv);
        if (POOL) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc8[k] += DT == XMC_BF16 ? (float)(xmc_h16)xv[k] : xv[k];
        }
    };
    const int p_end = min(HW, (int)(blockIdx.x + 1) * pix_per_block);
    if (g < groups)
        for (int p = blockIdx.x * pix_per_block + g; p < p_end; p += POOL ? groups : 2 * groups) {
            if (!POOL) {
                // two pixels' operands in flight per thread (one pixel per iteration measured 4.1 TB/s where the pooled form, which
                // has always loaded two, reaches 4.8)
                float xv[8], dv[8], o[8], xw[8], dw[8], ow[8];
                const size_t idx = ((size_t)n * HW + p) * C8 + cc;
                const bool second = p + groups < p_end;
                const size_t idx2 = second ? idx + (size_t)groups * C8 : idx;
                Vec8<DT>::load(x, idx, xv);
                Vec8<DT>::load(x, idx2, xw);
                Vec8<DT>::load(dy, idx, dv);
                Vec8<DT>::load(dy, idx2, dw);
                if (dx_in) { Vec8<DT>::load(dx_in, idx, o); Vec8<DT>::load(dx_in, idx2, ow); }
                pixel(idx, xv, dv, o, nullptr);
                if (second) pixel(idx2, xw, dw, ow, nullptr);
            } else {
                const int W = 2 * Wq;
                const int qy = p / W, col = p - qy * W;
                const size_t i0 = (((size_t)n * (HW / W) * 2 + 2 * qy) * W + col) * C8 + cc, i1 = i0 + (size_t)W * C8;
                float x0[8], d0[8], o0[8], x1[8], d1[8], o1[8];
                Vec8<DT>::load(x, i0, x0); Vec8<DT>::load(x, i1, x1);
                Vec8<DT>::load(dy, i0, d0); Vec8<DT>::load(dy, i1, d1);
                if (dx_in) { Vec8<DT>::load(dx_in, i0, o0); Vec8<DT>::load(dx_in, i1, o1); }
                float acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                pixel(i0, x0, d0, o0, acc8);
                pixel(i1, x1, d1, o1, acc8);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc8[k] += __shfl_xor(acc8[k], C8, 64);       // the pair of column col ^ 1: lane ^ C8
                if ((col & 1) == 0) Vec8<DT>::store(dx_pool, (((size_t)n * (HW / W) + qy) * Wq + (col >> 1)) * C8 + cc, acc8);
            }
        }
    if (dot) {
        dacc = wave_sum(dacc);
        if ((threadIdx.x & 63) == 0) atomicAdd(dot, dacc);
    }
    // block reduction over the `groups` pixel lanes that share a channel chunk
    __shared__ float red[NT * 8];
    float* outs[4] = {dg0, db0, dg1, db1};
    float* sums[4] = {sg0, sb0, sg1, sb1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!two && q >= 2) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = sums[q][k];
        __syncthreads();
        if (threadIdx.x < C8) {
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int gg = 0; gg < groups; ++gg)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] += red[(gg * C8 + threadIdx.x) * 8 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&outs[q][((size_t)n * C8 + threadIdx.x) * 8 + k], t[k]);
        }
    }
}

// ------------------------------------------------------------------ hinge
template <int DT>
__global__ void hinge_fwd_kernel(const void* x, int stride, float sign, float* out, int64_t n) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = DT == XMC_BF16 ? (float)reinterpret_cast<const xmc_h16*>(x)[i * stride] : reinterpret_cast<const float*>(x)[i * stride];
        s += fmaxf(1.f + sign * v, 0.f);
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        *out = t / (float)n;
    }
}
template <int DT>
__global__ void hinge_bwd_kernel(const void* x, int stride, float sign, const float* dloss, void* dx, int64_t n) {
    const float gscale = (*dloss) * sign / (float)n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = DT == XMC_BF16 ? (float)reinterpret_cast<const xmc_h16*>(x)[i * stride] : reinterpret_cast<const float*>(x)[i * stride];
        float gval = (1.f + sign * v) > 0.f ? gscale : 0.f;
        if (DT == XMC_BF16) reinterpret_cast<xmc_h16*>(dx)[i * stride] = (xmc_h16)gval;
        else reinterpret_cast<float*>(dx)[i * stride] = gval;
    }
}

// ------------------------------------------------------------------ weight pack / unpack
// value of packed element i of [KHW][rows_pad][cols_pad].  groups > 1: w is a grouped-convolution weight [Co][Ci/groups][KHW];
// the packed matrix is its block-diagonal expansion
__device__ __forceinline__ float pack_value(const float* __restrict__ w, int Co, int Ci, int KHW, int rows_pad, int cols_pad,
                                            int transpose, const int32_t* __restrict__ row_perm, int groups, int64_t i) {
    const int cog = Co / groups, cig = Ci / groups;
    int c = (int)(i % cols_pad);
    int64_t q = i / cols_pad;
    int r = (int)(q % rows_pad);
    int t = (int)(q / rows_pad);
    int co = transpose ? c : r, ci = transpose ? r : c;
    float v = 0.f;
    if (co < Co && ci < Ci) {
        int sco = row_perm ? row_perm[co] : co;
        if (groups == 1) v = w[((int64_t)sco * Ci + ci) * KHW + t];
        else if (sco / cog == ci / cig) v = w[((int64_t)sco * cig + ci % cig) * KHW + t];
    }
    return v;
}
// Fused nearest-x2 upsample + 3x3 convolution: output parity class (i,j) sees only a 2x2 neighbourhood of the
// low-resolution input, with weights that are sums of the 3x3 taps landing on the same low-res pixel:
//   rows: i=0: {kh=0} | {kh=1,2}     i=1: {kh=0,1} | {kh=2}        (same for columns)
// slice index = (i*2+j)*4 + th*2+tw.  Summation in f32, one rounding to the kernel dtype.
__device__ __forceinline__ float pack_upconv_value(const float* __restrict__ w, int Co, int Ci, int rows_pad, int cols_pad,
                                                   int transpose, int64_t idx) {
    int c = (int)(idx % cols_pad);
    int64_t q = idx / cols_pad;
    int r = (int)(q % rows_pad);
    int sl = (int)(q / rows_pad);
    int cls = sl >> 2, t = sl & 3;
    int i = cls >> 1, j = cls & 1, th = t >> 1, tw = t & 1;
    int co = transpose ? c : r, ci = transpose ? r : c;
    float v = 0.f;
    if (co < Co && ci < Ci) {
        int kh0 = (i == 0) ? (th == 0 ? 0 : 1) : (th == 0 ? 0 : 2), kh1 = (i == 0) ? (th == 0 ? 0 : 2) : (th == 0 ? 1 : 2);
        int kw0 = (j == 0) ? (tw == 0 ? 0 : 1) : (tw == 0 ? 0 : 2), kw1 = (j == 0) ? (tw == 0 ? 0 : 2) : (tw == 0 ? 1 : 2);
        const float* wp = w + ((int64_t)co * Ci + ci) * 9;
        for (int kh = kh0; kh <= kh1; ++kh)
            for (int kw = kw0; kw <= kw1; ++kw) v += wp[kh * 3 + kw];
    }
    return v;
}
template <int DT>
__global__ void pack_weight_kernel(const float* w, void* wpk, int Co, int Ci, int KHW, int rows_pad, int cols_pad,
                                   int transpose, const int32_t* row_perm, int groups) {
    const int64_t total = (int64_t)KHW * rows_pad * cols_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = pack_value(w, Co, Ci, KHW, rows_pad, cols_pad, transpose, row_perm, groups, i);
        if (DT == XMC_BF16) reinterpret_cast<xmc_h16*>(wpk)[i] = (xmc_h16)v;
        else reinterpret_cast<float*>(wpk)[i] = v;
    }
}
template <int DT>
__global__ void pack_upconv_kernel(const float* w, void* wpk, int Co, int Ci, int rows_pad, int cols_pad, int transpose) {
    const int64_t total = (int64_t)16 * rows_pad * cols_pad;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const float v = pack_upconv_value(w, Co, Ci, rows_pad, cols_pad, transpose, idx);
        if (DT == XMC_BF16) reinterpret_cast<xmc_h16*>(wpk)[idx] = (xmc_h16)v;
        else reinterpret_cast<float*>(wpk)[idx] = v;
    }
}
// All packed copies of a network's weights in one launch (after the optimizer step that changed them).  The job table travels
// in the kernel arguments (capturable, no device-side table).  Work is cut into chunks of PM_CHUNK (row, column) positions of
// the packed matrix; first[j] = first chunk of job j, so blocks are dealt in proportion to the size of a job.  A thread owns one
// (row, column) position for ALL taps: its KHW source values are contiguous (a wave reads one contiguous piece of the weight),
// and each tap's store is contiguous across the wave.
constexpr int PM_CHUNK = 2048;
struct PackJobs { XmcPackJob j[XMC_PACK_MULTI_MAX]; int32_t first[XMC_PACK_MULTI_MAX + 1]; int32_t njobs; };
template <int DT>
__device__ __forceinline__ void pack_store(void* p, int64_t i, float v, bool lo = false) {
    if (DT == XMC_BF16) reinterpret_cast<xmc_h16*>(p)[i] = lo ? (xmc_h16)(v - (float)(xmc_h16)v) : (xmc_h16)v;      // XmcPackJob.lo
    else reinterpret_cast<float*>(p)[i] = v;
}
template <int DT>
__device__ __forceinline__ void pack_position(const XmcPackJob& job, int pos) {
    const int c = pos % job.cols_pad, r = pos / job.cols_pad;
    const int co = job.transpose ? c : r, ci = job.transpose ? r : c;
    const int64_t plane = (int64_t)job.rows_pad * job.cols_pad;
    const bool in = co < job.Co && ci < job.Ci;
    if (job.upconv) {
        float w9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[t] = in ? job.w[((int64_t)co * job.Ci + ci) * 9 + t] : 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {      // slice (i*2+j)*4 + th*2+tw: rows i=0: {0} | {1,2}, i=1: {0,1} | {2}; same for columns
            const int i = sl >> 3, j = (sl >> 2) & 1, th = (sl >> 1) & 1, tw = sl & 1;
            const int kh0 = (i == 0) ? (th == 0 ? 0 : 1) : (th == 0 ? 0 : 2), kh1 = (i == 0) ? (th == 0 ? 0 : 2) : (th == 0 ? 1 : 2);
            const int kw0 = (j == 0) ? (tw == 0 ? 0 : 1) : (tw == 0 ? 0 : 2), kw1 = (j == 0) ? (tw == 0 ? 0 : 2) : (tw == 0 ? 1 : 2);
            float v = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (kh >= kh0 && kh <= kh1 && kw >= kw0 && kw <= kw1) v += w9[kh * 3 + kw];
            pack_store<DT>(job.wpk, sl * plane + pos, v);
        }
        return;
    }
    const int cog = job.Co / job.groups, cig = job.Ci / job.groups;
    const float* src = nullptr;
    if (in) {
        const int sco = job.row_perm ? job.row_perm[co] : co;
        if (job.groups == 1) src = job.w + ((int64_t)sco * job.Ci + ci) * job.KHW;
        else if (sco / cog == ci / cig) src = job.w + ((int64_t)sco * cig + ci % cig) * job.KHW;
    }
    for (int t = 0; t < job.KHW; ++t) pack_store<DT>(job.wpk, t * plane + pos, src ? src[t] : 0.f, job.lo != 0);
}
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobs J) {
    int jb = 0;
    while (jb + 1 < J.njobs && (int)blockIdx.x >= J.first[jb + 1]) ++jb;      // uniform; <= 47 steps
    const XmcPackJob& job = J.j[jb];
    const int total = job.rows_pad * job.cols_pad;
    const int p0 = ((int)blockIdx.x - J.first[jb]) * PM_CHUNK;
    const int p1 = p0 + PM_CHUNK < total ? p0 + PM_CHUNK : total;
    if (job.dtype == XMC_BF16) for (int pos = p0 + threadIdx.x; pos < p1; pos += 256) pack_position<XMC_BF16>(job, pos);
    else for (int pos = p0 + threadIdx.x; pos < p1; pos += 256) pack_position<XMC_F32>(job, pos);
}
// y[n,2h+i,2w+j,:] = a[n,h,w,:] + alpha * b[n,2h+i,2w+j,:]      (up(shortcut) + gamma*residual without materialising up())
template <int DT>
__global__ void axpby_up_kernel(const void* a, const void* b, const float* alpha, void* y, int N, int H, int W, int C8, int lrelu) {
    const float al = *alpha;
    const int OH = 2 * H, OW = 2 * W;
    const int64_t total = (int64_t)N * OH * OW * C8;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int cc = (int)(idx % C8);
        int64_t p = idx / C8;
        int ow = (int)(p % OW); p /= OW;
        int oh = (int)(p % OH);
        int n = (int)(p / OH);
        float u[8], v[8];
        Vec8<DT>::load(a, (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * C8 + cc, u);
        Vec8<DT>::load(b, idx, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] += al * v[k];
        if (lrelu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = lrelu_f(u[k]);
        }
        Vec8<DT>::store(y, idx, u);
    }
}

// Backward of y = up2(a) + alpha*b (and of y = a + alpha*b when up == 0) in one pass over dy and b:
//   db = alpha * dy,   da = 2x2 sum pool of dy (up) / nothing (plain: da is dy itself),   dot += <dy, b>  (= d alpha)
// instead of a scale pass, a pool pass and a dot pass (reads of dy: 3 -> 1).
template <int DT>
__global__ void axpby_bwd_kernel(const void* dy, const void* b, const float* alpha, void* db, void* da, float* dot,
                                 int N, int H, int W, int C8, int up, const void* ymask) {
    const float al = *alpha;
    float s = 0.f;
    const int64_t total = (int64_t)N * H * W * C8;                  // low-resolution positions (x channel units)
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        if (!up) {
            float u[8], v[8], o[8];
            Vec8<DT>::load(dy, idx, u);
            Vec8<DT>::load(b, idx, v);
            if (ymask) {                              // y = LeakyReLU(a + alpha*b): dy <- dy * LeakyReLU'(y)
                float mk[8];
                Vec8<DT>::load(ymask, idx, mk);
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] *= lrelu_slope(mk[k]);
                Vec8<DT>::store(da, idx, u);          // plain form with a mask: da is the masked dy
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += u[k] * v[k]; o[k] = al * u[k]; }
            if (db) Vec8<DT>::store(db, idx, o);
            continue;
        }
        const int cc = (int)(idx % C8);
        int64_t p = idx / C8;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t hi = ((((int64_t)n * 2 * H + 2 * h + i) * 2 * W) + 2 * w + j) * C8 + cc;
                float u[8], v[8], o[8];
                Vec8<DT>::load(dy, hi, u);
                Vec8<DT>::load(b, hi, v);
                if (ymask) {
                    float mk[8];
                    Vec8<DT>::load(ymask, hi, mk);
#pragma unroll
                    for (int k = 0; k < 8; ++k) u[k] *= lrelu_slope(mk[k]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += u[k] * v[k]; o[k] = al * u[k]; sum[k] += u[k]; }
                if (db) Vec8<DT>::store(db, hi, o);       // db == NULL: the consumers of alpha*dy apply alpha themselves
            }
        Vec8<DT>::store(da, idx, sum);
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w_ = 0; w_ < NT / 64; ++w_) t += part[w_];
        atomicAdd(dot, t);
    }
}

__global__ void unpack_wgrad_kernel(const float* dwp, float* gw, int Co, int Ci, int KHW, int rows_pad, int cols_pad,
                                    const float* scale_dev, const int32_t* row_perm, int accumulate,
                                    const float* gb_rep, float* gb, int CDb, int groups,
                                    const float* bias_dot, float* dot) {
    // groups > 1: gw is a grouped-convolution weight gradient [Co][Ci/groups][KHW] = the diagonal blocks of the dense one
    const float scale = scale_dev ? *scale_dev : 1.f;
    const int cog = Co / groups, cig = Ci / groups;
    const int64_t total = (int64_t)Co * cig * KHW;
    // bias gradient: sum of the XMC_BIAS_REPLICAS partial column sums the weight-gradient kernels accumulate into
    // (packed channel order, like the rows of dwp); rides along instead of a separate reduction launch
    if (gb_rep) {
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < CDb; c += gridDim.x * blockDim.x) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < XMC_BIAS_REPLICAS; ++r) sacc += gb_rep[r * CDb + c];
            gb[c] = scale * sacc;
            // <bias, UNSCALED bias gradient>: the bias term of d(alpha) for y = alpha * (conv(x) + bias) (see affine2_bwd_kernel)
            if (bias_dot && c < Co) atomicAdd(dot, bias_dot[c] * sacc);
        }
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % KHW);
        int64_t q = i / KHW;
        int ci = (int)(q % cig);
        int r = (int)(q / cig);             // packed row r holds parameter row row_perm[r]
        int dco = row_perm ? row_perm[r] : r;
        float v = scale * dwp[((int64_t)t * rows_pad + r) * cols_pad + (dco / cog) * cig + ci];
        int64_t o = ((int64_t)dco * cig + ci) * KHW + t;
        gw[o] = accumulate ? gw[o] + v : v;
    }
}

// ------------------------------------------------------------------ matching-aware gradient penalty (train_gan.py:241-247)
// ss[b] += sum_k x[b][k]^2 : f32 rows of `cols` elements (cols % 4 == 0), blockIdx.y = row, blockIdx.x strides over the row
__global__ void rows_sumsq_kernel(const float* __restrict__ x, float* __restrict__ ss, int64_t cols4) {
    const f32x4* row = reinterpret_cast<const f32x4*>(x) + (int64_t)blockIdx.y * cols4;
    float s = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cols4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = row[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        atomicAdd(&ss[blockIdx.y], t);
    }
}
// gp = mean_b ss[b]^3 (= mean ||g_b||^6), coef[b] = d gp / d ss[b] * 2 = 6 ss[b]^2 / B   (so that d gp / d g = coef[b] * g)
// inv_s2: the blocks hold s * g (the IEEE-half mode runs the inner backward of MA-GP on s x ones): ss / s^2 is the true sum of
// squares and d gp / d (s g) = coef / s^2 * (s g).  1 otherwise.
__global__ void gp_finish_kernel(const float* __restrict__ ss, int B, float* __restrict__ gp, float* __restrict__ coef, float inv_s2) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float v = ss[b] * inv_s2;
        s += v * v * v;
        coef[b] = 6.f * v * v / (float)B * inv_s2;
    }
    s = wave_sum(s);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        *gp = t / (float)B;
    }
}
// y[b][k] = (*g) * coef[b] * x[b][k]
__global__ void rows_scale_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ g,
                                  float* __restrict__ y, int64_t cols4) {
    const float c = coef[blockIdx.y] * (*g);
    const f32x4* row = reinterpret_cast<const f32x4*>(x) + (int64_t)blockIdx.y * cols4;
    f32x4* out = reinterpret_cast<f32x4*>(y) + (int64_t)blockIdx.y * cols4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cols4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = row[i];
        v[0] *= c; v[1] *= c; v[2] *= c; v[3] *= c;
        out[i] = v;
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int xmc_abi_version(void) { return XMC_ABI_VERSION; }
extern "C" int xmc_half_format(void) { return XMC_HALF_FORMAT; }

static thread_local char g_last_kernel[96] = "";
void xmc_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof g_last_kernel, fmt, ap);
    va_end(ap);
}
extern "C" const char* xmc_last_kernel(void) { return g_last_kernel; }

// XMC_DEBUG_DISPATCH=log_generic_epi: one line on stderr per (kernel, epilogue option mask) that ran through a descriptor-reading
// epilogue although its options are expressible as a mask -- the list of instantiations still worth adding (common.h: kEpi*)
void xmc_note_generic_epi(const char* kernel, int mask) {
    static const bool on = xmc_debug_off("log_generic_epi");
    if (!on || mask < 0) return;
    static char seen[64][96];
    static int nseen = 0;
    char key[96];
    snprintf(key, sizeof key, "%s epi=%d", kernel, mask);
    for (int i = 0; i < nseen; ++i)
        if (!strcmp(seen[i], key)) return;
    if (nseen < 64) strcpy(seen[nseen++], key);
    fprintf(stderr, "[xmc] generic epilogue: %s (bias %d lrelu %d round %d dst2 %d alpha %d mask %d res %d post %d pool %d)\n", key, mask & 1, (mask >> 1) & 1,
            (mask >> 2) & 1, (mask >> 3) & 1, (mask >> 4) & 1, (mask >> 5) & 1, (mask >> 6) & 1, (mask >> 7) & 1, (mask >> 8) & 1);
}

// One debugging switch for the kernel dispatchers: XMC_DEBUG_DISPATCH="tok1,tok2,..." disables the named specialised kernels
// (the dispatcher then falls through to the next, more general one).  Unset in production: every call returns false.
// Fixed-order reductions (test mode, xmc_set_fixed_order): the reductions whose results feed ACTIVATIONS -- the GroupNorm statistics and
// the attention logits' query gradient -- run with ONE workgroup per reduction target, so every f32 sum is formed in one order and an
// iteration is repeatable bit for bit up to the parameter-gradient atomics (which feed nothing inside the iteration).
static int g_fixed_order = 0;
extern "C" int xmc_set_fixed_order(int on) { const int was = g_fixed_order; g_fixed_order = on ? 1 : 0; return was; }
bool xmc_fixed_order() { return g_fixed_order != 0; }
// the caller's promise that "zeroed here" accumulators arrive zero (include/xmc_gan_hip.h): xmc_zero_acc is the one place that memsets them
static int g_prezeroed = 0;
extern "C" int xmc_set_prezeroed(int on) { const int was = g_prezeroed; g_prezeroed = on ? 1 : 0; return was; }
hipError_t xmc_zero_acc(void* p, size_t bytes, hipStream_t st) { return g_prezeroed ? hipSuccess : hipMemsetAsync(p, 0, bytes, st); }

bool xmc_debug_off(const char* token) {
    static const char* env = getenv("XMC_DEBUG_DISPATCH");
    if (!env || !*env) return false;
    const size_t n = strlen(token);
    for (const char* p = env; (p = strstr(p, token)) != nullptr; p += n)
        if ((p == env || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}

extern "C" int xmc_lrelu(const void* x, void* y, int64_t n, float slope, int dtype, void* s) { return run_map1(x, y, n, dtype, ST(s), FLrelu{slope}); }
extern "C" int xmc_tanh(const void* x, void* y, int64_t n, int dtype, void* s) { return run_map1(x, y, n, dtype, ST(s), FTanh{}); }
extern "C" int xmc_lrelu_mask(const void* dy, const void* ref, void* dx, int64_t n, float slope, int dtype, void* s) {
    return run_map2(dy, ref, dx, n, dtype, ST(s), FLreluMask{slope});
}
extern "C" int xmc_signmask_apply(const void* dy, const void* bits, void* dx, int64_t n, float slope, int dtype, void* s) {
    if (!dy || !bits || !dx) return XMC_EINVAL;
    if (n % 8) return XMC_EALIGN;
    if (n == 0) return 0;
    const int64_t n8 = n / 8;
    if (dtype == XMC_BF16) hipLaunchKernelGGL((signmask_kernel<XMC_BF16>), dim3(sblocks(n8)), dim3(NT), 0, ST(s), dy, (const unsigned char*)bits, dx, n8, slope);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((signmask_kernel<XMC_F32>), dim3(sblocks(n8)), dim3(NT), 0, ST(s), dy, (const unsigned char*)bits, dx, n8, slope);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* s) {
    return run_map2(dy, y, dx, n, dtype, ST(s), FTanhBwd{});
}
extern "C" int xmc_axpby(const void* a, const void* b, const float* alpha, void* y, int64_t n, int dtype, void* s) {
    if (!alpha) return XMC_EINVAL;
    return run_map2(a, b, y, n, dtype, ST(s), FAxpby{alpha});
}
extern "C" int xmc_scale(const void* x, const float* alpha, void* y, int64_t n, int dtype, void* s) {
    if (!alpha) return XMC_EINVAL;
    return run_map1(x, y, n, dtype, ST(s), FScale{alpha});
}
extern "C" int xmc_cast(const void* x, void* y, int64_t n, int src_dtype, int dst_dtype, void* s) {
    if (n % 8) return XMC_EALIGN;
    if (n == 0) return 0;
    int64_t n8 = n / 8;
    dim3 g(nblocks(n8)), b(NT);
    if (src_dtype == XMC_F32 && dst_dtype == XMC_BF16) hipLaunchKernelGGL((cast_kernel<XMC_F32, XMC_BF16>), g, b, 0, ST(s), x, y, n8);
    else if (src_dtype == XMC_BF16 && dst_dtype == XMC_F32) hipLaunchKernelGGL((cast_kernel<XMC_BF16, XMC_F32>), g, b, 0, ST(s), x, y, n8);
    else if (src_dtype == XMC_F32 && dst_dtype == XMC_F32) hipLaunchKernelGGL((cast_kernel<XMC_F32, XMC_F32>), g, b, 0, ST(s), x, y, n8);
    else if (src_dtype == XMC_BF16 && dst_dtype == XMC_BF16) hipLaunchKernelGGL((cast_kernel<XMC_BF16, XMC_BF16>), g, b, 0, ST(s), x, y, n8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_dot(const void* a, const void* b, float* out, int64_t n, int dtype, void* s) {
    if (n % 8) return XMC_EALIGN;
    int64_t n8 = n / 8;
    dim3 g(nblocks(n8, NT, 512)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((dot_kernel<XMC_BF16>), g, blk, 0, ST(s), a, b, out, n8);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((dot_kernel<XMC_F32>), g, blk, 0, ST(s), a, b, out, n8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_scale_mask_dot(const void* dy, const void* ref, const float* alpha, void* g, float* dot, int64_t n, int dtype, void* s) {
    if (n % 8) return XMC_EALIGN;
    int64_t n8 = n / 8;
    dim3 grd(sblocks(n8)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((scale_mask_dot_kernel<XMC_BF16>), grd, blk, 0, ST(s), dy, ref, alpha, g, dot, n8);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((scale_mask_dot_kernel<XMC_F32>), grd, blk, 0, ST(s), dy, ref, alpha, g, dot, n8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_colsum(const void* x, float* out, int64_t rows, int C, int dtype, void* s) {
    if (C % 8) return XMC_EALIGN;
    const int C8 = C / 8, groups = C8 >= NT ? 1 : NT / C8;
    dim3 g(nblocks(rows, groups * 8, 1024), (C8 + NT - 1) / NT), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((colsum_kernel<XMC_BF16>), g, blk, 0, ST(s), x, out, rows, C8);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((colsum_kernel<XMC_F32>), g, blk, 0, ST(s), x, out, rows, C8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
static int pool2(const void* x, void* y, int N, int H, int W, int C, float scale, int dtype, void* s) {
    if (C % 8 || H % 2 || W % 2) return XMC_EALIGN;
    int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    dim3 g(nblocks(total)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((pool2_kernel<XMC_BF16>), g, blk, 0, ST(s), x, y, N, H, W, C / 8, scale);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((pool2_kernel<XMC_F32>), g, blk, 0, ST(s), x, y, N, H, W, C / 8, scale);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_avgpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* s) {
    return pool2(x, y, N, H, W, C, 0.25f, dtype, s);
}
extern "C" int xmc_sumpool2(const void* x, void* y, int N, int H, int W, int C, float scale, int dtype, void* s) {
    return pool2(x, y, N, H, W, C, scale, dtype, s);
}
extern "C" int xmc_upsample2(const void* x, void* y, int N, int H, int W, int C, float scale, int dtype, void* s) {
    if (C % 8) return XMC_EALIGN;
    int64_t total = (int64_t)N * H * 2 * W * 2 * (C / 8);
    dim3 g(nblocks(total)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((up2_kernel<XMC_BF16>), g, blk, 0, ST(s), x, y, N, H, W, C / 8, scale);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((up2_kernel<XMC_F32>), g, blk, 0, ST(s), x, y, N, H, W, C / 8, scale);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_global_avgpool(const void* x, void* y, int N, int HW, int C, int dtype, int out_dtype, void* s) {
    if (C % 8) return XMC_EALIGN;
    if (HW >= 256 && out_dtype == XMC_F32 && C / 8 <= NT && (dtype == XMC_BF16 || dtype == XMC_F32)) {
        const int C8 = C / 8, groups = NT / C8;
        int64_t ppb = ((int64_t)N * HW + 2047) / 2048;          // ~2048 workgroups over the batch, >= 4 pixels per lane
        if (ppb < 4 * groups) ppb = 4 * groups;
        const int bx = (int)((HW + ppb - 1) / ppb);
        if (xmc_zero_acc(y, sizeof(float) * (size_t)N * C, ST(s)) != hipSuccess) return XMC_EINVAL;
        if (dtype == XMC_BF16) hipLaunchKernelGGL((gap_big_kernel<XMC_BF16>), dim3(bx, N), dim3(NT), 0, ST(s), x, (float*)y, HW, C8, (int)ppb);
        else hipLaunchKernelGGL((gap_big_kernel<XMC_F32>), dim3(bx, N), dim3(NT), 0, ST(s), x, (float*)y, HW, C8, (int)ppb);
        XMC_LAUNCH_CHECK();
        return 0;
    }
    dim3 g(nblocks((int64_t)N * (C / 8))), blk(NT);
    if (dtype == XMC_BF16 && out_dtype == XMC_F32) hipLaunchKernelGGL((gap_kernel<XMC_BF16, XMC_F32>), g, blk, 0, ST(s), x, y, N, HW, C / 8);
    else if (dtype == XMC_BF16 && out_dtype == XMC_BF16) hipLaunchKernelGGL((gap_kernel<XMC_BF16, XMC_BF16>), g, blk, 0, ST(s), x, y, N, HW, C / 8);
    else if (dtype == XMC_F32 && out_dtype == XMC_F32) hipLaunchKernelGGL((gap_kernel<XMC_F32, XMC_F32>), g, blk, 0, ST(s), x, y, N, HW, C / 8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_global_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, int in_dtype, void* s) {
    if (C % 8) return XMC_EALIGN;
    dim3 g(nblocks((int64_t)N * HW * (C / 8))), blk(NT);
    if (dtype == XMC_BF16 && in_dtype == XMC_F32) hipLaunchKernelGGL((gap_bwd_kernel<XMC_BF16, XMC_F32>), g, blk, 0, ST(s), dy, dx, N, HW, C / 8);
    else if (dtype == XMC_BF16 && in_dtype == XMC_BF16) hipLaunchKernelGGL((gap_bwd_kernel<XMC_BF16, XMC_BF16>), g, blk, 0, ST(s), dy, dx, N, HW, C / 8);
    else if (dtype == XMC_F32 && in_dtype == XMC_F32) hipLaunchKernelGGL((gap_bwd_kernel<XMC_F32, XMC_F32>), g, blk, 0, ST(s), dy, dx, N, HW, C / 8);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_nchw_to_nhwc8(const float* src, void* dst, int N, int C, int H, int W, int dtype, void* s) {
    if (C < 1 || C > 8) return XMC_ESHAPE;
    dim3 g(nblocks((int64_t)N * H * W)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((nchw_to_nhwc8_kernel<XMC_BF16>), g, blk, 0, ST(s), src, dst, N, C, H * W);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((nchw_to_nhwc8_kernel<XMC_F32>), g, blk, 0, ST(s), src, dst, N, C, H * W);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_nhwc8_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int dtype, void* s) {
    if (C < 1 || C > 8) return XMC_ESHAPE;
    dim3 g(nblocks((int64_t)N * H * W)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((nhwc8_to_nchw_kernel<XMC_BF16>), g, blk, 0, ST(s), src, dst, N, C, H * W);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((nhwc8_to_nchw_kernel<XMC_F32>), g, blk, 0, ST(s), src, dst, N, C, H * W);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}

// ppl = pixels per thread lane: 16 for the forward; 64 for the backward, whose workgroups end in an LDS reduction and 4*C atomics
// (N512 128x128x64: 637 -> 612 us, N256 256x256x32: 689 -> 606)
static inline int affine_grid(int HW, int C8, int N, dim3& g, int& ppb, int ppl) {
    const int groups = NT / C8;
    // small maps of a small batch (64 images of 32x32: one workgroup per image, 32 dependent load-compute-store rounds each -- 25 us for
    // 50 MB): fewer pixels per lane until the launch has a workgroup per CU
    static const bool wide = xmc_debug_off("affine_wide");
    while (!wide && ppl > 4 && (long long)N * ((HW + groups * ppl - 1) / (groups * ppl)) < 256) ppl >>= 1;
    int per = groups * ppl;
    int bx = (HW + per - 1) / per;
    if (bx < 1) bx = 1;
    ppb = (HW + bx - 1) / bx;
    g = dim3(bx, N);
    return 0;
}
extern "C" int xmc_affine2_act_fwd(const void* x, const float* g0, const float* b0, const float* g1, const float* b1,
                                   void* y, int N, int HW, int C, float slope, int dtype, void* s) {
    if (C % 8 || C / 8 > NT) return XMC_EALIGN;
    dim3 g; int ppb;
    affine_grid(HW, C / 8, N, g, ppb, 16);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((affine2_fwd_kernel<XMC_BF16>), g, dim3(NT), 0, ST(s), x, g0, b0, g1, b1, y, HW, C / 8, ppb, slope);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((affine2_fwd_kernel<XMC_F32>), g, dim3(NT), 0, ST(s), x, g0, b0, g1, b1, y, HW, C / 8, ppb, slope);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_affine2_act_bwd_dot_pool(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                            const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                            const void* dx_in, const float* alpha_dev, float* dot, void* dx_pool, int N, int H, int W,
                                            int C, float slope, int dtype, void* s) {
    if (C % 8 || C / 8 > NT) return XMC_EALIGN;
    if ((alpha_dev == nullptr) != (dot == nullptr)) return XMC_EINVAL;
    const int C8 = C / 8;
    if (dx_pool && ((H & 1) || (W & 1) || (C8 & (C8 - 1)) || C8 > 32)) return XMC_ESHAPE;
    const int HW = dx_pool ? (H / 2) * W : H * W;               // loop units: vertical pixel pairs or pixels
    dim3 g; int ppb;
    affine_grid(HW, C / 8, N, g, ppb, dx_pool ? 32 : 64);
    if (dx_pool && (ppb & 1)) {                                  // column neighbours (p, p ^ 1) must meet in one workgroup step
        ppb += 1;
        g.x = (HW + ppb - 1) / ppb;
    }
#define XMC_AFF_BWD(DTC, PL) hipLaunchKernelGGL((affine2_bwd_kernel<DTC, PL>), g, dim3(NT), 0, ST(s), x, dy, g0, b0, g1, b1, dx, dg0, db0, dg1, db1, \
                                                HW, C / 8, ppb, slope, dx_in, alpha_dev, dot, dx_pool, W / 2)
    if (dtype == XMC_BF16) { if (dx_pool) XMC_AFF_BWD(XMC_BF16, true); else XMC_AFF_BWD(XMC_BF16, false); }
    else if (dtype == XMC_F32) { if (dx_pool) XMC_AFF_BWD(XMC_F32, true); else XMC_AFF_BWD(XMC_F32, false); }
    else return XMC_EINVAL;
#undef XMC_AFF_BWD
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_affine2_act_bwd_dot(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                       const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                       const void* dx_in, const float* alpha_dev, float* dot, int N, int HW, int C, float slope,
                                       int dtype, void* s) {
    return xmc_affine2_act_bwd_dot_pool(x, dy, g0, b0, g1, b1, dx, dg0, db0, dg1, db1, dx_in, alpha_dev, dot, nullptr, N, 1, HW, C, slope, dtype, s);
}
extern "C" int xmc_affine2_act_bwd_acc(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                       const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                       const void* dx_in, int N, int HW, int C, float slope, int dtype, void* s) {
    return xmc_affine2_act_bwd_dot(x, dy, g0, b0, g1, b1, dx, dg0, db0, dg1, db1, dx_in, nullptr, nullptr, N, HW, C, slope, dtype, s);
}
extern "C" int xmc_affine2_act_bwd(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                   const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                   int N, int HW, int C, float slope, int dtype, void* s) {
    return xmc_affine2_act_bwd_acc(x, dy, g0, b0, g1, b1, dx, dg0, db0, dg1, db1, nullptr, N, HW, C, slope, dtype, s);
}
extern "C" int xmc_affine2_lrelu_fwd(const void* x, const float* g0, const float* b0, const float* g1, const float* b1,
                                     void* y, int N, int HW, int C, int dtype, void* s) {
    return xmc_affine2_act_fwd(x, g0, b0, g1, b1, y, N, HW, C, XMC_LRELU, dtype, s);
}
extern "C" int xmc_affine2_lrelu_bwd(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                     const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                                     int N, int HW, int C, int dtype, void* s) {
    return xmc_affine2_act_bwd(x, dy, g0, b0, g1, b1, dx, dg0, db0, dg1, db1, N, HW, C, XMC_LRELU, dtype, s);
}
extern "C" int xmc_hinge_fwd(const void* x, int stride, float sign, float* out, int64_t n, int dtype, void* s) {
    if (n < 1 || stride < 1) return XMC_ESHAPE;
    if (dtype == XMC_BF16) hipLaunchKernelGGL((hinge_fwd_kernel<XMC_BF16>), dim3(1), dim3(NT), 0, ST(s), x, stride, sign, out, n);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((hinge_fwd_kernel<XMC_F32>), dim3(1), dim3(NT), 0, ST(s), x, stride, sign, out, n);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_hinge_bwd(const void* x, int stride, float sign, const float* dloss, void* dx, int64_t n, int dtype, void* s) {
    if (n < 1 || stride < 1) return XMC_ESHAPE;
    dim3 g(nblocks(n)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((hinge_bwd_kernel<XMC_BF16>), g, blk, 0, ST(s), x, stride, sign, dloss, dx, n);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((hinge_bwd_kernel<XMC_F32>), g, blk, 0, ST(s), x, stride, sign, dloss, dx, n);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_pack_weight_grouped(const float* w, void* wpk, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                                       int transpose, int dtype, const int32_t* row_perm, int groups, void* s) {
    if (!w || !wpk || groups < 1 || Co % groups || Ci % groups) return XMC_EINVAL;
    if (transpose ? (rows_pad < Ci || cols_pad < Co) : (rows_pad < Co || cols_pad < Ci)) return XMC_ESHAPE;
    int64_t total = (int64_t)KH * KW * rows_pad * cols_pad;
    dim3 g(nblocks(total)), blk(NT);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<XMC_BF16>), g, blk, 0, ST(s), w, wpk, Co, Ci, KH * KW, rows_pad, cols_pad, transpose, row_perm, groups);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((pack_weight_kernel<XMC_F32>), g, blk, 0, ST(s), w, wpk, Co, Ci, KH * KW, rows_pad, cols_pad, transpose, row_perm, groups);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_pack_weight_multi(const XmcPackJob* jobs, int njobs, void* s) {
    if (!jobs || njobs < 0) return XMC_EINVAL;
    for (int j0 = 0; j0 < njobs; j0 += XMC_PACK_MULTI_MAX) {
        PackJobs J;
        const int n = njobs - j0 < XMC_PACK_MULTI_MAX ? njobs - j0 : XMC_PACK_MULTI_MAX;
        int nblk = 0;
        for (int k = 0; k < n; ++k) {
            const XmcPackJob& q = jobs[j0 + k];
            if (!q.w || !q.wpk || (q.dtype != XMC_BF16 && q.dtype != XMC_F32) || q.groups < 1 || q.Co % q.groups || q.Ci % q.groups ||
                q.rows_pad < 1 || q.cols_pad < 1 || q.KHW < 1 || (q.upconv && (q.groups != 1 || q.row_perm)))
                return XMC_EINVAL;
            if (q.transpose ? (q.rows_pad < q.Ci || q.cols_pad < q.Co) : (q.rows_pad < q.Co || q.cols_pad < q.Ci)) return XMC_ESHAPE;
            if ((int64_t)q.rows_pad * q.cols_pad >= (1ll << 30)) return XMC_ESHAPE;
            J.j[k] = q;
            J.first[k] = nblk;
            nblk += (q.rows_pad * q.cols_pad + PM_CHUNK - 1) / PM_CHUNK;
        }
        J.first[n] = nblk;
        J.njobs = n;
        hipLaunchKernelGGL(pack_multi_kernel, dim3(nblk), dim3(256), 0, ST(s), J);
        XMC_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int xmc_pack_weight(const float* w, void* wpk, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                               int transpose, int dtype, const int32_t* row_perm, void* s) {
    return xmc_pack_weight_grouped(w, wpk, Co, Ci, KH, KW, rows_pad, cols_pad, transpose, dtype, row_perm, 1, s);
}
extern "C" int xmc_unpack_wgrad_grouped(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                                        const float* scale_dev, const int32_t* row_perm, int accumulate, int groups, void* s) {
    if (!dwp || !gw || rows_pad < Co || cols_pad < Ci || groups < 1 || Co % groups || Ci % groups) return XMC_EINVAL;
    int64_t total = (int64_t)Co * (Ci / groups) * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(nblocks(total)), dim3(NT), 0, ST(s), dwp, gw, Co, Ci, KH * KW, rows_pad, cols_pad,
                       scale_dev, row_perm, accumulate, (const float*)nullptr, (float*)nullptr, 0, groups, (const float*)nullptr, (float*)nullptr);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_unpack_wgrad(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                                const float* scale_dev, const int32_t* row_perm, int accumulate, void* s) {
    return xmc_unpack_wgrad_grouped(dwp, gw, Co, Ci, KH, KW, rows_pad, cols_pad, scale_dev, row_perm, accumulate, 1, s);
}
extern "C" int xmc_unpack_wgrad_bias_dot(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                                         const float* scale_dev, const int32_t* row_perm, int accumulate,
                                         const float* gb_replicas, float* gb, int CD, const float* bias_dot, float* dot, void* s) {
    if (!dwp || !gw || rows_pad < Co || cols_pad < Ci || !gb_replicas || !gb || CD < 1) return XMC_EINVAL;
    if ((bias_dot == nullptr) != (dot == nullptr) || (bias_dot && row_perm)) return XMC_EINVAL;
    int64_t total = (int64_t)Co * Ci * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(nblocks(total)), dim3(NT), 0, ST(s), dwp, gw, Co, Ci, KH * KW, rows_pad, cols_pad,
                       scale_dev, row_perm, accumulate, gb_replicas, gb, CD, 1, bias_dot, dot);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_unpack_wgrad_bias(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                                     const float* scale_dev, const int32_t* row_perm, int accumulate,
                                     const float* gb_replicas, float* gb, int CD, void* s) {
    return xmc_unpack_wgrad_bias_dot(dwp, gw, Co, Ci, KH, KW, rows_pad, cols_pad, scale_dev, row_perm, accumulate, gb_replicas, gb, CD,
                                     nullptr, nullptr, s);
}

extern "C" int xmc_pack_weight_upconv(const float* w, void* wpk, int Co, int Ci, int rows_pad, int cols_pad, int transpose,
                                      int dtype, void* s) {
    if (!w || !wpk) return XMC_EINVAL;
    if (transpose ? (rows_pad < Ci || cols_pad < Co) : (rows_pad < Co || cols_pad < Ci)) return XMC_ESHAPE;
    int64_t total = (int64_t)16 * rows_pad * cols_pad;
    dim3 g(nblocks(total)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((pack_upconv_kernel<XMC_BF16>), g, blk, 0, ST(s), w, wpk, Co, Ci, rows_pad, cols_pad, transpose);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((pack_upconv_kernel<XMC_F32>), g, blk, 0, ST(s), w, wpk, Co, Ci, rows_pad, cols_pad, transpose);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_axpby_bwd(const void* dy, const void* b, const float* alpha, void* db, void* da, float* dot,
                             int N, int H, int W, int C, int up, const void* ymask, int dtype, void* s) {
    if (!dy || !b || !alpha || !dot || C % 8 || ((up || ymask) && !da)) return XMC_EINVAL;
    int64_t total = (int64_t)N * H * W * (C / 8);
    dim3 g(nblocks(total, NT, 2048)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((axpby_bwd_kernel<XMC_BF16>), g, blk, 0, ST(s), dy, b, alpha, db, da, dot, N, H, W, C / 8, up, ymask);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((axpby_bwd_kernel<XMC_F32>), g, blk, 0, ST(s), dy, b, alpha, db, da, dot, N, H, W, C / 8, up, ymask);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
static int axpby_up_impl(const void* a, const void* b, const float* alpha, void* y, int N, int H, int W, int C, int lrelu, int dtype, void* s) {
    if (!alpha || C % 8) return XMC_EINVAL;
    int64_t total = (int64_t)N * 4 * H * W * (C / 8);
    dim3 g(nblocks(total)), blk(NT);
    if (dtype == XMC_BF16) hipLaunchKernelGGL((axpby_up_kernel<XMC_BF16>), g, blk, 0, ST(s), a, b, alpha, y, N, H, W, C / 8, lrelu);
    else if (dtype == XMC_F32) hipLaunchKernelGGL((axpby_up_kernel<XMC_F32>), g, blk, 0, ST(s), a, b, alpha, y, N, H, W, C / 8, lrelu);
    else return XMC_EINVAL;
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_axpby_up(const void* a, const void* b, const float* alpha, void* y, int N, int H, int W, int C, int dtype, void* s) {
    return axpby_up_impl(a, b, alpha, y, N, H, W, C, 0, dtype, s);
}
extern "C" int xmc_axpby_up_lrelu(const void* a, const void* b, const float* alpha, void* y, int N, int H, int W, int C, int dtype, void* s) {
    return axpby_up_impl(a, b, alpha, y, N, H, W, C, 1, dtype, s);
}

// Matching-aware gradient penalty (train_gan.py:241-247): grad_l2norm^6 averaged over the batch, from the f32 gradient blocks
// [B, cols] (image gradient [B, 3*S*S], sentence gradient [B, cond]) without concatenating them.
extern "C" int xmc_rows_sumsq(const float* x, float* ss, int B, int64_t cols, void* s) {
    if (!x || !ss || B < 1 || cols < 4) return XMC_EINVAL;
    if (cols % 4) return XMC_EALIGN;
    const int64_t c4 = cols / 4;
    int gx = (int)((c4 + NT * 8 - 1) / (NT * 8));
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(rows_sumsq_kernel, dim3(gx, B), dim3(NT), 0, ST(s), x, ss, c4);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_gp_finish(const float* ss, int B, float* gp, float* coef, float inv_s2, void* s) {
    if (!ss || !gp || !coef || B < 1) return XMC_EINVAL;
    hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(NT), 0, ST(s), ss, B, gp, coef, inv_s2 == 0.f ? 1.f : inv_s2);
    XMC_LAUNCH_CHECK();
    return 0;
}
extern "C" int xmc_rows_scale(const float* x, const float* coef, const float* g, float* y, int B, int64_t cols, void* s) {
    if (!x || !coef || !g || !y || B < 1 || cols < 4) return XMC_EINVAL;
    if (cols % 4) return XMC_EALIGN;
    const int64_t c4 = cols / 4;
    int gx = (int)((c4 + NT * 8 - 1) / (NT * 8));
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(rows_scale_kernel, dim3(gx, B), dim3(NT), 0, ST(s), x, coef, g, y, c4);
    XMC_LAUNCH_CHECK();
    return 0;
}
