// Spectral normalisation of a discriminator layer weight (reference model/modules.py:3,16-17,31-32: the legacy
// torch.nn.utils.spectral_norm forward pre-hook, one power iteration per forward call in training mode).
//
// W is the f32 parameter viewed as a row-major [R, C] matrix (R = out channels, C = in channels * k * k; at most
// 512 x 8192 = 16 MB on this path).  Everything here is HBM-bound matrix-vector work: W is read once for W^T u, once
// for W v, once for the scaling (and once more, with the incoming gradient, in the backward), so the kernels are
// organised for coalesced 256-byte wave reads and enough workgroups to cover the 256 CUs -- no MFMA.
//
//   t = W^T u            sn_wtu_kernel     thread per column, rows split over blockIdx.y, f32 atomics into zeroed t
//   s = W t  (or W v)    sn_wv_kernel      one wave per row, lanes stride the columns, wave reduction
//   v = t/|t|, u = (s/|t|)/|s/|t||, sigma = u.(s/|t|)      sn_finish_kernel  (one workgroup; R, C <= a few thousand)
//   W_eff = W / sigma    sn_scale_kernel
//
// Backward (sigma = u^T W v with u, v constants):  dW = g/sigma - (<g, W>/sigma^2) u v^T   -> sn_bwd_kernel
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup; every thread gets the result.  red: >= blockDim/64 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += red[i];
    return tot;
}

__global__ __launch_bounds__(NT) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                    float* __restrict__ t, int R, int C, int rows_per) {
    const int c = blockIdx.x * NT + threadIdx.x;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    if (c >= C) return;
    const float* p = W + (size_t)r0 * C + c;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {            // four independent row reads in flight per lane
        const float w0 = p[0], w1 = p[C], w2 = p[2 * (size_t)C], w3 = p[3 * (size_t)C];
        acc0 += w0 * u[r];
        acc1 += w1 * u[r + 1];
        acc2 += w2 * u[r + 2];
        acc3 += w3 * u[r + 3];
        p += 4 * (size_t)C;
    }
    for (; r < r1; ++r, p += C) acc0 += p[0] * u[r];
    atomicAdd(t + c, (acc0 + acc1) + (acc2 + acc3));
}

__global__ __launch_bounds__(NT) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                   float* __restrict__ s, int R, int C) {
    const int r = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* row = W + (size_t)r * C;
    float acc = 0.f;
    if ((C & 3) == 0) {
        const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (int c = lane; c < C / 4; c += 64) {
            const f32x4 a = row4[c], b = x4[c];
            acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
    } else {
        for (int c = lane; c < C; c += 64) acc += row[c] * x[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) s[r] = acc;
}

// training: t = W^T u_old (unnormalised), s = W t.   eval: s = W v.   sig[0] = 1/sigma, sig[1] = sigma.
__global__ __launch_bounds__(1024) void sn_finish_kernel(const float* __restrict__ t, const float* __restrict__ s,
                                                         float* __restrict__ u, float* __restrict__ v,
                                                         float* __restrict__ sig, int R, int C, int training, float eps) {
    __shared__ float red[16];
    const int tid = threadIdx.x, n = blockDim.x;
    float sigma;
    if (training) {
        float a = 0.f;
        for (int c = tid; c < C; c += n) a += t[c] * t[c];
        const float nt = fmaxf(sqrtf(block_sum(a, red)), eps);
        const float rnt = 1.f / nt;
        for (int c = tid; c < C; c += n) v[c] = t[c] * rnt;
        float b = 0.f;
        for (int r = tid; r < R; r += n) {
            const float sv = s[r] * rnt;
            b += sv * sv;
        }
        const float ns = fmaxf(sqrtf(block_sum(b, red)), eps);
        const float rns = 1.f / ns;
        float d = 0.f;
        for (int r = tid; r < R; r += n) {
            const float sv = s[r] * rnt, un = sv * rns;
            u[r] = un;
            d += un * sv;
        }
        sigma = block_sum(d, red);
    } else {
        float d = 0.f;
        for (int r = tid; r < R; r += n) d += u[r] * s[r];
        sigma = block_sum(d, red);
    }
    if (tid == 0) {
        sig[0] = 1.f / sigma;
        sig[1] = sigma;
    }
}

__global__ __launch_bounds__(NT) void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, int64_t n) {
    __shared__ float red[NT / 64];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += a[i] * b[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, acc);
}

__global__ __launch_bounds__(NT) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                    const float* __restrict__ v, const float* __restrict__ sig,
                                                    const float* __restrict__ dot, float* __restrict__ dW, int R, int C) {
    const float is = sig[0], coef = dot[0] * is * is;
    const int r = blockIdx.y;
    const float ur = coef * u[r];
    const size_t base = (size_t)r * C;
    for (int c = blockIdx.x * NT + threadIdx.x; c < C; c += gridDim.x * NT) dW[base + c] = g[base + c] * is - ur * v[c];
}

__global__ __launch_bounds__(NT) void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sig,
                                                      float* __restrict__ y, int64_t n) {
    const float is = sig[0];
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * NT;
    const f32x4* W4 = reinterpret_cast<const f32x4*>(W);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) y4[i] = W4[i] * is;
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) y[i] = W[i] * is;
}

}  // namespace

extern "C" int xmc_spectral_sigma(const float* W, float* u, float* v, float* scratch, float* sig, float* w_eff, int R, int C,
                                  int training, float eps, void* stream) {
    if (!W || !u || !v || !scratch || !sig || R <= 0 || C <= 0) return XMC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* t = scratch;          // [C]
    float* s = scratch + C;      // [R]
    if (training) {
        if (hipMemsetAsync(t, 0, sizeof(float) * C, st) != hipSuccess) return XMC_EINVAL;
        const int cb = (C + NT - 1) / NT;
        int splits = 1024 / cb;
        if (splits < 1) splits = 1;
        if (splits > (R + 7) / 8) splits = (R + 7) / 8;
        const int rows_per = (R + splits - 1) / splits;
        splits = (R + rows_per - 1) / rows_per;
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(cb, splits), dim3(NT), 0, st, W, u, t, R, C, rows_per);
        XMC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((R + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, st, W, training ? t : v, s, R, C);
    XMC_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(1024), 0, st, t, s, u, v, sig, R, C, training, eps);
    XMC_LAUNCH_CHECK();
    if (w_eff) {
        const int64_t n = (int64_t)R * C;
        int64_t nb = (n / 4 + NT - 1) / NT;
        if (nb > 2048) nb = 2048;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(sn_scale_kernel, dim3((unsigned)nb), dim3(NT), 0, st, W, sig, w_eff, n);
        XMC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int xmc_spectral_bwd(const float* g, const float* W, const float* u, const float* v, const float* sig,
                                float* dot, float* dW, int R, int C, void* stream) {
    if (!g || !W || !u || !v || !sig || !dot || !dW || R <= 0 || C <= 0) return XMC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)R * C;
    if (hipMemsetAsync(dot, 0, sizeof(float), st) != hipSuccess) return XMC_EINVAL;
    int64_t nb = (n + NT * 8 - 1) / (NT * 8);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sn_dot_kernel, dim3((unsigned)nb), dim3(NT), 0, st, g, W, dot, n);
    XMC_LAUNCH_CHECK();
    int cb = (C + NT - 1) / NT;
    if (cb > 8) cb = 8;
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(cb, R), dim3(NT), 0, st, g, u, v, sig, dot, dW, R, C);
    XMC_LAUNCH_CHECK();
    return 0;
}
