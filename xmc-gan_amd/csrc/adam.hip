// Multi-tensor Adam: one launch updates every parameter tensor of a network
// (torch.optim.Adam.step at train_gan.py:229,252,289: eps 1e-8, no weight decay, no amsgrad).
// The table lives in device memory and is static while the parameter/gradient storage is, so the
// launch is hipGraph-capturable; per-tensor step counters are kept on the device as well.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int CHUNK = NT * 4 * 4;  // elements per block: 4 float4 per thread

// Dynamic loss scale (the IEEE-half mode; torch.cuda.amp.GradScaler's rule, device-resident so the step stays capturable):
//   sf[0] = scale, sf[1] = 1 / scale;  si[0] = a gradient of THIS step was not finite, si[1] = consecutive finite steps,
//   si[2] = si[0] of the last finished step, si[3] = steps skipped so far.
// adam_check_kernel raises si[0]; adam_kernel and adam_bump_kernel leave parameters, moments and step counters alone when it
// is set; scaler_update_kernel backs the scale off (or grows it after `interval` finite steps) and clears the flag.
__global__ void adam_check_kernel(const XmcAdamEntry* __restrict__ tab, const int2* __restrict__ chunks, int* __restrict__ si) {
    const int2 c = chunks[blockIdx.x];
    const XmcAdamEntry e = tab[c.x];
    const int64_t base = (int64_t)c.y * CHUNK;
    const int64_t lim = e.n - base < CHUNK ? e.n - base : CHUNK;
    const float* g = e.grad + base;
    bool bad = false;
    // |x| < inf is false for inf and NaN alike
    if ((e.n & 3) == 0) {
        for (int i = threadIdx.x * 4; i < lim; i += NT * 4) {
            const f32x4 G = *reinterpret_cast<const f32x4*>(g + i);
            bad |= !(fabsf(G[0]) < INFINITY) | !(fabsf(G[1]) < INFINITY) | !(fabsf(G[2]) < INFINITY) | !(fabsf(G[3]) < INFINITY);
        }
    } else {
        for (int i = threadIdx.x; i < lim; i += NT) bad |= !(fabsf(g[i]) < INFINITY);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(si, 1);
}
__global__ void scaler_update_kernel(float* __restrict__ sf, int* __restrict__ si, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sc = sf[0];
    const int bad = si[0];
    if (bad) {
        sc = fmaxf(sc * backoff, 1.f);
        si[1] = 0;
        si[3] += 1;
    } else if (++si[1] >= interval) {
        sc = fminf(sc * growth, 16777216.f);
        si[1] = 0;
    }
    sf[0] = sc;
    sf[1] = 1.f / sc;
    si[2] = bad;
    si[0] = 0;
}

__global__ void adam_kernel(const XmcAdamEntry* __restrict__ tab, const int2* __restrict__ chunks,
                            float lr, float b1, float b2, float eps, float gs, const float* __restrict__ sf,
                            const int* __restrict__ si) {
    if (si && *si) return;                      // a non-finite gradient: the whole step is skipped
    if (sf) gs *= sf[1];
    const int2 c = chunks[blockIdx.x];
    const XmcAdamEntry e = tab[c.x];
    const int t = *e.step + 1;
    const float bc1 = 1.f - powf(b1, (float)t), bc2s = sqrtf(1.f - powf(b2, (float)t));
    const float step_size = lr / bc1;
    const int64_t base = (int64_t)c.y * CHUNK;
    const int64_t lim = e.n - base < CHUNK ? e.n - base : CHUNK;
    float* p = e.param + base; const float* g = e.grad + base; float* m = e.m + base; float* v = e.v + base;
    if ((e.n & 3) == 0) {
        for (int i = threadIdx.x * 4; i < lim; i += NT * 4) {
            f32x4 P = *reinterpret_cast<f32x4*>(p + i), G = *reinterpret_cast<const f32x4*>(g + i) * gs;
            f32x4 M = *reinterpret_cast<f32x4*>(m + i), V = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                M[k] = b1 * M[k] + (1.f - b1) * G[k];
                V[k] = b2 * V[k] + (1.f - b2) * G[k] * G[k];
                P[k] -= step_size * M[k] / (sqrtf(V[k]) / bc2s + eps);
            }
            *reinterpret_cast<f32x4*>(p + i) = P; *reinterpret_cast<f32x4*>(m + i) = M; *reinterpret_cast<f32x4*>(v + i) = V;
        }
    } else {
        for (int i = threadIdx.x; i < lim; i += NT) {
            float G = g[i] * gs;
            float M = b1 * m[i] + (1.f - b1) * G;
            float V = b2 * v[i] + (1.f - b2) * G * G;
            p[i] -= step_size * M / (sqrtf(V) / bc2s + eps);
            m[i] = M; v[i] = V;
        }
    }
}
__global__ void adam_bump_kernel(const XmcAdamEntry* tab, int n, const int* __restrict__ si) {
    if (si && *si) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *tab[i].step += 1;
}
}  // namespace

extern "C" int xmc_adam_chunk_elems(void) { return CHUNK; }

extern "C" int xmc_adam_step(const XmcAdamEntry* table_dev, int ntensors, const int32_t* chunks_dev, int nchunks,
                             float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!table_dev || !chunks_dev || ntensors < 1 || nchunks < 1) return XMC_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(NT), 0, st, table_dev, reinterpret_cast<const int2*>(chunks_dev), lr, beta1, beta2, eps,
                       grad_scale, (const float*)nullptr, (const int*)nullptr);
    hipLaunchKernelGGL(adam_bump_kernel, dim3((ntensors + NT - 1) / NT), dim3(NT), 0, st, table_dev, ntensors, (const int*)nullptr);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_adam_step_scaled(const XmcAdamEntry* table_dev, int ntensors, const int32_t* chunks_dev, int nchunks,
                                    float lr, float beta1, float beta2, float eps, float* scale_dev, int32_t* flags_dev,
                                    int mode, float growth, float backoff, int interval, void* stream) {
    if (!table_dev || !chunks_dev || ntensors < 1 || nchunks < 1 || !scale_dev || !flags_dev || interval < 1 || (mode & 7) == 0) return XMC_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int2* ch = reinterpret_cast<const int2*>(chunks_dev);
    if (mode & XMC_ADAM_CHECK) hipLaunchKernelGGL(adam_check_kernel, dim3(nchunks), dim3(NT), 0, st, table_dev, ch, flags_dev);
    if (mode & XMC_ADAM_UPDATE) {
        hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(NT), 0, st, table_dev, ch, lr, beta1, beta2, eps, 1.f, (const float*)scale_dev,
                           (const int*)flags_dev);
        hipLaunchKernelGGL(adam_bump_kernel, dim3((ntensors + NT - 1) / NT), dim3(NT), 0, st, table_dev, ntensors, (const int*)flags_dev);
    }
    if (mode & XMC_ADAM_RESCALE) hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, st, scale_dev, flags_dev, growth, backoff, interval);
    XMC_LAUNCH_CHECK();
    return 0;
}
