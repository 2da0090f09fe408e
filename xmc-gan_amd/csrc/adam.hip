// Multi-tensor Adam: one launch updates every parameter tensor of a network
// (torch.optim.Adam.step at train_gan.py:229,252,289: eps 1e-8, no weight decay, no amsgrad).
// The table lives in device memory and is static while the parameter/gradient storage is, so the
// launch is hipGraph-capturable; per-tensor step counters are kept on the device as well.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int CHUNK = NT * 4 * 4;  // elements per block: 4 float4 per thread

__global__ void adam_kernel(const XmcAdamEntry* __restrict__ tab, const int2* __restrict__ chunks,
                            float lr, float b1, float b2, float eps, float gs) {
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
__global__ void adam_bump_kernel(const XmcAdamEntry* tab, int n) {
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
                       grad_scale);
    hipLaunchKernelGGL(adam_bump_kernel, dim3((ntensors + NT - 1) / NT), dim3(NT), 0, st, table_dev, ntensors);
    XMC_LAUNCH_CHECK();
    return 0;
}
