// Sustained MFMA issue rate of the two bf16 shapes on one chip, no memory traffic (diagnostic; built and run by mfma_probe.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    bf16x8 a[4], b[4];          // a 4 x 4 register tile as in the convolution kernels: distinct operand registers per column / row
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(0.001f * (threadIdx.x + i + q)); b[q][i] = (__bf16)(0.002f * (threadIdx.x + 3 * i + q)); }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    bf16x8 a[2], b[2];
    for (int q = 0; q < 2; ++q)
        for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(0.001f * (threadIdx.x + i + q)); b[q][i] = (__bf16)(0.002f * (threadIdx.x + 3 * i + q)); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 1], b[(j >> 1) & 1], acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> double timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
    float* out; hipMalloc(&out, 4 << 20);
    const int iters = 4000;
    for (int blocks : {256, 512, 1024}) {
        double t16 = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double t32 = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double f16 = 2.0 * 16 * 16 * 32 * 16 * iters * 4.0 * blocks, f32 = 2.0 * 32 * 32 * 16 * 4 * iters * 4.0 * blocks;
        printf("blocks %4d: 16x16x32 (16 accumulators) %7.1f TF/s   32x32x16 (4 accumulators) %7.1f TF/s\n", blocks, f16 / t16 / 1e9, f32 / t32 / 1e9);
    }
    return 0;
}
