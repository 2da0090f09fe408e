// Grouped small f32 GEMMs: every conditioning MLP of the generator (reference df_gan.py:232-241: Linear(cond,256) -> ReLU ->
// Linear(256,C), two per `affine`, four affines per G_Block, 5-7 blocks = 40-56 MLPs that all read the same sentence
// embedding) in ONE launch per layer and direction instead of one launch per Linear.
//
// Each problem is batch-sized (M = batch = 256, K, N <= 512: ~30-70 MFLOP), so a launch per problem leaves a handful of
// workgroups on a 256-CU chip and, worse, each of those launches has to find a free CU between the persistent one-workgroup-
// per-CU convolution kernels it was meant to overlap.  Here the tiles of all problems form one flat grid (a few hundred to a
// thousand 64x64 tiles), operands are addressed through element strides so the same kernel serves y = x W^T (+b, ReLU),
// dx = dy W (masked by ReLU') and dW = dy^T x (+ db as a row sum), and the arithmetic is plain f32 FMA: exact products, as
// the f32 MFMA path the per-layer version used, and far from any roofline that matters at 3 GFLOP per layer.
#include "common.h"

namespace {

constexpr int GT = 64;       // C tile is GT x GT
constexpr int GK = 64;       // reduction chunk per barrier pair: four 16-row loads per operand in flight at once (at 16 a launch was 16 serial
                             // global round trips: ~45 us for 2 GFLOP; the MLP bank is 10 launches per iteration)
constexpr int GLD = GT + 16; // LDS row pitch (floats): the four k-rows a wave reads per MFMA step (lane >> 4) land 16 banks apart

__device__ __forceinline__ void load_tile(float (*dst)[GLD], const float* __restrict__ base, int s_i, int s_r, int i0, int r0,
                                          int ni, int nr, int tid) {
    // dst[r][i] = base[(i0+i)*s_i + (r0+r)*s_r], zero outside [0,ni) x [0,nr)
    if (s_r == 1) {                       // reduction index contiguous in memory: 4 consecutive r per thread
        const int i = tid >> 2, rq = (tid & 3) * 4;
        const bool iok = i0 + i < ni;
        const float* p = base + (size_t)(i0 + i) * s_i + (r0 + rq);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[rq + k][i] = (iok && r0 + rq + k < nr) ? p[k] : 0.f;
    } else {                              // tile index contiguous (s_i == 1) or generic strides: 4 consecutive i per thread
        const int r = tid >> 4, iq = (tid & 15) * 4;
        const bool rok = r0 + r < nr;
        const float* p = base + (size_t)(r0 + r) * s_r + (size_t)(i0 + iq) * s_i;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[r][iq + k] = (rok && i0 + iq + k < ni) ? p[(size_t)k * s_i] : 0.f;
    }
}

constexpr int CHUNK = 32;    // problems per launch: the table travels in the kernel arguments (32 x 88 B < the 4 KB kernarg limit),
struct ProbChunk {           // so a launch needs no device-side table, no pinned staging, and is plain to capture in a hipGraph
    XmcGemmProblem p[CHUNK];
};

__global__ __launch_bounds__(256) void gemm_group_kernel(const ProbChunk chunk, int np) {
    const XmcGemmProblem* P = chunk.p;
    __shared__ __attribute__((aligned(16))) float As[GK][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[GK][GLD];
    __shared__ int s_g;
    const int tid = threadIdx.x, tile = blockIdx.x;
    for (int g = tid; g < np; g += 256) {
        const int t0 = P[g].tile0, t1 = g + 1 < np ? P[g + 1].tile0 : 0x7fffffff;
        if (tile >= t0 && tile < t1) s_g = g;
    }
    __syncthreads();
    const XmcGemmProblem p = P[s_g];
    const int ntn = (p.N + GT - 1) / GT;
    const int lt = tile - p.tile0, ti = lt / ntn, tj = lt - ti * ntn;
    const int i0 = ti * GT, j0 = tj * GT;
    // v_mfma_f32_16x16x4_f32 (exact f32 products and sums, twice the plain-FMA rate: the bank is 2 GFLOP per launch and was
    // VALU-bound at ~65 % of the vector peak, 44 us a launch).  Wave w owns rows [16 w, 16 w + 16) of the tile and its four 16-column
    // blocks; lane (c = lane & 15, kq = lane >> 4) feeds A[row c][k0 + kq] and B[k0 + kq][16 jb + c] and ends with rows 4 kq + r of
    // column c of each block.
    const int lane = tid & 63, wv = tid >> 6, c16 = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rs = 0.f;
    const bool want_rs = p.rowsum != nullptr && tj == 0;

    for (int r0 = 0; r0 < p.K; r0 += GK) {
#pragma unroll
        for (int q = 0; q < GK / 16; ++q) {
            load_tile(As + 16 * q, p.A, p.sa_i, p.sa_r, i0, r0 + 16 * q, p.M, p.K, tid);
            load_tile(Bs + 16 * q, p.B, p.sb_j, p.sb_r, j0, r0 + 16 * q, p.N, p.K, tid);
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < GK; k0 += 4) {
            const float a = As[k0 + kq][wv * 16 + c16];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[k0 + kq][b * 16 + c16], acc[b], 0, 0, 0);
            rs += a;
        }
        __syncthreads();
    }

    if (want_rs) {                            // row sums of A: the four k-quarters of a row sit in lanes c, c + 16, c + 32, c + 48
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        if (kq == 0 && i0 + wv * 16 + c16 < p.M) p.rowsum[i0 + wv * 16 + c16] = rs;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wv * 16 + 4 * kq + r;
        if (i >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + b * 16 + c16;
            if (j >= p.N) continue;
            float v = acc[b][r];
            if (p.flags & XMC_GP_BIAS) v += p.bias[j];
            if (p.flags & XMC_GP_RELU) v = fmaxf(v, 0.f);
            if (p.flags & XMC_GP_MASK) v = p.mask[(size_t)i * p.N + j] > 0.f ? v : 0.f;
            float* c = p.C + (size_t)i * p.N + j;
            if (p.flags & XMC_GP_ATOMIC) atomicAdd(c, v);
            else *c = v;
        }
    }
}

}  // namespace

extern "C" int xmc_gemm_group(const XmcGemmProblem* problems, int nproblems, void* stream) {
    if (!problems || nproblems < 1) return XMC_EINVAL;
    for (int g0 = 0; g0 < nproblems; g0 += CHUNK) {
        ProbChunk c;
        const int np = nproblems - g0 < CHUNK ? nproblems - g0 : CHUNK;
        int tiles = 0;
        for (int g = 0; g < np; ++g) {
            c.p[g] = problems[g0 + g];
            const XmcGemmProblem& q = c.p[g];
            if (!q.A || !q.B || !q.C || q.M < 1 || q.N < 1 || q.K < 1) return XMC_EINVAL;
            if (((q.flags & XMC_GP_BIAS) && !q.bias) || ((q.flags & XMC_GP_MASK) && !q.mask)) return XMC_EINVAL;
            c.p[g].tile0 = tiles;
            tiles += ((q.M + GT - 1) / GT) * ((q.N + GT - 1) / GT);
        }
        for (int g = np; g < CHUNK; ++g) c.p[g] = c.p[0];
        hipLaunchKernelGGL(gemm_group_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, c, np);
        XMC_LAUNCH_CHECK();
    }
    return 0;
}
