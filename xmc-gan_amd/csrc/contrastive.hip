// Contrastive head: cosine similarity matrix + symmetric InfoNCE (train_gan.py:85-139), forward and backward.
// Pipeline on one stream (all f32):
//   normalize rows of A and B (wave per row, shuffle reductions)            -> Ah, Bh, 1/|a|, 1/|b|
//   S = Ah Bh^T on the f32 MFMA path of the implicit-GEMM kernel            -> S [n][n8]
//   row pass / column pass: max + log-sum-exp with wave shuffles, label-weighted sums -> loss (atomic scalar)
// backward:
//   dS (and dS^T) from S, the two LSE vectors and the labels                 (one pass over S)
//   dAh = dS Bh, dBh = dS^T Ah on the MFMA weight-gradient kernel (K = n)
//   projection through the normalisation: dA = (dAh - Ah (Ah . dAh)) / |a|
#include "common.h"
#include <string.h>

namespace {

constexpr int NT = 256;

struct Ws {
    float *Ah, *Bh, *ina, *inb, *S, *lse_r, *lse_c, *dS, *dST, *dAh, *dBh;
    int n8, n32, Dp;
};
inline int64_t align64(int64_t x) { return (x + 63) / 64 * 64; }
inline Ws carve(void* ws, int n, int D) {
    Ws w;
    w.n8 = (n + 7) / 8 * 8; w.n32 = (n + 31) / 32 * 32; w.Dp = D;
    float* p = reinterpret_cast<float*>(ws);
    auto take = [&](int64_t cnt) { float* r = p; p += align64(cnt); return r; };
    w.Ah = take((int64_t)w.n32 * D); w.Bh = take((int64_t)w.n32 * D);
    w.ina = take(n); w.inb = take(n);
    w.S = take((int64_t)w.n32 * w.n8);
    w.lse_r = take(n); w.lse_c = take(n);
    w.dS = take((int64_t)w.n32 * w.n8); w.dST = take((int64_t)w.n32 * w.n8);
    w.dAh = take((int64_t)w.n32 * D); w.dBh = take((int64_t)w.n32 * D);
    return w;
}

// F.normalize(x, p=2, dim=1, eps=1e-12): one wave per row
__global__ void normalize_rows_kernel(const float* X, float* Xh, float* inv, int n, int n_pad, int D) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_pad) return;
    if (row >= n) {  // zero the padding rows (they act as zero weights in the GEMMs)
        for (int d = lane; d < D; d += 64) Xh[(size_t)row * D + d] = 0.f;
        return;
    }
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float v = X[(size_t)row * D + d]; s += v * v; }
    s = wave_sum(s);
    const float iv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int d = lane; d < D; d += 64) Xh[(size_t)row * D + d] = X[(size_t)row * D + d] * iv;
    if (lane == 0) inv[row] = iv;
}

// one wave per line (row if !col, column if col): lse, and the label-weighted sum  sum_k L * (S - lse) * inv_np
__global__ void lse_loss_kernel(const float* S, const float* labels, const float* inv_np, int n, int ld, int col,
                                float* lse_out, float* loss) {
    const int line = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float contrib = 0.f;
    if (line < n) {
        auto at = [&](int k) { return col ? S[(size_t)k * ld + line] : S[(size_t)line * ld + k]; };
        float mx = -INFINITY;
        for (int k = lane; k < n; k += 64) mx = fmaxf(mx, at(k));
        mx = wave_max(mx);
        float se = 0.f;
        for (int k = lane; k < n; k += 64) se += expf(at(k) - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        float acc = 0.f;
        if (labels) {
            for (int k = lane; k < n; k += 64) {
                float L = col ? labels[(size_t)k * n + line] : labels[(size_t)line * n + k];
                if (L != 0.f) acc += L * (at(k) - lse);
            }
        } else if (lane == 0) {
            acc = at(line) - lse;
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            lse_out[line] = lse;
            contrib = -acc * (inv_np ? inv_np[line] : 1.f) / (float)n;
        }
    }
    __shared__ float part[NT / 64];
    if (lane == 0) part[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += part[w];
        atomicAdd(loss, t);
    }
}

// label column/row sums (only needed when labels are given): cs[j] = sum_i L_ij, rs[i] = sum_j L_ij
__global__ void label_sums_kernel(const float* labels, int n, float* rs, float* cs) {
    const int line = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (line >= n) return;
    float r = 0.f, c = 0.f;
    for (int k = lane; k < n; k += 64) { r += labels[(size_t)line * n + k]; c += labels[(size_t)k * n + line]; }
    r = wave_sum(r); c = wave_sum(c);
    if (lane == 0) { rs[line] = r; cs[line] = c; }
}

// dS_ij = g/n * [ inv_np_j (softmax_col_ij * cs_j - L_ij) + inv_np_i (softmax_row_ij * rs_i - L_ij) ]
__global__ void ds_kernel(const float* S, const float* labels, const float* inv_np, const float* lse_r, const float* lse_c,
                          const float* rs, const float* cs, const float* dloss, int n, int ld, float* dS, float* dST) {
    const float g = (*dloss) / (float)n;
    const int64_t total = (int64_t)n * ld;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(id / ld), j = (int)(id - (int64_t)i * ld);
        float v = 0.f;
        if (j < n) {
            float s = S[(size_t)i * ld + j];
            float L = labels ? labels[(size_t)i * n + j] : (i == j ? 1.f : 0.f);
            float npj = inv_np ? inv_np[j] : 1.f, npi = inv_np ? inv_np[i] : 1.f;
            float csj = cs ? cs[j] : 1.f, rsi = rs ? rs[i] : 1.f;
            v = g * (npj * (expf(s - lse_c[j]) * csj - L) + npi * (expf(s - lse_r[i]) * rsi - L));
            dST[(size_t)j * ld + i] = v;
        }
        dS[(size_t)i * ld + j] = v;
    }
}
__global__ void zero_pad_kernel(float* M, int n, int n_rows, int ld) {  // zero rows >= n and cols >= n
    const int64_t total = (int64_t)n_rows * ld;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(id / ld), j = (int)(id - (int64_t)i * ld);
        if (i >= n || j >= n) M[id] = 0.f;
    }
}

// dX = (dXh - Xh * (Xh . dXh)) * inv   (one wave per row)
__global__ void normalize_bwd_kernel(const float* Xh, const float* dXh, const float* inv, float* dX, int n, int D) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += Xh[(size_t)row * D + d] * dXh[(size_t)row * D + d];
    dot = wave_sum(dot);
    const float iv = inv[row];
    for (int d = lane; d < D; d += 64)
        dX[(size_t)row * D + d] = (dXh[(size_t)row * D + d] - Xh[(size_t)row * D + d] * dot) * iv;
}

void fill_linear_desc(XmcConvDesc& d, const void* src, const void* w, void* dst, int rows, int K, int CD, int CDw) {
    d = XmcConvDesc{};
    d.src = src; d.wpk = w; d.dst = dst;
    d.N = rows; d.SH = d.SW = d.DH = d.DW = d.MH = d.MW = 1;
    d.CS = K; d.CD = CD; d.CDw = CDw;
    d.SA = d.DA = 1; d.ntaps = 1; d.nclass = 1;
    d.dtype = XMC_F32; d.out_dtype = XMC_F32; d.act = XMC_ACT_NONE;
}

}  // namespace

extern "C" int64_t xmc_contrastive_ws_bytes(int n, int D) {
    int64_t n8 = (n + 7) / 8 * 8, n32 = (n + 31) / 32 * 32;
    int64_t f = 4 * align64(n32 * D) + 3 * align64(n32 * n8) + 6 * align64(n) + 64;
    return f * 4;
}

// scratch for label sums lives at the tail of the workspace
static float* label_sum_scratch(const Ws& w, int n) { return w.dBh + align64((int64_t)w.n32 * w.Dp); }

extern "C" int xmc_contrastive_fwd(const float* A, const float* B, const float* labels, const float* inv_num_pos,
                                   int n, int D, float* loss, void* ws, void* stream) {
    if (!A || !B || !loss || !ws) return XMC_EINVAL;
    if (n < 1 || D < 4 || D % 4) return XMC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ws w = carve(ws, n, D);
    const int rb = (w.n32 + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(rb), dim3(NT), 0, st, A, w.Ah, w.ina, n, w.n32, D);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(rb), dim3(NT), 0, st, B, w.Bh, w.inb, n, w.n32, D);
    XMC_LAUNCH_CHECK();
    XmcConvDesc d;
    fill_linear_desc(d, w.Ah, w.Bh, w.S, n, D, w.n8, w.n32);
    int rc = xmc_conv_igemm(&d, stream);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), st);
    if (e != hipSuccess) return -(1000 + (int)e);
    const int lb = (n + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(lse_loss_kernel, dim3(lb), dim3(NT), 0, st, w.S, labels, inv_num_pos, n, w.n8, 1, w.lse_c, loss);
    hipLaunchKernelGGL(lse_loss_kernel, dim3(lb), dim3(NT), 0, st, w.S, labels, inv_num_pos, n, w.n8, 0, w.lse_r, loss);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_contrastive_bwd(const float* A, const float* B, const float* labels, const float* inv_num_pos,
                                   int n, int D, const float* dloss_dev, void* ws, float* dA, float* dB, void* stream) {
    (void)A; (void)B;
    if (!dloss_dev || !ws || !dA || !dB) return XMC_EINVAL;
    if (n < 1 || D < 4 || D % 4) return XMC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ws w = carve(ws, n, D);
    float *rs = nullptr, *cs = nullptr;
    const int lb = (n + NT / 64 - 1) / (NT / 64);
    if (labels) {
        rs = label_sum_scratch(w, n); cs = rs + align64(n);
        hipLaunchKernelGGL(label_sums_kernel, dim3(lb), dim3(NT), 0, st, labels, n, rs, cs);
    }
    int64_t total = (int64_t)w.n32 * w.n8;
    int blocks = (int)((total + NT - 1) / NT); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_pad_kernel, dim3(blocks), dim3(NT), 0, st, w.dS, n, w.n32, w.n8);
    hipLaunchKernelGGL(zero_pad_kernel, dim3(blocks), dim3(NT), 0, st, w.dST, n, w.n32, w.n8);
    hipLaunchKernelGGL(ds_kernel, dim3(blocks), dim3(NT), 0, st, w.S, labels, inv_num_pos, w.lse_r, w.lse_c, rs, cs, dloss_dev,
                       n, w.n8, w.dS, w.dST);
    XMC_LAUNCH_CHECK();
    // dAh[i][d] = sum_j dST[j][i] * Bh[j][d] ; dBh[j][d] = sum_i dS[i][j] * Ah[i][d]    (wgrad form, K = n rows)
    hipError_t e = hipMemsetAsync(w.dAh, 0, (size_t)w.n32 * D * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(w.dBh, 0, (size_t)w.n32 * D * 4, st);
    if (e != hipSuccess) return -(1000 + (int)e);
    XmcConvDesc d;
    fill_linear_desc(d, w.Bh, nullptr, w.dST, n, D, w.n8, w.n32);
    int rc = xmc_conv_wgrad(&d, w.dAh, stream);
    if (rc) return rc;
    fill_linear_desc(d, w.Ah, nullptr, w.dS, n, D, w.n8, w.n32);
    rc = xmc_conv_wgrad(&d, w.dBh, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(lb), dim3(NT), 0, st, w.Ah, w.dAh, w.ina, dA, n, D);
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(lb), dim3(NT), 0, st, w.Bh, w.dBh, w.inb, dB, n, D);
    XMC_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void copy_ld_kernel(const float* S, int ld, float* out, int n) {
    const int64_t total = (int64_t)n * n;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(id / n), j = (int)(id - (int64_t)i * n);
        out[id] = S[(size_t)i * ld + j];
    }
}
}  // namespace

// cosine_scores (train_gan.py:85-91) alone: S[n][n] = normalize(A) normalize(B)^T  (used by make_labels, 72-83)
extern "C" int xmc_cosine_scores(const float* A, const float* B, int n, int D, float* S, void* ws, void* stream) {
    if (!A || !B || !S || !ws) return XMC_EINVAL;
    if (n < 1 || D < 4 || D % 4) return XMC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ws w = carve(ws, n, D);
    const int rb = (w.n32 + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(rb), dim3(NT), 0, st, A, w.Ah, w.ina, n, w.n32, D);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(rb), dim3(NT), 0, st, B, w.Bh, w.inb, n, w.n32, D);
    XMC_LAUNCH_CHECK();
    XmcConvDesc d;
    fill_linear_desc(d, w.Ah, w.Bh, w.S, n, D, w.n8, w.n32);
    int rc = xmc_conv_igemm(&d, stream);
    if (rc) return rc;
    int64_t total = (int64_t)n * n;
    int blocks = (int)((total + NT - 1) / NT); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_ld_kernel, dim3(blocks), dim3(NT), 0, st, w.S, w.n8, S, n);
    XMC_LAUNCH_CHECK();
    return 0;
}
