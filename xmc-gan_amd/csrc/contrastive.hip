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


// =====================================================================================================================================
// Round 5: the head in 3 + 2 launches (was 7 + 9 dependent ones; n = 2048: 253-323 us forward + backward against ~20 us of traffic).
//   cn_prepare   rows of A and B: 1/|x|, xh = x / |x| (f32, for the projection in the backward) and xh as an IEEE-half PAIR hi + lo
//                (22 significant bits: |xh| <= 1, so the half range is no constraint), row-major [n][D] for the score GEMM and
//                transposed [D][n] for the two gradient GEMMs; zeroes the accumulators below.
//   cn_scores    64 x 64 tiles of S = Ah Bh^T on the f16 matrix pipeline, three MFMAs per K step (hi.hi + hi.lo + lo.hi: f32 grade at
//                16x the f32 MFMA rate); the tile goes through LDS once and leaves as coalesced row segments of S AND of S^T; on the
//                way: row and column sums of exp(S - 1) -- |S| <= 1, so the shift 1 replaces the row / column max and ONE pass serves
//                both directions --, the label-weighted sum of S and the labels' row / column sums.
//   cn_finish    lse_r = 1 + log(rowsum), lse_c likewise; the loss scalar.
//   cn_grad      both gradient GEMMs in one launch (blockIdx.y = side): a workgroup owns 32 output rows and walks the contraction in
//                steps of 32: dS is formed from the S (or S^T) tile as it is loaded -- never stored --, split into a half pair in LDS,
//                multiplied into the transposed half pairs of the other side's rows.  The contraction is cut into up to 8 ranges (one
//                workgroup each: 128 -> 1024 workgroups at n = 2048) whose partial tiles go to scratch.
//   cn_project   sums the ranges in order and projects through the normalisation, dx = (dxh - xh (xh . dxh)) / |x|.
// D % 32 == 0, D <= 512 (the head's 256 / 512); other widths take the f32 path above.
typedef _Float16 cn_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cn_h;

constexpr int CN_RS = 8;                               // at most this many ranges of the gradient GEMMs' contraction (one workgroup each)
struct Cn {
    float *Ah, *Bh, *ina, *inb, *S, *ST, *rowE, *colE, *rs, *cs, *lse_r, *lse_c, *T, *part;
    int* cnt;                                          // [0]: tiles of cn_scores done; [1 + side * np / 32 + cblock]: ranges of cn_grad done
    cn_h *Ahi, *Alo, *Bhi, *Blo, *AThi, *ATlo, *BThi, *BTlo;
    int np;                                            // n padded to 64
};
__device__ __forceinline__ float cn_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
inline Cn cn_carve(void* ws, int n, int D) {
    Cn w;
    w.np = (n + 63) / 64 * 64;
    float* p = reinterpret_cast<float*>(ws);
    auto take = [&](int64_t cnt) { float* r = p; p += align64(cnt); return r; };
    w.Ah = take((int64_t)w.np * D); w.Bh = take((int64_t)w.np * D);
    w.S = take((int64_t)w.np * w.np); w.ST = take((int64_t)w.np * w.np);
    w.ina = take(w.np); w.inb = take(w.np); w.rowE = take(w.np); w.colE = take(w.np); w.rs = take(w.np); w.cs = take(w.np);
    w.lse_r = take(w.np); w.lse_c = take(w.np); w.T = take(64);
    w.cnt = reinterpret_cast<int*>(take(w.np / 16 + 64));
    w.part = take((int64_t)2 * CN_RS * w.np * D);
    auto takeh = [&](int64_t cnt) { cn_h* r = reinterpret_cast<cn_h*>(p); p += align64((cnt + 1) / 2); return r; };
    const int64_t e = (int64_t)w.np * D;
    w.Ahi = takeh(e); w.Alo = takeh(e); w.Bhi = takeh(e); w.Blo = takeh(e);
    w.AThi = takeh(e); w.ATlo = takeh(e); w.BThi = takeh(e); w.BTlo = takeh(e);
    return w;
}
inline int64_t cn_bytes(int n, int D) {
    const int64_t np = (n + 63) / 64 * 64, e = np * D;
    return (2 * align64(e) + 2 * align64(np * np) + 8 * align64(np) + 64 + align64(np / 16 + 64) + align64(2 * CN_RS * e) +
            8 * align64((e + 1) / 2)) * 4 + 256;
}

// 8 rows per workgroup (256 threads): norms by wave, then every thread walks the columns of the 8 rows
__global__ __launch_bounds__(256) void cn_prepare_kernel(const float* __restrict__ A, const float* __restrict__ B, Cn w, int n, int D) {
    const int side = blockIdx.y, r0 = blockIdx.x * 8, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ X = side ? B : A;
    float* __restrict__ Xh = side ? w.Bh : w.Ah;
    float* __restrict__ inv = side ? w.inb : w.ina;
    cn_h* __restrict__ hi = side ? w.Bhi : w.Ahi;
    cn_h* __restrict__ lo = side ? w.Blo : w.Alo;
    cn_h* __restrict__ thi = side ? w.BThi : w.AThi;
    cn_h* __restrict__ tlo = side ? w.BTlo : w.ATlo;
    __shared__ float s_inv[8];
    for (int q = 0; q < 2; ++q) {
        const int row = r0 + wave * 2 + q;
        float s = 0.f;
        if (row < n)
            for (int d = lane; d < D; d += 64) { const float v = X[(size_t)row * D + d]; s += v * v; }
        s = wave_sum(s);
        if (lane == 0) {
            const float iv = row < n ? 1.f / fmaxf(sqrtf(s), 1e-12f) : 0.f;
            s_inv[wave * 2 + q] = iv;
            inv[row] = iv;
            if (side == 0) { w.rowE[row] = 0.f; w.rs[row] = 0.f; } else { w.colE[row] = 0.f; w.cs[row] = 0.f; }
        }
    }
    if (blockIdx.x == 0 && side == 0) {
        if (tid == 0) w.T[0] = 0.f;
        for (int q = tid; q < w.np / 16 + 1; q += 256) w.cnt[q] = 0;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        cn_h8 vh, vl;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = r0 + r;
            const float x = row < n ? X[(size_t)row * D + d] * s_inv[r] : 0.f;
            const cn_h h = (cn_h)x;
            const cn_h l = (cn_h)(x - (float)h);
            Xh[(size_t)row * D + d] = x;
            hi[(size_t)row * D + d] = h;
            lo[(size_t)row * D + d] = l;
            vh[r] = h; vl[r] = l;
        }
        *reinterpret_cast<cn_h8*>(thi + (size_t)d * w.np + r0) = vh;
        *reinterpret_cast<cn_h8*>(tlo + (size_t)d * w.np + r0) = vl;
    }
}

constexpr int CN_LD = 65;
__global__ __launch_bounds__(256) void cn_scores_kernel(Cn w, const float* __restrict__ labels, const float* __restrict__ inv_np, int n, int D,
                                                        float* __restrict__ loss) {
    __shared__ float Sl[64 * CN_LD], El[64 * CN_LD];
    __shared__ float s_t[4];
    (void)loss;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64, np = w.np;
    const int fr = lane & 15, fc = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t arow = (size_t)(i0 + wr * 32 + fr) * D + fc * 8, brow = (size_t)(j0 + wc * 32 + fr) * D + fc * 8;
    cn_h8 nah[2], nal[2], nbh[2], nbl[2];
    auto load_k = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            nah[q] = *reinterpret_cast<const cn_h8*>(w.Ahi + arow + (size_t)q * 16 * D + k0);
            nal[q] = *reinterpret_cast<const cn_h8*>(w.Alo + arow + (size_t)q * 16 * D + k0);
            nbh[q] = *reinterpret_cast<const cn_h8*>(w.Bhi + brow + (size_t)q * 16 * D + k0);
            nbl[q] = *reinterpret_cast<const cn_h8*>(w.Blo + brow + (size_t)q * 16 * D + k0);
        }
    };
    load_k(0);
    for (int k0 = 0; k0 < D; k0 += 32) {
        cn_h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { ah[q] = nah[q]; al[q] = nal[q]; bh[q] = nbh[q]; bl[q] = nbl[q]; }
        if (k0 + 32 < D) load_k(k0 + 32);              // the next K step's fragments fly during this step's MFMAs
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Sl[(wr * 32 + a * 16 + fc * 4 + r) * CN_LD + wc * 32 + b * 16 + fr] = acc[a][b][r];
    __syncthreads();
    // row pass: 4 threads per row, 16 columns each -> coalesced 256-byte segments of S; exp(S - 1) kept for the column pass
    const int row = tid >> 2, cq = (tid & 3) * 16, gi = i0 + row;
    float esum = 0.f, lsum = 0.f, tsum = 0.f;
    const float npi = (inv_np && gi < n) ? inv_np[gi] : 1.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        f32x4 sv, lv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (labels && gi < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int gj = j0 + cq + v * 4 + c; lv[c] = gj < n ? labels[(size_t)gi * n + gj] : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = cq + v * 4 + c, gj = j0 + col;
            const bool in = gi < n && gj < n;
            const float s = Sl[row * CN_LD + col];
            const float e = in ? __expf(s - 1.f) : 0.f;
            sv[c] = in ? s : 0.f;
            esum += e;
            float L = lv[c];
            if (!labels) L = (in && gi == gj) ? 1.f : 0.f;
            El[row * CN_LD + col] = e;
            if (L != 0.f) {
                const float npj = inv_np ? inv_np[gj] : 1.f;
                tsum += L * s * (npi + npj);
                lsum += L;
            }
            if (labels) Sl[row * CN_LD + col] = L;          // (the S value is in sv / was consumed: the tile now carries the labels for the column pass)
        }
        *reinterpret_cast<f32x4*>(w.S + (size_t)gi * np + j0 + cq + v * 4) = sv;
    }
    esum += __shfl_xor(esum, 1, 64); esum += __shfl_xor(esum, 2, 64);
    lsum += __shfl_xor(lsum, 1, 64); lsum += __shfl_xor(lsum, 2, 64);
    if ((tid & 3) == 0 && gi < n) {
        atomicAdd(&w.rowE[gi], esum);
        if (labels) atomicAdd(&w.rs[gi], lsum);
    }
    tsum = wave_sum(tsum);
    if (lane == 0) s_t[wave] = tsum;
    __syncthreads();
    if (tid == 0) { const float t = s_t[0] + s_t[1] + s_t[2] + s_t[3]; if (t != 0.f) atomicAdd(w.T, t); }
    // column pass: thread t < 64 sums column t of exp(S - 1) (and of the labels)
    if (tid < 64 && j0 + tid < n) {
        float e = 0.f, l = 0.f;
        for (int r = 0; r < 64; ++r) { e += El[r * CN_LD + tid]; if (labels) l += Sl[r * CN_LD + tid]; }
        atomicAdd(&w.colE[j0 + tid], e);
        if (labels) atomicAdd(&w.cs[j0 + tid], l);
    }
    // S^T: thread (j = tid / 4, 16 rows i) reads a column piece of the tile -- from the copy in global memory when the LDS tile was
    // overwritten by the labels
    {
        const int jr = tid >> 2, iq = (tid & 3) * 16, gj = j0 + jr;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 tv;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ir = iq + v * 4 + c;
                const bool in = (i0 + ir) < n && gj < n;
                // exp(S - 1) is in El: S = 1 + log(E) would lose bits; keep S itself: unless labels overwrote it, it is still in Sl
                tv[c] = in ? (labels ? 0.f : Sl[ir * CN_LD + jr]) : 0.f;
            }
            if (!labels) *reinterpret_cast<f32x4*>(w.ST + (size_t)gj * np + i0 + iq + v * 4) = tv;
        }
    }
}

// The same for n >= 1024 on 128 x 128 tiles (one per CU at n = 2048): the 64 x 64 form reads its MFMA fragments straight from L2, 32 KB per tile
// and K step through the CU's vector-memory path (~22 B/clk: 23 us of the launch at n = 2048); here a K step's operands (128 rows x 32 k x hi, lo
// of both sides: 32 KB for FOUR times the MACs) are staged through LDS once, double buffered, and every wave reads its fragments from there.
constexpr int CN_LD2 = 129, CN_SROW = 80;                  // epilogue tile row (floats); staged operand row (bytes: 64 + 16 pad, conflict-free b128 reads)
__global__ __launch_bounds__(512) void cn_scores128_kernel(Cn w, const float* __restrict__ labels, const float* __restrict__ inv_np, int n, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cn_smem[];
    float* const Sl = reinterpret_cast<float*>(cn_smem);                    // [128][129]   (aliases the staging buffers: used after the K loop)
    float* const El = Sl + 128 * CN_LD2;
    __shared__ float s_t[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128, np = w.np;
    const int fr = lane & 15, fc = lane >> 4;
    constexpr int ARR = 128 * CN_SROW;                                       // one staged array: 10 KB
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lrow = tid >> 2, lch = tid & 3;
    const cn_h* const src[4] = {w.Ahi + (size_t)(i0 + lrow) * D + lch * 8, w.Alo + (size_t)(i0 + lrow) * D + lch * 8,
                                w.Bhi + (size_t)(j0 + lrow) * D + lch * 8, w.Blo + (size_t)(j0 + lrow) * D + lch * 8};
    u32x4 ld[4];
    auto load_k = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ld[q] = *reinterpret_cast<const u32x4*>(src[q] + k0);
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(cn_smem + buf * 4 * ARR + q * ARR + lrow * CN_SROW + lch * 16) = ld[q];
    };
    load_k(0);
    store_k(0);
    __syncthreads();
    const int nst = D / 32;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) load_k((st + 1) * 32);
        const unsigned char* base = cn_smem + buf * 4 * ARR;
        cn_h8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int o = (wr * 32 + a * 16 + fr) * CN_SROW + fc * 16;
            ah[a] = *reinterpret_cast<const cn_h8*>(base + o);
            al[a] = *reinterpret_cast<const cn_h8*>(base + ARR + o);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int o = (wc * 64 + b * 16 + fr) * CN_SROW + fc * 16;
            bh[b] = *reinterpret_cast<const cn_h8*>(base + 2 * ARR + o);
            bl[b] = *reinterpret_cast<const cn_h8*>(base + 3 * ARR + o);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            }
        if (st + 1 < nst) store_k(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Sl[(wr * 32 + a * 16 + fc * 4 + r) * CN_LD2 + wc * 64 + b * 16 + fr] = acc[a][b][r];
    __syncthreads();
    // row pass: 4 threads per row, 32 columns each
    const int row = tid >> 2, cq = (tid & 3) * 32, gi = i0 + row;
    float esum = 0.f, lsum = 0.f, tsum = 0.f;
    const float npi = (inv_np && gi < n) ? inv_np[gi] : 1.f;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        f32x4 sv, lv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (labels && gi < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int gj = j0 + cq + v * 4 + c; lv[c] = gj < n ? labels[(size_t)gi * n + gj] : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = cq + v * 4 + c, gj = j0 + col;
            const bool in = gi < n && gj < n;
            const float s_ = Sl[row * CN_LD2 + col];
            const float e = in ? __expf(s_ - 1.f) : 0.f;
            sv[c] = in ? s_ : 0.f;
            esum += e;
            float L = lv[c];
            if (!labels) L = (in && gi == gj) ? 1.f : 0.f;
            El[row * CN_LD2 + col] = e;
            if (L != 0.f) {
                const float npj = inv_np ? inv_np[gj] : 1.f;
                tsum += L * s_ * (npi + npj);
                lsum += L;
            }
            if (labels) Sl[row * CN_LD2 + col] = L;
        }
        *reinterpret_cast<f32x4*>(w.S + (size_t)gi * np + j0 + cq + v * 4) = sv;
    }
    esum += __shfl_xor(esum, 1, 64); esum += __shfl_xor(esum, 2, 64);
    lsum += __shfl_xor(lsum, 1, 64); lsum += __shfl_xor(lsum, 2, 64);
    if ((tid & 3) == 0 && gi < n) {
        atomicAdd(&w.rowE[gi], esum);
        if (labels) atomicAdd(&w.rs[gi], lsum);
    }
    tsum = wave_sum(tsum);
    if (lane == 0) s_t[wave] = tsum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int q = 0; q < 8; ++q) t += s_t[q];
        if (t != 0.f) atomicAdd(w.T, t);
    }
    if (tid < 128 && j0 + tid < n) {
        float e = 0.f, l = 0.f;
        for (int r = 0; r < 128; ++r) { e += El[r * CN_LD2 + tid]; if (labels) l += Sl[r * CN_LD2 + tid]; }
        atomicAdd(&w.colE[j0 + tid], e);
        if (labels) atomicAdd(&w.cs[j0 + tid], l);
    }
    if (!labels) {                                  // S^T from the tile's columns (labels given: cn_transpose_kernel)
        const int jr = tid >> 2, iq = (tid & 3) * 32, gj = j0 + jr;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            f32x4 tv;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ir = iq + v * 4 + c;
                tv[c] = ((i0 + ir) < n && gj < n) ? Sl[ir * CN_LD2 + jr] : 0.f;
            }
            *reinterpret_cast<f32x4*>(w.ST + (size_t)gj * np + i0 + iq + v * 4) = tv;
        }
    }
}

// labels given: S^T by a plain tiled transpose of S (the rare B_GLOBAL path; keeps cn_scores' LDS tile free for the labels)
__global__ __launch_bounds__(256) void cn_transpose_kernel(const float* __restrict__ S, float* __restrict__ ST, int np) {
    __shared__ float tl[64 * CN_LD];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int e = tid; e < 4096; e += 256) { const int r = e >> 6, c = e & 63; tl[r * CN_LD + c] = S[(size_t)(i0 + r) * np + j0 + c]; }
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) { const int r = e >> 6, c = e & 63; ST[(size_t)(j0 + r) * np + i0 + c] = tl[c * CN_LD + r]; }
}

// (a ticket counter that let the LAST tile of cn_scores do this -- and the last range of cn_grad its projection -- was built and
// measured: the agent-scope release fence in front of the ticket writes back the XCD's L2 on this 8-XCD part, per workgroup:
// forward 61 -> 213 us at n = 2048.  Kernel boundaries do that write-back once.)
__global__ __launch_bounds__(256) void cn_finish_kernel(Cn w, const float* __restrict__ inv_np, int has_labels, int n, float* __restrict__ loss) {
    __shared__ float part[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float lr = 1.f + logf(w.rowE[i]), lc = 1.f + logf(w.colE[i]);
        w.lse_r[i] = lr; w.lse_c[i] = lc;
        const float np_ = inv_np ? inv_np[i] : 1.f;
        a += np_ * (lr * (has_labels ? w.rs[i] : 1.f) + lc * (has_labels ? w.cs[i] : 1.f));
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = -(w.T[0] - (part[0] + part[1] + part[2] + part[3])) / (float)n;
}

constexpr int CN_GLD = 40;                      // halfs per row of the transposed dS tile (80 bytes: 16-byte aligned, staggered banks)
template <int NB, int CB>                       // NB: 16-column blocks of a wave's slice of D (D / 64); CB: output rows per workgroup (32 or 64)
__global__ __launch_bounds__(256) void cn_grad_kernel(Cn w, const float* __restrict__ labels, const float* __restrict__ inv_np, int has_labels,
                                                      const float* __restrict__ dloss, int n, int D, float* __restrict__ dA, float* __restrict__ dB) {
    // side 0: dB[j] = sum_i dS[i][j] Ah[i]   (M = S,   rows r = i, outputs c = j);   side 1: dA[i] = sum_j dS[i][j] Bh[j]  (M = S^T)
    // (CB = 64 where the accumulators allow it: the other side's transposed half pairs -- 32 KB per step through the vector-memory path,
    // the bound of this launch -- then feed twice the MFMAs)
    constexpr int NA = CB / 16, RP = 256 / CB, NQ = 32 / RP;        // A blocks; contraction rows per pass of the tile build; passes
    const int side = blockIdx.y, c0 = blockIdx.x * CB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, np = w.np;
    const int RS = gridDim.z, rng = blockIdx.z;                     // this workgroup walks range `rng` of the contraction
    const float* __restrict__ M = side ? w.ST : w.S;
    const cn_h* __restrict__ Xhi = side ? w.BThi : w.AThi;
    const cn_h* __restrict__ Xlo = side ? w.BTlo : w.ATlo;
    const float* __restrict__ Rl = side ? w.lse_c : w.lse_r;       // per contraction row r
    const float* __restrict__ Rs = side ? w.cs : w.rs;
    const float* __restrict__ Cl = side ? w.lse_r : w.lse_c;       // per output row c
    const float* __restrict__ Cs = side ? w.rs : w.cs;
    __shared__ __attribute__((aligned(16))) cn_h Gh[2][CB * CN_GLD], Gl[2][CB * CN_GLD];
    const float g = dloss[0] / (float)n;
    const int fr = lane & 15, fc = lane >> 4;
    const int ec = tid % CB, er = tid / CB;                         // element (r = er + RP q, c = ec) of a 32 x CB step tile
    const int gc = c0 + ec;
    const float cl = gc < n ? Cl[gc] : 0.f, ccs = has_labels ? (gc < n ? Cs[gc] : 0.f) : 1.f, cnp = (inv_np && gc < n) ? inv_np[gc] : 1.f;
    const int dbase = wave * (D / 4);
    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mv[NQ];
    cn_h8 xh[NB], xl[NB];
    auto load_step = [&](int r0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) mv[q] = M[(size_t)(r0 + er + RP * q) * np + gc];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const size_t o = (size_t)(dbase + b * 16 + fr) * np + r0 + fc * 8;
            xh[b] = *reinterpret_cast<const cn_h8*>(Xhi + o);
            xl[b] = *reinterpret_cast<const cn_h8*>(Xlo + o);
        }
    };
    const int nall = np / 32, st0 = (int)((int64_t)nall * rng / RS), st1 = (int)((int64_t)nall * (rng + 1) / RS);
    load_step(st0 * 32);
    for (int st = st0; st < st1; ++st) {
        const int r0 = st * 32, buf = st & 1;
        // dS of this thread's elements, as a half pair, into the TRANSPOSED tile [c][r]
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = er + RP * q, gr = r0 + r;
            float v = 0.f;
            if (gr < n && gc < n) {
                const float s_ = mv[q];
                float L;
                if (has_labels) L = side ? labels[(size_t)gc * n + gr] : labels[(size_t)gr * n + gc];
                else L = gr == gc ? 1.f : 0.f;
                const float rnp = inv_np ? inv_np[gr] : 1.f, rrs = has_labels ? Rs[gr] : 1.f;
                v = g * (cnp * (__expf(s_ - cl) * ccs - L) + rnp * (__expf(s_ - Rl[gr]) * rrs - L));
            }
            const cn_h h = (cn_h)v;
            Gh[buf][ec * CN_GLD + r] = h;
            Gl[buf][ec * CN_GLD + r] = (cn_h)(v - (float)h);
        }
        cn_h8 bh[NB], bl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { bh[b] = xh[b]; bl[b] = xl[b]; }
        if (st + 1 < st1) load_step(r0 + 32);                         // next step's loads fly during this step's MFMAs
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const cn_h8 ah = *reinterpret_cast<const cn_h8*>(&Gh[buf][(a * 16 + fr) * CN_GLD + fc * 8]);
            const cn_h8 al = *reinterpret_cast<const cn_h8*>(&Gl[buf][(a * 16 + fr) * CN_GLD + fc * 8]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    // this range's partial sums -> scratch [side][range][c][d] (cn_project_kernel adds them up and projects)
    float* __restrict__ part = w.part + ((size_t)(side * CN_RS + rng) * np) * D;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + a * 16 + fc * 4 + r;
#pragma unroll
            for (int b = 0; b < NB; ++b) part[(size_t)c * D + dbase + b * 16 + fr] = acc[a][b][r];
        }
}

// sums the ranges' partial tiles IN ORDER (no atomics on the data: repeatable) and projects through the normalisation,
// out[c] = (dxh[c] - xh[c] (xh[c] . dxh[c])) / |x_c|: a wave per row
__global__ __launch_bounds__(256) void cn_project_kernel(Cn w, int RS, int n, int D, float* __restrict__ dA, float* __restrict__ dB) {
    const int side = blockIdx.y, lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), np = w.np;
    if (c >= n) return;
    const float* __restrict__ Xc = side ? w.Ah : w.Bh;
    const float* __restrict__ invc = side ? w.ina : w.inb;
    float* __restrict__ out = side ? dA : dB;
    const float* __restrict__ p0 = w.part + ((size_t)side * CN_RS * np) * D;
    float v[8], x[8], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = lane + 64 * k;
        v[k] = 0.f; x[k] = 0.f;
        if (d < D) {
            for (int z = 0; z < RS; ++z) v[k] += p0[((size_t)z * np + c) * D + d];
            x[k] = Xc[(size_t)c * D + d];
            dot += x[k] * v[k];
        }
    }
    dot = wave_sum(dot);
    const float iv = invc[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = lane + 64 * k;
        if (d < D) out[(size_t)c * D + d] = (v[k] - x[k] * dot) * iv;
    }
}

inline bool cn_ok(int n, int D) {
    static const bool off = xmc_debug_off("no_contrastive_fused");
    return !off && D % 64 == 0 && D <= 512 && n >= 1;
}
}  // namespace

extern "C" int64_t xmc_contrastive_ws_bytes(int n, int D) {
    int64_t n8 = (n + 7) / 8 * 8, n32 = (n + 31) / 32 * 32;
    int64_t f = 4 * align64(n32 * D) + 3 * align64(n32 * n8) + 6 * align64(n) + 64;
    const int64_t fused = cn_ok(n, D) ? cn_bytes(n, D) : 0;
    return f * 4 > fused ? f * 4 : fused;
}

// scratch for label sums lives at the tail of the workspace
static float* label_sum_scratch(const Ws& w, int n) { return w.dBh + align64((int64_t)w.n32 * w.Dp); }

extern "C" int xmc_contrastive_fwd(const float* A, const float* B, const float* labels, const float* inv_num_pos,
                                   int n, int D, float* loss, void* ws, void* stream) {
    if (!A || !B || !loss || !ws) return XMC_EINVAL;
    if (n < 1 || D < 4 || D % 4) return XMC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (cn_ok(n, D)) {
        Cn c = cn_carve(ws, n, D);
        hipLaunchKernelGGL(cn_prepare_kernel, dim3(c.np / 8, 2), dim3(256), 0, st, A, B, c, n, D);
        static const bool no128 = xmc_debug_off("no_contrastive_128");
        if (c.np % 128 == 0 && c.np >= 1024 && !no128) {
            const size_t lds = (size_t)2 * 128 * CN_LD2 * 4;
            XMC_ALLOW_BIG_LDS(cn_scores128_kernel);
            hipLaunchKernelGGL(cn_scores128_kernel, dim3(c.np / 128, c.np / 128), dim3(512), lds, st, c, labels, inv_num_pos, n, D);
        } else {
            hipLaunchKernelGGL(cn_scores_kernel, dim3(c.np / 64, c.np / 64), dim3(256), 0, st, c, labels, inv_num_pos, n, D, loss);
        }
        if (labels) hipLaunchKernelGGL(cn_transpose_kernel, dim3(c.np / 64, c.np / 64), dim3(256), 0, st, c.S, c.ST, c.np);
        hipLaunchKernelGGL(cn_finish_kernel, dim3(1), dim3(256), 0, st, c, inv_num_pos, labels ? 1 : 0, n, loss);
        XMC_LAUNCH_CHECK();
        return 0;
    }
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
    if (cn_ok(n, D)) {
        Cn c = cn_carve(ws, n, D);
        const int hl = labels ? 1 : 0;
        static const bool no_cb64 = xmc_debug_off("no_contrastive_cb64");
        const int cb = (D <= 256 && c.np >= 1024 && !no_cb64) ? 64 : 32;   // 64 output rows per workgroup where the accumulators allow it
        // ranges: up to 8, at least one 32-row step each, two workgroups per CU at most (n = 2048, measured forward + backward: D = 256,
        // 32 output blocks per side: 2 / 4 / 8 ranges 137 / 99 / 92 us; D = 512, 64 blocks: 161 / 150 / 163 us -- the K walk of a workgroup
        // is a chain of memory round trips, so ranges pay until the CUs hold two workgroups each)
        int rs_ = c.np / 32 < CN_RS ? c.np / 32 : CN_RS;
        while (rs_ > 1 && (c.np / cb) * 2 * rs_ > 512) rs_ >>= 1;
        const dim3 grid(c.np / cb, 2, rs_);
#define CN_GRAD(NB_, CB_) hipLaunchKernelGGL((cn_grad_kernel<NB_, CB_>), grid, dim3(256), 0, st, c, labels, inv_num_pos, hl, dloss_dev, n, D, dA, dB)
        switch (D / 64) {
            case 1: if (cb == 64) CN_GRAD(1, 64); else CN_GRAD(1, 32); break;
            case 2: if (cb == 64) CN_GRAD(2, 64); else CN_GRAD(2, 32); break;
            case 4: if (cb == 64) CN_GRAD(4, 64); else CN_GRAD(4, 32); break;
            case 8: CN_GRAD(8, 32); break;
            default: return XMC_ESHAPE;
        }
#undef CN_GRAD
        hipLaunchKernelGGL(cn_project_kernel, dim3((n + 3) / 4, 2), dim3(256), 0, st, c, rs_, n, D, dA, dB);
        XMC_LAUNCH_CHECK();
        return 0;
    }
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
