// Grouped convolutions of the attention-modulation blocks (df_concept_gan.py:146 trans_gconv 3x3, 16 groups of 8 -> 8 channels;
// 267/546 key_gconv 1x1, 8 -> 4 per group), forward and data gradient, bf16.
//
// On the dense kernels these layers run as their block-diagonal expansion: 16x the MACs (a 3x3 128 -> 128 layer at 64x64,
// batch 64, is 77 GFLOP dense = 85 us of MFMA time for 134 MB of tensors = 17 us of HBM time).  Here the block structure is
// kept at the granularity the MFMA offers: one 16-byte pixel unit (8 channels) IS one K block of v_mfma_f32_16x16x32_bf16,
// so a wave that owns 16 output channels (= 2 groups of a 3x3 layer) multiplies only the K blocks (tap, unit) whose input
// channels can reach those outputs: 18 K blocks = 5 MFMAs per 16 pixels instead of 36, half of them useful.  That puts the
// arithmetic (5 us) well under the memory time, so there is no staging at all: a lane's 16-byte unit goes from global memory
// (L1/L2 serve the 9-fold tap re-use) straight into the MFMA B operand, the weights -- read from the SAME packed block-diagonal
// matrix the dense kernels use -- stay in registers as the A operand, and a lane ends up with 4 consecutive output channels of
// one pixel (8-byte stores; the 8 waves of a workgroup cover the 256-byte pixel).
//
// General form: column block cb = output channels [16cb, 16cb+16) belongs to groups g_lo..g_hi, whose input channels are the
// units [u0, u0+nun); K block kbi = j*4 + (lane >> 4) of MFMA j is (tap kbi / nun, unit u0 + kbi % nun), zero weights beyond
// ntaps*nun.  3x3 8->8: nun 2, 5 MFMAs; key 1x1 8->4: nun 4, 1 MFMA; its data gradient (4 -> 8 per group): nun 1, 1 MFMA.
#include "common.h"

namespace {

constexpr int TI = 4;      // 16-pixel tiles in flight per wave

template <int NMF>
__global__ __launch_bounds__(512) void gconv_kernel(const XmcConvDesc d, int nun, int cig, int cog, int chunks) {
    const int lane = threadIdx.x & 63, cb = threadIdx.x >> 6;
    const int col = lane & 15, kb = lane >> 4;
    const int g_lo = (cb * 16) / cog;
    const int u0 = (g_lo * cig) >> 3;
    const int nk = d.ntaps * nun;
    const int csu = d.CS >> 3;                              // 16-byte units per source pixel
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ wpk = reinterpret_cast<const u32x4*>(d.wpk);

    // A operand: weights [16 output channels][K 32]; lane (row = col, K block kb) holds wpk[slice][cb*16 + row][unit*8 .. +8]
    bf16x8 wa[NMF];
    int toff_h[NMF], toff_w[NMF], unit[NMF];
    bool kval[NMF];
#pragma unroll
    for (int j = 0; j < NMF; ++j) {
        const int kbi = j * 4 + kb;
        kval[j] = kbi < nk;
        const int tap = kval[j] ? kbi / nun : 0;
        unit[j] = u0 + (kval[j] ? kbi - tap * nun : 0);
        toff_h[j] = d.dh[0][tap];
        toff_w[j] = d.dw[0][tap];
        u32x4 w = {0, 0, 0, 0};
        if (kval[j]) w = wpk[((size_t)d.wi[0][tap] * d.CDw + cb * 16 + col) * csu + unit[j]];
        wa[j] = __builtin_bit_cast(bf16x8, w);
    }

    const int total = d.N * d.MH * d.MW;
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        u32x4 xb[TI][NMF];
        bool ok[TI][NMF];
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const int p = (chunk * TI + t) * 16 + col;
            const int pc = p < total ? p : total - 1;
            const int n = pc / (d.MH * d.MW), r = pc - n * (d.MH * d.MW);
            const int y = r / d.MW, x = r - y * d.MW;
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                const int sy = y + toff_h[j], sx = x + toff_w[j];
                ok[t][j] = kval[j] && p < total && (unsigned)sy < (unsigned)d.SH && (unsigned)sx < (unsigned)d.SW;
                const int cy = min(max(sy, 0), d.SH - 1), cx = min(max(sx, 0), d.SW - 1);
                xb[t][j] = src[(((size_t)n * d.SH + cy) * d.SW + cx) * csu + unit[j]];
            }
        }
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                const u32x4 z = {0, 0, 0, 0};
                const u32x4 v = ok[t][j] ? xb[t][j] : z;
                acc = XMC_MFMA_16x16x32(wa[j], __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
            }
            // D[row = output channel kb*4 + i][col = pixel]
            const int p = (chunk * TI + t) * 16 + col;
            if (p < total) {
                const size_t e = (size_t)p * d.CD + cb * 16 + kb * 4;
                if (d.res) {                                   // residual in the destination layout (another gradient of the same tensor)
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const xmc_h16*>(d.res) + e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += (float)r[i];
                }
                bf16x4 o = {(xmc_h16)acc[0], (xmc_h16)acc[1], (xmc_h16)acc[2], (xmc_h16)acc[3]};
                *reinterpret_cast<bf16x4*>(reinterpret_cast<xmc_h16*>(d.dst) + e) = o;
            }
        }
    }
}

// 1x1 grouped layers with one MFMA per 16-channel column block (key / query 1x1: 8 -> 4 per group, and its data gradient 4 -> 8), Cout a
// multiple of 64.  In the general kernel a wave owns ONE column block, so a pixel's 128 / 256 output bytes are written by 4 / 8 different
// waves, 32 bytes each, at different times (the data gradient ran at 2.7 TB/s against the forward's 3.9), and the data gradient loaded the
// same 16-byte unit in all four lane groups.  Here a wave owns FOUR consecutive column blocks (one 128-byte line of every pixel, written
// by four back-to-back stores) and the waves of a workgroup split the pixels.  nun == 4 (forward): one load per column block, all four
// K blocks live.  nun == 1 (data gradient): the four column blocks read four CONSECUTIVE units, so lane group kb loads unit u0 + kb ONCE
// and column block b multiplies it with a weight fragment that is zero outside K block b.
template <int NUN>
__global__ __launch_bounds__(256) void gconv1x1_kernel(const XmcConvDesc d, int cig, int cog, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kb = lane >> 4;
    const int csu = d.CS >> 3;
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ wpk = reinterpret_cast<const u32x4*>(d.wpk);
    const int ng4 = d.CD >> 6;                                // 64-channel groups of column blocks
    const int total = d.N * d.MH * d.MW;
    for (int g4 = 0; g4 < ng4; ++g4) {
        bf16x8 wa[4];
        int unit[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int cb = g4 * 4 + b;
            const int u0 = (((cb * 16) / cog) * cig) >> 3;
            unit[b] = u0 + (NUN == 1 ? 0 : kb);
            u32x4 w = {0, 0, 0, 0};
            if (NUN == 1 ? kb == b : true) w = wpk[((size_t)d.wi[0][0] * d.CDw + cb * 16 + col) * csu + unit[b]];
            wa[b] = __builtin_bit_cast(bf16x8, w);
        }
        const int ubase = unit[0] + (NUN == 1 ? kb : 0);       // NUN == 1: lane group kb carries the unit of column block kb
        for (int t0 = (blockIdx.x * 4 + wave) * TI; t0 < tiles; t0 += gridDim.x * 4 * TI) {
            u32x4 xb[TI][NUN == 1 ? 1 : 4];
            bf16x4 rv[TI][4];                                 // the residual (another gradient of the same tensor), requested with the operands
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int p = (t0 + t) * 16 + col;
                const size_t pq = (size_t)(p < total ? p : total - 1);
                const size_t pc = pq * csu;
                if (NUN == 1) xb[t][0] = src[pc + ubase];
                else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) xb[t][b] = src[pc + unit[b]];
                }
                if (d.res) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        rv[t][b] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const xmc_h16*>(d.res) + pq * d.CD + (g4 * 4 + b) * 16 + kb * 4);
                }
            }
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int p = (t0 + t) * 16 + col;
                if (t0 + t >= tiles) break;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = XMC_MFMA_16x16x32(wa[b], __builtin_bit_cast(bf16x8, xb[t][NUN == 1 ? 0 : b]), acc, 0, 0, 0);
                    if (p < total) {
                        const size_t e = (size_t)p * d.CD + (g4 * 4 + b) * 16 + kb * 4;
                        if (d.res) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] += (float)rv[t][b][i];
                        }
                        bf16x4 o = {(xmc_h16)acc[0], (xmc_h16)acc[1], (xmc_h16)acc[2], (xmc_h16)acc[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<xmc_h16*>(d.dst) + e) = o;
                    }
                }
            }
        }
    }
}

// 3x3 (taps within [-1, 1]^2) on maps with H % 8 == 0, W % 16 == 0: the direct form above pulls every unit through the
// vector-memory path once per tap (9x, ~22 B/clk per CU: 104 us per call on average where the dense kernel took 111), so the
// 8 x 16-pixel tile's halo patch (10 x 18 pixels, each byte once + 41 % halo) is staged in LDS and the B operands are
// ds_read_b128s at the tap-shifted pixel; pixel stride = channels + 16 bytes, which spreads the 16 pixels of a lane group over
// all banks.  Persistent workgroups, the next tile's patch prefetched into registers during the MFMAs; 46 KB of LDS, two or three
// workgroups per CU cover each other's staging.
constexpr int GT_H = 8, GT_W = 16, GP_H = GT_H + 2, GP_W = GT_W + 2;

template <int NW>      // waves = 16-wide output-channel blocks
__global__ __launch_bounds__(NW * 64, 2) void gconv_patch_kernel(const XmcConvDesc d, int nun, int cig, int cog, int tiles_x,
                                                                 int tiles_y, int ntiles) {
    constexpr int NT_ = NW * 64, NMF = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, cb = threadIdx.x >> 6, tid = threadIdx.x;
    const int col = lane & 15, kb = lane >> 4;
    const int csu = d.CS >> 3, pstr = d.CS * 2 + 16;
    const int g_lo = (cb * 16) / cog, u0 = (g_lo * cig) >> 3, nk = d.ntaps * nun;
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ wpk = reinterpret_cast<const u32x4*>(d.wpk);
    bf16x8 wa[NMF];
    int loff[NMF];
#pragma unroll
    for (int j = 0; j < NMF; ++j) {
        const int kbi = j * 4 + kb;
        const bool kv = kbi < nk;
        const int tap = kv ? kbi / nun : 0;
        const int unit = u0 + (kv ? kbi - tap * nun : 0);
        loff[j] = ((d.dh[0][tap] + 1) * GP_W + d.dw[0][tap] + 1 + col) * pstr + unit * 16;
        u32x4 w = {0, 0, 0, 0};
        if (kv) w = wpk[((size_t)d.wi[0][tap] * d.CDw + cb * 16 + col) * csu + unit];
        wa[j] = __builtin_bit_cast(bf16x8, w);
    }
    const int nunits = GP_H * GP_W * csu;
    constexpr int MAXU = (GP_H * GP_W * 16 + NT_ - 1) / NT_;      // csu <= 16
    u32x4 pv[MAXU];
    auto prefetch = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * GT_H - 1, x0 = (tr % tiles_x) * GT_W - 1;
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int id = tid + it * NT_;
            const int pp = id / csu, ch = id - pp * csu;
            const int py = pp / GP_W, px = pp - py * GP_W;
            const int sy = y0 + py, sx = x0 + px;
            const bool ok = id < nunits && (unsigned)sy < (unsigned)d.SH && (unsigned)sx < (unsigned)d.SW;
            const u32x4 z = {0, 0, 0, 0};
            pv[it] = ok ? src[(((size_t)n * d.SH + sy) * d.SW + sx) * csu + ch] : z;
        }
    };
    const XcdWalk xw = xmc_xcd_walk(ntiles);
    int tile = xw.first;
    if (tile < xw.end) prefetch(tile);
    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();                                          // the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int id = tid + it * NT_;
            const int pp = id / csu, ch = id - pp * csu;
            if (id < nunits) *reinterpret_cast<u32x4*>(smem + pp * pstr + ch * 16) = pv[it];
        }
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * GT_H, x0 = (tr % tiles_x) * GT_W;
#pragma unroll
        for (int r = 0; r < GT_H; ++r) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NMF; ++j) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + r * GP_W * pstr + loff[j]);
                acc = XMC_MFMA_16x16x32(wa[j], b, acc, 0, 0, 0);
            }
            const size_t e = (((size_t)n * d.DH + y0 + r) * d.DW + x0 + col) * d.CD + cb * 16 + kb * 4;
            if (d.res) {
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const xmc_h16*>(d.res) + e);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += (float)rr[i];
            }
            bf16x4 o = {(xmc_h16)acc[0], (xmc_h16)acc[1], (xmc_h16)acc[2], (xmc_h16)acc[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<xmc_h16*>(d.dst) + e) = o;
        }
    }
}

}  // namespace

// 0 = launched, 1 = not this kernel's case, < 0 = error.  `groups` block-diagonal weight in the dense packed layout.
int xmc_conv_group_try(const XmcConvDesc* d, void* stream) {
    static const bool off = xmc_debug_off("no_gconv");
    const int G = d->groups;
    if (off || G <= 1) return 1;
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16) return 1;
    if (d->nclass != 1 || d->SA != 1 || d->DA != 1 || d->src_shift != 0 || d->dph[0] != 0 || d->dpw[0] != 0) return 1;
    if (d->bias || d->alpha_dev || d->mask || d->dst2 || d->dst_pool || d->post_act || d->sign_bits || d->dot || d->act != XMC_ACT_NONE) return 1;
    if (d->res && (d->res_mode != 0 || (d->res_scale != 0.f && d->res_scale != 1.f))) return 1;
    if (d->MH != d->DH || d->MW != d->DW || d->MH != d->SH || d->MW != d->SW) return 1;
    if (d->CS % G || d->CD % G || d->CD % 16 || d->CS % 8 || d->CD > 128) return 1;
    const int cig = d->CS / G, cog = d->CD / G;
    if (cog > 16 || 16 % cog) return 1;                          // a column block holds whole groups
    const int gpb = 16 / cog;                                     // groups per column block
    if ((gpb * cig) % 8) return 1;                                // ... whose input channels are whole 16-byte units
    const int nun = gpb * cig / 8;
    const int nk = d->ntaps * nun, nmf = (nk + 3) / 4;
    if (nun > 4 || (nmf != 1 && nmf != 5 && nmf != 3)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool no_patch = xmc_debug_off("no_gconv_patch");
    if (!no_patch && nmf == 5 && d->ntaps == 9 && d->MH % GT_H == 0 && d->MW % GT_W == 0 && d->CS <= 128 && (d->CD == 128 || d->CD == 64)) {
        bool in3 = true;
        for (int k = 0; k < 9; ++k) in3 = in3 && d->dh[0][k] >= -1 && d->dh[0][k] <= 1 && d->dw[0][k] >= -1 && d->dw[0][k] <= 1;
        if (in3) {
            const int tx = d->MW / GT_W, ty = d->MH / GT_H, ntiles = d->N * tx * ty;
            const size_t lds = (size_t)GP_H * GP_W * (d->CS * 2 + 16);
            const int grid = ntiles < 256 * 3 ? ntiles : 256 * 3;
            if (d->CD == 128) hipLaunchKernelGGL(gconv_patch_kernel<8>, dim3(xmc_ab_grid(grid)), dim3(512), lds, st, *d, nun, cig, cog, tx, ty, ntiles);
            else hipLaunchKernelGGL(gconv_patch_kernel<4>, dim3(xmc_ab_grid(grid)), dim3(256), lds, st, *d, nun, cig, cog, tx, ty, ntiles);
            xmc_note_kernel("gconv_patch_kernel<%d>", d->CD / 16);
            XMC_LAUNCH_CHECK();
            return 0;
        }
    }
    const int64_t total = (int64_t)d->N * d->MH * d->MW;
    static const bool no_1x1 = xmc_debug_off("no_gconv1x1");
    if (!no_1x1 && d->ntaps == 1 && nmf == 1 && d->CD % 64 == 0 && (nun == 1 || nun == 4) && d->dh[0][0] == 0 && d->dw[0][0] == 0 &&
        total < (1ll << 27)) {
        bool consecutive = true;                              // nun == 1: the four column blocks of a wave read consecutive units
        if (nun == 1)
            for (int cb = 0; cb < d->CD / 16; ++cb) consecutive = consecutive && ((((cb * 16) / cog) * cig) >> 3) == cb;
        if (consecutive) {
            const int tiles = (int)((total + 15) / 16);
            int grid = (tiles + 4 * TI - 1) / (4 * TI);
            if (grid > 256 * 8) grid = 256 * 8;
            if (nun == 1) hipLaunchKernelGGL(gconv1x1_kernel<1>, dim3(grid), dim3(256), 0, st, *d, cig, cog, tiles);
            else hipLaunchKernelGGL(gconv1x1_kernel<4>, dim3(grid), dim3(256), 0, st, *d, cig, cog, tiles);
            xmc_note_kernel("gconv1x1_kernel<%d>", nun);
            XMC_LAUNCH_CHECK();
            return 0;
        }
    }
    const int chunks = (int)((total + 16 * TI - 1) / (16 * TI));
    const int threads = d->CD / 16 * 64;
    int grid = chunks < 256 * 8 ? chunks : 256 * 8;
    if (nmf == 1) hipLaunchKernelGGL(gconv_kernel<1>, dim3(grid), dim3(threads), 0, st, *d, nun, cig, cog, chunks);
    else if (nmf == 3) hipLaunchKernelGGL(gconv_kernel<3>, dim3(grid), dim3(threads), 0, st, *d, nun, cig, cog, chunks);
    else hipLaunchKernelGGL(gconv_kernel<5>, dim3(grid), dim3(threads), 0, st, *d, nun, cig, cog, chunks);
    xmc_note_kernel("gconv_kernel<%d>", nmf);
    XMC_LAUNCH_CHECK();
    return 0;
}
