// Convolution whose SOURCE has 8 stored channels (an image: 3 channels padded to 8), unit stride, bf16 -- the discriminator's
// first layer (df_gan.py:130 `conv_img`, 3 -> ndf) and the data gradient of the generator's last layer (df_gan.py:86-90,
// ngf <- 3).  The generic implicit-GEMM kernel spends one 64-byte K sub-step per tap on 16 bytes of data and re-gathers every
// pixel nine times (measured 29 TFLOP/s, 4x the HBM time of the layer); this kernel is built for what the layer is, a stream:
//
//   * a persistent workgroup walks 8x32-pixel output tiles; the (8+2)x(32+2) source patch is 5.4 KB of LDS;
//   * K = taps x 8 channels: one 16-byte unit of a pixel IS one lane's 8 K-values of a 16x16x32 MFMA operand, so a 3x3 kernel
//     is three MFMA K-steps (taps 0-3, 4-7, 8 + zero weights) whose pixel operand is a single ds_read_b128 at the tap-shifted
//     patch position; the weights (<= 12 fragments) stay in registers for the life of the workgroup;
//   * MFMA roles as in conv_tile.hip's ptile3: A = weight rows (permuted while loading), B = pixels, so each lane ends up with
//     BN/4 consecutive output channels of one pixel and stores 16-byte units straight from registers.
// HBM traffic = source once (+6 % halo) + destination once.
#include "common.h"
#include <stdlib.h>

namespace {

struct ThinCfg {
    int tiles_y, tiles_x, PW, PH, dh0, dw0;
};

template <int BN>
__global__ __launch_bounds__(256) void thin_in_kernel(const XmcConvDesc d, const ThinCfg t, int ntiles) {
    constexpr int TH = 8, TW = 32, TN = BN / 16, UPL = BN / 32, KS = 3;
    __shared__ __attribute__((aligned(16))) unsigned char patch[(TH + 4) * (TW + 4) * 16];
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int PW = t.PW, PH = t.PH, dh0 = t.dh0, dw0 = t.dw0;
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // weights -> registers.  K-step ks, lane group fc <-> tap 4*ks + fc; physical row (j, q) <-> channel (q/4)*(BN/4) + j*4 + q%4
    u32x4 wf[KS][TN];
    int toff[KS];                                 // byte offset of this lane's tap in the patch, per K-step
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks * 4 + fc;
        const bool live = tap < d.ntaps;
        const int tt = live ? tap : 0;
        toff[ks] = ((d.dh[0][tt] - dh0) * PW + (d.dw[0][tt] - dw0)) * 16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int lrow = (fr >> 2) * (BN / 4) + j * 4 + (fr & 3);
            wf[ks][j] = live ? w16[(size_t)d.wi[0][tt] * d.CDw + lrow] : u32x4{0, 0, 0, 0};
        }
    }
    // patch staging: thread -> up to 2 patch pixels; halo bits as in ptile3 (1 top, 2 bottom, 4 left, 8 right)
    int ppix[2], psrc[2];
    unsigned halo[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pp = tid + it * 256;
        const int py = pp / PW, px = pp - py * PW;
        const bool in = pp < PH * PW;
        ppix[it] = in ? pp : -1;
        psrc[it] = in ? (dh0 + py) * d.SW + (dw0 + px) : 0;
        halo[it] = !in ? 0u : ((py < -dh0 ? 1u : 0u) | (py >= TH - dh0 ? 2u : 0u) | (px < -dw0 ? 4u : 0u) | (px >= TW - dw0 ? 8u : 0u));
    }
    int pbase[4], eoff[4];
    const int cd8 = d.CD / 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wm * 2 + (i >> 1), c = (i & 1) * 16 + fr;
        pbase[i] = (r * PW + c) * 16;
        eoff[i] = (r * d.DW + c) * cd8 + fc * UPL;
    }
    const int ch0 = fc * (BN / 4);
    float bias8[UPL][8];
#pragma unroll
    for (int u = 0; u < UPL; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) bias8[u][c] = (d.bias && ch0 + u * 8 < d.CD) ? d.bias[ch0 + u * 8 + c] : 0.f;
    const float slope = d.act == XMC_ACT_LRELU ? XMC_LRELU : 1.f;
    const int tpi = t.tiles_y * t.tiles_x;
    bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
    const bf16x8* __restrict__ mask8 = reinterpret_cast<const bf16x8*>(d.mask);
    const float pscale = d.pool_scale == 0.f ? 0.25f : d.pool_scale;

    u32x4 pv[2];
    unsigned okbits = 0;
    auto issue = [&](int tile) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int ty = trem / t.tiles_x, tx = trem - ty * t.tiles_x;
        const int a0 = ty * TH, b0 = tx * TW;
        const int base = (img * d.SH + a0) * d.SW + b0;
        const unsigned border = (a0 + dh0 < 0 ? 1u : 0u) | (a0 + PH + dh0 > d.SH ? 2u : 0u) | (b0 + dw0 < 0 ? 4u : 0u) | (b0 + PW + dw0 > d.SW ? 8u : 0u);
        okbits = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool ok = (halo[it] & border) == 0;
            pv[it] = src16[(unsigned)(base + (ok ? psrc[it] : 0))];
            okbits |= ok ? (1u << it) : 0u;
        }
    };
    const XcdWalk xw = xmc_xcd_walk(ntiles);
    int tile = xw.first;
    if (tile < xw.end) issue(tile);
    for (; tile < xw.end; tile += xw.step) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int ty = trem / t.tiles_x, tx = trem - ty * t.tiles_x;
        __syncthreads();                          // previous tile's reads are done
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            u32x4 v = pv[it];
            if (!((okbits >> it) & 1)) v = u32x4{0, 0, 0, 0};
            if (ppix[it] >= 0) *reinterpret_cast<u32x4*>(patch + ppix[it] * 16) = v;
        }
        __syncthreads();
        if (tile + xw.step < xw.end) issue(tile + xw.step);
        const int dbase = ((img * d.DH + ty * TH) * d.DW + tx * TW) * cd8;
        // pixel blocks i and i+2 are vertical neighbours (rows 2*wm and 2*wm+1, same columns) of the same lane: they are
        // finished together so that the optional third output (2x2 average of the rounded result, XmcConvDesc.dst_pool) is a
        // sum of two registers plus one exchange with lane ^ 1
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            float fin[2][UPL][8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = ip + 2 * h;
                f32x4 acc[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 pf = *reinterpret_cast<const u32x4*>(patch + pbase[i] + toff[ks]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, wf[ks][j]), __builtin_bit_cast(bf16x8, pf), acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 8 >= d.CD) continue;
                    bf16x8 o;
                    if (mask8 == nullptr) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x0 = acc[2 * u][q] + bias8[u][q], x1 = acc[2 * u + 1][q] + bias8[u][4 + q];
                            o[q] = (xmc_h16)fmaxf(x0, x0 * slope);
                            o[4 + q] = (xmc_h16)fmaxf(x1, x1 * slope);
                        }
                    } else {              // data gradient of the layer above a LeakyReLU: times LeakyReLU'(mask), mask in the dst layout
                        const bf16x8 mk = mask8[dbase + eoff[i] + u];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x0 = acc[2 * u][q] + bias8[u][q], x1 = acc[2 * u + 1][q] + bias8[u][4 + q];
                            o[q] = (xmc_h16)(fmaxf(x0, x0 * slope) * lrelu_slope((float)mk[q]));
                            o[4 + q] = (xmc_h16)(fmaxf(x1, x1 * slope) * lrelu_slope((float)mk[4 + q]));
                        }
                    }
                    dst8[dbase + eoff[i] + u] = o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) fin[h][u][q] = (float)o[q];
                }
            }
            if (d.dst_pool) {
                bf16x8* __restrict__ pool8 = reinterpret_cast<bf16x8*>(d.dst_pool);
                const int prow = ty * (TH / 2) + wm, pcol = tx * (TW / 2) + ((ip * 16 + fr) >> 1);
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 8 >= d.CD) continue;
                    bf16x8 o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float sm = fin[0][u][q] + fin[1][u][q];
                        sm += xmc_xor1(sm);
                        o[q] = (xmc_h16)(pscale * sm);
                    }
                    if ((fr & 1) == 0) pool8[((img * (d.DH >> 1) + prow) * (d.DW >> 1) + pcol) * cd8 + fc * UPL + u] = o;
                }
            }
        }
    }
}


// 1x1 convolution with few channels (the discriminator's shortcut convolutions on the pooled input, df_gan.py:280,286-291, and
// their data gradients): nothing to stage -- a lane's 16-byte unit of a pixel IS its K-fragment, so the source goes straight
// from global memory into the MFMA B operand (16 pixels x 64 bytes per wave load, fully coalesced), the weights (<= 16
// fragments) live in registers, and the result leaves in 64-byte-per-pixel contiguous stores.  Pure stream: HBM bound.
// By-product for the backward pass of a discriminator block (ops.ResDBwdFn): the kernel that streams `dout` as the SOURCE
// of the shortcut's data gradient also writes `dout x LeakyReLU'(branch)` from the block's sign bits (MASKED: src_masked[unit] =
// src[unit] with the lanes of cleared bits times `slope`, rounded like xmc_signmask_apply) -- the separate mask pass read dout again.
__device__ __forceinline__ u32x4 mask_unit(u32x4 v, unsigned bits, float slope) {
    bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float f = (float)h[k];
        h[k] = (xmc_h16)(((bits >> k) & 1u) ? f : slope * f);
    }
    return __builtin_bit_cast(u32x4, h);
}

// SPLIT (round 5, XmcConvDesc.wpk_lo): the weights are the pair hi + lo = round16(w) + round16(w - round16(w)); a second register
// set and a second MFMA per K step into the same accumulator (the launch is bound by its HBM stream, not by its matrix work).
template <int KS, int TN, bool MASKED = false, bool SPLIT = false>   // KS = Cin / 32 K-steps, TN = Cout(padded) / 16 row blocks
__global__ __launch_bounds__(256) void pw1x1_kernel(const XmcConvDesc d, int ngroups, const unsigned char* __restrict__ sbits, u32x4* __restrict__ smasked,
                                                    float mslope) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fc = lane >> 4;
    const int cs_units = d.CS >> 3, cd8 = d.CD >> 3;
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);
    bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
    u32x4 wf[KS][TN], wl[SPLIT ? KS : 1][SPLIT ? TN : 1];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int lrow = (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3);      // see conv_tile.hip: 64-byte stores per pixel
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wf[ks][j] = w16[((size_t)d.wi[0][0] * d.CDw + lrow) * cs_units + ks * 4 + fc];
            if (SPLIT) wl[ks][j] = reinterpret_cast<const u32x4*>(d.wpk_lo)[((size_t)d.wi[0][0] * d.CDw + lrow) * cs_units + ks * 4 + fc];
        }
    }
    float bias8[TN / 2][8];
#pragma unroll
    for (int u = 0; u < TN / 2; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) bias8[u][c] = (d.bias && u * 32 + fc * 8 < d.CD) ? d.bias[u * 32 + fc * 8 + c] : 0.f;
    const float slope = d.act == XMC_ACT_LRELU ? XMC_LRELU : 1.f;
    const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int UNR = 4;                     // pixel groups in flight per wave
    for (int g0 = wave_g * UNR; g0 < ngroups; g0 += nwaves * UNR) {
        u32x4 pf[UNR][KS];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int g = g0 + r < ngroups ? g0 + r : ngroups - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pf[r][ks] = src16[(size_t)(g * 16 + fr) * cs_units + ks * 4 + fc];
        }
        if (MASKED) {
#pragma unroll
            for (int r = 0; r < UNR; ++r) {
                if (g0 + r >= ngroups) break;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const size_t idx = (size_t)((g0 + r) * 16 + fr) * cs_units + ks * 4 + fc;
                    smasked[idx] = mask_unit(pf[r][ks], sbits[idx], mslope);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            if (g0 + r >= ngroups) break;
            f32x4 acc[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SPLIT) acc[j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, wl[ks][j]), __builtin_bit_cast(bf16x8, pf[r][ks]), acc[j], 0, 0, 0);
                    acc[j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, wf[ks][j]), __builtin_bit_cast(bf16x8, pf[r][ks]), acc[j], 0, 0, 0);
                }
            const size_t pix = (size_t)(g0 + r) * 16 + fr;
#pragma unroll
            for (int u = 0; u < TN / 2; ++u) {
                if (u * 32 + fc * 8 >= d.CD) continue;
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x0 = acc[2 * u][q] + bias8[u][q], x1 = acc[2 * u + 1][q] + bias8[u][4 + q];
                    o[q] = (xmc_h16)fmaxf(x0, x0 * slope);
                    o[4 + q] = (xmc_h16)fmaxf(x1, x1 * slope);
                }
                dst8[pix * cd8 + u * 4 + fc] = o;
            }
        }
    }
}

template <int KS, int TN>
int launch_pw(const XmcConvDesc& d, int ngroups, hipStream_t st, const unsigned char* sbits, void* smasked, float mslope) {
    int nb = (ngroups + 4 * 4 - 1) / (4 * 4);           // 4 waves per block, 4 groups per wave and iteration
    if (nb > 256 * 8) nb = 256 * 8;
    if (d.wpk_lo) {      // the pair form, forward: D's 64 -> 128 shortcut, G's 128 -> 64 and 64 -> 32 ones (the wider ones: launch_pww)
        if constexpr ((KS == 2 && TN == 8) || (KS == 4 && TN == 4) || (KS == 2 && TN == 2)) {
            if (smasked) return 1;
            hipLaunchKernelGGL((pw1x1_kernel<KS, TN, false, true>), dim3(nb), dim3(256), 0, st, d, ngroups, nullptr, nullptr, 0.f);
            xmc_note_kernel("pw1x1_kernel<%d, %d, false, true>", KS, TN);
            XMC_LAUNCH_CHECK();
            return 0;
        } else {
            return 1;
        }
    }
    if (smasked) hipLaunchKernelGGL((pw1x1_kernel<KS, TN, true>), dim3(nb), dim3(256), 0, st, d, ngroups, sbits, (u32x4*)smasked, mslope);
    else hipLaunchKernelGGL((pw1x1_kernel<KS, TN, false>), dim3(nb), dim3(256), 0, st, d, ngroups, nullptr, nullptr, 0.f);
    xmc_note_kernel(smasked ? "pw1x1_kernel<%d, %d, true>" : "pw1x1_kernel<%d, %d>", KS, TN);
    XMC_LAUNCH_CHECK();
    return 0;
}

// 1x1 convolution with MANY channels (the learned shortcuts of the deeper discriminator blocks, 128 -> 256 and 256 -> 512, and their
// data gradients): the weights no longer fit in registers, so they sit in LDS -- the workgroup's column slice, staged once, rows
// padded by 16 bytes (16 consecutive rows then cover every bank exactly once per ds_read_b128 of a lane group) -- and everything
// else is pw1x1_kernel: the source goes from global memory straight into the MFMA B operand, four 16-pixel groups in flight per
// wave, each weight fragment read once per 64 pixels, 64-byte-per-pixel stores.  8 waves per workgroup (2 per SIMD) share the
// slice.  85-170 FLOP/byte: an HBM stream (the gather kernel ran these at 1.7-2.6 TB/s: two K steps of prologue and an LDS
// transposition of the result per 256x256 tile).
template <int KS, bool MASKED = false, bool SPLIT = false>   // KS = Cin / 32 K-steps (4 or 8); MASKED: see pw1x1_kernel (the first column slice writes it)
__global__ __launch_bounds__(512) void pw1x1w_kernel(const XmcConvDesc d, int ngroups, int ncols, const unsigned char* __restrict__ sbits,
                                                     u32x4* __restrict__ smasked, float mslope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
    constexpr int RSTR = KS * 64 * (SPLIT ? 2 : 1) + 16;         // bytes per weight row (SPLIT: the row's hi half, then its lo half)
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fc = lane >> 4;
    const int cs_units = d.CS >> 3, cd8 = d.CD >> 3;
    const int c0 = blockIdx.y * ncols;         // first output channel of this workgroup's slice
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);
    bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
    // physical row (block j, r) <- logical channel (j/2)*32 + (r/4)*8 + (j%2)*4 + r%4 (conv_tile.hip: lane group fc ends with 8
    // consecutive channels in its two accumulator blocks -> 16-byte stores, 64 bytes per pixel per 32 channels)
    for (int id = tid; id < ncols * KS * 4; id += 512) {
        const int row = id / (KS * 4), ch = id - row * (KS * 4);
        const int j = row >> 4, r = row & 15;
        const int lrow = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        *reinterpret_cast<u32x4*>(wlds + row * RSTR + ch * 16) = w16[((size_t)d.wi[0][0] * d.CDw + c0 + lrow) * cs_units + ch];
        if (SPLIT) *reinterpret_cast<u32x4*>(wlds + row * RSTR + KS * 64 + ch * 16) =
                reinterpret_cast<const u32x4*>(d.wpk_lo)[((size_t)d.wi[0][0] * d.CDw + c0 + lrow) * cs_units + ch];
    }
    __syncthreads();
    const float slope = d.act == XMC_ACT_LRELU ? XMC_LRELU : 1.f;
    const int wave_g = blockIdx.x * 8 + (tid >> 6), nwaves = gridDim.x * 8;
    constexpr int UNR = 4;
    const int nu = ncols >> 5;                 // 32-channel units of the slice
    for (int g0 = wave_g * UNR; g0 < ngroups; g0 += nwaves * UNR) {
        u32x4 pf[UNR][KS];
#pragma unroll
        for (int r = 0; r < UNR; ++r) {
            const int g = g0 + r < ngroups ? g0 + r : ngroups - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pf[r][ks] = src16[(size_t)(g * 16 + fr) * cs_units + ks * 4 + fc];
        }
        if (MASKED && blockIdx.y == 0) {
#pragma unroll
            for (int r = 0; r < UNR; ++r) {
                if (g0 + r >= ngroups) break;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const size_t idx = (size_t)((g0 + r) * 16 + fr) * cs_units + ks * 4 + fc;
                    smasked[idx] = mask_unit(pf[r][ks], sbits[idx], mslope);
                }
            }
        }
        for (int u = 0; u < nu; ++u) {
            f32x4 acc[UNR][2];
#pragma unroll
            for (int r = 0; r < UNR; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* const wr = wlds + (u * 32 + fr) * RSTR + fc * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wa = *reinterpret_cast<const bf16x8*>(wr + ks * 64), wb = *reinterpret_cast<const bf16x8*>(wr + 16 * RSTR + ks * 64);
                if (SPLIT) {
                    const bf16x8 la = *reinterpret_cast<const bf16x8*>(wr + KS * 64 + ks * 64), lb = *reinterpret_cast<const bf16x8*>(wr + 16 * RSTR + KS * 64 + ks * 64);
#pragma unroll
                    for (int r = 0; r < UNR; ++r) {
                        acc[r][0] = XMC_MFMA_16x16x32(la, __builtin_bit_cast(bf16x8, pf[r][ks]), acc[r][0], 0, 0, 0);
                        acc[r][1] = XMC_MFMA_16x16x32(lb, __builtin_bit_cast(bf16x8, pf[r][ks]), acc[r][1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < UNR; ++r) {
                    acc[r][0] = XMC_MFMA_16x16x32(wa, __builtin_bit_cast(bf16x8, pf[r][ks]), acc[r][0], 0, 0, 0);
                    acc[r][1] = XMC_MFMA_16x16x32(wb, __builtin_bit_cast(bf16x8, pf[r][ks]), acc[r][1], 0, 0, 0);
                }
            }
            const int ch = c0 + u * 32 + fc * 8;
            if (ch >= d.CD) continue;
            float b8[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) b8[c] = d.bias ? d.bias[ch + c] : 0.f;
#pragma unroll
            for (int r = 0; r < UNR; ++r) {
                if (g0 + r >= ngroups) break;
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x0 = acc[r][0][q] + b8[q], x1 = acc[r][1][q] + b8[4 + q];
                    o[q] = (xmc_h16)fmaxf(x0, x0 * slope);
                    o[4 + q] = (xmc_h16)fmaxf(x1, x1 * slope);
                }
                dst8[((size_t)(g0 + r) * 16 + fr) * cd8 + (ch >> 3)] = o;
            }
        }
    }
}

template <int KS>
int launch_pww(const XmcConvDesc& d, int ngroups, hipStream_t st, const unsigned char* sbits, void* smasked, float mslope) {
    // the column slice of a workgroup: as many 32-channel units as fit in LDS, dividing CDw evenly (256 x 512: two slices of 135 KB)
    const bool split = d.wpk_lo != nullptr;
    if (split && smasked) return 1;
    const int RSTR = KS * 64 * (split ? 2 : 1) + 16;
    int ny = 1;
    while ((size_t)(d.CDw / ny) * RSTR > XMC_MAX_DYN_LDS || d.CDw % ny != 0 || (d.CDw / ny) % 32 != 0) {
        if (++ny > d.CDw / 32) return 1;
    }
    const int ncols = d.CDw / ny;
    const size_t lds = (size_t)ncols * RSTR;
    int nb = (ngroups + 8 * 4 - 1) / (8 * 4);           // 8 waves per block, 4 groups per wave and iteration
    if (nb > 256 / ny) nb = 256 / ny;                   // one workgroup per CU (the slice is staged once per workgroup)
    if (nb < 1) nb = 1;
    if (split) {
        XMC_ALLOW_BIG_LDS((pw1x1w_kernel<KS, false, true>));
        hipLaunchKernelGGL((pw1x1w_kernel<KS, false, true>), dim3(nb, ny), dim3(512), lds, st, d, ngroups, ncols, nullptr, nullptr, 0.f);
        xmc_note_kernel("pw1x1w_kernel<%d, false, true>", KS);
        XMC_LAUNCH_CHECK();
        return 0;
    }
    if (smasked) {
        XMC_ALLOW_BIG_LDS((pw1x1w_kernel<KS, true>));
        hipLaunchKernelGGL((pw1x1w_kernel<KS, true>), dim3(nb, ny), dim3(512), lds, st, d, ngroups, ncols, sbits, (u32x4*)smasked, mslope);
    } else {
        XMC_ALLOW_BIG_LDS((pw1x1w_kernel<KS, false>));
        hipLaunchKernelGGL((pw1x1w_kernel<KS, false>), dim3(nb, ny), dim3(512), lds, st, d, ngroups, ncols, nullptr, nullptr, 0.f);
    }
    xmc_note_kernel(smasked ? "pw1x1w_kernel<%d, true>" : "pw1x1w_kernel<%d>", KS);
    XMC_LAUNCH_CHECK();
    return 0;
}

// ---- 8 stored output channels (conv_out of the generator: 32 -> 3, df_gan.py:85-87, + bias + tanh).  On the 32-wide tile
// kernel this layer multiplies 32 output channels for the 8 it stores and is bound by that waste (0.66 ms where its 1.34 GB
// of tensors take 0.27 ms).  Here the output channels are the 16 rows of the MFMA (8 used), the B operand is the lane's
// 16-byte unit of the tap-shifted pixel read from a small LDS halo patch (pixel stride = channels + 16 bytes: conflict-free
// ds_read_b128), the weights (9 taps x Cin/32 fragments) stay in registers; 4 waves x 2 rows of an 8 x 16-pixel tile,
// 14-29 KB of LDS, persistent with the next patch prefetched into registers.
constexpr int TO_H = 8, TO_W = 16, TO_PH = TO_H + 2, TO_PW = TO_W + 2;
template <int KC>      // 32-channel K chunks per tap (Cin = 32 * KC)
__global__ __launch_bounds__(256, 2) void thin_out_kernel(const XmcConvDesc d, int tiles_x, int tiles_y, int ntiles) {
    constexpr int NKS = 9 * KC, CSU = 4 * KC, PSTR = CSU * 16 + 16;
    constexpr int NUN = TO_PH * TO_PW * CSU, MAXU = (NUN + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TO_PH * TO_PW * PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kb = lane >> 4;
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ wpk = reinterpret_cast<const u32x4*>(d.wpk);
    bf16x8 wa[NKS];
    int loff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        loff[t] = ((d.dh[0][t] + 1) * TO_PW + d.dw[0][t] + 1 + col) * PSTR + kb * 16;
#pragma unroll
        for (int c = 0; c < KC; ++c)
            wa[t * KC + c] = __builtin_bit_cast(bf16x8, wpk[((size_t)d.wi[0][t] * d.CDw + col) * CSU + c * 4 + kb]);
    }
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.bias && kb < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = d.bias[kb * 4 + i];
    }
    u32x4 pv[MAXU];
    auto prefetch = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * TO_H - 1, x0 = (tr % tiles_x) * TO_W - 1;
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int id = tid + it * 256;
            const int pp = id / CSU, ch = id - pp * CSU;
            const int py = pp / TO_PW, px = pp - py * TO_PW;
            const int sy = y0 + py, sx = x0 + px;
            const bool ok = id < NUN && (unsigned)sy < (unsigned)d.SH && (unsigned)sx < (unsigned)d.SW;
            const u32x4 z = {0, 0, 0, 0};
            pv[it] = ok ? src[(((size_t)n * d.SH + sy) * d.SW + sx) * CSU + ch] : z;
        }
    };
    const XcdWalk xw = xmc_xcd_walk(ntiles);
    int tile = xw.first;
    if (tile < xw.end) prefetch(tile);
    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int id = tid + it * 256;
            const int pp = id / CSU, ch = id - pp * CSU;
            if (id < NUN) *reinterpret_cast<u32x4*>(smem + pp * PSTR + ch * 16) = pv[it];
        }
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * TO_H, x0 = (tr % tiles_x) * TO_W;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wave * 2 + rr;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + r * TO_PW * PSTR + loff[t] + c * 64);
                    acc = XMC_MFMA_16x16x32(wa[t * KC + c], b, acc, 0, 0, 0);
                }
            if (kb < 2) {                                  // D[row = output channel kb*4 + i][col = pixel]; 8 channels stored
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[i] + bias4[i];
                    if (d.act == XMC_ACT_TANH) v = tanh_fast(v);
                    else if (d.act == XMC_ACT_LRELU) v = lrelu_f(v);
                    o[i] = (xmc_h16)v;
                }
                const size_t p = ((size_t)n * d.DH + y0 + r) * d.DW + x0 + col;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<xmc_h16*>(d.dst) + p * 8 + kb * 4) = o;
            }
        }
    }
}

}  // namespace

// 0 = launched, 1 = not this kernel's case, < 0 = error
int xmc_conv_thin_out_try(const XmcConvDesc* d, void* stream) {
    static const bool off = xmc_debug_off("no_thin_out");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16 || d->CD != 8 || (d->CS != 32 && d->CS != 64) || d->CDw < 16) return 1;
    if (d->SA != 1 || d->DA != 1 || d->src_shift != 0 || d->nclass != 1 || d->ntaps != 9 || d->groups > 1) return 1;
    if (d->res || d->mask || d->alpha_dev || d->dst2 || d->dst_pool || d->post_act || d->sign_bits || d->dot) return 1;
    if (d->act != XMC_ACT_NONE && d->act != XMC_ACT_TANH && d->act != XMC_ACT_LRELU) return 1;
    if (d->MH % TO_H != 0 || d->MW % TO_W != 0 || d->SH != d->MH || d->SW != d->MW || d->DH != d->MH || d->DW != d->MW) return 1;
    if (d->dph[0] != 0 || d->dpw[0] != 0) return 1;
    for (int k = 0; k < 9; ++k)
        if (d->dh[0][k] < -1 || d->dh[0][k] > 1 || d->dw[0][k] < -1 || d->dw[0][k] > 1) return 1;
    const int tx = d->MW / TO_W, ty = d->MH / TO_H, ntiles = d->N * tx * ty;
    const int grid = ntiles < 256 * 8 ? ntiles : 256 * 8;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->CS == 32) hipLaunchKernelGGL((thin_out_kernel<1>), dim3(xmc_ab_grid(grid)), dim3(256), 0, st, *d, tx, ty, ntiles);
    else hipLaunchKernelGGL((thin_out_kernel<2>), dim3(xmc_ab_grid(grid)), dim3(256), 0, st, *d, tx, ty, ntiles);
    xmc_note_kernel("thin_out_kernel<%d>", d->CS / 32);
    XMC_LAUNCH_CHECK();
    return 0;
}

// 0 = launched, 1 = not this kernel's case, < 0 = error
int xmc_conv_thin_try(const XmcConvDesc* d, void* stream) {
    static const bool off = xmc_debug_off("no_thin");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16 || d->CS != 8) return 1;
    if (d->SA != 1 || d->DA != 1 || d->src_shift != 0 || d->nclass != 1 || d->ntaps > 12) return 1;
    if (d->CDw != 32 && d->CDw != 64) return 1;
    if (d->res || d->alpha_dev || d->dst2 || d->post_act || d->sign_bits || d->dot || (d->act != XMC_ACT_NONE && d->act != XMC_ACT_LRELU)) return 1;
    if (d->MH % 8 != 0 || d->MW % 32 != 0 || d->SH != d->MH || d->SW != d->MW || d->DH != d->MH || d->DW != d->MW) return 1;
    if (d->dph[0] != 0 || d->dpw[0] != 0) return 1;
    int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
    for (int k = 0; k < d->ntaps; ++k) {
        const int h = d->dh[0][k], w = d->dw[0][k];
        hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax;
        wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax;
    }
    if (hmax - hmin > 4 || wmax - wmin > 4 || hmin > 0 || wmin > 0 || hmax < 0 || wmax < 0) return 1;
    ThinCfg t;
    t.tiles_y = d->MH / 8; t.tiles_x = d->MW / 32;
    t.PH = 8 + (hmax - hmin); t.PW = 32 + (wmax - wmin);
    t.dh0 = hmin; t.dw0 = wmin;
    if (t.PH * t.PW > 512) return 1;
    const int ntiles = d->N * t.tiles_y * t.tiles_x;
    int gx = 256 * 5;
    if (gx > ntiles) gx = ntiles;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->CDw == 32) {
        hipLaunchKernelGGL((thin_in_kernel<32>), dim3(xmc_ab_grid(gx)), dim3(256), 0, st, *d, t, ntiles);
    } else {
        hipLaunchKernelGGL((thin_in_kernel<64>), dim3(xmc_ab_grid(gx)), dim3(256), 0, st, *d, t, ntiles);
    }
    xmc_note_kernel("thin_in_kernel<%d>", d->CDw);
    XMC_LAUNCH_CHECK();
    return 0;
}

// 0 = launched, 1 = not this kernel's case, < 0 = error
static int pw1x1_go(const XmcConvDesc* d, void* stream, const unsigned char* sbits, void* smasked, float mslope) {
    static const bool off = xmc_debug_off("no_pw1x1");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16) return 1;
    if (d->ntaps != 1 || d->nclass != 1 || d->SA != 1 || d->DA != 1 || d->src_shift != 0) return 1;
    if (d->dh[0][0] != 0 || d->dw[0][0] != 0 || d->dph[0] != 0 || d->dpw[0] != 0) return 1;
    if (d->SH != d->MH || d->SW != d->MW || d->DH != d->MH || d->DW != d->MW) return 1;
    if (d->post_act || d->sign_bits || d->dot) return 1;
    if (d->res || d->mask || d->alpha_dev || d->dst2 || (d->act != XMC_ACT_NONE && d->act != XMC_ACT_LRELU)) return 1;
    if (d->CS % 32 != 0 || d->CDw % 32 != 0 || d->CD % 8 != 0) return 1;
    const int64_t M = (int64_t)d->N * d->MH * d->MW;
    if (M % 16 != 0 || M / 16 >= (1ll << 31) || M < 16 * 1024) return 1;
    const int ngroups = (int)(M / 16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ks = d->CS / 32, tn = d->CDw / 16;
    // more weight fragments than a wave's registers hold: weights in LDS (Cin 128 or 256, any Cout)
    static const bool no_w = xmc_debug_off("no_pw1x1_lds");
    if (!no_w && (ks == 4 || ks == 8) && ks * tn > 16 && d->CDw >= 128 && M >= 32 * 1024)
        return ks == 4 ? launch_pww<4>(*d, ngroups, st, sbits, smasked, mslope) : launch_pww<8>(*d, ngroups, st, sbits, smasked, mslope);
    if (d->CS > 128 || d->CDw > 128) return 1;
    if (ks * tn > 16 || ks == 3) return 1;
#define PW_CASE(K, T) if (ks == K && tn == T) return launch_pw<K, T>(*d, ngroups, st, sbits, smasked, mslope);
    PW_CASE(1, 2) PW_CASE(1, 4) PW_CASE(1, 8) PW_CASE(2, 2) PW_CASE(2, 4) PW_CASE(2, 8) PW_CASE(4, 2) PW_CASE(4, 4)
#undef PW_CASE
    return 1;
}

int xmc_conv_pw1x1_try(const XmcConvDesc* d, void* stream) { return d->wpk_lo ? 1 : pw1x1_go(d, stream, nullptr, nullptr, 0.f); }

// The same kernels on a weight PAIR (XmcConvDesc.wpk_lo, ABI 12): two MFMAs per K step.  1 = not one of their shapes.
extern "C" int xmc_conv_pw1x1_split(const XmcConvDesc* d, void* stream) {
    if (!d || !d->src || !d->wpk || !d->wpk_lo || !d->dst) return XMC_EINVAL;
    if (d->ntaps != 1 || d->CS % 8 != 0 || d->CD % 8 != 0 || d->CDw % 32 != 0 || d->CDw < d->CD || d->N < 1 || d->MH < 1 || d->MW < 1) return XMC_ESHAPE;
    if (d->mask_bits || d->sc_img) return XMC_EINVAL;
    static const bool off = xmc_debug_off("no_pw1x1_split");
    if (off) return 1;
    return pw1x1_go(d, stream, nullptr, nullptr, 0.f);
}

// The 1x1 convolution of `d` on the streaming kernels only, with the masked copy of its SOURCE as a by-product (see mask_unit):
// src_masked[unit] = src[unit] x LeakyReLU'(bit) from one sign byte per 8-channel unit (XmcConvDesc.sign_bits layout of the source
// tensor).  Returns 1 and launches nothing when the shape is not one of theirs -- the caller then runs xmc_conv_igemm and
// xmc_signmask_apply separately.
extern "C" int xmc_conv_pw1x1_masked_src(const XmcConvDesc* d, const void* src_bits, void* src_masked, float slope, void* stream) {
    if (!d || !d->src || !d->wpk || !d->dst || !src_bits || !src_masked) return XMC_EINVAL;
    if (d->ntaps != 1 || d->CS % 8 != 0 || d->CD % 8 != 0 || d->CDw % 32 != 0 || d->CDw < d->CD || d->N < 1 || d->MH < 1 || d->MW < 1) return XMC_ESHAPE;
    static const bool off = xmc_debug_off("no_pw1x1_masked_src");
    if (off) return 1;
    return pw1x1_go(d, stream, static_cast<const unsigned char*>(src_bits), src_masked, slope);
}
