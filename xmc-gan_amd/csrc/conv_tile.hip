// Halo-tile convolution (forward / data-gradient, unit source stride) for gfx950, bf16.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) gathers every K sub-step of its A tile from
// global memory, i.e. a 3x3 convolution pulls each input pixel through L2 nine times per output
// tile.  For the high-resolution, few-channel layers of the generator/discriminator (Cout 32..128)
// that gather, not the MFMA pipe, is the limit (32 flop per gathered byte at Cout=32).  Here one
// workgroup owns a TH x TW block of output pixels of ONE image (TH*TW = 256), stages the input patch
// (tile + halo) for a slab of <= 64 input channels in LDS ONCE, and all taps read their A fragments
// from that patch at shifted positions; only the (small, L2-resident) per-tap weight tile is
// streamed, register-prefetched one tap ahead.  Global traffic per tile drops to ~1.3x the input
// tile + the output tile, which is the HBM floor for these layers.
//
// LDS rows are (slab bytes + 32 B) apart: with 16-byte chunks this stride makes the ds_read_b128
// fragment pattern (16 consecutive pixels x 4 chunks) conflict-free for ANY patch alignment
// (96 B and 160 B strides; checked exhaustively).
// An optional prologue applies DF-GAN's conditional affine pair + LeakyReLU (df_gan.py:213-216)
// to the patch while it is staged, so that tensor never makes a round trip through HBM.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

struct TileCfg {
    int TH, TW, log2TW;          // output tile (TH*TW == 256)
    int tiles_y, tiles_x;        // tiles per image
    int PH[XMC_MAX_CLASSES], PW[XMC_MAX_CLASSES];        // patch size per class
    int dh0[XMC_MAX_CLASSES], dw0[XMC_MAX_CLASSES];      // min tap offsets per class
    int slab;                    // channels per slab (32 or 64)
    const float* pro[4];         // optional prologue params g0,b0,g1,b1 : f32 [N][CS]; pro[0]==nullptr -> none
    int dbg;                     // XMC_TILE_DBG experiment bits (0 in production)
    unsigned long long* stamp;   // XMC_TILE_STAMP diagnostic: s_memtime stamps of workgroup 0 (nullptr in production)
};

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) void tile_kernel(const XmcConvDesc d, const TileCfg t) {
    constexpr int NT = 256, BM = 256;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int EP_ROWS = 128, EP_LD = BN + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int tpi = t.tiles_y * t.tiles_x;
    // tap tables live in LDS: indexing the kernel-argument arrays with a run-time tap would go through vector memory
    __shared__ int s_toff[XMC_MAX_TAPS], s_twi[XMC_MAX_TAPS];
    if (tid < XMC_MAX_TAPS) {
        const int tt = tid < d.ntaps ? tid : 0;
        s_toff[tid] = (d.dh[cls][tt] - t.dh0[cls]) * t.PW[cls] + (d.dw[cls][tt] - t.dw0[cls]);
        s_twi[tid] = d.wi[cls][tt];
    }
    __syncthreads();
    const int img = blockIdx.x / tpi, trem = blockIdx.x - img * tpi;
    const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
    const int PH = t.PH[cls], PW = t.PW[cls], dh0 = t.dh0[cls], dw0 = t.dw0[cls];
    const int slab = t.slab;
    const int cps = slab / 8;                     // 16-byte chunks per pixel per slab
    const int pstride = slab * 2 + 32;            // bytes between patch pixels / weight rows
    const int nslab = d.CS / slab;
    const int cs_units = d.CS / 8;                // 16-byte units per source pixel
    unsigned char* patch = smem;
    const int patch_bytes = (PH * PW * pstride + 15) & ~15;
    unsigned char* wbuf = smem + patch_bytes;     // 2 x [BN][pstride]
    const int wbytes = BN * pstride;
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // weight staging: thread -> (row, chunk) pairs, BN*cps chunks per tap
    constexpr int WL_MAX = (BN * 8 + NT - 1) / NT;        // cps <= 8
    u32x4 wr[WL_MAX];
    auto load_w = [&](int tap, int sl) {
        const int twi = s_twi[tap];
#pragma unroll
        for (int j = 0; j < WL_MAX; ++j) {
            int id = tid + j * NT;
            int row = id / cps, ch = id - row * cps;
            u32x4 z = {0, 0, 0, 0};
            wr[j] = (id < BN * cps) ? w16[((size_t)twi * d.CDw + n0 + row) * cs_units + sl * cps + ch] : z;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WL_MAX; ++j) {
            int id = tid + j * NT;
            int row = id / cps, ch = id - row * cps;
            if (id < BN * cps) *reinterpret_cast<u32x4*>(wbuf + buf * wbytes + row * pstride + ch * 16) = wr[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fc = lane >> 4;
    // per-fragment patch pixel of lane's row for tap offset (0,0) relative to (dh0,dw0)
    int apix[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int ml = wm * WTM + i * 16;               // 16 consecutive pixels of one tile row
        int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
        apix[i] = ty * PW + tx + fr;
    }
    const int pchunk = tid % cps;                 // this thread's chunk index while staging the patch (NT % cps == 0)
    const int ppix0 = tid / cps, ppix_step = NT / cps;

    int wcur = 0;
    for (int sl = 0; sl < nslab; ++sl) {
        __syncthreads();                          // previous slab's compute finished with patch + wbuf
        // ---- stage the patch for this slab (optionally through the fused affine pair)
        float P0[8], P1[8], P2[8], P3[8];
        const bool has_pro = t.pro[0] != nullptr;
        if (has_pro) {
            const size_t pb = (size_t)img * d.CS + (size_t)sl * slab + pchunk * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { P0[k] = t.pro[0][pb + k]; P1[k] = t.pro[1][pb + k]; P2[k] = t.pro[2][pb + k]; P3[k] = t.pro[3][pb + k]; }
        }
        // all of this thread's patch loads are issued before the first one is consumed (latency overlap)
        constexpr int PIT = 12;                    // >= ceil(max patch pixels (10*34) / (256/8))
        u32x4 pv[PIT];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            int pp = ppix0 + it * ppix_step;
            int py = pp / PW, px = pp - py * PW;
            int sy = a0 + dh0 + py, sx = b0 + dw0 + px;
            bool ok = pp < PH * PW && (unsigned)sy < (unsigned)d.SH && (unsigned)sx < (unsigned)d.SW;
            u32x4 z = {0, 0, 0, 0};
            pv[it] = ok ? src16[(((size_t)img * d.SH + sy) * d.SW + sx) * cs_units + sl * cps + pchunk] : z;
            if (has_pro && ok) {
                bf16x8 h = __builtin_bit_cast(bf16x8, pv[it]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)h[k];
                    f = lrelu_f(lrelu_f(f * P0[k] + P1[k]) * P2[k] + P3[k]);
                    h[k] = (__bf16)f;
                }
                pv[it] = __builtin_bit_cast(u32x4, h);
            }
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            int pp = ppix0 + it * ppix_step;
            if (pp < PH * PW) *reinterpret_cast<u32x4*>(patch + pp * pstride + pchunk * 16) = pv[it];
        }
        load_w(0, sl);
        store_w(wcur);
        __syncthreads();
        for (int tap = 0; tap < d.ntaps; ++tap) {
            const bool more = tap + 1 < d.ntaps;
            if (more) load_w(tap + 1, sl);
            const int toff = s_toff[tap];
            const unsigned char* wb = wbuf + wcur * wbytes;
            for (int s = 0; s < slab / 32; ++s) {
                u32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(patch + (apix[i] + toff) * pstride + (s * 4 + fc) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const u32x4*>(wb + (wn * WTN + j * 16 + fr) * pstride + (s * 4 + fc) * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                             __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
            }
            if (more) {
                store_w(wcur ^ 1);                // other buffer: nobody reads it during this tap
                __syncthreads();
                wcur ^= 1;
            }
        }
    }

    // ---- epilogue in two halves of 128 rows through LDS (f32), coalesced 8-channel stores
    float* ep = reinterpret_cast<float*>(smem);
    const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
    constexpr int CPR = BN / 8;
    const int dph = d.dph[cls], dpw = d.dpw[cls];
    for (int half = 0; half < BM / EP_ROWS; ++half) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int rbase = wm * WTM + i * 16;
            if (rbase / EP_ROWS == half) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ep[(rbase - half * EP_ROWS + fc * 4 + r) * EP_LD + wn * WTN + j * 16 + fr] = acc[i][j][r];
            }
        }
        __syncthreads();
        for (int id = tid; id < EP_ROWS * CPR; id += NT) {
            int row = id / CPR, cc = id - row * CPR;
            int ml = half * EP_ROWS + row, ch = n0 + cc * 8;
            if (ch >= d.CD) continue;
            int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
            size_t pix = ((size_t)img * d.DH + (a0 + ty) * d.DA + dph) * d.DW + (b0 + tx) * d.DA + dpw;
            size_t idx8 = (pix * d.CD + ch) >> 3;
            float v[8];
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8]);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8 + 4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = e0[k]; v[4 + k] = e1[k]; }
            if (d.bias) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += d.bias[ch + k];
            }
            if (d.act == XMC_ACT_LRELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = lrelu_f(v[k]);
            } else if (d.act == XMC_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            } else if (d.act == XMC_ACT_TANH) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tanhf(v[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= alpha;
            if (d.out_dtype == XMC_BF16) {
                if (d.res) {
                    float rr[8];
                    Vec8<XMC_BF16>::load(d.res, idx8, rr);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += rr[k];
                }
                Vec8<XMC_BF16>::store(d.dst, idx8, v);
            } else {
                if (d.res) {
                    float rr[8];
                    Vec8<XMC_F32>::load(d.res, idx8, rr);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += rr[k];
                }
                Vec8<XMC_F32>::store(d.dst, idx8, v);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Persistent variant for the highest-resolution layers (Cin <= 64, Cout <= 64): ALL taps' weights stay
// resident in LDS for the life of the workgroup, which walks over output tiles; the next tile's patch is
// prefetched into registers while the current one is being multiplied, so per tile the only exposed
// global traffic is the output store.  These layers are HBM-bound (<= 150 flop/byte), the kernel's job
// is to keep ~50 KB per tile in flight per CU without ever re-reading an input pixel.
template <int BN>
__global__ __launch_bounds__(256) void ptile_kernel(const XmcConvDesc d, const TileCfg t, int ntiles) {
    constexpr int NT = 256, BM = 256, WM = 4;
    constexpr int WTM = BM / WM;                 // 64 rows per wave, all BN columns
    constexpr int TM = WTM / 16, TN = BN / 16;
    constexpr int EP_ROWS = 128, EP_LD = BN + 4;
    constexpr int PIT = 12;
    constexpr int CPR = BN / 8;                  // 8-channel chunks per output row
    constexpr int EIT = EP_ROWS * CPR / NT;      // epilogue chunks per thread per half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PH = t.PH[cls], PW = t.PW[cls], dh0 = t.dh0[cls], dw0 = t.dw0[cls];
    __shared__ int s_toff[XMC_MAX_TAPS];          // tap -> patch byte offset (kernel-argument arrays indexed with a
    const int slab = t.slab;                      // run-time tap would be fetched through vector memory)
    const int cps = slab / 8;
    const int pstride = slab * 2 + 32;
    if (tid < XMC_MAX_TAPS) {
        const int tt = tid < d.ntaps ? tid : 0;
        s_toff[tid] = ((d.dh[cls][tt] - dh0) * PW + (d.dw[cls][tt] - dw0)) * pstride;
    }
    const int cs_units = d.CS / 8;
    unsigned char* patch = smem;
    const int patch_bytes = (PH * PW * pstride + 15) & ~15;
    const int ep_bytes = EP_ROWS * EP_LD * 4;
    unsigned char* wall = smem + (patch_bytes > ep_bytes ? patch_bytes : ep_bytes);    // [ntaps][BN][pstride]
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    for (int id = tid; id < d.ntaps * BN * cps; id += NT) {
        int ch = id % cps, row = (id / cps) % BN, tap = id / (cps * BN);
        *reinterpret_cast<u32x4*>(wall + (tap * BN + row) * pstride + ch * 16) =
            w16[((size_t)d.wi[cls][tap] * d.CDw + n0 + row) * cs_units + ch];
    }

    const int fr = lane & 15, fc = lane >> 4;
    // All per-thread index arithmetic that does not depend on the tile is done ONCE here: with one or two waves per
    // SIMD every VALU instruction costs ~4 issue cycles, and the first version of this kernel spent 890 VALU
    // instructions per tile against 72 MFMAs (rocprof SQ_INSTS_VALU / SQ_INSTS_MFMA).
    int abyte[TM];                                // patch byte offset of this lane's A-fragment rows (tap (0,0), k-chunk fc)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int ml = wm * WTM + i * 16;
        abyte[i] = ((ml >> t.log2TW) * PW + (ml & (t.TW - 1)) + fr) * pstride + fc * 16;
    }
    const int bbyte = fr * pstride + fc * 16;     // weight-row byte offset of this lane's B fragment
    const int pchunk = tid % cps, ppix0 = tid / cps, ppix_step = NT / cps;
    int pyx[PIT];                                 // (py << 16) | px of this thread's patch pixels, -1 if beyond the patch
    int psrc[PIT];                                // source offset (16-byte units) relative to the tile origin pixel
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        int pp = ppix0 + it * ppix_step;
        int py = pp / PW, px = pp - py * PW;
        pyx[it] = pp < PH * PW ? ((py << 16) | px) : -1;
        psrc[it] = ((dh0 + py) * d.SW + (dw0 + px)) * cs_units + pchunk;
    }
    int eoff[EIT], erow[EIT], ecc[EIT];           // epilogue: destination offset (8-channel units) relative to tile origin
#pragma unroll
    for (int k = 0; k < EIT; ++k) {
        int id = tid + k * NT;
        erow[k] = id / CPR; ecc[k] = id - erow[k] * CPR;
    }
    const int dph = d.dph[cls], dpw = d.dpw[cls];
    const int cd8 = d.CD / 8;
    const bool has_pro = t.pro[0] != nullptr;

    u32x4 pv[PIT];
    auto prefetch = [&](int tile) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        const int base = ((img * d.SH + a0) * d.SW + b0) * cs_units;
        const int ymin = -(a0 + dh0), ymax = d.SH - (a0 + dh0), xmin = -(b0 + dw0), xmax = d.SW - (b0 + dw0);
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int py = pyx[it] >> 16, px = pyx[it] & 0xffff;
            const bool ok = pyx[it] >= 0 && py >= ymin && py < ymax && px >= xmin && px < xmax;
            u32x4 z = {0, 0, 0, 0};
            pv[it] = ok ? src16[(unsigned)(base + psrc[it])] : z;
            if (has_pro && !ok) pv[it] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // marks padding
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
    float* ep = reinterpret_cast<float*>(smem);
    float bias8[EIT][8];
#pragma unroll
    for (int k = 0; k < EIT; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) bias8[k][c] = (d.bias && n0 + ecc[k] * 8 < d.CD) ? d.bias[n0 + ecc[k] * 8 + c] : 0.f;
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        __syncthreads();                          // previous tile's epilogue is done with the patch region
        if (has_pro) {
            float P0[8], P1[8], P2[8], P3[8];
            const size_t pb = (size_t)img * d.CS + pchunk * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { P0[k] = t.pro[0][pb + k]; P1[k] = t.pro[1][pb + k]; P2[k] = t.pro[2][pb + k]; P3[k] = t.pro[3][pb + k]; }
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                if (pv[it][0] == 0xFFFFFFFFu && pv[it][3] == 0xFFFFFFFFu) { pv[it] = u32x4{0, 0, 0, 0}; continue; }
                bf16x8 h = __builtin_bit_cast(bf16x8, pv[it]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)h[k];
                    f = lrelu_f(lrelu_f(f * P0[k] + P1[k]) * P2[k] + P3[k]);
                    h[k] = (__bf16)f;
                }
                pv[it] = __builtin_bit_cast(u32x4, h);
            }
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            int pp = ppix0 + it * ppix_step;
            if (pyx[it] >= 0) *reinterpret_cast<u32x4*>(patch + pp * pstride + pchunk * 16) = pv[it];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);     // in flight during the MFMAs below

        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tap = 0; tap < d.ntaps; ++tap) {
            const unsigned char* pa = patch + s_toff[tap];
            const unsigned char* wb = wall + tap * BN * pstride + bbyte;
            for (int s = 0; s < slab / 32; ++s) {
                u32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(pa + abyte[i] + s * 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * pstride + s * 64);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                             __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
            }
        }
        const int dbase = (((img * d.DH + a0 * d.DA + dph) * d.DW) + b0 * d.DA + dpw) * cd8 + (n0 >> 3);
        for (int half = 0; half < BM / EP_ROWS; ++half) {
            __syncthreads();
            if (wm / 2 == half) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ep[((wm & 1) * WTM + i * 16 + fc * 4 + r) * EP_LD + j * 16 + fr] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EIT; ++k) {
                const int row = erow[k], cc = ecc[k];
                if (n0 + cc * 8 >= d.CD) continue;
                const int ml = half * EP_ROWS + row;
                const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
                const size_t idx8 = (size_t)(dbase + ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + cc);
                float v[8];
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8]);
                const f32x4 e1 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8 + 4]);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = e0[q] + bias8[k][q]; v[4 + q] = e1[q] + bias8[k][4 + q]; }
                if (d.act == XMC_ACT_LRELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = lrelu_f(v[q]);
                } else if (d.act == XMC_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (d.act == XMC_ACT_TANH) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = tanhf(v[q]);
                }
                if (d.alpha_dev) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= alpha;
                }
                if (d.out_dtype == XMC_BF16) {
                    if (d.res) {
                        float rr[8];
                        Vec8<XMC_BF16>::load(d.res, idx8, rr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rr[q];
                    }
                    Vec8<XMC_BF16>::store(d.dst, idx8, v);
                } else {
                    if (d.res) {
                        float rr[8];
                        Vec8<XMC_F32>::load(d.res, idx8, rr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rr[q];
                    }
                    Vec8<XMC_F32>::store(d.dst, idx8, v);
                }
            }
        }
    }
}

// Second version of the persistent kernel: same staging, but
//  * the MFMA roles are swapped (A = weight rows, B = pixels), so a lane's accumulators are CONSECUTIVE output channels
//    of ONE pixel (weight rows are permuted while they are copied to LDS) and the epilogue -- bias, activation, alpha,
//    residual, 16-byte stores -- runs straight from registers: no LDS round trip, 2 barriers per tile instead of 6;
//  * the (tap, k-chunk) loop is software pipelined: the fragments of step i+1 are read from LDS while the MFMAs of
//    step i issue (one wave per SIMD cannot hide an un-pipelined ds_read -> wait -> MFMA chain behind anything).
__device__ unsigned g_zero16[4];              // zero-initialised, global address space (a const one would make the select flat)

template <int BN, int SLAB, int NTAPS, int NT>       // NTAPS == 0: run-time tap count (loop not unrolled); NT threads
__global__ __launch_bounds__(NT) void ptile2_kernel(const XmcConvDesc d, const TileCfg t, int ntiles) {
    constexpr int WTM = 256 / (NT / 64);         // pixels per wave: 64 (4 waves) or 32 (8 waves: two per SIMD, so the
    constexpr int TM = WTM / 16, TN = BN / 16;   // VALU-bound staging/epilogue phases issue from two waves at once)
    constexpr int PIT = 12 * 256 / NT;
    constexpr int UPL = BN / 32;                  // 8-channel output units per lane (lane owns BN/4 consecutive channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PH = t.PH[cls], PW = t.PW[cls], dh0 = t.dh0[cls], dw0 = t.dw0[cls];
    __shared__ int s_toff[XMC_MAX_TAPS];
    constexpr int slab = SLAB;
    constexpr int cps = slab / 8;
    constexpr int pstride = slab * 2 + 32;
    if (tid < XMC_MAX_TAPS) {
        const int tt = tid < d.ntaps ? tid : 0;
        s_toff[tid] = ((d.dh[cls][tt] - dh0) * PW + (d.dw[cls][tt] - dw0)) * pstride;
    }
    const int cs_units = d.CS / 8;
    unsigned char* patch = smem;
    const int patch_bytes = (PH * PW * pstride + 15) & ~15;
    unsigned char* wall = smem + patch_bytes;                                            // [ntaps][BN][pstride]
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // physical weight row (n-block j, row q) holds logical output channel (q/4)*(BN/4) + j*4 + q%4
    for (int id = tid; id < d.ntaps * BN * cps; id += NT) {
        const int ch = id % cps, prow = (id / cps) % BN, tap = id / (cps * BN);
        const int j = prow >> 4, q = prow & 15;
        const int lrow = (q >> 2) * (BN / 4) + j * 4 + (q & 3);
        *reinterpret_cast<u32x4*>(wall + (tap * BN + prow) * pstride + ch * 16) =
            w16[((size_t)d.wi[cls][tap] * d.CDw + n0 + lrow) * cs_units + ch];
    }

    const int fr = lane & 15, fc = lane >> 4;
    int abyte[TM];                                // patch byte offset of this lane's pixel fragments (tap (0,0), k-chunk fc)
    int eoff[TM];                                 // destination offset (8-channel units) of this lane's pixel, relative to the tile
    const int cd8 = d.CD / 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WTM + i * 16 + fr;
        const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
        abyte[i] = (ty * PW + tx) * pstride + fc * 16;
        eoff[i] = ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + fc * UPL;
    }
    const int bbyte = fr * pstride + fc * 16;     // weight-row byte offset of this lane's weight fragment
    const int pchunk = tid % cps, ppix0 = tid / cps, ppix_step = NT / cps;
    int pyx[PIT], psrc[PIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        int pp = ppix0 + it * ppix_step;
        int py = pp / PW, px = pp - py * PW;
        pyx[it] = pp < PH * PW ? ((py << 16) | px) : -1;
        psrc[it] = ((dh0 + py) * d.SW + (dw0 + px)) * cs_units + pchunk;
    }
    const int dph = d.dph[cls], dpw = d.dpw[cls];
    const bool has_pro = t.pro[0] != nullptr;
    const int ch0 = n0 + fc * (BN / 4);           // first output channel of this lane

    const u32x4* zero16 = reinterpret_cast<const u32x4*>(g_zero16);
    u32x4 pv[PIT];
    unsigned okmask = 0;                          // nothing may touch pv between the loads and the next tile's LDS store:
                                                  // a select on the loaded value would put s_waitcnt vmcnt(0) ahead of the MFMAs
    auto prefetch = [&](int tile) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        const int base = ((img * d.SH + a0) * d.SW + b0) * cs_units;
        const int ymin = -(a0 + dh0), ymax = d.SH - (a0 + dh0), xmin = -(b0 + dw0), xmax = d.SW - (b0 + dw0);
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int py = pyx[it] >> 16, px = pyx[it] & 0xffff;
            const bool ok = pyx[it] >= 0 && py >= ymin && py < ymax && px >= xmin && px < xmax;
            // branch-free: padding (and lanes beyond the patch) read a 16-byte block of zeros instead of being masked off,
            // so the 12 loads issue back to back and nothing touches pv until the next tile stores it to LDS
            const u32x4* ptr = ok ? src16 + (unsigned)(base + psrc[it]) : zero16;
            pv[it] = __builtin_nontemporal_load(ptr);
            okmask = ok ? (okmask | (1u << it)) : (okmask & ~(1u << it));      // the prologue must leave padding at zero
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
    float bias8[UPL][8];
#pragma unroll
    for (int u = 0; u < UPL; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) bias8[u][c] = (d.bias && ch0 + u * 8 < d.CD) ? d.bias[ch0 + u * 8 + c] : 0.f;
    __syncthreads();
    int toffr[NTAPS > 0 ? NTAPS : 1];            // uniform -> scalar registers
    if constexpr (NTAPS > 0) {
#pragma unroll
        for (int k = 0; k < NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);
    }

    for (; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tpi, trem = tile - img * tpi;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        __syncthreads();                          // every wave is done reading the previous patch
        if (has_pro) {
            float P0[8], P1[8], P2[8], P3[8];
            const size_t pb = (size_t)img * d.CS + pchunk * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { P0[k] = t.pro[0][pb + k]; P1[k] = t.pro[1][pb + k]; P2[k] = t.pro[2][pb + k]; P3[k] = t.pro[3][pb + k]; }
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                if (!((okmask >> it) & 1)) continue;
                bf16x8 h = __builtin_bit_cast(bf16x8, pv[it]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)h[k];
                    f = lrelu_f(lrelu_f(f * P0[k] + P1[k]) * P2[k] + P3[k]);
                    h[k] = (__bf16)f;
                }
                pv[it] = __builtin_bit_cast(u32x4, h);
            }
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            int pp = ppix0 + it * ppix_step;
            if (pyx[it] >= 0) *reinterpret_cast<u32x4*>(patch + pp * pstride + pchunk * 16) = pv[it];
        }
        // Explicit vmcnt(0): the previous tile's stores have had the patch write to drain, and with nothing pending here the
        // compiler's wait-count merge at the loop head no longer puts conservative vmcnt waits (i.e. waits for the
        // prefetch just issued) in front of the MFMAs.
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);     // in flight during the MFMAs below

        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int S = SLAB / 32;
        auto ldfrag = [&](int toff, int tap, int s, u32x4* p, u32x4* w) {
            const unsigned char* pa = patch + toff + s * 64;
            const unsigned char* wb = wall + tap * BN * pstride + bbyte + s * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i) p[i] = *reinterpret_cast<const u32x4*>(pa + abyte[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * pstride);
        };
        auto mma = [&](const u32x4* p, const u32x4* w) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j]),
                                                                         __builtin_bit_cast(bf16x8, p[i]), acc[i][j], 0, 0, 0);
        };
        if constexpr (NTAPS > 0) {
            // fully unrolled: tap offsets sit in scalar registers, the scheduler hoists the LDS reads of later steps
            u32x4 pf[NTAPS * S][TM], wf[NTAPS * S][TN];
#pragma unroll
            for (int st = 0; st < NTAPS * S; ++st) {
                if (st == 0) ldfrag(toffr[0], 0, 0, pf[0], wf[0]);
                if (st + 1 < NTAPS * S) ldfrag(toffr[(st + 1) / S], (st + 1) / S, (st + 1) % S, pf[st + 1], wf[st + 1]);
                mma(pf[st], wf[st]);
            }
        } else {
            u32x4 pf[2][TM], wf[2][TN];
            const int nsteps = d.ntaps * S;
            ldfrag(s_toff[0], 0, 0, pf[0], wf[0]);
            for (int step = 0; step < nsteps; step += 2) {
                if (step + 1 < nsteps) ldfrag(s_toff[(step + 1) / S], (step + 1) / S, (step + 1) % S, pf[1], wf[1]);
                mma(pf[0], wf[0]);
                if (step + 2 < nsteps) ldfrag(s_toff[(step + 2) / S], (step + 2) / S, (step + 2) % S, pf[0], wf[0]);
                if (step + 1 < nsteps) mma(pf[1], wf[1]);
            }
        }

        // epilogue from registers: acc[i][j][r] = pixel (m-block i, fr), channel ch0 + j*4 + r
        const int dbase = (((img * d.DH + a0 * d.DA + dph) * d.DW) + b0 * d.DA + dpw) * cd8 + (n0 >> 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int u = 0; u < UPL; ++u) {
                if (ch0 + u * 8 >= d.CD) continue;
                const size_t idx8 = (size_t)(dbase + eoff[i] + u);
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = acc[i][2 * u][q] + bias8[u][q]; v[4 + q] = acc[i][2 * u + 1][q] + bias8[u][4 + q]; }
                if (d.act == XMC_ACT_LRELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = lrelu_f(v[q]);
                } else if (d.act == XMC_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (d.act == XMC_ACT_TANH) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = tanhf(v[q]);
                }
                if (d.alpha_dev) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= alpha;
                }
                if (d.out_dtype == XMC_BF16) {
                    if (d.res) {
                        float rr[8];
                        Vec8<XMC_BF16>::load(d.res, idx8, rr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rr[q];
                    }
                    Vec8<XMC_BF16>::store(d.dst, idx8, v);
                } else {
                    if (d.res) {
                        float rr[8];
                        Vec8<XMC_F32>::load(d.res, idx8, rr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rr[q];
                    }
                    Vec8<XMC_F32>::store(d.dst, idx8, v);
                }
            }
        }
    }
}

// Third version: role-split workgroup of 8 waves.  One wave per SIMD computes and the other stages, and the two roles run
// different phases at the same time (measured on the 4-wave kernel above: a tile spends 4.6 k cycles in MFMAs and ~8 k in
// address arithmetic, LDS stores, the epilogue and waiting for memory, none of which a single wave per SIMD overlaps):
//   waves 0-3 ("compute"): B1 | MFMAs over the staged patch           | B2 | epilogue from registers, global stores
//   waves 4-7 ("stage")  : B1 | addresses + global loads of next tile | B2 | registers -> LDS patch
// Two barriers per tile, both reached by all 8 waves.  MFMA roles as in ptile2 (lane = pixel, accumulators = consecutive
// channels), tap loop fully unrolled when NTAPS > 0.
template <int BN, int SLAB, int NTAPS>
__global__ __launch_bounds__(512) void ptile3_kernel(const XmcConvDesc d, const TileCfg t, int ntiles) {
    constexpr int NS = 256;                      // threads per role
    constexpr int TM = 4, TN = BN / 16;
    constexpr int cps = SLAB / 8;
    constexpr int pstride = SLAB * 2 + 32;
    constexpr int PIT = 384 * cps / NS;          // patch units per staging thread: patches of up to 384 pixels
    constexpr int UPL = BN / 32;
    constexpr int S = SLAB / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool stager = wave >= 4;
    const int rt = tid & (NS - 1);               // thread index within the role
    const int wm = wave & 3;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PH = t.PH[cls], PW = t.PW[cls], dh0 = t.dh0[cls], dw0 = t.dw0[cls];
    __shared__ int s_toff[XMC_MAX_TAPS];
    if (tid < XMC_MAX_TAPS) {
        const int tt = tid < d.ntaps ? tid : 0;
        s_toff[tid] = ((d.dh[cls][tt] - dh0) * PW + (d.dw[cls][tt] - dw0)) * pstride;
    }
    const int cs_units = d.CS / 8;
    unsigned char* patch = smem;
    const int patch_bytes = (PH * PW * pstride + 15) & ~15;
    unsigned char* wall = smem + patch_bytes;                                            // [ntaps][BN][pstride]
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // physical weight row (n-block j, row q) holds logical output channel (q/4)*(BN/4) + j*4 + q%4
    for (int id = tid; id < d.ntaps * BN * cps; id += 512) {
        const int ch = id % cps, prow = (id / cps) % BN, tap = id / (cps * BN);
        const int j = prow >> 4, q = prow & 15;
        const int lrow = (q >> 2) * (BN / 4) + j * 4 + (q & 3);
        *reinterpret_cast<u32x4*>(wall + (tap * BN + prow) * pstride + ch * 16) =
            w16[((size_t)d.wi[cls][tap] * d.CDw + n0 + lrow) * cs_units + ch];
    }
    const int tile0 = blockIdx.x, tstep = gridDim.x;

    if (stager) {
        // ------------------------------------------------------------------------------------------------ staging role
        const int pchunk = rt % cps, ppix0 = rt / cps;
        constexpr int ppix_step = NS / cps;
        // Per staged unit: source offset relative to the tile origin and 4 halo bits (1 top, 2 bottom, 4 left, 8 right rows /
        // columns of the patch that fall outside the image when the tile touches that image border).  Per tile the bounds
        // test is then `halo & border_mask` with a scalar mask: 2 VALU instead of 4 compares + 4 ands per unit, and the loads
        // are unconditional (a unit outside the image re-reads the tile's first pixel and is zeroed when it is stored to LDS),
        // so there is no branch per load.  The staging waves share their SIMDs' issue slots with the MFMA waves: every
        // instruction here is taken from the matrix pipe's feed.
        int psrc[PIT];
        unsigned halo[PIT];
        unsigned inpatch = 0;                     // bit it: this thread stages a unit of the patch
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int pp = ppix0 + it * ppix_step;
            const int py = pp / PW, px = pp - py * PW;
            const bool in = pp < PH * PW;
            inpatch |= in ? (1u << it) : 0u;
            psrc[it] = in ? ((dh0 + py) * d.SW + (dw0 + px)) * cs_units + pchunk : 0;
            halo[it] = !in ? 0u : ((py < -dh0 ? 1u : 0u) | (py >= t.TH - dh0 ? 2u : 0u) | (px < -dw0 ? 4u : 0u) | (px >= t.TW - dw0 ? 8u : 0u));
        }
        const bool has_pro = t.pro[0] != nullptr;
        u32x4 pv[PIT];
        unsigned okmask = 0;
        auto issue = [&](int tile) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int ty = trem / t.tiles_x, tx = trem - ty * t.tiles_x;
            const int a0 = ty * t.TH, b0 = tx * t.TW;
            const int base = ((img * d.SH + a0) * d.SW + b0) * cs_units;
            // a halo row/column is outside the image only for tiles on that border (halo depth <= tile size)
            const unsigned border = (a0 + dh0 < 0 ? 1u : 0u) | (a0 + t.TH + (PH - t.TH + dh0) > d.SH ? 2u : 0u) |
                                    (b0 + dw0 < 0 ? 4u : 0u) | (b0 + t.TW + (PW - t.TW + dw0) > d.SW ? 8u : 0u);
            okmask = 0;
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const bool ok = (halo[it] & border) == 0;
                pv[it] = src16[(unsigned)(base + (ok ? psrc[it] : 0))];
                okmask |= ok ? (1u << it) : 0u;
            }
            okmask &= inpatch;
        };
        auto transform = [&](int tile) {          // DF-GAN affine pair + LeakyReLU on the staged values (padding stays 0)
            const int img = tile / tpi;
            float P0[8], P1[8], P2[8], P3[8];
            const size_t pb = (size_t)img * d.CS + pchunk * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { P0[k] = t.pro[0][pb + k]; P1[k] = t.pro[1][pb + k]; P2[k] = t.pro[2][pb + k]; P3[k] = t.pro[3][pb + k]; }
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                if (!((okmask >> it) & 1)) continue;
                bf16x8 h = __builtin_bit_cast(bf16x8, pv[it]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)h[k];
                    f = lrelu_f(lrelu_f(f * P0[k] + P1[k]) * P2[k] + P3[k]);
                    h[k] = (__bf16)f;
                }
                pv[it] = __builtin_bit_cast(u32x4, h);
            }
        };
        auto commit = [&]() {
            const bool all_ok = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(okmask != inpatch) == 0);
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int pp = ppix0 + it * ppix_step;
                u32x4 v = pv[it];
                if (!all_ok && !((okmask >> it) & 1)) v = u32x4{0, 0, 0, 0};       // padding (border tiles only)
                if ((inpatch >> it) & 1) *reinterpret_cast<u32x4*>(patch + pp * pstride + pchunk * 16) = v;
            }
        };
        if (tile0 < ntiles) {
            issue(tile0);
            if (has_pro) transform(tile0);
            commit();
        }
        __syncthreads();                          // weights + first patch staged
        const bool stamping = t.stamp && blockIdx.x == 0 && blockIdx.z == 0 && rt == 0;
        int sit = 0;
        for (int tile = tile0; tile < ntiles; tile += tstep, ++sit) {
            const int next = tile + tstep;
            __syncthreads();                      // B1
            if (stamping && sit < 32) t.stamp[sit * 16 + 8] = __builtin_amdgcn_s_memtime();
            if (next < ntiles) {
                issue(next);
                if (has_pro) transform(next);
            }
            if (stamping && sit < 32) t.stamp[sit * 16 + 9] = __builtin_amdgcn_s_memtime();
            __syncthreads();                      // B2: the compute waves are done reading the patch
            if (stamping && sit < 32) t.stamp[sit * 16 + 10] = __builtin_amdgcn_s_memtime();
            if (stamping) __builtin_amdgcn_s_waitcnt(0x0f70);
            if (stamping && sit < 32) t.stamp[sit * 16 + 11] = __builtin_amdgcn_s_memtime();
            if (next < ntiles) commit();
            if (stamping && sit < 32) t.stamp[sit * 16 + 12] = __builtin_amdgcn_s_memtime();
        }
    } else {
        // ------------------------------------------------------------------------------------------------ compute role
        const int fr = lane & 15, fc = lane >> 4;
        const int cd8 = d.CD / 8;
        int abyte[TM], eoff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * 64 + i * 16 + fr;
            const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
            abyte[i] = (ty * PW + tx) * pstride + fc * 16;
            eoff[i] = ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + fc * UPL;
        }
        const int bbyte = fr * pstride + fc * 16;
        const int dph = d.dph[cls], dpw = d.dpw[cls];
        const int ch0 = n0 + fc * (BN / 4);
        const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
        float bias8[UPL][8];
#pragma unroll
        for (int u = 0; u < UPL; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) bias8[u][c] = (d.bias && ch0 + u * 8 < d.CD) ? d.bias[ch0 + u * 8 + c] : 0.f;
        // one uniform decision instead of a chain of branches per stored unit
        const bool fast = d.out_dtype == XMC_BF16 && d.res == nullptr && d.alpha_dev == nullptr &&
                          (d.act == XMC_ACT_NONE || d.act == XMC_ACT_LRELU);
        const float slope = d.act == XMC_ACT_LRELU ? XMC_LRELU : 1.f;
        __syncthreads();                          // weights + first patch staged
        int toffr[NTAPS > 0 ? NTAPS : 1];
        if constexpr (NTAPS > 0) {
#pragma unroll
            for (int k = 0; k < NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);
        }
        const bool stamping = t.stamp && blockIdx.x == 0 && blockIdx.z == 0 && tid == 0;
        int sit = 0;
        for (int tile = tile0; tile < ntiles; tile += tstep, ++sit) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
            if (stamping && sit < 32) t.stamp[sit * 16 + 0] = __builtin_amdgcn_s_memtime();
            __syncthreads();                      // B1: patch of this tile is in LDS
            if (stamping && sit < 32) t.stamp[sit * 16 + 1] = __builtin_amdgcn_s_memtime();
            f32x4 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto ldfrag = [&](int toff, int tap, int s, u32x4* p, u32x4* w) {
                const unsigned char* pa = patch + toff + s * 64;
                const unsigned char* wb = wall + tap * BN * pstride + bbyte + s * 64;
#pragma unroll
                for (int i = 0; i < TM; ++i) p[i] = *reinterpret_cast<const u32x4*>(pa + abyte[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * pstride);
            };
            auto mma = [&](const u32x4* p, const u32x4* w) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j]),
                                                                             __builtin_bit_cast(bf16x8, p[i]), acc[i][j], 0, 0, 0);
            };
            if constexpr (NTAPS > 0) {
                // Fully unrolled and hand-interleaved: while the MFMAs of step st issue, the TM+TN fragment reads of step
                // st+1 are dealt out between them (one read, then its share of MFMAs), and sched_barrier pins that order --
                // left alone, the scheduler sinks each read to just before its first use and every group of MFMAs then
                // waits ~100 cycles on lgkmcnt(0) (measured: 28 cycles per MFMA instead of 16).
                constexpr int NST = NTAPS * S, G = TM + TN, M = TM * TN;
                u32x4 fr_[2][G];                 // [parity][0..TM) pixel fragments, [TM..G) weight fragments
                auto rd = [&](int st, int g) -> u32x4 {
                    const int tap = st / S, sub = st % S;
                    if (g < TM) return *reinterpret_cast<const u32x4*>(patch + toffr[tap] + sub * 64 + abyte[g]);
                    return *reinterpret_cast<const u32x4*>(wall + tap * BN * pstride + bbyte + sub * 64 + (g - TM) * 16 * pstride);
                };
                // read order: p0, w0, p1, w1, ... so that the first MFMAs of the next step find their operands first
                auto gslot = [&](int k) -> int { return (k < 2 * (TM < TN ? TM : TN)) ? ((k & 1) ? TM + k / 2 : k / 2) : (TM >= TN ? k - TN : k); };
#pragma unroll
                for (int g = 0; g < G; ++g) fr_[0][gslot(g)] = rd(0, gslot(g));
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int cur = st & 1, nxt = cur ^ 1;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (st + 1 < NST) fr_[nxt][gslot(g)] = rd(st + 1, gslot(g));
#pragma unroll
                        for (int m = g * M / G; m < (g + 1) * M / G; ++m) {
                            const int mi = m % TM, mj = m / TM;
                            acc[mi][mj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr_[cur][TM + mj]),
                                                                                   __builtin_bit_cast(bf16x8, fr_[cur][mi]), acc[mi][mj], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                u32x4 pf[2][TM], wf[2][TN];
                const int nsteps = d.ntaps * S;
                ldfrag(s_toff[0], 0, 0, pf[0], wf[0]);
                for (int step = 0; step < nsteps; step += 2) {
                    if (step + 1 < nsteps) ldfrag(s_toff[(step + 1) / S], (step + 1) / S, (step + 1) % S, pf[1], wf[1]);
                    mma(pf[0], wf[0]);
                    if (step + 2 < nsteps) ldfrag(s_toff[(step + 2) / S], (step + 2) / S, (step + 2) % S, pf[0], wf[0]);
                    if (step + 1 < nsteps) mma(pf[1], wf[1]);
                }
            }
            if (stamping && sit < 32) t.stamp[sit * 16 + 2] = __builtin_amdgcn_s_memtime();
            __syncthreads();                      // B2: patch may be overwritten
            if (stamping && sit < 32) t.stamp[sit * 16 + 3] = __builtin_amdgcn_s_memtime();
            // epilogue from registers: acc[i][j][r] = pixel (m-block i, fr), channel ch0 + j*4 + r
            const int dbase = (((img * d.DH + a0 * d.DA + dph) * d.DW) + b0 * d.DA + dpw) * cd8 + (n0 >> 3);
            if (fast) {
                bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst) + dbase;
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 8 >= d.CD) continue;            // same for every pixel of the lane: one branch per unit column
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        bf16x8 o;
                        if (slope != 1.f) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x0 = acc[i][2 * u][q] + bias8[u][q], x1 = acc[i][2 * u + 1][q] + bias8[u][4 + q];
                                o[q] = (__bf16)fmaxf(x0, x0 * slope);
                                o[4 + q] = (__bf16)fmaxf(x1, x1 * slope);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                o[q] = (__bf16)(acc[i][2 * u][q] + bias8[u][q]);
                                o[4 + q] = (__bf16)(acc[i][2 * u + 1][q] + bias8[u][4 + q]);
                            }
                        }
                        dst8[eoff[i] + u] = o;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int u = 0; u < UPL; ++u) {
                        if (ch0 + u * 8 >= d.CD) continue;
                        const size_t idx8 = (size_t)(dbase + eoff[i] + u);
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q] = acc[i][2 * u][q] + bias8[u][q]; v[4 + q] = acc[i][2 * u + 1][q] + bias8[u][4 + q]; }
                        if (d.act == XMC_ACT_LRELU) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = lrelu_f(v[q]);
                        } else if (d.act == XMC_ACT_RELU) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                        } else if (d.act == XMC_ACT_TANH) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = tanhf(v[q]);
                        }
                        if (d.alpha_dev) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] *= alpha;
                        }
                        if (d.out_dtype == XMC_BF16) {
                            if (d.res) {
                                float rr[8];
                                Vec8<XMC_BF16>::load(d.res, idx8, rr);
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] += rr[q];
                            }
                            Vec8<XMC_BF16>::store(d.dst, idx8, v);
                        } else {
                            if (d.res) {
                                float rr[8];
                                Vec8<XMC_F32>::load(d.res, idx8, rr);
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] += rr[q];
                            }
                            Vec8<XMC_F32>::store(d.dst, idx8, v);
                        }
                    }
            }
        }
    }
}

template <int BN>
int launch_ptile(const XmcConvDesc& d, const TileCfg& t, hipStream_t st) {
    int maxpatch = 0;
    for (int z = 0; z < d.nclass; ++z) maxpatch = t.PH[z] * t.PW[z] > maxpatch ? t.PH[z] * t.PW[z] : maxpatch;
    const int pstride = t.slab * 2 + 32;
    size_t pb = (size_t)((maxpatch * pstride + 15) & ~15), eb = (size_t)128 * (BN + 4) * 4;
    size_t lds = (pb > eb ? pb : eb) + (size_t)d.ntaps * BN * pstride;
    if (lds > 160 * 1024) return XMC_ESHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ptile_kernel<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static const bool v1 = getenv("XMC_PTILE_V1") != nullptr;
    const int ntiles = d.N * t.tiles_y * t.tiles_x;
    static const bool v2 = getenv("XMC_PTILE_V2") != nullptr;
    if (!v1 && !v2 && maxpatch <= 384) {                              // 384 = staging registers of ptile3 (PIT * 256 / cps)
        const size_t lds3 = pb + (size_t)d.ntaps * BN * pstride;
        const int per_cu3 = (lds3 <= 80 * 1024 && BN == 32 && t.slab == 32) ? 2 : 1;    // 8-wave workgroups; 2 fit when <= 128 VGPRs
        int g3 = 256 * per_cu3 / (int)((d.CDw / BN) * d.nclass);
        if (g3 < 1) g3 = 1;
        if (g3 > ntiles) g3 = ntiles;
        dim3 grid3((unsigned)g3, (unsigned)(d.CDw / BN), (unsigned)d.nclass);
        TileCfg ts = t;
        static const bool stamp_on = getenv("XMC_TILE_STAMP") != nullptr;
        static unsigned long long* stamp_dev = nullptr;
        if (stamp_on) {
            if (!stamp_dev) (void)hipMalloc(&stamp_dev, 32 * 16 * 8);
            (void)hipMemset(stamp_dev, 0, 32 * 16 * 8);
            ts.stamp = stamp_dev;
        }
#define XMC_PT3(SL, NTP)                                                                                                      \
    do {                                                                                                                      \
        static bool once = false;                                                                                             \
        if (!once) {                                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ptile3_kernel<BN, SL, NTP>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
            once = true;                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL((ptile3_kernel<BN, SL, NTP>), grid3, dim3(512), lds3, st, d, ts, ntiles);                          \
        xmc_note_kernel("ptile3_kernel<%d, %d, %d>", BN, SL, NTP);                                                            \
    } while (0)
        if (t.slab == 64) {
            if (d.ntaps == 9) XMC_PT3(64, 9); else if (d.ntaps == 4) XMC_PT3(64, 4); else XMC_PT3(64, 0);
        } else {
            if (d.ntaps == 9) XMC_PT3(32, 9); else if (d.ntaps == 4) XMC_PT3(32, 4); else XMC_PT3(32, 0);
        }
#undef XMC_PT3
        XMC_LAUNCH_CHECK();
        if (stamp_on) {
            static int printed = 0;
            unsigned long long h[32 * 16];
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, stamp_dev, sizeof h, hipMemcpyDeviceToHost);
            if (printed++ < 2) {
                double acc[8] = {0};
                int n = 0;
                for (int k = 4; k < 30; ++k) {
                    const unsigned long long* r = h + k * 16;
                    if (!r[0] || !r[16]) continue;
                    acc[0] += (double)(r[1] - r[0]);      // compute: wait at B1
                    acc[1] += (double)(r[2] - r[1]);      // compute: MFMA phase
                    acc[2] += (double)(r[3] - r[2]);      // compute: wait at B2
                    acc[3] += (double)(r[16] - r[3]);     // compute: epilogue
                    acc[4] += (double)(r[9] - r[8]);      // stage: address + issue
                    acc[5] += (double)(r[10] - r[9]);     // stage: wait at B2
                    acc[6] += (double)(r[11] - r[10]);    // stage: wait for loads
                    acc[7] += (double)(r[12] - r[11]);    // stage: LDS stores
                    ++n;
                }
                if (n) fprintf(stderr, "[ptile3 stamps, cycles/tile over %d tiles] compute: B1 wait %.0f  mfma %.0f  B2 wait %.0f  epilogue %.0f | "
                               "stage: issue %.0f  B2 wait %.0f  load wait %.0f  lds store %.0f\n", n, acc[0] / n, acc[1] / n, acc[2] / n,
                               acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n);
            }
        }
        return 0;
    }
    if (!v1) {
        const size_t lds2 = pb + (size_t)d.ntaps * BN * pstride;       // no epilogue staging
        const int per_cu2 = lds2 <= 80 * 1024 ? 2 : 1;
        int g2 = 256 * per_cu2 / (int)((d.CDw / BN) * d.nclass);
        if (g2 < 1) g2 = 1;
        if (g2 > ntiles) g2 = ntiles;
        dim3 grid2((unsigned)g2, (unsigned)(d.CDw / BN), (unsigned)d.nclass);
        static const bool no_unroll = getenv("XMC_PTILE_NOUNROLL") != nullptr;
        static const bool w8 = getenv("XMC_PTILE_W8") != nullptr;
        const int nt = no_unroll ? 0 : d.ntaps;
        if (w8 && t.slab == 64 && BN == 64 && nt == 9) {
            static bool once8 = false;
            if (!once8) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ptile2_kernel<BN, 64, 9, 512>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                once8 = true;
            }
            hipLaunchKernelGGL((ptile2_kernel<BN, 64, 9, 512>), grid2, dim3(512), lds2, st, d, t, ntiles);
            xmc_note_kernel("ptile2_kernel<%d, 64, 9, 512>", BN);
            XMC_LAUNCH_CHECK();
            return 0;
        }
#define XMC_PT2(SL, NTP)                                                                                                      \
    do {                                                                                                                      \
        static bool once = false;                                                                                             \
        if (!once) {                                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ptile2_kernel<BN, SL, NTP, 256>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
            once = true;                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL((ptile2_kernel<BN, SL, NTP, 256>), grid2, dim3(256), lds2, st, d, t, ntiles);                           \
        xmc_note_kernel("ptile2_kernel<%d, %d, %d, 256>", BN, SL, NTP);                                                            \
    } while (0)
        if (t.slab == 64) {
            if (nt == 9) XMC_PT2(64, 9); else if (nt == 4) XMC_PT2(64, 4); else XMC_PT2(64, 0);
        } else {
            if (nt == 9) XMC_PT2(32, 9); else if (nt == 4) XMC_PT2(32, 4); else XMC_PT2(32, 0);
        }
#undef XMC_PT2
        XMC_LAUNCH_CHECK();
        return 0;
    }
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    int gx = 256 * per_cu / (int)((d.CDw / BN) * d.nclass);
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    dim3 grid((unsigned)gx, (unsigned)(d.CDw / BN), (unsigned)d.nclass);
    hipLaunchKernelGGL((ptile_kernel<BN>), grid, dim3(256), lds, st, d, t, ntiles);
    xmc_note_kernel("ptile_kernel<%d>", BN);
    XMC_LAUNCH_CHECK();
    return 0;
}

template <int BN, int WM, int WN>
int launch_tile(const XmcConvDesc& d, const TileCfg& t, hipStream_t st) {
    int maxpatch = 0;
    for (int z = 0; z < d.nclass; ++z) maxpatch = t.PH[z] * t.PW[z] > maxpatch ? t.PH[z] * t.PW[z] : maxpatch;
    const int pstride = t.slab * 2 + 32;
    size_t lds = (size_t)((maxpatch * pstride + 15) & ~15) + 2 * (size_t)BN * pstride;
    size_t ep = (size_t)128 * (BN + 4) * 4;
    if (ep > lds) lds = ep;
    if (lds > 160 * 1024) return XMC_ESHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((unsigned)(d.N * t.tiles_y * t.tiles_x), (unsigned)(d.CDw / BN), (unsigned)d.nclass);
    hipLaunchKernelGGL((tile_kernel<BN, WM, WN>), grid, dim3(256), lds, st, d, t);
    xmc_note_kernel("tile_kernel<%d, %d, %d>", BN, WM, WN);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Returns 1 if the descriptor is eligible for the halo-tile kernel (and fills cfg), 0 otherwise.
static int tile_plan(const XmcConvDesc* d, TileCfg* t) {
    if (d->dtype != XMC_BF16 || d->SA != 1 || d->src_shift != 0) return 0;
    if (d->CS % 32 != 0 || d->MW % 16 != 0) return 0;
    if (d->ntaps < 2) return 0;                       // 1x1: nothing to reuse, the gather kernel streams it
    if (d->CDw > 64 && d->CS > 64) return 0;          // wide layers are MFMA-bound: 128x128 gather tiles win
    int TW = d->MW >= 32 ? 32 : 16;
    int TH = 256 / TW;
    if (d->MH % TH != 0 || d->MW % TW != 0) return 0;
    t->TH = TH; t->TW = TW; t->log2TW = TW == 32 ? 5 : 4;
    t->tiles_y = d->MH / TH; t->tiles_x = d->MW / TW;
    t->slab = (d->CS % 64 == 0) ? 64 : 32;
    for (int z = 0; z < d->nclass; ++z) {
        int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
        for (int k = 0; k < d->ntaps; ++k) {
            int h = d->dh[z][k], w = d->dw[z][k];
            hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax;
            wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax;
        }
        t->dh0[z] = hmin; t->dw0[z] = wmin;
        t->PH[z] = TH + (hmax - hmin); t->PW[z] = TW + (wmax - wmin);
        if (t->PH[z] * t->PW[z] > 12 * (256 / (t->slab / 8))) return 0;   // staging registers (PIT)
    }
    for (int k = 0; k < 4; ++k) t->pro[k] = nullptr;
    t->stamp = nullptr;
    static const int dbg = getenv("XMC_TILE_DBG") ? atoi(getenv("XMC_TILE_DBG")) : 0;
    t->dbg = dbg;
    return 1;
}

// entry used by xmc_conv_igemm's dispatcher (conv_igemm.hip) and by the fused-prologue ABI call
int xmc_conv_tile_try(const XmcConvDesc* d, const float* const* pro, void* stream) {
    TileCfg t;
    if (!tile_plan(d, &t)) return 1;   // not eligible
    if (pro) for (int k = 0; k < 4; ++k) t.pro[k] = pro[k];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    static const bool no_pt = getenv("XMC_NO_PTILE") != nullptr;
    if (!no_pt && d->CS <= 64 && t.slab == d->CS && d->CDw <= 64) {          // persistent, weights resident
        rc = d->CDw == 64 ? launch_ptile<64>(*d, t, st) : launch_ptile<32>(*d, t, st);
        if (rc != XMC_ESHAPE) return rc;
    }
    if (d->CDw % 128 == 0) rc = launch_tile<128, 2, 2>(*d, t, st);
    else if (d->CDw % 64 == 0) rc = launch_tile<64, 4, 1>(*d, t, st);
    else rc = launch_tile<32, 4, 1>(*d, t, st);
    return rc == XMC_ESHAPE ? 1 : rc;
}
