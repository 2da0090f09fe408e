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

namespace {

struct TileCfg {
    int TH, TW, log2TW;          // output tile (TH*TW == 256)
    int tiles_y, tiles_x;        // tiles per image
    int PH[XMC_MAX_CLASSES], PW[XMC_MAX_CLASSES];        // patch size per class
    int dh0[XMC_MAX_CLASSES], dw0[XMC_MAX_CLASSES];      // min tap offsets per class
    int slab;                    // channels per slab (32 or 64)
    const float* pro[4];         // optional prologue params g0,b0,g1,b1 : f32 [N][CS]; pro[0]==nullptr -> none
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
    const int ntiles = d.N * t.tiles_y * t.tiles_x;
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
