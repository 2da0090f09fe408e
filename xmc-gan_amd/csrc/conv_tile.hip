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
// (Round 1 carried an optional prologue here that applied DF-GAN's affine pair + LeakyReLU while the patch was staged.  No
// caller ever used it: the weight gradient of the same layer needs the activated tensor as its operand, so the tensor has to
// exist in HBM anyway and the fusion only moved a pass from the forward to the backward.  Removed in round 2.)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct TileCfg {
    int TH, TW, log2TW;          // output tile (TH*TW == 256)
    int tiles_y, tiles_x;        // tiles per image
    int PH[XMC_MAX_CLASSES], PW[XMC_MAX_CLASSES];        // patch size per class
    int dh0[XMC_MAX_CLASSES], dw0[XMC_MAX_CLASSES];      // min tap offsets per class
    int PHu, PWu, dh0u, dw0u;    // union of the classes' patches (one staged patch serves all classes)
    int slab;                    // channels per slab (32 or 64)
    int no_xcd;                  // XMC_DEBUG_DISPATCH=no_xcd_map: plain tile = blockIdx.x + k * gridDim.x walk (A/B)
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
        // ---- stage the patch for this slab
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
                        acc[i][j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, af[i]),
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
    float dacc = 0.f;
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
            const size_t ridx8 = res_index8(d, idx8, img, (a0 + ty) * d.DA + dph, (b0 + tx) * d.DA + dpw, a0 + ty, b0 + tx, (n0 >> 3) + cc);
            if (d.out_dtype == XMC_BF16) epilogue_tail<XMC_BF16>(d, idx8, ridx8, v, alpha, &dacc);
            else epilogue_tail<XMC_F32>(d, idx8, ridx8, v, alpha, &dacc);
        }
    }
    if (d.dot) {
        dacc = wave_sum(dacc);
        if (lane == 0) atomicAdd(d.dot, dacc);
    }
}


// Persistent kernel for Cin, Cout <= 64: ALL taps' weights stay in LDS for the life of the workgroup, which walks output tiles.
// Role-split workgroup of 8 waves: one wave per SIMD computes, the other stages, and the two run different phases at the same
// time (measured with s_memtime on the earlier single-role 4-wave version: a tile spent 4.6 k cycles in MFMAs and ~8 k in
// address arithmetic, LDS stores, the LDS-staged epilogue and waiting for memory, none of which one wave per SIMD overlaps;
// 64->64 @128^2: 0.64 ms -> 0.35 ms, 32->32 @256^2: 0.82 -> 0.46 ms = 4.6 TB/s):
//   waves 0-3 ("compute"): B1 | MFMAs over the staged patch           | B2 | epilogue from registers, global stores
//   waves 4-7 ("stage")  : B1 | addresses + global loads of next tile | B2 | registers -> LDS patch
// Two barriers per tile, both reached by all 8 waves.  The MFMA roles are swapped (A = weight rows, B = pixels), so a lane's
// accumulators are CONSECUTIVE output channels of ONE pixel (weight rows are permuted while they are copied to LDS) and the
// epilogue runs straight from registers with 16-byte stores.  Tap loop fully unrolled when NTAPS > 0.
// MC > 1: the MC output-parity classes of a stride-2 data gradient / fused upsample-conv are done by ONE workgroup from one
// staged patch (the union of their halos) instead of MC launches that each stage the same pixels: the source is read once.
// SA == 2: stride-2 forward (the 4x4 stride-2 layer of a discriminator block with <= 64 channels): tiles of 8x16 output pixels
// (two pixel blocks per compute wave), the (2*8+2) x (2*16+2) source patch stored as two column-parity planes per patch row so
// that the pixels of consecutive output columns for a fixed tap are consecutive (and bank-conflict free) LDS rows.
// M32: the compute waves use v_mfma_f32_32x32x16_bf16 (2 pixel blocks of 32 x BN/32 channel blocks of 32 per wave).  This kernel
// runs ONE matrix wave per SIMD, and a single wave issues the 16x16x32 form at only ~63 % of the pipe's rate (tests/diag/
// mfma_probe.hip: 1277 TF/s against 1948 for 32x32x16); same fragment bytes, half the MFMA instructions.  bf16 output, SA == 1,
// unrolled tap loop only.
// EPI (32x32x16 role): the epilogue's option set as a compile-time bit mask (kEpi*), or -1 = read the descriptor at run time.  The
// epilogue of this role is ~500 VALU instructions per wave and tile -- more issue time than the tile's 144 MFMAs -- and with every
// option a run-time branch per 8-channel unit it measured 0.535 ms where the same launch with its options folded takes 0.450
// (64 -> 64 @128^2, batch 256, block sum): the option sets that occur in the training step get an instantiation each.
template <int BN, int SLAB, int NTAPS, int MC, int SA = 1, bool M32 = false, int EPI = -1>
__global__ __launch_bounds__(512) void ptile3_kernel(const XmcConvDesc d, const TileCfg t, int ntiles) {
    static_assert(!M32 || (SA == 1 && NTAPS > 0), "32x32x16 form: unit stride, unrolled taps");
    constexpr int NS = 256;                      // threads per role
    constexpr int TM = SA == 1 ? 4 : 2, TN = BN / 16;
    constexpr int cps = SLAB / 8;
    // bytes between patch pixels / weight rows.  The 16-row fragment pattern of v_mfma_f32_16x16x32 is bank-conflict free at
    // 128 + 32 bytes; the 32-row pattern of the 32x32x16 form is NOT (rows r and r + 8 land on one 16-byte slot: 2-way on every
    // read, SQ_LDS_BANK_CONFLICT 0.44 per active cycle in round 2) but is at 128 + 16 (slot = 9 r mod 16: a permutation of any
    // 16 of 32 consecutive rows; checked by enumeration over every patch alignment)
    constexpr int pstride = SLAB * 2 + (M32 ? 16 : 32);
    constexpr int PIT = ((SA == 1 ? 384 : 18 * 34) * cps + NS - 1) / NS;     // patch units per staging thread (patches of <= 384 / 612 pixels)
    constexpr int UPL = BN / 32;
    constexpr int S = SLAB / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool stager = wave >= 4;
    const int rt = tid & (NS - 1);               // thread index within the role
    const int wm = wave & 3;
    const int cls = MC > 1 ? 0 : blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PH = MC > 1 ? t.PHu : t.PH[cls], PW = MC > 1 ? t.PWu : t.PW[cls];
    const int dh0 = MC > 1 ? t.dh0u : t.dh0[cls], dw0 = MC > 1 ? t.dw0u : t.dw0[cls];
    __shared__ int s_toff[XMC_MAX_TAPS];          // [class * ntaps + tap]
    // 32x32x16 role: the bias vectors of this workgroup's channels (the convolution's, and the recomputed shortcut's) in LDS -- read from
    // global memory inside the epilogue they were a round trip per tile between the last MFMA and the first store
    __shared__ __attribute__((aligned(16))) float s_bias[M32 ? 2 * BN : 4];
    if (M32 && tid >= 64 && tid < 64 + BN) {
        const int ch = n0 + tid - 64;
        s_bias[tid - 64] = (d.bias && ch < d.CD) ? d.bias[ch] : 0.f;
        s_bias[BN + tid - 64] = (d.sc_bias && ch < d.CD) ? d.sc_bias[ch] : 0.f;
    }
    if (tid < XMC_MAX_TAPS) {
        const int sl = tid < MC * d.ntaps ? tid : 0;
        const int c = MC > 1 ? sl / d.ntaps : cls, tt = MC > 1 ? sl % d.ntaps : sl;
        const int th = d.dh[c][tt] - dh0, tw = d.dw[c][tt] - dw0;
        s_toff[tid] = (SA == 1 ? th * PW + tw : (th * 2 + (tw & 1)) * (PW >> 1) + (tw >> 1)) * pstride;
    }
    const int cs_units = d.CS / 8;
    unsigned char* patch = smem;
    const int patch_bytes = (PH * PW * pstride + 15) & ~15;
    unsigned char* wall = smem + patch_bytes;                                            // [ntaps][BN][pstride]
    // SCI: the residual is the composed stem's shortcut, recomputed from the image (XmcConvDesc.sc_img): an 18 x 66 pixel image patch
    // (rows 2 a0 - 1 .., columns 2 b0 - 1 .., the first four channels = 8 bytes of each pixel, row-major) behind the weights
    constexpr bool SCI = M32 && EPI >= 0 && (EPI & kEpiScImg) != 0;
    constexpr int IPH = 18, IPW = 66, IPU = IPH * IPW;
    unsigned char* ipatch = wall + MC * (NTAPS > 0 ? NTAPS : 1) * BN * pstride;
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // physical weight row (n-block j, row q) holds logical output channel (j/2)*32 + (q/4)*8 + (j%2)*4 + q%4: lane group fc = q/4
    // then owns, for unit u = j/2, channels u*32 + fc*8 .. +7 and the four lane groups of a pixel store 64 contiguous bytes
    for (int id = tid; id < MC * d.ntaps * BN * cps; id += 512) {
        const int ch = id % cps, prow = (id / cps) % BN, tap = id / (cps * BN);          // tap = slice index c * ntaps + tap
        const int j = prow >> 4, q = prow & 15;
        // M32: MFMA row rho = 8a + 4h + r of a 32-block lands in lane half h, accumulator 4a + r <- channel 16h + 4a + r
        const int rho = prow & 31;
        const int lrow = M32 ? (prow >> 5) * 32 + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3)
                             : (j >> 1) * 32 + (q >> 2) * 8 + (j & 1) * 4 + (q & 3);
        *reinterpret_cast<u32x4*>(wall + (tap * BN + prow) * pstride + ch * 16) =
            w16[((size_t)d.wi[MC > 1 ? tap / d.ntaps : cls][MC > 1 ? tap % d.ntaps : tap] * d.CDw + n0 + lrow) * cs_units + ch];
    }
    // XCD-aware tile walk.  Workgroups are dealt to the chip's 8 XCDs round-robin by blockIdx, each XCD has its own L2, and a tile shares
    // its halo rows / columns with its neighbours: with tile = blockIdx.x + k * gridDim.x two neighbouring tiles NEVER meet in one L2 and
    // every halo is fetched from memory again (FETCH_SIZE = 1.33x the source tensor).  Here XCD x owns the contiguous tile range
    // [x T8, (x+1) T8) and its gridDim.x / 8 workgroups walk it side by side, so a halo row is in L2 when the tile below asks for it.
    const bool xaware = (gridDim.x & 7) == 0 && gridDim.y == 1 && gridDim.z == 1 && !t.no_xcd;
    const int T8 = (ntiles + 7) >> 3, xper = (int)gridDim.x >> 3;
    const int xbase = xaware ? ((int)blockIdx.x & 7) * T8 : 0, xlim = xaware ? (xbase + T8 < ntiles ? xbase + T8 : ntiles) : ntiles;
    const int tile0 = xaware ? xbase + ((int)blockIdx.x >> 3) : (int)blockIdx.x, tstep = xaware ? xper : (int)gridDim.x;

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
            halo[it] = !in ? 0u : ((py < -dh0 ? 1u : 0u) | (py >= t.TH * SA - dh0 ? 2u : 0u) | (px < -dw0 ? 4u : 0u) | (px >= t.TW * SA - dw0 ? 8u : 0u));
        }
        constexpr int IIT = SCI ? (IPU + NS - 1) / NS : 1;           // image-patch units per staging thread (5)
        u32x2 iv[IIT];
        unsigned ihalo = 0;                       // 4 bits per unit: the patch's first / last row / column (outside the image on that border)
        if (SCI) {
#pragma unroll
            for (int it = 0; it < IIT; ++it) {
                const int u = rt + it * NS, iy = u / IPW, ix = u - iy * IPW;
                ihalo |= (u < IPU ? ((iy == 0 ? 1u : 0u) | (iy == IPH - 1 ? 2u : 0u) | (ix == 0 ? 4u : 0u) | (ix == IPW - 1 ? 8u : 0u)) : 15u) << (4 * it);
            }
        }
        const u32x4* __restrict__ img16 = reinterpret_cast<const u32x4*>(d.sc_img);
        u32x4 pv[PIT];
        unsigned char pb8[PIT];                   // sign bytes of the staged units (XmcConvDesc.mask_bits: the source is a masked gradient)
        const unsigned char* __restrict__ bits8 = reinterpret_cast<const unsigned char*>(d.mask_bits);
        unsigned okmask = 0;
        auto issue = [&](int tile) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int ty = trem / t.tiles_x, tx = trem - ty * t.tiles_x;
            const int a0 = ty * t.TH * SA, b0 = tx * t.TW * SA;        // tile origin in the source
            const int base = ((img * d.SH + a0) * d.SW + b0) * cs_units;
            // a halo row/column is outside the image only for tiles on that border (halo depth <= tile size)
            const unsigned border = (a0 + dh0 < 0 ? 1u : 0u) | (a0 + PH + dh0 > d.SH ? 2u : 0u) |
                                    (b0 + dw0 < 0 ? 4u : 0u) | (b0 + PW + dw0 > d.SW ? 8u : 0u);
            okmask = 0;
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const bool ok = (halo[it] & border) == 0;
                pv[it] = src16[(unsigned)(base + (ok ? psrc[it] : 0))];
                if (bits8) pb8[it] = bits8[(unsigned)(base + (ok ? psrc[it] : 0))];
                okmask |= ok ? (1u << it) : 0u;
            }
            okmask &= inpatch;
            if (SCI) {
                const int IW = 2 * d.SW;
                const long long ibase = ((long long)img * (2 * d.SH) + 2 * a0 - 1) * IW + 2 * b0 - 1;
#pragma unroll
                for (int it = 0; it < IIT; ++it) {
                    const int u = rt + it * NS, iy = u / IPW, ix = u - iy * IPW;
                    const bool ok = u < IPU && (((ihalo >> (4 * it)) & 15u) & border) == 0;
                    iv[it] = ok ? *reinterpret_cast<const u32x2*>(img16 + (ibase + (long long)iy * IW + ix)) : u32x2{0, 0};
                }
            }
        };
        auto commit = [&]() {
            const bool all_ok = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(okmask != inpatch) == 0);
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int pp = ppix0 + it * ppix_step;
                int lp = pp;
                if (SA == 2) {                    // patch pixel (py, px) -> row py, plane px & 1, column px >> 1
                    const int py = pp / PW, px = pp - py * PW;
                    lp = (py * 2 + (px & 1)) * (PW >> 1) + (px >> 1);
                }
                u32x4 v = pv[it];
                if (bits8) v = xmc_apply_sign_bits(v, pb8[it]);
                if (!all_ok && !((okmask >> it) & 1)) v = u32x4{0, 0, 0, 0};       // padding (border tiles only)
                if ((inpatch >> it) & 1) *reinterpret_cast<u32x4*>(patch + lp * pstride + pchunk * 16) = v;
            }
            if (SCI) {
#pragma unroll
                for (int it = 0; it < IIT; ++it)
                    if (rt + it * NS < IPU) *reinterpret_cast<u32x2*>(ipatch + (rt + it * NS) * 8) = iv[it];
            }
        };
        if (tile0 < xlim) {
            issue(tile0);
            commit();
        }
        __syncthreads();                          // weights + first patch staged
        for (int tile = tile0; tile < xlim; tile += tstep) {
            const int next = tile + tstep;
            __syncthreads();                      // B1
            if (next < xlim) issue(next);
            __syncthreads();                      // B2: the compute waves are done reading the patch
            if (next < xlim) commit();
        }
    } else if constexpr (M32) {
        // ------------------------------------------------------------------------------------------------ compute role, 32x32x16
        typedef __attribute__((ext_vector_type(16))) float f32x16;
        constexpr int NB = BN / 32, S16 = SLAB / 16, NST = NTAPS * S16, G = 2 + NB, M = 2 * NB;
        const int l32 = lane & 31, hh = lane >> 5;
        const int cd8 = d.CD / 8;
        int abyte[2], pty[2], ptx[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int ml = wm * 64 + pb * 32 + l32;
            pty[pb] = ml >> t.log2TW; ptx[pb] = ml & (t.TW - 1);
            abyte[pb] = (pty[pb] * PW + ptx[pb]) * pstride + hh * 16;
        }
        const int bbyte = l32 * pstride + hh * 16;
        const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
        const float pscale = d.pool_scale == 0.f ? 0.25f : d.pool_scale;
        constexpr bool RT = EPI < 0;                       // options read from the descriptor
        const bool e_bias = RT ? d.bias != nullptr : (EPI & kEpiBias) != 0;
        const bool e_tanh = RT ? d.act == XMC_ACT_TANH : false;
        const bool e_round = RT ? (d.dst2 != nullptr || d.round_act != 0) : (EPI & (kEpiRound | kEpiDst2)) != 0;
        const bool e_dst2 = RT ? d.dst2 != nullptr : (EPI & kEpiDst2) != 0;
        const bool e_alpha = RT ? d.alpha_dev != nullptr : (EPI & kEpiAlpha) != 0;
        const bool e_mask = RT ? d.mask != nullptr : (EPI & kEpiMask) != 0;
        const bool e_res = SCI ? true : (RT ? d.res != nullptr : (EPI & kEpiRes) != 0);        // SCI: the residual vectors are computed, not loaded
        const bool e_post = RT ? d.post_act == XMC_ACT_LRELU : (EPI & kEpiPost) != 0;
        const bool e_pool = RT ? d.dst_pool != nullptr : (EPI & kEpiPool) != 0;
        constexpr bool e_sign = !RT && (EPI & kEpiSign) != 0;       // sign bits / dot: compile-time sets only (the launcher declines otherwise)
        constexpr bool e_dot = !RT && (EPI & kEpiDot) != 0;
        float dacc = 0.f;                                  // running sum for XmcConvDesc.dot over this workgroup's tiles
        const float slope = RT ? (d.act == XMC_ACT_LRELU ? XMC_LRELU : (d.act == XMC_ACT_RELU ? 0.f : 1.f)) : ((EPI & kEpiLrelu) ? XMC_LRELU : 1.f);
        const float rs = SCI ? 1.f : (d.res_scale == 0.f ? 1.f : d.res_scale);
        // SCI: the shortcut's weights W_B as A fragments (K step = window row u: lane half hh holds taps (u, 2 hh), (u, 2 hh + 1) x 4 channels),
        // in registers for the whole launch; the image fragment of (pixel block, u) is ONE ds_read_b128 at row 2 py + u, column 2 px + 2 hh
        u32x4 scw[SCI ? 4 : 1][SCI ? NB : 1];
        int ibyte[2] = {0, 0};
        if (SCI) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < NB; ++c) scw[u][c] = reinterpret_cast<const u32x4*>(d.sc_frag)[(u * NB + c) * 64 + lane];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) ibyte[pb] = ((2 * pty[pb]) * IPW + 2 * ptx[pb] + 2 * hh) * 8;
        }
        __syncthreads();                          // weights + first patch staged
        int toffr[MC * NTAPS];
#pragma unroll
        for (int k = 0; k < MC * NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);
        for (int tile = tile0; tile < xlim; tile += tstep) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
            __syncthreads();                      // B1: patch of this tile is in LDS
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) {
                const int ccls = MC > 1 ? mc : cls;
                const int dph = d.dph[ccls], dpw = d.dpw[ccls];
                f32x16 acc[2][NB];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int c = 0; c < NB; ++c)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[pb][c][e] = 0.f;
                // The epilogue's mask / residual vectors are requested NOW, before the tile's MFMAs: their memory latency (one
                // round trip per channel unit when they were requested in the epilogue, ~15 % of a tile) hides behind the K loop.
                const int dbase = (((img * d.DH + a0 * d.DA + dph) * d.DW) + b0 * d.DA + dpw) * cd8 + (n0 >> 3);
                const int rbase = ((img * d.MH + a0) * d.MW + b0) * cd8 + (n0 >> 3);
                int eo[2];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) eo[pb] = dbase + ((pty[pb] * d.DA) * d.DW + ptx[pb] * d.DA) * cd8;
                bf16x8 mkv[NB][2][2], rrv[NB][2][2];
                if (SCI) {
                    // shortcut of this tile from the staged image patch: 4 K steps x 2 pixel blocks x NB channel blocks, then + bias and
                    // rounded to the 16-bit format -- the values the stem kernel used to store and this epilogue used to load
                    f32x16 sca[2][NB];
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int c = 0; c < NB; ++c)
#pragma unroll
                            for (int e = 0; e < 16; ++e) sca[pb][c][e] = 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb) {
                            const u32x4 xi = *reinterpret_cast<const u32x4*>(ipatch + u * IPW * 8 + ibyte[pb]);
#pragma unroll
                            for (int c = 0; c < NB; ++c)
                                sca[pb][c] = XMC_MFMA_32x32x16(__builtin_bit_cast(bf16x8, scw[u][c]), __builtin_bit_cast(bf16x8, xi), sca[pb][c], 0, 0, 0);
                        }
#pragma unroll
                    for (int c = 0; c < NB; ++c)
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            const float* bp = s_bias + BN + 32 * c + 16 * hh + 8 * v2;
                            const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp), b1v = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    rrv[c][pb][v2][r] = (xmc_h16)(sca[pb][c][8 * v2 + r] + b0v[r]);
                                    rrv[c][pb][v2][4 + r] = (xmc_h16)(sca[pb][c][8 * v2 + 4 + r] + b1v[r]);
                                }
                        }
                }
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const int ub = c * 4 + hh * 2;
                    if (n0 + ub * 8 >= d.CD) continue;
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            const size_t idx8 = (size_t)(eo[pb] + ub + v2);
                            if (e_mask) mkv[c][pb][v2] = reinterpret_cast<const bf16x8*>(d.mask)[idx8];
                            if (e_res && !SCI) {
                                size_t rix = idx8;
                                if (d.res_mode == 1) rix = (size_t)(rbase + (pty[pb] * d.MW + ptx[pb]) * cd8 + ub + v2);
                                else if (d.res_mode == 2)
                                    rix = res_index8(d, idx8, img, (a0 + pty[pb]) * d.DA + dph, (b0 + ptx[pb]) * d.DA + dpw, 0, 0, (n0 >> 3) + ub + v2);
                                rrv[c][pb][v2] = reinterpret_cast<const bf16x8*>(d.res)[rix];
                            }
                        }
                }
                // K16 step st = (tap, 16-channel piece s): fragments [0, 2) pixels, [2, G) weights; the reads of step st+1 are dealt
                // out between the MFMAs of step st and pinned there
                u32x4 fr_[2][G];
                auto rd = [&](int st, int g) -> u32x4 {
                    const int tap = st / S16, sI = st % S16;
                    if (g < 2) return *reinterpret_cast<const u32x4*>(patch + toffr[mc * NTAPS + tap] + sI * 32 + abyte[g]);
                    return *reinterpret_cast<const u32x4*>(wall + (mc * NTAPS + tap) * BN * pstride + bbyte + sI * 32 + (g - 2) * 32 * pstride);
                };
#pragma unroll
                for (int g = 0; g < G; ++g) fr_[0][g] = rd(0, g);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int cur = st & 1, nxt = cur ^ 1;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (st + 1 < NST) fr_[nxt][g] = rd(st + 1, g);
#pragma unroll
                        for (int m = g * M / G; m < (g + 1) * M / G; ++m) {
                            const int pb = m % 2, c = m / 2;
                            acc[pb][c] = XMC_MFMA_32x32x16(__builtin_bit_cast(bf16x8, fr_[cur][2 + c]),
                                                                                  __builtin_bit_cast(bf16x8, fr_[cur][pb]), acc[pb][c], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (mc == MC - 1) __syncthreads();    // B2: patch may be overwritten
                // epilogue from registers: acc[pb][c][8v + r] = pixel wm*64 + pb*32 + l32, channel n0 + 32c + 16hh + 8v + r
                bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
                unsigned sgn[2] = {0u, 0u};           // packed sign bytes of this lane's four units, per pixel block
                const bool pack_sign = e_sign && NB == 2 && d.CD == 64;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const int ub = c * 4 + hh * 2;        // first of this lane's two 8-channel units of the block
                    if (n0 + ub * 8 >= d.CD) continue;
                    float fin[2][2][8];
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            const size_t idx8 = (size_t)(eo[pb] + ub + v2);
                            float v[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = acc[pb][c][8 * v2 + r];
                            if (e_bias) {
                                const float* bp = s_bias + (ub + v2) * 8;
                                const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp), b1v = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) { v[r] += b0v[r]; v[4 + r] += b1v[r]; }
                            }
                            if (e_tanh) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] = d.res || d.mask || d.dst2 ? tanhf(v[r]) : tanh_fast(v[r]);
                            } else if (slope != 1.f) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], v[r] * slope);
                            }
                            if (e_sign) {
                                unsigned sb = 0;
#pragma unroll
                                for (int r = 0; r < 8; ++r) sb |= (v[r] > 0.f ? 1u : 0u) << r;
                                // one byte per 8-channel unit.  A lane owns units hh*2 + {0,1} (c = 0) and 4 + hh*2 + {0,1} (c = 1) of its
                                // pixel: with all 8 units of the pixel in play (CD == 64) the bytes are collected and the pixel's 8 sign
                                // bytes leave as ONE dword per lane (below) instead of four byte stores per lane and pixel block
                                if (pack_sign) sgn[pb] |= sb << (8 * (2 * c + v2));
                                else reinterpret_cast<unsigned char*>(d.sign_bits)[idx8] = (unsigned char)sb;
                            }
                            if (e_round) {
                                bf16x8 o2;
#pragma unroll
                                for (int r = 0; r < 8; ++r) { o2[r] = (xmc_h16)v[r]; v[r] = (float)o2[r]; }
                                if (e_dst2) reinterpret_cast<bf16x8*>(d.dst2)[idx8] = o2;
                            }
                            if (e_dot && e_mask) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) dacc += v[r] * (float)mkv[c][pb][v2][r];
                            }
                            if (e_alpha) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] *= alpha;
                            }
                            if (e_mask) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] *= lrelu_slope((float)mkv[c][pb][v2][r]);
                            }
                            if (e_res) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] += rs * (float)rrv[c][pb][v2][r];
                            }
                            if (e_post) {
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], v[r] * XMC_LRELU);
                            }
                            bf16x8 o;
#pragma unroll
                            for (int r = 0; r < 8; ++r) { o[r] = (xmc_h16)v[r]; fin[pb][v2][r] = (float)o[r]; }
                            dst8[idx8] = o;
                        }
                    if (MC == 1 && e_pool) {
                        // third output: 2x2 average of the rounded block output.  8x32 tiles: the wave's two 32-pixel blocks are tile rows
                        // 2wm and 2wm+1 (vertical pair = the two blocks of a lane); 16x16 tiles: a block is two rows of 16 (vertical pair
                        // = lane ^ 16).  Horizontal neighbour: lane ^ 1.
                        bf16x8* __restrict__ pool8 = reinterpret_cast<bf16x8*>(d.dst_pool);
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            if (t.log2TW == 5) {
                                bf16x8 o;
#pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    float sm = fin[0][v2][r] + fin[1][v2][r];
                                    sm += xmc_xor1(sm);
                                    o[r] = (xmc_h16)(pscale * sm);
                                }
                                if ((l32 & 1) == 0)
                                    pool8[((img * (d.DH >> 1) + ((a0 + pty[0]) >> 1)) * (d.DW >> 1) + ((b0 + ptx[0]) >> 1)) * cd8 + (n0 >> 3) + ub + v2] = o;
                            } else {
#pragma unroll
                                for (int pb = 0; pb < 2; ++pb) {
                                    bf16x8 o;
#pragma unroll
                                    for (int r = 0; r < 8; ++r) {
                                        float sm = fin[pb][v2][r];
                                        sm += __shfl_xor(sm, 16, 64);
                                        sm += xmc_xor1(sm);
                                        o[r] = (xmc_h16)(pscale * sm);
                                    }
                                    if ((l32 & 17) == 0)
                                        pool8[((img * (d.DH >> 1) + ((a0 + pty[pb]) >> 1)) * (d.DW >> 1) + ((b0 + ptx[pb]) >> 1)) * cd8 + (n0 >> 3) + ub + v2] = o;
                                }
                            }
                        }
                    }
                }
                if (e_sign && pack_sign) {
                    // lane halves hh = 0 / 1 of a pixel hold bytes {0,1 | 4,5} / {2,3 | 6,7}: swap the halves they do not store, then
                    // lane (l32, hh) writes bytes 4*hh .. 4*hh+3 -- 32 pixels x 8 bytes = 256 contiguous bytes per store instruction
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const unsigned keep = hh == 0 ? (sgn[pb] & 0xFFFFu) : (sgn[pb] >> 16);
                        const unsigned send = hh == 0 ? (sgn[pb] >> 16) : (sgn[pb] & 0xFFFFu);
                        const unsigned recv = (unsigned)__shfl_xor((int)send, 32, 64);
                        const unsigned word = hh == 0 ? (keep | (recv << 16)) : (recv | (keep << 16));
                        reinterpret_cast<unsigned*>(d.sign_bits)[(size_t)(eo[pb] >> 2) + hh] = word;      // eo = pixel * 8 sign bytes
                    }
                }
            }
        }
        if (e_dot) {                              // one atomic per wave for the whole launch
            dacc = wave_sum(dacc);
            if (lane == 0) atomicAdd(d.dot, dacc);
        }
    } else {
        // ------------------------------------------------------------------------------------------------ compute role
        const int fr = lane & 15, fc = lane >> 4;
        const int cd8 = d.CD / 8;
        int abyte[TM], eoff[TM], roff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * (16 * TM) + i * 16 + fr;
            const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
            abyte[i] = ((SA == 1 ? ty * PW : 2 * ty * PW) + tx) * pstride + fc * 16;      // SA == 2: row 2*ty, plane 0, column tx
            eoff[i] = ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + fc;
            roff[i] = (ty * d.MW + tx) * cd8 + fc;                // same pixel in the [N,MH,MW,CD] grid (res_mode 1)
        }
        const int bbyte = fr * pstride + fc * 16;
        const int ch0 = n0 + fc * 8;                          // unit u of this lane: channels ch0 + 32*u .. +7
        const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
        const float pscale = d.pool_scale == 0.f ? 0.25f : d.pool_scale;
        float bias8[UPL][8];
#pragma unroll
        for (int u = 0; u < UPL; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) bias8[u][c] = (d.bias && ch0 + u * 32 < d.CD) ? d.bias[ch0 + u * 32 + c] : 0.f;
        // one uniform decision instead of a chain of branches per stored unit
        constexpr bool RT = EPI < 0;                       // epilogue options read from the descriptor (common.h: kEpi*)
        constexpr int EB = RT ? 0 : (EPI & ~kEpiBias);     // (the bias vector is always added: zeros when there is none)
        const bool fast = RT ? (d.out_dtype == XMC_BF16 && d.res == nullptr && d.mask == nullptr && d.alpha_dev == nullptr && d.dst2 == nullptr &&
                                d.dst_pool == nullptr && d.post_act == XMC_ACT_NONE && d.sign_bits == nullptr && d.dot == nullptr &&
                                (d.act == XMC_ACT_NONE || d.act == XMC_ACT_LRELU || d.act == XMC_ACT_TANH))
                             : (EB == 0 || EB == kEpiLrelu);
        const bool do_tanh = RT ? d.act == XMC_ACT_TANH : false;
        const float slope = RT ? (d.act == XMC_ACT_LRELU ? XMC_LRELU : 1.f) : ((EPI & kEpiLrelu) ? XMC_LRELU : 1.f);
        const bool e_lrelu = RT ? d.act == XMC_ACT_LRELU : (EPI & kEpiLrelu) != 0;
        const bool e_relu = RT ? d.act == XMC_ACT_RELU : false;
        const bool e_round = RT ? (d.dst2 != nullptr || d.round_act != 0) : (EPI & (kEpiRound | kEpiDst2)) != 0;
        const bool e_dst2 = RT ? d.dst2 != nullptr : (EPI & kEpiDst2) != 0;
        const bool e_alpha = RT ? d.alpha_dev != nullptr : (EPI & kEpiAlpha) != 0;
        const bool e_mask = RT ? d.mask != nullptr : (EPI & kEpiMask) != 0;
        const bool e_res = RT ? d.res != nullptr : (EPI & kEpiRes) != 0;
        const bool e_post = RT ? d.post_act == XMC_ACT_LRELU : (EPI & kEpiPost) != 0;
        const bool e_pool = RT ? d.dst_pool != nullptr : (EPI & kEpiPool) != 0;
        constexpr bool e_sign = !RT && (EPI & kEpiSign) != 0;       // sign bits / dot: compile-time sets only (the launcher declines otherwise)
        constexpr bool e_dot = !RT && (EPI & kEpiDot) != 0;
        float dacc = 0.f;
        __syncthreads();                          // weights + first patch staged
        int toffr[NTAPS > 0 ? MC * NTAPS : 1];
        if constexpr (NTAPS > 0) {
#pragma unroll
            for (int k = 0; k < MC * NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);
        }
        for (int tile = tile0; tile < xlim; tile += tstep) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
            __syncthreads();                      // B1: patch of this tile is in LDS
            constexpr bool HOIST = UPL == 1;
            // epilogue operands requested ahead of the MFMAs (below).  They live across the class loop: a ROW-indexed residual
            // (res_mode 1: the shortcut gradient of a discriminator block, one vector per 2x2 block of dst) is the same for all MC
            // classes of a tile and is requested once, with the first class -- not once per class in front of a 16-MFMA K loop
            bf16x8 mkv[HOIST ? TM : 1], rrv[HOIST ? TM : 1];
#pragma unroll
          for (int mc = 0; mc < MC; ++mc) {
            const int ccls = MC > 1 ? mc : cls;
            const int dph = d.dph[ccls], dpw = d.dpw[ccls];
            f32x4 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // bf16 epilogue operands (LeakyReLU' mask, residual) are requested NOW, before the tile's MFMAs, as in the 32x32x16 role:
            // requested in the epilogue, one memory round trip per tile stood between the last MFMA and the first store -- at 32
            // channels as long as the tile's whole K loop (32 -> 32 @ 256^2 with the block sum: 1.01 ms against 0.55 without).
            const bool pre = RT ? d.out_dtype == XMC_BF16 : true;
            const int dbase = (((img * d.DH + a0 * d.DA + dph) * d.DW) + b0 * d.DA + dpw) * cd8 + (n0 >> 3);
            const int rbase = ((img * d.MH + a0) * d.MW + b0) * cd8 + (n0 >> 3);
            // (one 32-channel unit per lane only: with two, the 64 extra live registers spill; the 64-channel layers with such an
            // epilogue run in the 32x32x16 role anyway)
            const bool res_once = MC > 1 && d.res_mode == 1;
            if (HOIST && pre && (e_mask || e_res)) {
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 32 >= d.CD) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const size_t idx8 = (size_t)(dbase + eoff[i] + u * 4);
                        size_t rix = d.res_mode == 1 ? (size_t)(rbase + roff[i] + u * 4) : idx8;
                        if (d.res_mode == 2) {
                            const int ml_ = wm * (16 * TM) + i * 16 + fr;
                            const int y_ = (a0 + (ml_ >> t.log2TW)) * d.DA + dph, x_ = (b0 + (ml_ & (t.TW - 1))) * d.DA + dpw;
                            rix = res_index8(d, idx8, img, y_, x_, 0, 0, (n0 >> 3) + fc + u * 4);
                        }
                        if (e_mask) mkv[HOIST ? i : 0] = reinterpret_cast<const bf16x8*>(d.mask)[idx8];
                        if (e_res && (mc == 0 || !res_once)) rrv[HOIST ? i : 0] = reinterpret_cast<const bf16x8*>(d.res)[rix];
                    }
                }
            }
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
                        acc[i][j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, w[j]),
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
                    if (g < TM) return *reinterpret_cast<const u32x4*>(patch + toffr[mc * NTAPS + tap] + sub * 64 + abyte[g]);
                    return *reinterpret_cast<const u32x4*>(wall + (mc * NTAPS + tap) * BN * pstride + bbyte + sub * 64 + (g - TM) * 16 * pstride);
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
                            acc[mi][mj] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, fr_[cur][TM + mj]),
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
            if (mc == MC - 1) __syncthreads();    // B2: patch may be overwritten
            // epilogue from registers: acc[i][j][r] = pixel (m-block i, fr), channel ch0 + j*4 + r
            if (fast) {
                bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst) + dbase;
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 32 >= d.CD) continue;           // same for every pixel of the lane: one branch per unit column
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        bf16x8 o;
                        if (slope != 1.f) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x0 = acc[i][2 * u][q] + bias8[u][q], x1 = acc[i][2 * u + 1][q] + bias8[u][4 + q];
                                o[q] = (xmc_h16)fmaxf(x0, x0 * slope);
                                o[4 + q] = (xmc_h16)fmaxf(x1, x1 * slope);
                            }
                        } else if (do_tanh) {     // generator output layer (df_gan.py:88-90)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                o[q] = (xmc_h16)tanh_fast(acc[i][2 * u][q] + bias8[u][q]);
                                o[4 + q] = (xmc_h16)tanh_fast(acc[i][2 * u + 1][q] + bias8[u][4 + q]);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                o[q] = (xmc_h16)(acc[i][2 * u][q] + bias8[u][q]);
                                o[4 + q] = (xmc_h16)(acc[i][2 * u + 1][q] + bias8[u][4 + q]);
                            }
                        }
                        dst8[eoff[i] + u * 4] = o;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UPL; ++u) {
                    if (ch0 + u * 32 >= d.CD) continue;
                    float fin[TM][8];
                    size_t rix[TM];           // f32 destination: epilogue_tail loads its own operands; two units: one round trip per unit
                    bf16x8 mkl[HOIST ? 1 : TM], rrl[HOIST ? 1 : TM];
                    if (!pre || !HOIST) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const size_t idx8 = (size_t)(dbase + eoff[i] + u * 4);
                            rix[i] = d.res_mode == 1 ? (size_t)(rbase + roff[i] + u * 4) : idx8;
                            if (d.res_mode == 2) {
                                const int ml_ = wm * (16 * TM) + i * 16 + fr;
                                const int y_ = (a0 + (ml_ >> t.log2TW)) * d.DA + dph, x_ = (b0 + (ml_ & (t.TW - 1))) * d.DA + dpw;
                                rix[i] = res_index8(d, idx8, img, y_, x_, 0, 0, (n0 >> 3) + fc + u * 4);
                            }
                            if (pre && e_mask) mkl[HOIST ? 0 : i] = reinterpret_cast<const bf16x8*>(d.mask)[idx8];
                            if (pre && e_res) rrl[HOIST ? 0 : i] = reinterpret_cast<const bf16x8*>(d.res)[rix[i]];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const size_t idx8 = (size_t)(dbase + eoff[i] + u * 4);
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q] = acc[i][2 * u][q] + bias8[u][q]; v[4 + q] = acc[i][2 * u + 1][q] + bias8[u][4 + q]; }
                        if (e_lrelu) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = lrelu_f(v[q]);
                        } else if (e_relu) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                        } else if (do_tanh) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = tanhf(v[q]);
                        }
                        if (pre) {      // epilogue_tail<XMC_BF16> with the loads hoisted
                            bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
                            if (e_sign) {
                                unsigned sb = 0;
#pragma unroll
                                for (int q = 0; q < 8; ++q) sb |= (v[q] > 0.f ? 1u : 0u) << q;
                                reinterpret_cast<unsigned char*>(d.sign_bits)[idx8] = (unsigned char)sb;
                            }
                            if (e_round) {
                                bf16x8 o2;
#pragma unroll
                                for (int q = 0; q < 8; ++q) { o2[q] = (xmc_h16)v[q]; v[q] = (float)o2[q]; }
                                if (e_dst2) reinterpret_cast<bf16x8*>(d.dst2)[idx8] = o2;
                            }
                            if (e_dot && e_mask) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) dacc += v[q] * (float)(HOIST ? mkv[HOIST ? i : 0] : mkl[HOIST ? 0 : i])[q];
                            }
                            if (e_alpha) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] *= alpha;
                            }
                            if (e_mask) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] *= lrelu_slope((float)(HOIST ? mkv[HOIST ? i : 0] : mkl[HOIST ? 0 : i])[q]);
                            }
                            if (e_res) {
                                const float rs = d.res_scale == 0.f ? 1.f : d.res_scale;
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] += rs * (float)(HOIST ? rrv[HOIST ? i : 0] : rrl[HOIST ? 0 : i])[q];
                            }
                            if (e_post) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], v[q] * XMC_LRELU);
                            }
                            bf16x8 o;
#pragma unroll
                            for (int q = 0; q < 8; ++q) { o[q] = (xmc_h16)v[q]; fin[i][q] = (float)o[q]; }
                            dst8[idx8] = o;
                        } else {
                            epilogue_tail<XMC_F32>(d, idx8, rix[i], v, alpha, &dacc);
#pragma unroll
                            for (int q = 0; q < 8; ++q) fin[i][q] = v[q];
                        }
                    }
                    if (SA == 1 && MC == 1 && TM == 4 && e_pool) {
                        // third output: 2x2 average of the rounded block output (the launcher admits it for bf16, DA == 1 only).
                        // Vertical neighbour = pixel block i+2 (8x32 tiles) / i+1 (16x16 tiles) of this lane, horizontal = lane ^ 1.
                        bf16x8* __restrict__ pool8 = reinterpret_cast<bf16x8*>(d.dst_pool);
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const int i0 = t.log2TW == 5 ? pr : 2 * pr;
                            const int ml_ = wm * 64 + i0 * 16 + fr;
                            const int ty_ = ml_ >> t.log2TW, tx_ = ml_ & (t.TW - 1);
                            bf16x8 o;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                float sm = t.log2TW == 5 ? fin[pr][q] + fin[(pr + 2) % TM][q] : fin[(2 * pr) % TM][q] + fin[(2 * pr + 1) % TM][q];
                                sm += xmc_xor1(sm);
                                o[q] = (xmc_h16)(pscale * sm);
                            }
                            if ((fr & 1) == 0)
                                pool8[((img * (d.DH >> 1) + ((a0 + ty_) >> 1)) * (d.DW >> 1) + ((b0 + tx_) >> 1)) * cd8 + (n0 >> 3) + fc + u * 4] = o;
                        }
                    }
                }
            }
          }
        }
        if (e_dot) {
            dacc = wave_sum(dacc);
            if (lane == 0) atomicAdd(d.dot, dacc);
        }
    }
}

template <int BN>
int launch_ptile(const XmcConvDesc& d, const TileCfg& t, hipStream_t st) {
    // (a descriptor with sc_img is served by the 32x32x16 block-end sets below or not at all)
    if (d.sc_img && !(d.SA == 1 && d.ntaps == 9 && d.nclass == 1 && t.slab == 64 && d.out_dtype == XMC_BF16 && (d.dst_pool || d.sign_bits))) return XMC_ESHAPE;
    int maxpatch = 0;
    for (int z = 0; z < d.nclass; ++z) maxpatch = t.PH[z] * t.PW[z] > maxpatch ? t.PH[z] * t.PW[z] : maxpatch;
    if (d.SA == 2) {                                                  // 4x4 stride 2, 32 input channels: 612-pixel patch in planes
        const size_t lds2 = (size_t)((maxpatch * 96 + 15) & ~15) + (size_t)16 * BN * 96;
        if (lds2 > XMC_MAX_DYN_LDS || t.slab != 32) return XMC_ESHAPE;
        const int nt2 = d.N * t.tiles_y * t.tiles_x;
        int g2 = 256 / (int)(d.CDw / BN);
        if (g2 > nt2) g2 = nt2;
        static const bool no_epi2 = xmc_debug_off("no_ptile_epi");
        const int epi2 = no_epi2 ? -1 : (xmc_epi_mask(d) & ~kEpiBias);          // (< 0 stays < 0: bit 0 of -1 cleared is -2)
#define XMC_PT3S2(E)                                                                                                        \
        if (epi2 == (E)) {                                                                                                  \
            XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 32, 16, 1, 2, false, (E)>));                                               \
            hipLaunchKernelGGL((ptile3_kernel<BN, 32, 16, 1, 2, false, (E)>), dim3(g2, d.CDw / BN, 1), dim3(512), lds2, st, d, t, nt2); \
            xmc_note_kernel("ptile3_kernel<%d, 32, 16, 1, 2>", BN);                                                         \
            XMC_LAUNCH_CHECK();                                                                                             \
            return 0;                                                                                                       \
        }
        XMC_PT3S2(kEpiLrelu) XMC_PT3S2(0) XMC_PT3S2(kEpiMask)
#undef XMC_PT3S2
        if (d.sign_bits || d.dot) return XMC_ESHAPE;     // only compile-time option sets carry these two
        xmc_note_generic_epi("ptile3<s2>", epi2);
        XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 32, 16, 1, 2>));
        hipLaunchKernelGGL((ptile3_kernel<BN, 32, 16, 1, 2>), dim3(g2, d.CDw / BN, 1), dim3(512), lds2, st, d, t, nt2);
        xmc_note_kernel("ptile3_kernel<%d, 32, 16, 1, 2>", BN);
        XMC_LAUNCH_CHECK();
        return 0;
    }
    if (maxpatch > 384) return XMC_ESHAPE;                            // staging registers: PIT * 256 / cps pixels
    const int pstride = t.slab * 2 + 32;
    const size_t pb = (size_t)((maxpatch * pstride + 15) & ~15);
    const size_t lds = pb + (size_t)d.ntaps * BN * pstride;
    if (lds > XMC_MAX_DYN_LDS) return XMC_ESHAPE;
    const int ntiles = d.N * t.tiles_y * t.tiles_x;
    const int per_cu = (lds <= 80 * 1024 && BN == 32 && t.slab == 32) ? 2 : 1;    // 8-wave workgroups; 2 fit when <= 128 VGPRs
    int gx = 256 * per_cu / (int)((d.CDw / BN) * d.nclass);
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    dim3 grid((unsigned)gx, (unsigned)(d.CDw / BN), (unsigned)d.nclass);
    if (t.slab == 128) {                                                // one class per blockIdx.z, its four 128-deep slices resident
        if (BN != 64 || d.ntaps != 4 || d.sign_bits || d.dot) return XMC_ESHAPE;
        const int epis = xmc_debug_off("no_ptile_epi") ? -1 : (xmc_epi_mask(d) & ~kEpiBias);
#define XMC_PT3W(E)                                                                                                         \
        if (epis == (E)) {                                                                                                  \
            XMC_ALLOW_BIG_LDS((ptile3_kernel<64, 128, 4, 1, 1, false, (E)>));                                               \
            hipLaunchKernelGGL((ptile3_kernel<64, 128, 4, 1, 1, false, (E)>), grid, dim3(512), lds, st, d, t, ntiles);      \
        } else
        XMC_PT3W(kEpiRes) XMC_PT3W(0)
#undef XMC_PT3W
        {
            xmc_note_generic_epi("ptile3<128>", epis);
            XMC_ALLOW_BIG_LDS((ptile3_kernel<64, 128, 4, 1>));
            hipLaunchKernelGGL((ptile3_kernel<64, 128, 4, 1>), grid, dim3(512), lds, st, d, t, ntiles);
        }
        xmc_note_kernel("ptile3_kernel<64, 128, 4, 1>");
        XMC_LAUNCH_CHECK();
        return 0;
    }
    // all 4 output-parity classes from one staged patch when their 16 weight slices fit beside it
    if (d.nclass == 4 && d.ntaps == 4 && t.PHu * t.PWu <= 384) {
        const size_t ldsm = (size_t)((t.PHu * t.PWu * pstride + 15) & ~15) + (size_t)16 * BN * pstride;
        static const bool no_merge = xmc_debug_off("no_class_merge");
        if (ldsm <= XMC_MAX_DYN_LDS && !no_merge) {
            int gm = 256 / (int)(d.CDw / BN);
            if (gm > ntiles) gm = ntiles;
            dim3 gridm((unsigned)gm, (unsigned)(d.CDw / BN), 1);
            if (d.sign_bits || d.dot) return XMC_ESHAPE;
            static const bool no_epim = xmc_debug_off("no_ptile_epi");
            const int epim = no_epim ? -1 : (xmc_epi_mask(d) & ~kEpiBias);
#define XMC_PT3M(SL, E)                                                                                                     \
            if (t.slab == (SL) && epim == (E)) {                                                                            \
                XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, SL, 4, 4, 1, false, (E)>));                                            \
                hipLaunchKernelGGL((ptile3_kernel<BN, SL, 4, 4, 1, false, (E)>), gridm, dim3(512), ldsm, st, d, t, ntiles); \
            } else
            XMC_PT3M(64, kEpiRes) XMC_PT3M(64, 0) XMC_PT3M(32, kEpiRes) XMC_PT3M(32, 0)
#undef XMC_PT3M
            if (t.slab == 64) {
                xmc_note_generic_epi("ptile3<4,4>", epim);
                XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 64, 4, 4>));
                hipLaunchKernelGGL((ptile3_kernel<BN, 64, 4, 4>), gridm, dim3(512), ldsm, st, d, t, ntiles);
            } else {
                XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 32, 4, 4>));
                hipLaunchKernelGGL((ptile3_kernel<BN, 32, 4, 4>), gridm, dim3(512), ldsm, st, d, t, ntiles);
            }
            xmc_note_kernel("ptile3_kernel<%d, %d, 4, 4>", BN, t.slab);
            XMC_LAUNCH_CHECK();
            return 0;
        }
    }
    static const bool no_m32 = xmc_debug_off("no_ptile_m32");
    static const bool m32_all = xmc_debug_off("ptile_m32_all");
    if (!no_m32 && d.ntaps == 9 && d.out_dtype == XMC_BF16 && t.slab == 64 &&
        (m32_all || d.res || d.dst2 || d.dst_pool || d.mask || d.sign_bits)) {      // one matrix wave per SIMD: 32x32x16 form
        // the epilogue option sets of the training step as compile-time instantiations (kEpi*), anything else through the
        // descriptor-reading one
        int epi = xmc_epi_mask(d);
        static const bool no_epi = xmc_debug_off("no_ptile_epi");
        if (no_epi) epi = -1;
        if (d.sc_img) {                           // the residual recomputed from the image (xmc_conv_ptile_scimg): two block-end sets, 8 x 32 tiles
            if (BN != 64 || d.res || t.TW != 32 || d.CD != 64 || lds + 18 * 66 * 8 > XMC_MAX_DYN_LDS) return XMC_ESHAPE;
#define XMC_PT3_SCI(E)                                                                                                   \
            if (epi == (E)) {                                                                                            \
                XMC_ALLOW_BIG_LDS((ptile3_kernel<64, 64, 9, 1, 1, true, (E)>));                                          \
                hipLaunchKernelGGL((ptile3_kernel<64, 64, 9, 1, 1, true, (E)>), grid, dim3(512), lds + 18 * 66 * 8, st, d, t, ntiles); \
                xmc_note_kernel("ptile3_kernel<64, 64, 9, 1, 1, true, sc>");                                             \
                XMC_LAUNCH_CHECK();                                                                                      \
                return 0;                                                                                                \
            }
            XMC_PT3_SCI((kEpiDKeepS & ~kEpiRes) | kEpiScImg) XMC_PT3_SCI((kEpiDFwd & ~kEpiRes) | kEpiScImg)
#undef XMC_PT3_SCI
            return XMC_ESHAPE;
        }
#define XMC_PT3_EPI(E)                                                                                                   \
        if (epi == (E)) {                                                                                                \
            XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 64, 9, 1, 1, true, (E)>));                                              \
            hipLaunchKernelGGL((ptile3_kernel<BN, 64, 9, 1, 1, true, (E)>), grid, dim3(512), lds, st, d, t, ntiles);     \
            xmc_note_kernel("ptile3_kernel<%d, 64, 9, 1, 1, true>", BN);                                                 \
            XMC_LAUNCH_CHECK();                                                                                          \
            return 0;                                                                                                    \
        }
        XMC_PT3_EPI(kEpiGSum) XMC_PT3_EPI(kEpiDKeep) XMC_PT3_EPI(kEpiDFwd) XMC_PT3_EPI(kEpiDLast) XMC_PT3_EPI(kEpiDLin)
        XMC_PT3_EPI(kEpiDKeepS) XMC_PT3_EPI(kEpiDLastS) XMC_PT3_EPI(kEpiDgDot)
        XMC_PT3_EPI(kEpiMask)                                                                // data gradient through a LeakyReLU
#undef XMC_PT3_EPI
        if (d.sign_bits || d.dot) return XMC_ESHAPE;
        xmc_note_generic_epi("ptile3<M32>", epi);
        XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, 64, 9, 1, 1, true>));
        hipLaunchKernelGGL((ptile3_kernel<BN, 64, 9, 1, 1, true>), grid, dim3(512), lds, st, d, t, ntiles);
        xmc_note_kernel("ptile3_kernel<%d, 64, 9, 1, 1, true>", BN);
        XMC_LAUNCH_CHECK();
        return 0;
    }
#define XMC_PT3(SL, NTP)                                                                                                      \
    do {                                                                                                                      \
        XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, SL, NTP, 1>));                                                                   \
        hipLaunchKernelGGL((ptile3_kernel<BN, SL, NTP, 1>), grid, dim3(512), lds, st, d, t, ntiles);                          \
        xmc_note_kernel("ptile3_kernel<%d, %d, %d, 1>", BN, SL, NTP);                                                         \
    } while (0)
    static const bool no_epi9 = xmc_debug_off("no_ptile_epi");
    const int epi9 = (no_epi9 || d.ntaps != 9) ? -1 : (xmc_epi_mask(d) & ~kEpiBias);
#define XMC_PT3E(SL, E)                                                                                                       \
    if (t.slab == (SL) && epi9 == (E)) {                                                                                      \
        XMC_ALLOW_BIG_LDS((ptile3_kernel<BN, SL, 9, 1, 1, false, (E)>));                                                      \
        hipLaunchKernelGGL((ptile3_kernel<BN, SL, 9, 1, 1, false, (E)>), grid, dim3(512), lds, st, d, t, ntiles);             \
        xmc_note_kernel("ptile3_kernel<%d, %d, %d, 1>", BN, SL, 9);                                                           \
        XMC_LAUNCH_CHECK();                                                                                                   \
        return 0;                                                                                                             \
    }
    // plain / bias only (the data gradients, the 64 -> 64 forward without a block end) and the generator's last block: c2 + block sum
    // + the tail's LeakyReLU
    XMC_PT3E(64, 0) XMC_PT3E(32, 0) XMC_PT3E(32, (kEpiGSum | kEpiPost) & ~kEpiBias) XMC_PT3E(64, (kEpiGSum | kEpiPost) & ~kEpiBias)
#undef XMC_PT3E
    if (d.sign_bits || d.dot) return XMC_ESHAPE;
    xmc_note_generic_epi("ptile3<16x16>", epi9);
    if (t.slab == 64) {
        if (d.ntaps == 9) XMC_PT3(64, 9); else if (d.ntaps == 4) XMC_PT3(64, 4); else XMC_PT3(64, 0);
    } else {
        if (d.ntaps == 9) XMC_PT3(32, 9); else if (d.ntaps == 4) XMC_PT3(32, 4); else XMC_PT3(32, 0);
    }
#undef XMC_PT3
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
    if (lds > XMC_MAX_DYN_LDS) return XMC_ESHAPE;
    XMC_ALLOW_BIG_LDS((tile_kernel<BN, WM, WN>));
    dim3 grid((unsigned)(d.N * t.tiles_y * t.tiles_x), (unsigned)(d.CDw / BN), (unsigned)d.nclass);
    hipLaunchKernelGGL((tile_kernel<BN, WM, WN>), grid, dim3(256), lds, st, d, t);
    xmc_note_kernel("tile_kernel<%d, %d, %d>", BN, WM, WN);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Returns 1 if the descriptor is eligible for the halo-tile kernel (and fills cfg), 0 otherwise.
static int tile_plan(const XmcConvDesc* d, TileCfg* t) {
    if (d->dtype != XMC_BF16 || d->src_shift != 0) return 0;
    static const bool no_s2 = xmc_debug_off("no_ptile_s2");
    const bool s2 = d->SA == 2;                       // stride-2 forward: weights-resident persistent kernel only
    if (d->SA != 1 && !(s2 && !no_s2 && d->nclass == 1 && d->ntaps == 16 && d->CS == 32 && d->CDw <= 64 && d->DA == 1)) return 0;
    if (d->CS % 32 != 0 || d->MW % 16 != 0) return 0;
    if (d->ntaps < 2) return 0;                       // 1x1: nothing to reuse, the gather kernel streams it
    if (d->CDw > 64 && d->CS > 64) return 0;          // wide layers: the streamed-weights kernel (conv_wtile.hip) or the gather kernel
    int TW = (d->MW >= 32 && !s2) ? 32 : 16;
    int TH = s2 ? 8 : 256 / TW;
    if (d->MH % TH != 0 || d->MW % TW != 0) return 0;
    t->TH = TH; t->TW = TW; t->log2TW = TW == 32 ? 5 : 4;
    t->tiles_y = d->MH / TH; t->tiles_x = d->MW / TW;
    t->slab = (d->CS % 64 == 0) ? 64 : 32;
    static const bool no_xcd = xmc_debug_off("no_xcd_map");
    t->no_xcd = no_xcd ? 1 : 0;
    // 128 -> 64 channels, 2x2 tap classes (the data gradient of a block's 4x4 stride-2 convolution, df_gan.py:272-273): ALL 128 source
    // channels of a class's four weight slices stay in LDS (4 x 64 x 288 B beside a 9 x 33 pixel patch: 159 KB), so nothing is streamed
    // per tile but the patch -- the streamed-weights kernel moved 140 KB per class tile through the vector-memory path for 128 MFMAs a wave
    static const bool no_s128 = xmc_debug_off("no_ptile_slab128");
    if (!no_s128 && d->CS == 128 && d->SA == 1 && d->ntaps == 4 && d->nclass == 4 && d->CDw == 64 && d->out_dtype == XMC_BF16) t->slab = 128;
    for (int z = 0; z < d->nclass; ++z) {
        int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
        for (int k = 0; k < d->ntaps; ++k) {
            int h = d->dh[z][k], w = d->dw[z][k];
            hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax;
            wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax;
        }
        t->dh0[z] = hmin; t->dw0[z] = wmin;
        t->PH[z] = d->SA * (TH - 1) + (hmax - hmin + 1); t->PW[z] = d->SA * (TW - 1) + (wmax - wmin + 1);
        if (s2 ? (t->PH[z] > 18 || t->PW[z] > 34 || (t->PW[z] & 1)) : (t->PH[z] * t->PW[z] > (t->slab == 128 ? 384 : 12 * (256 / (t->slab / 8))))) return 0;   // staging registers (PIT)
    }
    {
        int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
        for (int z = 0; z < d->nclass; ++z) {
            hmin = t->dh0[z] < hmin ? t->dh0[z] : hmin; wmin = t->dw0[z] < wmin ? t->dw0[z] : wmin;
            const int he = t->dh0[z] + t->PH[z] - TH, we = t->dw0[z] + t->PW[z] - TW;
            hmax = he > hmax ? he : hmax; wmax = we > wmax ? we : wmax;
        }
        t->dh0u = hmin; t->dw0u = wmin; t->PHu = TH + (hmax - hmin); t->PWu = TW + (wmax - wmin);
    }
    return 1;
}

// the weights-resident persistent kernel writes the pooled third output (XmcConvDesc.dst_pool) from its epilogue for these cases
int xmc_conv_ptile_pool_try(const XmcConvDesc* d, void* stream) {
    TileCfg t;
    if (!d->dst_pool || d->out_dtype != XMC_BF16 || d->SA != 1 || d->DA != 1 || d->nclass != 1 || (d->DH & 1) || (d->DW & 1)) return 1;
    if (!tile_plan(d, &t)) return 1;
    if (!(d->CS <= 64 && t.slab == d->CS && d->CDw <= 64) || xmc_debug_off("no_ptile")) return 1;
    const int rc = d->CDw == 64 ? launch_ptile<64>(*d, t, reinterpret_cast<hipStream_t>(stream))
                                : launch_ptile<32>(*d, t, reinterpret_cast<hipStream_t>(stream));
    return rc == XMC_ESHAPE ? 1 : rc;
}

// block end whose residual is recomputed from the image (XmcConvDesc.sc_img): the 32x32x16 role of the persistent kernel only
extern "C" int xmc_conv_ptile_scimg(const XmcConvDesc* d, void* stream) {
    if (!d || !d->src || !d->wpk || !d->dst || !d->sc_img || !d->sc_frag || !d->sc_bias) return XMC_EINVAL;
    static const bool off = xmc_debug_off("no_scimg");
    if (off || d->res || d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16 || d->SA != 1 || d->DA != 1 || d->nclass != 1 || d->ntaps != 9 ||
        d->src_shift != 0 || d->CS != 64 || d->CDw != 64 || d->CD != 64 || d->MW % 32 != 0 || d->MH % 8 != 0 || d->SH != d->MH || d->SW != d->MW ||
        d->DH != d->MH || d->DW != d->MW || xmc_debug_off("no_ptile") || xmc_debug_off("no_ptile_m32") || xmc_debug_off("no_ptile_epi"))
        return 1;
    for (int k = 0; k < 9; ++k)
        if (d->dh[0][k] < -1 || d->dh[0][k] > 1 || d->dw[0][k] < -1 || d->dw[0][k] > 1) return 1;
    TileCfg t;
    if (!tile_plan(d, &t) || t.slab != 64) return 1;
    const int rc = launch_ptile<64>(*d, t, reinterpret_cast<hipStream_t>(stream));
    return rc == XMC_ESHAPE ? 1 : rc;
}

// data gradient whose source is masked by sign bytes while it is staged (XmcConvDesc.mask_bits): the persistent kernel only
extern "C" int xmc_conv_ptile_bits(const XmcConvDesc* d, void* stream) {
    if (!d || !d->src || !d->wpk || !d->dst || !d->mask_bits) return XMC_EINVAL;
    static const bool off = xmc_debug_off("no_stage_bits");
    if (off || d->dtype != XMC_BF16 || d->SA != 1 || d->DA != 1 || d->nclass != 1 || d->src_shift != 0 || d->dst_pool) return 1;
    TileCfg t;
    if (!tile_plan(d, &t)) return 1;
    if (!(d->CS <= 64 && t.slab == d->CS && d->CDw <= 64) || xmc_debug_off("no_ptile")) return 1;
    const int rc = d->CDw == 64 ? launch_ptile<64>(*d, t, reinterpret_cast<hipStream_t>(stream))
                                : launch_ptile<32>(*d, t, reinterpret_cast<hipStream_t>(stream));
    return rc == XMC_ESHAPE ? 1 : rc;
}

// the 128 -> 64 channel class data gradient with its weights resident (tile_plan: slab 128), asked BEFORE the streamed-weights kernels
int xmc_conv_ptile_slab128_try(const XmcConvDesc* d, void* stream) {
    if (d->CS != 128 || d->CDw != 64 || d->ntaps != 4 || d->nclass != 4 || d->dst_pool || xmc_debug_off("no_ptile")) return 1;
    TileCfg t;
    if (!tile_plan(d, &t) || t.slab != 128) return 1;
    const int rc = launch_ptile<64>(*d, t, reinterpret_cast<hipStream_t>(stream));
    return rc == XMC_ESHAPE ? 1 : rc;
}

// entry used by xmc_conv_igemm's dispatcher (conv_igemm.hip)
int xmc_conv_tile_try(const XmcConvDesc* d, void* stream) {
    TileCfg t;
    if (!tile_plan(d, &t)) return 1;   // not eligible
    if (t.slab == 128) return 1;       // (that case was offered to xmc_conv_ptile_slab128_try and declined)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    static const bool no_pt = xmc_debug_off("no_ptile");
    if (!no_pt && d->CS <= 64 && t.slab == d->CS && d->CDw <= 64) {          // persistent, weights resident
        rc = d->CDw == 64 ? launch_ptile<64>(*d, t, st) : launch_ptile<32>(*d, t, st);
        if (rc != XMC_ESHAPE) return rc;
    }
    // 32 -> 128 channels, 3x3 (the data gradient of the attention blocks' 128 -> 32 output convolutions, df_concept_gan.py:146-160): two
    // 64-channel halves over blockIdx.y, each with its half of the weights resident (the non-persistent tile kernel ran it at 227 TF/s)
    static const bool no_pt128 = xmc_debug_off("no_ptile_cd128");
    if (!no_pt && !no_pt128 && d->CS == 32 && t.slab == 32 && d->CDw == 128 && d->SA == 1 && d->nclass == 1 && d->ntaps == 9) {
        rc = launch_ptile<64>(*d, t, st);
        if (rc != XMC_ESHAPE) return rc;
    }
    if (d->SA != 1) return 1;                         // stride 2 exists only in the persistent kernel
    if (d->CDw % 128 == 0) rc = launch_tile<128, 2, 2>(*d, t, st);
    else if (d->CDw % 64 == 0) rc = launch_tile<64, 4, 1>(*d, t, st);
    else rc = launch_tile<32, 4, 1>(*d, t, st);
    return rc == XMC_ESHAPE ? 1 : rc;
}
