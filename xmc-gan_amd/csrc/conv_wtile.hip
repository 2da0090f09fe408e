// Streamed-weights halo-patch convolution (forward / data gradient) for gfx950, bf16: the wide layers of D and G.
//
// One 8-wave workgroup per CU owns 256 output pixels (TH x TW of one image) x BN = 16*TN output channels and walks its tiles
// persistently.  K is cut into slabs of 64 source channels; per slab the input patch (tile + halo) is staged in LDS ONCE and
// every tap of the slab reads its pixel fragments from it at a shifted offset, so only the weights of each (slab, tap) STAGE
// (BN x 64 bf16 = 16 KB at BN = 128) are streamed: ~21 KB through the CU's vector-memory path per 128 MFMAs of a wave instead
// of the 48 KB an im2col gather needs (that path, ~22 B/clk per CU, is what bounds the gather kernel).
//
//   waves 4-7 ("stage")  : weights of stage g+4 global -> registers (two register sets, so a load has TWO stages to arrive),
//                          stage g+2 registers -> LDS ring (3 slots); the patch of slab q+2 global -> registers during slab q,
//                          written into the OTHER of two patch buffers at the start of slab q+1 -- the slab change costs the
//                          compute waves nothing (the single-buffered predecessor stalled two barriers per slab, which is
//                          what kept its 2x2-tap forms at the gather kernel's speed);
//   waves 0-3 ("compute"): 64 pixels x BN channels each; per 32-deep K sub-step 4 + TN fragment reads and 4*TN MFMAs
//                          (v_mfma_f32_16x16x32_bf16, A = weight rows, B = pixels), the reads of the next sub-step dealt out
//                          between the MFMAs of this one with the order pinned; the first fragments of the next stage are
//                          fetched across the barrier; epilogue from registers (a lane owns 8 consecutive channels of a pixel).
// One barrier per stage.  All staging is branch-free (past the end of the stream it re-loads valid addresses and writes slots
// nobody reads), so the compiler's s_waitcnt counts stay exact and nothing drains the prefetch.
//
// MODE 0: unit source stride, tap sets of 2x2 (the four output-parity classes of a stride-2 data gradient or of the fused
//         upsample + 3x3 convolution, one class per blockIdx.z) or 3x3.
// MODE 1: 4x4 taps at source stride 2 (resD conv_r[0], df_gan.py:272-273; also the data gradient of the fused upsample conv).
//         out(a,b) reads rows 2a-1..2a+2 = the 2x2 blocks (a,b)..(a+1,b+1) of the space-to-depth view shifted by one pixel, so
//         the layer is a DENSE 2x2-tap unit-stride convolution over 4*CS channels.  That view is never written: slab
//         (dy, dx, 64-channel block) is gathered from the NHWC source at stride 2 while it is staged ((TH+1) x (TW+1) pixels
//         of 128 contiguous bytes each), and stage (slab, (ta,tb)) uses the weight slice of tap (2ta+dy, 2tb+dx).
//
// LDS: weight ring 3 x BN x 128 B, rows XOR-swizzled by (row & 7) on the 16-byte chunk (conflict-free ds_read_b128 for the
// lane -> (row l&15, chunk l>>4) fragment pattern); two patch buffers of <= 340 pixels x 160 B (128 B of data + 32 B pad: the
// stride that makes the same fragment pattern conflict-free for ANY patch alignment).  3*16 + 2*53.1 KB = 154 KB.
//
// Takes over: F.conv2d at df_gan.py:187-188 (G_Block c1/c2), 273,276 (resD conv_r) for Cin % 64 == 0 and the matching halves
// of errD.backward() / errG.backward() (train_gan.py:228,288).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef WT_ABL
#define WT_ABL 0     // ablation builds for timing experiments only (results are garbage): 1 no global loads, 2 no LDS stores in the
#endif               // staging role, 4 no fragment reads in the compute role, 8 no stage barriers, 16 no epilogue stores

namespace {

struct WtCfg {
    int TH, TW, log2TW;
    int tiles_y, tiles_x;
    int PH[XMC_MAX_CLASSES], PW[XMC_MAX_CLASSES];        // patch size per class
    int dh0[XMC_MAX_CLASSES], dw0[XMC_MAX_CLASSES];      // min tap offsets per class (MODE 0)
    int nslab;                                           // K slabs per tile: CS/64 (MODE 0), 4*CS/64 (MODE 1)
    int patch_bytes;                                     // one patch buffer (max over classes), multiple of 16
    int8_t tsel[4][4];                                   // MODE 1: tap index of (dy*2+dx, ta*2+tb)
};

constexpr int kPStride = 160;       // bytes between patch pixels
constexpr int kPIT = 12;            // patch pixels per staging thread (32 pixel rows of 8 units per pass): <= 384 pixels

// CW = number of compute waves: 4 (one per SIMD, 64 pixels x BN channels each) or 8 (two per SIMD, 64 pixels x BN/2 channels each:
// the two waves of a SIMD interleave their MFMAs, so one's LDS latency and barrier skew are covered by the other's matrix work)
template <int NTAPS, int MODE, int TN, int CW, int EPI = -1>
__global__ __launch_bounds__(64 * CW + 256) void wtile2_kernel(const XmcConvDesc d, const WtCfg t, int ntiles) {
    constexpr int BN = 16 * TN, TM = 4, NS = 64 * CW;
    constexpr int TNW = TN * 4 / CW;             // 16-channel blocks per compute wave
    static_assert(CW == 4 || CW == 8, "4 or 8 compute waves");
    static_assert(TNW % 2 == 0, "a lane's 8-channel unit needs a pair of blocks");
    constexpr int WL = BN / 32;                  // weight rows per staging thread and stage
    constexpr int WSTG = BN * 128;               // bytes of one weight stage in the ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_toff[XMC_MAX_TAPS];         // tap -> patch byte offset
    __shared__ int s_wbase[XMC_MAX_TAPS];        // (slab group, tap) -> first unit of the weight slice
    __shared__ int s_slab[64];                   // slab -> (source shift in units) | choff; see below

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * BN;
    const int cls = blockIdx.z;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PW = t.PW[cls];
    const int dh0 = MODE == 0 ? t.dh0[cls] : 0, dw0 = MODE == 0 ? t.dw0[cls] : 0;
    const int cs_units = d.CS / 8;
    const int nslab = t.nslab;
    const int cb = d.CS / 64;
    if (tid < XMC_MAX_TAPS) {
        if (MODE == 0) {
            const int tt = tid < NTAPS ? tid : 0;
            s_toff[tid] = ((d.dh[cls][tt] - dh0) * PW + (d.dw[cls][tt] - dw0)) * kPStride;
            s_wbase[tid] = d.wi[cls][tt] * d.CDw * cs_units;
        } else {
            const int g4 = tid >> 2, tp = tid & 3;                     // tid = (dy*2+dx)*4 + (ta*2+tb)
            s_toff[tid] = ((tp >> 1) * PW + (tp & 1)) * kPStride;       // same for every slab group
            s_wbase[tid] = d.wi[0][t.tsel[g4][tp]] * d.CDw * cs_units;
        }
    }
    if (tid >= 64 && tid < 128) {
        // slab sl -> { low 16 bits: index of the slab group's first entry in s_wbase; high bits: unused } and choff; packed as
        // two ints would need two tables: keep (group << 8 | channel block) and derive the rest where it is used
        const int sl = tid - 64;
        s_slab[sl] = MODE == 0 ? sl : (((sl / cb) << 8) | (sl % cb));
    }
    // Persistent: this workgroup's tiles are blockIdx.x, + gridDim.x, ...  The K stages of all of them form ONE stream
    // (patch sequence q = (tile, slab), NTAPS stages each).
    const XcdWalk xw = xmc_xcd_walk(ntiles);      // XCD-aware tile walk (common.h): this workgroup's tiles are xw.first, + xw.step, ... < xw.end
    const int mytiles = xw.first < xw.end ? (xw.end - xw.first + xw.step - 1) / xw.step : 0;
    const int Q = mytiles * nslab;
    unsigned char* const wring = smem;                           // [3][BN][128]
    unsigned char* const patch0 = smem + 3 * WSTG;               // [2][patch_bytes]
    const int pbytes = t.patch_bytes;
    __syncthreads();
    if (mytiles <= 0) return;

    if (wave >= CW) {
        // ================================================================================================ staging role
        const int rt = tid - NS;
        const int unit = rt & 7, r32 = rt >> 3;   // weights: 8 units per row, rows r32 + 32*i; patch: pixels r32 + 32*it
        const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
        const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);
        const int PH = t.PH[cls];
        int psrc[kPIT];
        unsigned halo[kPIT], inpatch = 0;
        {
            int py = r32 / PW, px = r32 - (r32 / PW) * PW;       // pixel r32 + 32*it, stepped without a division per unit
#pragma unroll
            for (int it = 0; it < kPIT; ++it) {
                const int pp = r32 + it * 32;
                const bool in = pp < PH * PW;
                inpatch |= in ? (1u << it) : 0u;
                if (MODE == 0) {
                    psrc[it] = in ? ((dh0 + py) * d.SW + (dw0 + px)) * cs_units + unit : 0;
                    halo[it] = !in ? 0u : ((py < -dh0 ? 1u : 0u) | (py >= t.TH - dh0 ? 2u : 0u) | (px < -dw0 ? 4u : 0u) | (px >= t.TW - dw0 ? 8u : 0u));
                } else {
                    psrc[it] = in ? ((2 * py - 1) * d.SW + (2 * px - 1)) * cs_units + unit : 0;
                    halo[it] = !in ? 0u : ((py == 0 ? 1u : 0u) | (py == t.TH ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == t.TW ? 8u : 0u));
                }
                px += 32;
                if (px >= PW) { px -= PW; ++py; }     // PW >= 16: at most two wraps
                if (px >= PW) { px -= PW; ++py; }
            }
        }
        int wrow[WL], wdst[WL];                   // weight rows of this thread: source row offset (units), LDS byte offset
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            // physical ring row (n-block j, row r) <- logical channel (j/2)*32 + (r/4)*8 + (j%2)*4 + r%4: lane group fc = r/4 then
            // holds, for unit u = j/2, channels u*32 + fc*8 .. +7, so the four lane groups of a pixel store 64 contiguous bytes
            const int prow = r32 + 32 * i;
            const int j = prow >> 4, r = prow & 15;
            const int lrow = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
            wrow[i] = (n0 + lrow) * cs_units + unit;
            wdst[i] = prow * 128 + ((unit ^ (prow & 7)) << 4);
        }
        u32x4 pv[kPIT] = {}, wA[WL] = {}, wB[WL] = {};
        if (WT_ABL & 1) {     // no loads: stage random-looking bf16 values (all-zero operands would let the chip clock higher)
            unsigned h = (unsigned)tid * 2654435761u;
            auto junk = [&]() { u32x4 v; for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; v[k] = (h & 0x807f807fu) | 0x3f003f00u; } return v; };
            for (int it = 0; it < kPIT; ++it) pv[it] = junk();
            for (int i = 0; i < WL; ++i) { wA[i] = junk(); wB[i] = junk(); }
        }
        unsigned okmask = 0;                      // of the patch held in pv
        auto issue_patch = [&](int q) {           // q is clamped by the caller: always a patch of this workgroup
            const int tk = q / nslab, sl = q - tk * nslab;
            const int tile = xw.first + tk * xw.step;
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
            int base;
            unsigned border;
            if (MODE == 0) {
                base = ((img * d.SH + a0) * d.SW + b0) * cs_units + sl * 8;
                border = (a0 + dh0 < 0 ? 1u : 0u) | (a0 + PH + dh0 > d.SH ? 2u : 0u) | (b0 + dw0 < 0 ? 4u : 0u) | (b0 + PW + dw0 > d.SW ? 8u : 0u);
            } else {
                const int si = s_slab[sl];
                const int grp = si >> 8, cbi = si & 0xff, dy = grp >> 1, dx = grp & 1;
                base = ((img * d.SH + 2 * a0 + dy) * d.SW + 2 * b0 + dx) * cs_units + cbi * 8;
                border = ((a0 == 0 && dy == 0) ? 1u : 0u) | ((a0 + t.TH == d.MH && dy == 1) ? 2u : 0u) |
                         ((b0 == 0 && dx == 0) ? 4u : 0u) | ((b0 + t.TW == d.MW && dx == 1) ? 8u : 0u);
            }
            okmask = 0;
#pragma unroll
            for (int it = 0; it < kPIT; ++it) {
                const bool ok = (halo[it] & border) == 0 && ((inpatch >> it) & 1);
                okmask |= ok ? (1u << it) : 0u;
                if (!(WT_ABL & 1)) pv[it] = src16[(unsigned)(base + (ok ? psrc[it] : 0))];
            }
        };
        auto commit_patch = [&](unsigned char* patch) {
#pragma unroll
            for (int it = 0; it < kPIT; ++it) {
                u32x4 v = pv[it];
                if (!((okmask >> it) & 1)) v = u32x4{0, 0, 0, 0};
                if (((inpatch >> it) & 1) && !(WT_ABL & 2)) *reinterpret_cast<u32x4*>(patch + (r32 + it * 32) * kPStride + unit * 16) = v;
            }
        };
        // weights of the stage (slab sl, tap): the same for every tile
        auto issue_w = [&](int sl, int tap, u32x4 (&wv)[WL]) {
            int wb;
            if (MODE == 0) {
                wb = s_wbase[tap] + sl * 8;
            } else {
                const int si = s_slab[sl];
                wb = s_wbase[(si >> 8) * 4 + tap] + (si & 0xff) * 8;
            }
#pragma unroll
            for (int i = 0; i < WL; ++i) if (!(WT_ABL & 1)) wv[i] = w16[(unsigned)(wb + wrow[i])];
        };
        auto commit_w = [&](int slot, const u32x4 (&wv)[WL]) {
            unsigned char* wb = wring + slot * WSTG;
#pragma unroll
            for (int i = 0; i < WL; ++i) if (!(WT_ABL & 2)) *reinterpret_cast<u32x4*>(wb + wdst[i]) = wv[i];
        };
        // ---- prologue: everything the first two stages need, all loads in flight together
        {
            u32x4 w2[WL] = {}, w3[WL] = {};
            issue_patch(0);
            issue_w(0, 0, wA);
            issue_w(0, 1, wB);
            issue_w(0, 2, w2);
            issue_w(0, 3, w3);
            commit_patch(patch0);
            commit_w(0, wA);
            commit_w(1, wB);
#pragma unroll
            for (int i = 0; i < WL; ++i) { wA[i] = w2[i]; wB[i] = w3[i]; }
            issue_patch(Q > 1 ? 1 : 0);
        }
        __syncthreads();                          // patch 0 and weight stages 0, 1 are in LDS; wA / wB hold stages 2 / 3
        // ---- the stream.  Stage g = (q, tap): commit the weights of g+2 (loaded two stages ago), load those of g+4 into the
        // same registers; at the first stage of a slab write the patch of slab q+1 (loaded during slab q-1) into the buffer the
        // compute waves left at the previous barrier and start loading the patch of slab q+2.
        int sl = 0, slot2 = 2;                    // slab of stage g within its tile; ring slot of stage g+2
        auto slab = [&](int q, auto par) {
            constexpr int PAR = decltype(par)::value;
            const int sl1 = sl + 1 == nslab ? 0 : sl + 1;
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                const bool useA = ((PAR + tap) & 1) == 0;
                const int t4 = tap + 4 >= NTAPS ? tap + 4 - NTAPS : tap + 4;        // NTAPS >= 4: one wrap at most
                const int s4 = tap + 4 >= NTAPS ? sl1 : sl;
                if (useA) { commit_w(slot2, wA); issue_w(s4, t4, wA); }
                else      { commit_w(slot2, wB); issue_w(s4, t4, wB); }
                slot2 = slot2 == 2 ? 0 : slot2 + 1;
                if (tap == 0) {
                    commit_patch(patch0 + ((q + 1) & 1) * pbytes);
                    issue_patch(q + 2 < Q ? q + 2 : Q - 1);
                }
                if (!(WT_ABL & 8)) __syncthreads();                  // end of stage g
            }
            sl = sl1;
        };
        for (int q = 0;;) {
            slab(q, std::integral_constant<int, 0>{});
            if (++q >= Q) break;
            slab(q, std::integral_constant<int, NTAPS & 1>{});
            if (++q >= Q) break;
        }
    } else {
        // ================================================================================================ compute role
        const int wm = wave & 3, wn = wave >> 2;  // pixel group (64 pixels), channel half (CW == 8)
        const int fr = lane & 15, fc = lane >> 4;
        const int cd8 = d.CD / 8;
        int abyte[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * 64 + i * 16 + fr;
            const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
            abyte[i] = (ty * PW + tx) * kPStride + fc * 16;
        }
        // weight fragment (n-block mj, row fr, chunk sub*4+fc) of a ring slot: chunk index XOR (row & 7)
        const int bb0 = fr * 128 + ((fc ^ (fr & 7)) << 4) + wn * TNW * 2048, bb1 = bb0 ^ 64;
        f32x4 acc[TM][TNW];
        float dacc = 0.f;                        // running sum for XmcConvDesc.dot over this workgroup's tiles
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Fragment registers: pixel fragments double buffered (every MFMA column of a sub-step uses all four), weight fragments
        // in ONE set: the MFMAs run column by column (all four pixel blocks against weight fragment mj), and as soon as the four
        // MFMAs of column mj have issued, slot mj is re-loaded with the NEXT sub-step's fragment.
        u32x4 P[2][TM], Wf[TNW];
        auto rdp = [&](const unsigned char* pa, int sub, int mi) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(pa + sub * 64 + abyte[mi]);
        };
        auto rdw = [&](const unsigned char* wb, int sub, int mj) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(wb + (sub ? bb1 : bb0) + mj * 2048);
        };
        const int ch0 = n0 + wn * (BN / 2) * (CW / 4 - 1) + fc * 8;      // first channel of this lane's unit 0
        const int nw8 = (n0 + wn * (BN / 2) * (CW / 4 - 1)) >> 3;         // ... in 8-channel units
        // epilogue from registers: acc[i][j][r] = pixel (m-block i, fr), channel n0 + (j/2)*32 + fc*8 + (j%2)*4 + r; clears acc.
        // bf16 destination only (plan()); order: bias, activation, alpha, LeakyReLU' mask, (row-indexed, scaled) residual.
        auto epilogue = [&](int tile) {
            // Everything the epilogue needs besides the accumulators is recomputed here from an opaque lane index: hoisted out
            // of the tile loop as "invariant", the store addresses and bias values spill.
            int lane_op = fr;
            asm volatile("" : "+v"(lane_op) :: "memory");
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
            const int dbase = (((img * d.DH + a0 * d.DA + d.dph[cls]) * d.DW) + b0 * d.DA + d.dpw[cls]) * cd8 + nw8;
            const int rbase = d.res_mode ? ((img * d.MH + a0) * d.MW + b0) * cd8 + nw8 : dbase;
            const int rsy = d.res_mode ? d.MW : d.DA * d.DW, rsx = d.res_mode ? 1 : d.DA;
            constexpr bool RT = EPI < 0;               // epilogue options read from the descriptor (common.h: kEpi*)
            const bool e_bias = RT ? d.bias != nullptr : (EPI & kEpiBias) != 0;
            const bool e_tanh = RT ? d.act == XMC_ACT_TANH : false;
            const bool e_round = RT ? (d.dst2 != nullptr || d.round_act != 0) : (EPI & (kEpiRound | kEpiDst2)) != 0;
            const bool e_dst2 = RT ? d.dst2 != nullptr : (EPI & kEpiDst2) != 0;
            const bool e_alpha = RT ? d.alpha_dev != nullptr : (EPI & kEpiAlpha) != 0;
            const bool e_mask = RT ? d.mask != nullptr : (EPI & kEpiMask) != 0;
            const bool e_res = RT ? d.res != nullptr : (EPI & kEpiRes) != 0;
            const bool e_post = RT ? d.post_act == XMC_ACT_LRELU : (EPI & kEpiPost) != 0;
            const bool e_pool = RT ? d.dst_pool != nullptr : (EPI & kEpiPool) != 0;
            constexpr bool e_sign = !RT && (EPI & kEpiSign) != 0;       // sign bits / dot: compile-time sets only (the launcher declines otherwise)
            constexpr bool e_dot = !RT && (EPI & kEpiDot) != 0;
            const float slope = RT ? (d.act == XMC_ACT_LRELU ? XMC_LRELU : (d.act == XMC_ACT_RELU ? 0.f : 1.f)) : ((EPI & kEpiLrelu) ? XMC_LRELU : 1.f);
            const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
            const float rs = d.res_scale == 0.f ? 1.f : d.res_scale;
            bf16x8* __restrict__ dst8 = reinterpret_cast<bf16x8*>(d.dst);
            const bf16x8* __restrict__ mask8 = reinterpret_cast<const bf16x8*>(d.mask);
            const bf16x8* __restrict__ res8 = reinterpret_cast<const bf16x8*>(d.res);
            bf16x8* __restrict__ dst2_8 = reinterpret_cast<bf16x8*>(d.dst2);
            bf16x8* __restrict__ pool8 = reinterpret_cast<bf16x8*>(d.dst_pool);
            int eo[TM], ro[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * 64 + i * 16 + lane_op;
                const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
                eo[i] = dbase + ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + fc;
                if (d.res_mode == 2)      // residual at half resolution, nearest-upsampled (DA == 1)
                    ro[i] = ((img * (d.DH >> 1) + ((a0 + ty) >> 1)) * (d.DW >> 1) + ((b0 + tx) >> 1)) * cd8 + nw8 + fc;
                else
                    ro[i] = rbase + (ty * rsy + tx * rsx) * cd8 + fc;
            }
#pragma unroll
            for (int u = 0; u < TNW / 2; ++u) {
                if (ch0 + u * 32 >= d.CD) continue;
                float fin[TM][8];
                // every mask / residual vector of this channel unit is requested before the first one is used: one memory latency
                // per unit instead of one per vector (they were 2 x 16 dependent round trips per tile, longer than its MFMAs)
                bf16x8 mkv[TM], rrv[TM];
                if (e_mask) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) mkv[i] = mask8[eo[i] + u * 4];
                }
                if (e_res) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) rrv[i] = res8[ro[i] + u * 4];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * u][r]; v[4 + r] = acc[i][2 * u + 1][r]; }
                    if (e_bias && !(WT_ABL & 32)) {
                        const f32x4 b0v = *reinterpret_cast<const f32x4*>(d.bias + ch0 + u * 32), b1v = *reinterpret_cast<const f32x4*>(d.bias + ch0 + u * 32 + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] += b0v[r]; v[4 + r] += b1v[r]; }
                    }
                    if (e_tanh) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = tanhf(v[r]);
                    } else if (slope != 1.f) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], v[r] * slope);
                    }
                    if (e_sign) {
                        unsigned sb = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) sb |= (v[r] > 0.f ? 1u : 0u) << r;
                        reinterpret_cast<unsigned char*>(d.sign_bits)[eo[i] + u * 4] = (unsigned char)sb;
                    }
                    if (e_round) {
                        bf16x8 o2;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { o2[r] = (xmc_h16)v[r]; v[r] = (float)o2[r]; }
                        if (e_dst2) dst2_8[eo[i] + u * 4] = o2;
                    }
                    if (e_dot && e_mask) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) dacc += v[r] * (float)mkv[i][r];
                    }
                    if (e_alpha) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= alpha;
                    }
                    if (e_mask) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= lrelu_slope((float)mkv[i][r]);
                    }
                    if (e_res) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += rs * (float)rrv[i][r];
                    }
                    if (e_post) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], v[r] * XMC_LRELU);
                    }
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) { o[r] = (xmc_h16)v[r]; fin[i][r] = (float)o[r]; }
                    if (!(WT_ABL & 16)) dst8[eo[i] + u * 4] = o;
                    else asm volatile("" :: "v"(o));
                }
                if (e_pool) {
                    // 2x2 average of the ROUNDED block output (== F.avg_pool2d of dst): the vertical neighbour is pixel block
                    // i+2 (8x32 tiles) or i+1 (16x16 tiles) of the same lane, the horizontal one the same block of lane ^ 1
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int i0 = t.log2TW == 5 ? pr : 2 * pr, i1 = t.log2TW == 5 ? pr + 2 : 2 * pr + 1;
                        const int ml = wm * 64 + i0 * 16 + lane_op;
                        const int ty = ml >> t.log2TW, tx = ml & (t.TW - 1);
                        bf16x8 o;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            float sm = (t.log2TW == 5 ? fin[pr][r] + fin[pr + 2][r] : fin[2 * pr][r] + fin[2 * pr + 1][r]);
                            sm += xmc_xor1(sm);
                            o[r] = (xmc_h16)((d.pool_scale == 0.f ? 0.25f : d.pool_scale) * sm);
                        }
                        (void)i1;
                        if ((lane_op & 1) == 0)
                            pool8[((img * (d.DH >> 1) + ((a0 + ty) >> 1)) * (d.DW >> 1) + ((b0 + tx) >> 1)) * cd8 + nw8 + fc + u * 4] = o;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        int toffr[NTAPS];                         // tap -> patch byte offset, in scalar registers
#pragma unroll
        for (int k = 0; k < NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);
        __syncthreads();                          // patch 0 and weight stages 0, 1 are in LDS
        {   // sub-step 0 of stage 0: nothing in flight yet
            const unsigned char* pa = patch0 + toffr[0];
#pragma unroll
            for (int k = 0; k < TM; ++k) P[0][k] = rdp(pa, 0, k);
#pragma unroll
            for (int k = 0; k < TNW; ++k) Wf[k] = rdw(wring, 0, k);
        }
        // one stage: the MFMAs of both K sub-steps; during the second, the first fragments of the NEXT stage are fetched (its
        // weights were committed a stage ago, its patch -- at a slab change -- NTAPS-1 stages ago).  Past the end of the stream
        // those reads return bytes nobody uses.
        auto stage = [&](const unsigned char* pa, const unsigned char* wb, const unsigned char* npa, const unsigned char* nwb) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const unsigned char* rpa = sub == 0 ? pa : npa;
                const unsigned char* rwb = sub == 0 ? wb : nwb;
                const int rsub = sub == 0 ? 1 : 0;
#pragma unroll
                for (int mj = 0; mj < TNW; ++mj) {
#pragma unroll
                    for (int k = mj; k < TM; k += TNW)          // the TM pixel-fragment reads of the sub-step, dealt over its columns
                        if (!(WT_ABL & 4)) P[sub ^ 1][k] = rdp(rpa, rsub, k);
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[mi][mj] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, Wf[mj]),
                                                                               __builtin_bit_cast(bf16x8, P[sub][mi]), acc[mi][mj], 0, 0, 0);
                    if (!(WT_ABL & 4)) Wf[mj] = rdw(rwb, rsub, mj);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        int slot = 0;
        for (int q = 0, sl = 0, tk = 0; q < Q; ++q) {
            const unsigned char* pcur = patch0 + (q & 1) * pbytes;
            const unsigned char* pnxt = patch0 + ((q + 1) & 1) * pbytes;
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                const int nslot = slot == 2 ? 0 : slot + 1;
                const unsigned char* npa = tap + 1 < NTAPS ? pcur + toffr[tap + 1 < NTAPS ? tap + 1 : 0] : pnxt + toffr[0];
                stage(pcur + toffr[tap], wring + slot * WSTG, npa, wring + nslot * WSTG);
                slot = nslot;
                // End of stage g.  A bare s_barrier: __syncthreads() would first wait for every outstanding LDS read
                // (s_waitcnt lgkmcnt(0)), i.e. for the next stage's fragments fetched just above, and drain the MFMA pipe once
                // per stage.  Nothing needs that wait: the reads of THIS stage's ring slot and patch have been consumed by
                // MFMAs already issued, and what the staging waves overwrite after this barrier (ring slot g % 3; the other
                // patch buffer at a slab start) is not what the outstanding reads address (slot (g+1) % 3, the live patch).
                if (!(WT_ABL & 8)) __builtin_amdgcn_s_barrier();
            }
            if (++sl == nslab) {
                epilogue(xw.first + tk * xw.step);
                sl = 0; ++tk;
            }
        }
        if ((EPI < 0 ? d.dot != nullptr : (EPI & kEpiDot) != 0)) {
            dacc = wave_sum(dacc);
            if (lane == 0) atomicAdd(d.dot, dacc);
        }
    }
}

// Fills the plan; returns 1 when the descriptor is this kernel's case.
int plan(const XmcConvDesc* d, WtCfg* t, int* mode, int* tn) {
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16 || d->src_shift != 0) return 0;
    if (d->CS % 64 != 0 || d->CS > 64 * 64 / 4) return 0;
    if (d->CDw % 128 == 0) *tn = 8;
    else if (d->CDw == 64 && d->CS > 64) *tn = 4;      // 64 -> 64 stays on the weights-resident kernel (conv_tile.hip)
    else return 0;
    if (d->MW % 16 != 0) return 0;
    const int TW = d->MW >= 32 ? 32 : 16, TH = 256 / TW;
    if (d->MH % TH != 0 || d->MW % TW != 0) return 0;
    t->TH = TH; t->TW = TW; t->log2TW = TW == 32 ? 5 : 4;
    t->tiles_y = d->MH / TH; t->tiles_x = d->MW / TW;
    int maxpix = 0;
    if (d->SA == 1) {
        if (d->ntaps != 9 && d->ntaps != 4) return 0;
        *mode = 0;
        t->nslab = d->CS / 64;
        for (int z = 0; z < d->nclass; ++z) {
            int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
            for (int k = 0; k < d->ntaps; ++k) {
                const int h = d->dh[z][k], w = d->dw[z][k];
                hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax;
                wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax;
            }
            // halo depth <= tile size (a halo row is outside the image only for tiles on that border)
            if (hmin < -TH || hmax > TH || wmin < -TW || wmax > TW) return 0;
            t->dh0[z] = hmin; t->dw0[z] = wmin;
            t->PH[z] = TH + (hmax - hmin); t->PW[z] = TW + (wmax - wmin);
            if (t->PH[z] * t->PW[z] > 32 * kPIT) return 0;
            if (d->SH != d->MH || d->SW != d->MW) return 0;
            maxpix = t->PH[z] * t->PW[z] > maxpix ? t->PH[z] * t->PW[z] : maxpix;
        }
    } else if (d->SA == 2) {
        if (d->ntaps != 16 || d->nclass != 1 || d->DA != 1 || d->SH != 2 * d->MH || d->SW != 2 * d->MW) return 0;
        *mode = 1;
        t->nslab = 4 * (d->CS / 64);
        if (t->nslab > 64) return 0;
        for (int g = 0; g < 4; ++g)
            for (int tp = 0; tp < 4; ++tp) {
                const int wh = 2 * (tp >> 1) + (g >> 1) - 1, ww = 2 * (tp & 1) + (g & 1) - 1;
                int found = -1;
                for (int k = 0; k < 16; ++k)
                    if (d->dh[0][k] == wh && d->dw[0][k] == ww) found = found < 0 ? k : 99;
                if (found < 0 || found > 15) return 0;
                t->tsel[g][tp] = (int8_t)found;
            }
        t->PH[0] = TH + 1; t->PW[0] = TW + 1; t->dh0[0] = t->dw0[0] = 0;
        maxpix = t->PH[0] * t->PW[0];
        if (maxpix > 32 * kPIT) return 0;
    } else {
        return 0;
    }
    if (d->dst_pool && (d->DA != 1 || d->nclass != 1 || (d->DH & 1) || (d->DW & 1))) return 0;
    if (d->res_mode == 2 && d->DA != 1) return 0;
    t->patch_bytes = (maxpix * kPStride + 15) & ~15;
    // 32-bit unit offsets in the kernel
    if ((int64_t)d->N * d->SH * d->SW * (d->CS / 8) >= (1ll << 31) || (int64_t)d->N * d->DH * d->DW * (d->CD / 8) >= (1ll << 31)) return 0;
    return 1;
}

template <int NTAPS, int MODE, int TN, int CW>
int launch(const XmcConvDesc& d, const WtCfg& t, hipStream_t st) {
    constexpr int BN = 16 * TN;
    const size_t lds = (size_t)3 * BN * 128 + 2 * (size_t)t.patch_bytes;
    if (lds > XMC_MAX_DYN_LDS) return XMC_ESHAPE;
    const int ntiles = d.N * t.tiles_y * t.tiles_x, ny = d.CDw / BN;
    int gx = 256 / (ny * d.nclass);               // one 8-wave workgroup per CU, persistent over its tiles
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    gx = xmc_ab_grid(gx);
    // epilogue option sets of the training step as compile-time instantiations (common.h: kEpi*), eight compute waves only
    static const bool no_epi = xmc_debug_off("no_wtile_epi");
    const int epi = (no_epi || CW != 8) ? -1 : xmc_epi_mask(d);
#define XMC_W2_EPI(E)                                                                                                                          \
    if (epi == (E)) {                                                                                                                          \
        XMC_ALLOW_BIG_LDS((wtile2_kernel<NTAPS, MODE, TN, CW, (E)>));                                                                          \
        hipLaunchKernelGGL((wtile2_kernel<NTAPS, MODE, TN, CW, (E)>), dim3((unsigned)gx, (unsigned)ny, (unsigned)d.nclass), dim3(64 * CW + 256), lds, st, \
                           d, t, ntiles);                                                                                                      \
        xmc_note_kernel("wtile2_kernel<%d, %d, %d, %d>", NTAPS, MODE, TN, CW);                                                                 \
        XMC_LAUNCH_CHECK();                                                                                                                    \
        return 0;                                                                                                                              \
    }
    if constexpr (CW == 8 && NTAPS == 9 && TN == 8) {
        XMC_W2_EPI(kEpiGSum) XMC_W2_EPI(kEpiDKeep) XMC_W2_EPI(kEpiDFwd) XMC_W2_EPI(kEpiDLast) XMC_W2_EPI(kEpiMask) XMC_W2_EPI(0) XMC_W2_EPI(kEpiDLin)
        XMC_W2_EPI(kEpiDKeepS) XMC_W2_EPI(kEpiDLastS) XMC_W2_EPI(kEpiDgDot)
        XMC_W2_EPI(kEpiBias) XMC_W2_EPI(kEpiBias | kEpiLrelu)            // the attention-modulation blocks' convolutions
    } else if constexpr (CW == 8 && NTAPS == 9) {
        XMC_W2_EPI(kEpiBias) XMC_W2_EPI(kEpiBias | kEpiLrelu) XMC_W2_EPI(0)
    } else if constexpr (CW == 8 && MODE == 1 && TN == 8) {
        XMC_W2_EPI(kEpiLrelu) XMC_W2_EPI(0) XMC_W2_EPI(kEpiMask)
    } else if constexpr (CW == 8 && NTAPS == 4 && MODE == 0) {
        XMC_W2_EPI(kEpiRes) XMC_W2_EPI(kEpiBias) XMC_W2_EPI(0)
    }
#undef XMC_W2_EPI
    if (d.sign_bits || d.dot) return 1;          // only the compile-time sets above carry these two; the next kernel in line takes it
    xmc_note_generic_epi(NTAPS == 9 ? (TN == 8 ? "wtile2<9,0,8>" : "wtile2<9,0,4>") : MODE == 1 ? (TN == 8 ? "wtile2<4,1,8>" : "wtile2<4,1,4>") : "wtile2<4,0>", CW == 8 ? xmc_epi_mask(d) : -1);
    XMC_ALLOW_BIG_LDS((wtile2_kernel<NTAPS, MODE, TN, CW>));
    hipLaunchKernelGGL((wtile2_kernel<NTAPS, MODE, TN, CW>), dim3((unsigned)gx, (unsigned)ny, (unsigned)d.nclass), dim3(64 * CW + 256), lds, st, d, t,
                       ntiles);
    xmc_note_kernel("wtile2_kernel<%d, %d, %d, %d>", NTAPS, MODE, TN, CW);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// entry used by xmc_conv_igemm's dispatcher: 0 = launched, 1 = not this kernel's case, < 0 = error
int xmc_conv_wtile_try(const XmcConvDesc* d, void* stream) {
    WtCfg t;
    int mode = 0, tn = 8;
    if (!plan(d, &t, &mode, &tn)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    static const bool cw4 = xmc_debug_off("wtile_cw4");      // A/B: one compute wave per SIMD
#define WT_GO(NT_, MD_, TN_) (cw4 ? launch<NT_, MD_, TN_, 4>(*d, t, st) : launch<NT_, MD_, TN_, 8>(*d, t, st))
    if (mode == 1) rc = tn == 8 ? WT_GO(4, 1, 8) : WT_GO(4, 1, 4);
    else if (d->ntaps == 9) rc = tn == 8 ? WT_GO(9, 0, 8) : WT_GO(9, 0, 4);
    else rc = tn == 8 ? WT_GO(4, 0, 8) : WT_GO(4, 0, 4);
#undef WT_GO
    return rc == XMC_ESHAPE ? 1 : rc;
}
