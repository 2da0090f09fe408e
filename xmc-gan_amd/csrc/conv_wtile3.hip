// Streamed-weights halo-patch convolution, third form (round 3): two MFMA waves per SIMD in ping-pong, LDS-DMA staging, no staging
// waves.  Forward / data gradient of the wide layers of D and G (Cin % 64 == 0, Cout % 128 == 0), 16-bit storage.
//
// What it keeps from conv_wtile.hip: a persistent workgroup per CU owns 256 output pixels (TH x TW of one image) x BN output
// channels; K is cut into slabs of 64 source channels; per slab the input patch (tile + halo) is staged in LDS ONCE and every tap
// reads its pixel fragments from it at a shifted offset, so only the weights of each (slab, tap) STAGE are streamed.
// What changes (the role-split kernel sat at 41 % MfmaUtil with LDS ~2/3 busy and staging waves that cannot
// issue faster than the vector-memory path accepts):
//   * ALL eight waves are MFMA waves.  Wave tile 128 pixels x 64 channels (BN = 256: 2 x 4 waves) -- 24 fragment reads per 64
//     MFMAs = 37 % of the LDS read bandwidth at the MFMA rate instead of 50-67 % -- or 64 x 64 (BN = 128: 4 x 2 waves).
//   * The two waves of a SIMD (w and w + 4) run the SAME program one barrier apart: while one is in a LOAD phase (fragment
//     ds_reads for its next 16 MFMAs + its share of the staging) the other is in its MFMA phase (16 back-to-back MFMAs, nothing
//     else), then they swap.  The matrix pipe of every SIMD always has one wave feeding it; LDS latency, address arithmetic and
//     the staging issue are all in the partner's shadow.  Two barriers per 16 MFMAs per wave, raw s_barrier (no vmcnt drain).
//   * Staging is LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write -- the 79 B/clk store path and the staging
//     registers are out of the loop).  LDS images are lane-linear; the bank-conflict swizzle (16-byte chunk index XOR (row & 7),
//     conflict-free ds_read_b128 for the row = lane & 15 fragment pattern at ANY row alignment, checked by enumeration) is
//     applied to the per-lane SOURCE address and again on the read.  Halo pixels outside the image read a 16-byte zero page.
//   * Weight ring: 2 slots of BN x 128 B at BN = 256 (stage g+1 is fetched during stage g, issued in phase 1, awaited in phase 3),
//     3 slots at BN = 128 (two phases per stage: stage g+2 issued in phase 1 of stage g, awaited in phase 1 of stage g+1).
//     Two patch buffers (slab q+1 is fetched during slab q, one or two 1 KB pieces per wave and stage).
//     2 x 32 + 2 x 42.5 KB = 149 KB.
// Hazard bookkeeping (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"): interval i is the
// time between barriers i-1 and i; the first-half waves (0-3) run LOAD(k) in interval 2k and MFMA(k) in 2k+1, the second half
// LOAD(k) in 2k+1 and MFMA(k) in 2k+2.  RAW: a DMA is awaited (counted vmcnt) before the barrier that ends the awaiting phase, and
// the first read of that data is at least one barrier later for either half.  WAR: a slot is re-filled no earlier than two
// intervals after the last ds_read of it was ISSUED, and those reads were retired (lgkmcnt(0) in front of the MFMAs) one interval
// earlier.
//
// MODE 0: unit source stride, 3x3 or 2x2 tap sets (the parity classes of a stride-2 data gradient / fused upsample conv: one
//         class per blockIdx.z).  MODE 1: the 4x4 stride-2 forward as a dense 2x2-tap convolution over the space-to-depth view,
//         gathered from NHWC while it is staged (see conv_wtile.hip).
// Takes over: F.conv2d at df_gan.py:187-188 (G_Block c1/c2), 273,276 (resD conv_r) for the wide layers and the matching halves of
// errD.backward() / errG.backward() (train_gan.py:228,288).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct W3Cfg {
    int TH, TW, log2TW;
    int tiles_y, tiles_x;
    int PH[XMC_MAX_CLASSES], PW[XMC_MAX_CLASSES];        // patch size per class
    int dh0[XMC_MAX_CLASSES], dw0[XMC_MAX_CLASSES];      // min tap offsets per class (MODE 0)
    int nslab;                                           // K slabs per tile: CS/64 (MODE 0), 4*CS/64 (MODE 1)
    int patch_bytes;                                     // one patch buffer: 384 pixel slots x 128 B
    int8_t tsel[4][4];                                   // MODE 1: tap index of (dy*2+dx, ta*2+tb)
    int ipt;                                             // images per tile: 1, or 4 (8 x 8 maps: a tile is four whole images side by side)
};

constexpr int kPI = 6;              // 1 KB patch pieces (8 pixels x 128 B) per wave and slab, at most
constexpr int kPieces = 47;         // pieces of a patch buffer: 376 pixel slots (10 x 34 = 340 pixels; four 8 x 8 images with their halos, 10 x 37 = 370)
constexpr int kPatchSlots = kPieces * 8;

__device__ __attribute__((aligned(16))) unsigned char g_zero16[16];     // what a halo pixel outside the image reads

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) void* gptr_t;

// one LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to the wave-uniform LDS address `dst` + lane * 16
__device__ __forceinline__ void glds16(const void* src, lds_u8* dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// WM = wave rows (pixel direction): 2 -> BN = 256, wave tile 128 x 64, four phases per stage, ring of 2
//                                   4 -> BN = 128, wave tile  64 x 64, two phases per stage, ring of 4
template <int NTAPS, int MODE, int WM, int EPI = -1>
__global__ __launch_bounds__(512) void wtile3_kernel(const XmcConvDesc d, const W3Cfg t, int ntiles) {
    constexpr int WN = 8 / WM, BN = 64 * WN;
    constexpr int TMW = 16 / WM;                 // 16-pixel blocks per wave (8 or 4)
    constexpr int TNW = 4;                       // 16-channel blocks per wave
    constexpr int NPH = TMW / 2;                 // phases (clusters of 16 MFMAs) per stage: 4 or 2
    constexpr int RING = WM == 2 ? 2 : 4;
    constexpr int WSTG = BN * 128;               // bytes of one weight stage
    constexpr int NI = BN / 64;                  // 1 KB weight pieces per wave and stage (4 or 2)
    constexpr int PPS = (kPI + NTAPS - 1) / NTAPS;   // patch pieces per wave and stage (1 for 3x3, 2 for 2x2 tap sets)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // everything lives in ONE LDS array (a second __shared__ object beside an LDS-DMA target can make hipcc drain vmcnt in front
    // of every ds_read): [ring][2 patches][tables]
    unsigned char* const wring = smem;
    unsigned char* const patch0 = smem + RING * WSTG;
    const int pbytes = t.patch_bytes;
    int* const s_toff = reinterpret_cast<int*>(smem + RING * WSTG + 2 * pbytes);      // [16] tap -> patch PIXEL offset
    int* const s_wbase = s_toff + XMC_MAX_TAPS;                                       // [16] (slab group, tap) -> first unit of the weight slice
    int* const s_slab = s_wbase + XMC_MAX_TAPS;                                       // [64] MODE 1: slab -> (group << 8) | channel block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;                  // SIMD partner: waves w and w + 4 share a SIMD
    const int wr = WM == 2 ? (wave >> 2) : (wave & 3);      // pixel slice of the tile
    const int wc = WM == 2 ? (wave & 3) : (wave >> 2);      // 64-channel slice of BN
    const int n0 = blockIdx.y * BN;
    const int cls = blockIdx.z;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PW = t.PW[cls], PH = t.PH[cls];
    const int dh0 = MODE == 0 ? t.dh0[cls] : 0, dw0 = MODE == 0 ? t.dw0[cls] : 0;
    const int cs_units = d.CS / 8;
    const int nslab = t.nslab;
    const int cb = d.CS / 64;
    const bool mi = t.ipt > 1;                   // multi-image tiles (8 x 8 maps)
    if (tid < XMC_MAX_TAPS) {
        if (MODE == 0) {
            const int tt = tid < NTAPS ? tid : 0;
            s_toff[tid] = ((d.dh[cls][tt] - dh0) * PW + (d.dw[cls][tt] - dw0)) | ((d.dw[cls][tt] - dw0) << 16);
            s_wbase[tid] = d.wi[cls][tt] * d.CDw * cs_units;
        } else {
            const int g4 = tid >> 2, tp = tid & 3;                     // tid = (dy*2+dx)*4 + (ta*2+tb)
            s_toff[tid] = ((tp >> 1) * PW + (tp & 1)) | ((tp & 1) << 16);
            s_wbase[tid] = d.wi[0][t.tsel[g4][tp]] * d.CDw * cs_units;
        }
    }
    if (tid >= 64 && tid < 128) {
        const int sl = tid - 64;
        s_slab[sl] = MODE == 0 ? sl : (((sl / cb) << 8) | (sl % cb));
    }
    const XcdWalk xw = xmc_xcd_walk(ntiles);      // XCD-aware tile walk (common.h): this workgroup's tiles are xw.first, + xw.step, ... < xw.end
    const int mytiles = xw.first < xw.end ? (xw.end - xw.first + xw.step - 1) / xw.step : 0;
    const int Q = mytiles * nslab;               // patches (tile, slab) of this workgroup, one stream
    __syncthreads();
    if (mytiles <= 0) return;

    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);
    const u32x4* const zero16 = reinterpret_cast<const u32x4*>(g_zero16);

    // ------------------------------------------------------------------------------------------------ staging addresses
    // weights: piece i of this wave covers physical ring rows (wave * NI + i) * 8 + (lane >> 3), LDS chunk slot lane & 7, which
    // holds logical chunk (lane & 7) ^ (row & 7) of logical channel lrow(row):
    //   physical row (n-block j, r) <- logical channel (j/2)*32 + (r/4)*8 + (j%2)*4 + r%4, so that lane group fc = r/4 ends with
    //   channels u*32 + fc*8 .. +7 of unit u = j/2 in its two accumulator blocks (16-byte stores in the epilogue)
    // (one per-lane register + wave-uniform piece terms: the row of piece gi = wave * NI + i is gi * 8 + (lane >> 3))
    const int wlane = (n0 + (lane >> 5) * 8 + ((lane >> 3) & 3)) * cs_units + ((lane & 7) ^ (lane >> 3));
    auto wpiece_off = [&](int i) {               // uniform: (j/2)*32 + (j%2)*4 + (r/8)*16 channels, j = gi / 2, r/8 = gi % 2
        const int gi = wave * NI + i, j = gi >> 1;
        return ((j >> 1) * 32 + (j & 1) * 4 + (gi & 1) * 16) * cs_units;
    };
    // patch: piece i covers pixel slots (wave * kPI + i) * 8 + (lane >> 3), chunk slot lane & 7 <- logical chunk ^ (pixel & 7)
    int psrc[kPI];
    unsigned hpack = 0;                          // 5 bits per piece: halo nibble (which tile borders the pixel lies beyond) | 16 = not a patch pixel
#pragma unroll
    for (int i = 0; i < kPI; ++i) {
        const int pp = (wave * kPI + i) * 8 + (lane >> 3);
        const bool in = pp < PH * PW;
        const int py = pp / PW, px = pp - py * PW;
        // swizzle key = the pixel's column in ITS image's patch (mod 8): the reader's XOR term then depends on lane and tap only
        unsigned hb;
        if (!mi) {
            const int ch = (lane & 7) ^ (px & 7);
            if (MODE == 0) {
                psrc[i] = in ? ((dh0 + py) * d.SW + (dw0 + px)) * cs_units + ch : 0;
                hb = (py < -dh0 ? 1u : 0u) | (py >= t.TH - dh0 ? 2u : 0u) | (px < -dw0 ? 4u : 0u) | (px >= t.TW - dw0 ? 8u : 0u);
            } else {
                psrc[i] = in ? ((2 * py - 1) * d.SW + (2 * px - 1)) * cs_units + ch : 0;
                hb = (py == 0 ? 1u : 0u) | (py == t.TH ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == t.TW ? 8u : 0u);
            }
        } else if (MODE == 0) {
            // four 8 x 8 images side by side, 9 patch columns apart: [z] img0 [z] img1 [z] img2 [z] img3 [z] -- a zero column between
            // neighbours serves as the right halo of one and the left halo of the next; every tile border is an image border, so a
            // pixel is either data (bits 0) or always zero (bits 15 against pborder = 15)
            const int q = px + dw0, kq = q >= 0 ? q / 9 : 0, x = q - 9 * kq, row = dh0 + py;
            const bool data = in && q >= 0 && x < 8 && kq < 4 && row >= 0 && row < d.SH;
            const int ch = (lane & 7) ^ ((x - dw0) & 7);
            psrc[i] = data ? ((kq * d.SH + row) * d.SW + x) * cs_units + ch : 0;
            hb = data ? 0u : 15u;
        } else {
            // space-to-depth patches of four 16 x 16 images: 9 columns each (both outer columns carry data for one of the groups)
            const int kq = px / 9, x9 = px - 9 * kq;
            const int ch = (lane & 7) ^ (x9 & 7);
            psrc[i] = in ? ((kq * d.SH + 2 * py - 1) * d.SW + (2 * x9 - 1)) * cs_units + ch : 0;
            hb = (py == 0 ? 1u : 0u) | (py == t.TH ? 2u : 0u) | (x9 == 0 ? 4u : 0u) | (x9 == 8 ? 8u : 0u);
        }
        hpack |= (in ? hb : 16u) << (5 * i);
    }
    lds_u8* const lds0 = (lds_u8*)smem;
    const int wpiece0 = wave * NI * 1024;

    // patch q = (tile, slab): base unit of its pixel (0,0) / channel block, and which tile borders lie on the image border
    int pbase = 0;
    unsigned pborder = 0;
    auto patch_setup = [&](int q) {
        const int tk = q / nslab, sl = q - tk * nslab;
        const int tile = xw.first + tk * xw.step;
        const int timg = tile / tpi, trem = tile - timg * tpi, img = timg * t.ipt;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        if (MODE == 0) {
            pbase = ((img * d.SH + a0) * d.SW + b0) * cs_units + sl * 8;
            pborder = (a0 + dh0 < 0 ? 1u : 0u) | (a0 + PH + dh0 > d.SH ? 2u : 0u) | (b0 + dw0 < 0 ? 4u : 0u) | (b0 + PW + dw0 > d.SW ? 8u : 0u);
            if (mi) pborder = 15u;
        } else {
            const int si = s_slab[sl];
            const int grp = si >> 8, cbi = si & 0xff, dy = grp >> 1, dx = grp & 1;
            pbase = ((img * d.SH + 2 * a0 + dy) * d.SW + 2 * b0 + dx) * cs_units + cbi * 8;
            pborder = ((a0 == 0 && dy == 0) ? 1u : 0u) | ((a0 + t.TH == d.MH && dy == 1) ? 2u : 0u) |
                      ((b0 == 0 && dx == 0) ? 4u : 0u) | (((mi || b0 + t.TW == d.MW) && dx == 1) ? 8u : 0u);
        }
    };
    // Every lane of every piece issues (halo pixels outside the image and slots past the end of the patch copy the zero page).  The
    // eight waves cover 48 pieces, the buffer has 44: pieces 44-47 are re-directed onto piece 43, which -- like them -- lies wholly
    // past the largest patch (340 pixels < 43 x 8) and only ever receives zeros.  No branch, the same count on every wave.
    auto patch_piece = [&](int i, int buf, int z) {     // i: compile-time piece index; z: opaque zero (keeps the 64-bit address
        // arithmetic at the point of use instead of in hoisted, spilled registers)
        const bool ok = (((hpack + (unsigned)z) >> (5 * i)) & (pborder | 16u)) == 0;
        const u32x4* s = ok ? src16 + (unsigned)(pbase + psrc[i] + z) : zero16;
        const int g = wave * kPI + i;
        glds16(s, lds0 + RING * WSTG + buf * pbytes + (g < kPieces ? g : kPieces - 1) * 1024);
    };
    // weights of stage (slab sl, tap): the same for every tile
    int wbr[NTAPS];                              // MODE 0: first unit of each tap's weight slice, in scalar registers (an LDS table read
#pragma unroll                                   // inside a LOAD phase would wait for the fragment reads issued before it)
    for (int k = 0; k < NTAPS; ++k) wbr[k] = __builtin_amdgcn_readfirstlane(s_wbase[k]);
    // MODE 1: the weight slice of (slab, tap) depends on the slab's group; looked up once per slab for this slab and the next (the
    // stages fetched ahead wrap into it), not inside the phases
    int wbm[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto wbm_setup = [&](int which, int sl) {
        const int si = __builtin_amdgcn_readfirstlane(s_slab[sl]);
#pragma unroll
        for (int k = 0; k < 4; ++k) wbm[which][k] = __builtin_amdgcn_readfirstlane(s_wbase[(si >> 8) * 4 + k]) + (si & 0xff) * 8;
    };
    auto weights = [&](int sl, int tap, int slot, int z, int i0 = 0, int i1 = 16, int nxt = 0) {     // pieces [i0, i1) of the stage
        const int wb = MODE == 0 ? wbr[tap < NTAPS ? tap : 0] + sl * 8 : wbm[nxt][tap & 3];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i >= i0 && i < i1) glds16(w16 + (unsigned)(wb + wpiece_off(i) + wlane + z), lds0 + slot * WSTG + wpiece0 + i * 1024);
    };

    // ------------------------------------------------------------------------------------------------ fragment addresses
    const int fr = lane & 15, fc = lane >> 4;
    // patch PIXEL index of (pixel block i, lane fr) at tap offset 0: ppl0 + (i / 2) * ps2 + (i % 2) * ps1 (8 x 32 tiles: block i is
    // half a tile row; 16 x 16 tiles: a whole row) -- one per-lane register, the block terms are wave-uniform
    int ppl0;
    {
        const int ml = wr * (256 / WM) + fr;
        const int col = ml & (t.TW - 1);
        ppl0 = (ml >> t.log2TW) * PW + (mi ? col + (col >> 3) : col);       // multi-image: image k's columns start 9 k patch columns in
    }
    const int ps1 = t.log2TW == 5 ? (mi ? 18 : 16) : PW, ps2 = t.log2TW == 5 ? PW : 2 * PW;
    // weight fragment (n-block j of this wave, row fr, chunk ksub*4 + fc): chunk slot = chunk ^ (row & 7), row & 7 == fr & 7
    const int wf0 = (wc * TNW * 16 + fr) * 128 + ((fc ^ (fr & 7)) << 4), wf1 = wf0 ^ 64;      // K sub-step 0 / 1

    f32x4 acc[TMW][TNW];
    float dacc = 0.f;                            // running sum for XmcConvDesc.dot over this workgroup's tiles
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cd8 = d.CD / 8;
    const int ch0 = n0 + wc * 64 + fc * 8;       // first channel of this lane's unit 0
    const int nw8 = (n0 + wc * 64) >> 3;
    // epilogue from registers: acc[i][j][r] = pixel (block i, fr), channel n0 + wc*64 + (j/2)*32 + fc*8 + (j%2)*4 + r; clears acc.
    // order: bias, activation, [second output], alpha, LeakyReLU' mask, (row-indexed / half-resolution, scaled) residual, pool.
    auto epilogue = [&](int tile) {
        int lane_op = fr;                        // opaque: keeps the store addresses from being hoisted out of the tile loop (spills)
        asm volatile("" : "+v"(lane_op) :: "memory");
        const int timg = tile / tpi, trem = tile - timg * tpi, img = timg * t.ipt;
        const int a0 = (trem / t.tiles_x) * t.TH, b0 = (trem % t.tiles_x) * t.TW;
        const int dbase = (((img * d.DH + a0 * d.DA + d.dph[cls]) * d.DW) + b0 * d.DA + d.dpw[cls]) * cd8 + nw8;
        // multi-image tiles: tile column tx is column tx % 8 of image img + tx / 8
        const int dimg = d.DH * d.DW * cd8, rimg = d.res_mode ? d.MH * d.MW * cd8 : dimg;
        const int rbase = d.res_mode ? ((img * d.MH + a0) * d.MW + b0) * cd8 + nw8 : dbase;
        const int rsy = d.res_mode ? d.MW : d.DA * d.DW, rsx = d.res_mode ? 1 : d.DA;
        constexpr bool RT = EPI < 0;                   // epilogue options read from the descriptor (common.h: kEpi*)
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
        // four pixel blocks at a time (register pressure: the accumulators of the whole wave tile are live here)
#pragma unroll
        for (int g = 0; g < TMW / 4; ++g) {
            int eo[4], ro[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = wr * (256 / WM) + (g * 4 + i) * 16 + lane_op;
                const int ty = ml >> t.log2TW, txt = ml & (t.TW - 1);
                const int tx = mi ? (txt & 7) : txt, ki = mi ? (txt >> 3) : 0;
                eo[i] = dbase + ((ty * d.DA) * d.DW + tx * d.DA) * cd8 + ki * dimg + fc;
                if (d.res_mode == 2)
                    ro[i] = (((img + ki) * (d.DH >> 1) + ((a0 + ty) >> 1)) * (d.DW >> 1) + ((b0 + tx) >> 1)) * cd8 + nw8 + fc;
                else
                    ro[i] = rbase + (ty * rsy + tx * rsx) * cd8 + ki * rimg + fc;
            }
#pragma unroll
            for (int u = 0; u < TNW / 2; ++u) {
                if (ch0 + u * 32 >= d.CD) continue;
                float fin[4][8];
                bf16x8 mkv[4], rrv[4];
                if (e_mask) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mkv[i] = mask8[eo[i] + u * 4];
                }
                if (e_res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rrv[i] = res8[ro[i] + u * 4];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[g * 4 + i][2 * u][r]; v[4 + r] = acc[g * 4 + i][2 * u + 1][r]; }
                    if (e_bias) {
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
                    dst8[eo[i] + u * 4] = o;
                }
                if (e_pool) {
                    // 2x2 average of the ROUNDED output (== F.avg_pool2d of dst): the vertical neighbour is pixel block i+2 (8x32
                    // tiles) or i+1 (16x16 tiles) of the same lane -- both inside this group of four -- the horizontal one lane ^ 1
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int i0 = t.log2TW == 5 ? pr : 2 * pr;
                        const int ml = wr * (256 / WM) + (g * 4 + i0) * 16 + lane_op;
                        const int ty = ml >> t.log2TW, txt = ml & (t.TW - 1);
                        const int tx = mi ? (txt & 7) : txt, ki = mi ? (txt >> 3) : 0;
                        bf16x8 o;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            float sm = (t.log2TW == 5 ? fin[pr][r] + fin[pr + 2][r] : fin[2 * pr][r] + fin[2 * pr + 1][r]);
                            sm += xmc_xor1(sm);
                            o[r] = (xmc_h16)((d.pool_scale == 0.f ? 0.25f : d.pool_scale) * sm);
                        }
                        if ((lane_op & 1) == 0)
                            pool8[(((img + ki) * (d.DH >> 1) + ((a0 + ty) >> 1)) * (d.DW >> 1) + ((b0 + tx) >> 1)) * cd8 + nw8 + fc + u * 4] = o;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    int toffr[NTAPS];                            // tap -> patch pixel offset, in scalar registers
#pragma unroll
    for (int k = 0; k < NTAPS; ++k) toffr[k] = __builtin_amdgcn_readfirstlane(s_toff[k]);

    // ------------------------------------------------------------------------------------------------ prologue
    patch_setup(0);
    if (MODE == 1) wbm_setup(0, 0);
#pragma unroll
    for (int i = 0; i < kPI; ++i) patch_piece(i, 0, 0);
#pragma unroll
    for (int s0 = 0; s0 < RING - 1; ++s0) weights(0, s0, s0, 0);      // the loop fetches RING - 1 stages ahead (RING - 1 <= 3 < NTAPS)
    patch_setup(Q > 1 ? 1 : 0);                  // the loop fetches patch q+1 during slab q
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (half == 1) __builtin_amdgcn_s_barrier();     // the second wave of every SIMD runs one interval behind the first

    // ------------------------------------------------------------------------------------------------ the stream
    u32x4 Wf[TNW], Pf[4];
    int slot = 0;                                // ring slot of the running stage
    int sl = 0, tk = 0;
    for (int q = 0; q < Q; ++q) {
        const int pcur = (q & 1) * pbytes, pnxt = ((q + 1) & 1) * pbytes;
        // The fragment addresses of a tap do not depend on q: left alone, loop-invariant code motion hoists all NTAPS x TMW of them
        // out of the slab loop and spills them (and every scratch re-load drains the LDS-DMA queue with vmcnt(0)).  An opaque zero
        // per slab keeps them where they are used: three VALU instructions per fragment read, in the partner's MFMA shadow.
        int zq = 0;
        asm volatile("" : "+v"(zq));
        const int sl1 = sl + 1 == nslab ? 0 : sl + 1;
        if (MODE == 1) { wbm_setup(0, sl); wbm_setup(1, sl1); }
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const unsigned char* const wb = wring + slot * WSTG;
            const unsigned char* const pb = patch0 + pcur;
            // stage to fetch: RING - 1 stages ahead (past the end of the stream it re-fetches valid addresses into a slot nobody reads)
            constexpr int D = RING - 1;
            const int ft = tap + D >= NTAPS ? tap + D - NTAPS : tap + D;
            const int fs = tap + D >= NTAPS ? sl1 : sl;
            const int fslot = slot + D >= RING ? slot + D - RING : slot + D;
#pragma unroll
            for (int p = 0; p < NPH; ++p) {
                const int ksub = NPH == 4 ? (p >> 1) : p;
                const int mb = NPH == 4 ? (p & 1) * 4 : 0;          // first pixel block of this cluster
                // ---------------- LOAD phase of cluster (stage, p)
                if (NPH == 4 ? (p & 1) == 0 : true) {
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
                    {   // inline asm + hand-placed counted waits: with LDS-DMA in the kernel hipcc answers every fragment use with
                        // lgkmcnt(0), i.e. the first MFMA of a cluster would wait for the LAST of its eight reads
                        const unsigned wa = (unsigned)(size_t)(wb + (ksub ? wf1 : wf0) + zq - smem);
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Wf[j]) : "v"(wa), "n"(j * 2048));
                    }
                }
                {
                    // chunk fc + 4 ksub of a pixel sits at slot (fc + 4 ksub) ^ (patch column & 7), and the column of (block i, lane
                    // fr) at this tap is fr + tdx (mod 8) for every block: two VALU instructions per (tap, sub-step), one per read
                    const int slot16 = (((fc + 4 * ksub) ^ (fr + zq + (toffr[tap] >> 16))) & 7) << 4;
                    const unsigned char* const pl = pb + ((ppl0 + (toffr[tap] & 0xffff)) * 128 + slot16);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                    {
                        const unsigned pa = (unsigned)(size_t)(pl + (((mb + i) >> 1) * ps2 + ((mb + i) & 1) * ps1) * 128 - smem);
                        asm volatile("ds_read_b128 %0, %1" : "=v"(Pf[i]) : "v"(pa));
                    }
                }
                // Staging.  Every wait is a full vmcnt(0) on operations issued at least two intervals earlier by this wave, and the
                // first ds_read of what it retires is at least two barriers later for either half.
                if (NPH == 4) {
                    // at most two LDS-DMA per phase (their issue costs the wave 60-185 cycles each).  The slot of stage g-1 was last
                    // read in phase 2 of g-1, three intervals before the first piece; the next patch's buffer in the last phase of
                    // the previous slab, three intervals before phase 1.
                    if (p == 0) weights(fs, ft, fslot, zq, 0, 1, tap + D >= NTAPS);
                    if (p == 1) {
                        weights(fs, ft, fslot, zq, 1, 2, tap + D >= NTAPS);
#pragma unroll
                        for (int e = 0; e < PPS; ++e)
                            if (tap * PPS + e < kPI) patch_piece(tap * PPS + e, (q + 1) & 1, zq);
                    }
                    if (p == 2) weights(fs, ft, fslot, zq, 2, 4, tap + D >= NTAPS);
                    if (p == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage g+1 (first read: after this phase's barrier)
                } else {
                    // ring of 4: stage g+3 goes into the slot of stage g-1 (last read in phase 1 of g-1: three intervals before phase 0)
                    if (p == 0) weights(fs, ft, fslot, zq, 0, 1, tap + D >= NTAPS);
                    if (p == 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // everything this wave issued up to phase 0 (first read: two stages on)
                        weights(fs, ft, fslot, zq, 1, 2, tap + D >= NTAPS);
#pragma unroll
                        for (int e = 0; e < PPS; ++e)
                            if (tap * PPS + e < kPI) patch_piece(tap * PPS + e, (q + 1) & 1, zq);
                    }
                }
                (void)pnxt;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- MFMA phase: 16 back-to-back MFMAs, nothing else
                // (no explicit wait: the compiler counts -- lgkmcnt(3) before the first MFMA, then one fewer per pixel fragment --
                // so the cluster starts as soon as its first two operands are there)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // LDS returns in order (weights first, then the four pixel fragments): pixel block i is usable once all but the
                    // 3 - i youngest reads have returned
                    if (i == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                    if (i == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    if (i == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                    if (i == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
                        acc[mb + i][j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, Wf[j]), __builtin_bit_cast(bf16x8, Pf[i]), acc[mb + i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
        }
        // end of a tile: the epilogue takes the place of (the start of) this wave's next LOAD phase, i.e. it runs beside the SIMD
        // partner's MFMA phase
        if (sl == nslab - 1) {
            epilogue(xw.first + tk * xw.step);
            ++tk;
        }
        sl = sl1;
        patch_setup(q + 2 < Q ? q + 2 : Q - 1);
    }
    if (half == 0) __builtin_amdgcn_s_barrier();     // same barrier count for every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
    if ((EPI < 0 ? d.dot != nullptr : (EPI & kEpiDot) != 0)) {
        dacc = wave_sum(dacc);
        if (lane == 0) atomicAdd(d.dot, dacc);
    }
}

// Fills the plan; returns 1 when the descriptor is this kernel's case.
int plan3(const XmcConvDesc* d, W3Cfg* t, int* mode, int* wm) {
    if (d->dtype != XMC_BF16 || d->out_dtype != XMC_BF16 || d->src_shift != 0) return 0;
    if (d->CS % 64 != 0 || d->CS > 64 * 64 / 4) return 0;
    if (d->CDw % 256 == 0) *wm = 2;
    else if (d->CDw % 128 == 0) *wm = 4;
    else return 0;
    // 8 x 8 maps (the last discriminator blocks, the generator's second block): a tile is four whole images side by side
    static const bool no_mi = xmc_debug_off("no_wtile3_multi_image");
    const bool mi = !no_mi && d->MW == 8 && d->MH == 8 && d->N % 4 == 0;
    if (!mi && d->MW % 16 != 0) return 0;
    const int TW = (mi || d->MW >= 32) ? 32 : 16, TH = 256 / TW;
    if (!mi && (d->MH % TH != 0 || d->MW % TW != 0)) return 0;
    t->TH = TH; t->TW = TW; t->log2TW = TW == 32 ? 5 : 4;
    t->tiles_y = mi ? 1 : d->MH / TH; t->tiles_x = mi ? 1 : d->MW / TW;
    t->ipt = mi ? 4 : 1;
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
            if (hmin < -TH || hmax > TH || wmin < -TW || wmax > TW) return 0;
            t->dh0[z] = hmin; t->dw0[z] = wmin;
            t->PH[z] = TH + (hmax - hmin); t->PW[z] = TW + (wmax - wmin);
            if (mi) {
                // image k's columns start at patch column 9 k - dw0; one shared zero column between neighbours
                if (wmax - wmin < 1 || wmax - wmin > 2 || wmin > 0 || wmax < 0 || hmin > 0 || hmax < 0) return 0;
                t->PW[z] = 36 + (wmax - wmin == 2 ? 1 : 0);
                // rows below the declared patch are only ever read as zeros: every slot past the patch is zero-filled on each fill, so
                // the bottom halo row may (and, for 3x3, does: 10 x 37 = 370 > 368) reach into the last piece
                if (t->PH[z] * t->PW[z] > kPieces * 8 || (hmax > 0 ? (t->PH[z] - 1) * t->PW[z] : t->PH[z] * t->PW[z]) > (kPieces - 1) * 8) return 0;
            }
            if (d->SH != d->MH || d->SW != d->MW) return 0;
            if (!mi) maxpix = t->PH[z] * t->PW[z] > maxpix ? t->PH[z] * t->PW[z] : maxpix;
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
        t->PH[0] = TH + 1; t->PW[0] = mi ? 36 : TW + 1; t->dh0[0] = t->dw0[0] = 0;
        maxpix = t->PH[0] * t->PW[0];
    } else {
        return 0;
    }
    if (maxpix > (kPieces - 1) * 8) return 0;      // the last piece must stay past the end of every patch (patch_piece)
    if (d->dst_pool && (d->DA != 1 || d->nclass != 1 || (d->DH & 1) || (d->DW & 1))) return 0;
    if (d->res_mode == 2 && d->DA != 1) return 0;
    t->patch_bytes = kPatchSlots * 128;
    if ((int64_t)d->N * d->SH * d->SW * (d->CS / 8) >= (1ll << 31) || (int64_t)d->N * d->DH * d->DW * (d->CD / 8) >= (1ll << 31)) return 0;
    return 1;
}

template <int NTAPS, int MODE, int WM>
int launch3(const XmcConvDesc& d, const W3Cfg& t, hipStream_t st) {
    constexpr int BN = 64 * (8 / WM), RING = WM == 2 ? 2 : 4;
    const size_t lds = (size_t)RING * BN * 128 + 2 * (size_t)t.patch_bytes + (size_t)(2 * XMC_MAX_TAPS + 64) * sizeof(int);
    if (lds > XMC_MAX_DYN_LDS) return XMC_ESHAPE;
    const int ntiles = d.N / t.ipt * t.tiles_y * t.tiles_x, ny = d.CDw / BN;
    int gx = 256 / (ny * d.nclass);               // one 8-wave workgroup per CU, persistent over its tiles
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    gx = xmc_ab_grid(gx);
    // epilogue option sets of the training step as compile-time instantiations (common.h: kEpi*)
    static const bool no_epi = xmc_debug_off("no_wtile_epi");
    const int epi = no_epi ? -1 : xmc_epi_mask(d);
#define XMC_W3_EPI(E)                                                                                                                        \
    if (epi == (E)) {                                                                                                                        \
        XMC_ALLOW_BIG_LDS((wtile3_kernel<NTAPS, MODE, WM, (E)>));                                                                            \
        hipLaunchKernelGGL((wtile3_kernel<NTAPS, MODE, WM, (E)>), dim3((unsigned)gx, (unsigned)ny, (unsigned)d.nclass), dim3(512), lds, st, d, t, ntiles); \
        xmc_note_kernel("wtile3_kernel<%d, %d, %d>", NTAPS, MODE, WM);                                                                       \
        XMC_LAUNCH_CHECK();                                                                                                                  \
        return 0;                                                                                                                            \
    }
    if constexpr (NTAPS == 9) {
        XMC_W3_EPI(kEpiGSum) XMC_W3_EPI(kEpiDKeep) XMC_W3_EPI(kEpiDFwd) XMC_W3_EPI(kEpiDLast) XMC_W3_EPI(kEpiMask) XMC_W3_EPI(0) XMC_W3_EPI(kEpiDLin)
        XMC_W3_EPI(kEpiDKeepS) XMC_W3_EPI(kEpiDLastS) XMC_W3_EPI(kEpiDgDot)
    } else if constexpr (MODE == 1) {
        XMC_W3_EPI(kEpiLrelu) XMC_W3_EPI(0) XMC_W3_EPI(kEpiMask)       // forward; data gradient of the fused upsample conv; MA-GP's linearised forward
    } else {
        XMC_W3_EPI(kEpiRes) XMC_W3_EPI(kEpiBias) XMC_W3_EPI(0)
    }
#undef XMC_W3_EPI
    if (d.sign_bits || d.dot) return 1;          // only the compile-time sets above carry these two; the next kernel in line takes it
    xmc_note_generic_epi(NTAPS == 9 ? "wtile3<9,0>" : MODE == 1 ? "wtile3<4,1>" : "wtile3<4,0>", epi);
    XMC_ALLOW_BIG_LDS((wtile3_kernel<NTAPS, MODE, WM>));
    hipLaunchKernelGGL((wtile3_kernel<NTAPS, MODE, WM>), dim3((unsigned)gx, (unsigned)ny, (unsigned)d.nclass), dim3(512), lds, st, d, t, ntiles);
    xmc_note_kernel("wtile3_kernel<%d, %d, %d>", NTAPS, MODE, WM);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// entry used by xmc_conv_igemm's dispatcher: 0 = launched, 1 = not this kernel's case, < 0 = error
int xmc_conv_wtile3_try(const XmcConvDesc* d, void* stream) {
    W3Cfg t;
    int mode = 0, wm = 2;
    if (!plan3(d, &t, &mode, &wm)) return 1;
    // BN = 128 (64 x 64 wave tiles: the same fragment-read load as the role-split kernel, without its dedicated staging waves) measures
    // 3-6 % behind conv_wtile.hip on the 128-channel layers: kept for A/B runs only
    static const bool bn128 = xmc_debug_off("wtile3_bn128"), bn128_epi = xmc_debug_off("wtile3_bn128_epi"), bn128_plain = xmc_debug_off("wtile3_bn128_plain");
    const bool heavy = d->res || d->dst2 || d->dst_pool || d->mask || d->sign_bits;
    static const bool any_count = xmc_debug_off("wtile3_any_tile_count");      // tests: small batches through this kernel
    const long long tiles = (long long)d->N / t.ipt * t.tiles_y * t.tiles_x * d->nclass;
    if (t.ipt > 1) {
        // 8 x 8 maps: nothing but the generic kernel (~550 TF/s) behind this one.  256-channel tiles when they give one workgroup per
        // CU, else 128-channel tiles (twice the workgroups), else not worth it
        if (wm == 2 && tiles * (d->CDw / 256) < 256 && !any_count) wm = 4;
        if (wm == 4 && tiles * (d->CDw / 128) < 256 && !any_count) return 1;
    } else {
        if (wm == 4 && !(bn128 || (bn128_epi && heavy) || (bn128_plain && !heavy))) return 1;
        // 256-channel tiles halve the number of workgroups: below one tile per CU the role-split kernel (128-channel tiles, twice the
        // workgroups) wins -- 128x128 / batch 64: 10.8 vs 10.4 ms per iteration
        if (wm == 2 && tiles * (d->CDw / 256) < 256 && !any_count) return 1;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
#define W3_GO(NT_, MD_) (wm == 2 ? launch3<NT_, MD_, 2>(*d, t, st) : launch3<NT_, MD_, 4>(*d, t, st))
    if (mode == 1) rc = W3_GO(4, 1);
    else if (d->ntaps == 9) rc = W3_GO(9, 0);
    else rc = W3_GO(4, 0);
#undef W3_GO
    return rc == XMC_ESHAPE ? 1 : rc;
}
