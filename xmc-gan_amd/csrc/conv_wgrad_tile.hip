// Weight gradient for the high-resolution, few-channel layers (Cin <= 64, Cout <= 64, unit stride, bf16).
//
// The generic wgrad kernel gives every (tap, co-tile, ci-tile) its own workgroups, so a 3x3 layer streams both
// activations nine times; at 256x256 that is 2.4 GB per launch for tensors of 0.27 GB (rocprof FETCH_SIZE).
// Here a persistent workgroup walks over 8x32-pixel tiles: dy tile and the x patch (tile + halo) are staged in
// LDS once per tile (next tile prefetched into registers during the MFMAs) and ALL taps are accumulated from
// them into register-resident f32 accumulators (up to 9 x 64 x 64); one atomic add per weight per workgroup at
// the very end.  HBM traffic = each activation read ~1.3x, once.
// MFMA operands are pixel-major in LDS, fragments come from ds_read_b64_tr_b16 (see conv_wgrad.hip).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// tile: 8 x 32 pixels, or (T16) 16 x 16 for maps whose width is 16 (mod 32): a K step of 32 pixels is then two tile rows

struct WTCfg {
    int tiles_y, tiles_x, ntiles;
    int PH, PW, dh0, dw0;
};

// The small-channel variants are pure streaming (a tile is ~2 k cycles of MFMA work behind ~20 KB of loads): cap their
// registers so three workgroups share a CU and one's load latency hides behind the others' tiles (3 -> 32 channels: 0.86 -> 0.4 ms).
template <int NCO, int NCI> struct WtOcc {
    static constexpr int v = (NCO == 2 && NCI == 1) ? 4 : ((NCO * NCI <= 2 || (NCO == 2 && NCI == 2)) ? 3 : (NCO * NCI == 8 ? 2 : 1));     // no spills at these caps
    // (2,4) / (4,2): 252 registers at two workgroups per CU, no spills; one wave per SIMD left the 64 -> 32 layer through the
    // upsample (the largest single weight gradient of the step) at 1.33 ms = 464 TF/s, two reach 0.78 ms = 789
};

// SA == 2 (the 4x4 stride-2 layers, KT = 16 taps): the x patch covers (2*TH+2) x (2*TW+2) source pixels and is stored as two
// column-parity planes per patch row, so that the pixels of consecutive output columns for a fixed tap -- source columns 2k+kw --
// are consecutive LDS rows and the tr16 fragment reads stay unit-stride (and bank-conflict free) exactly as for SA == 1.
// CBW: 16-wide Cout blocks per wave.  Every (ci block, tap) item needs its own shifted x fragment (2 transposing reads); a wave
// that multiplies it against CBW dy fragments instead of one does CBW MFMAs per 2 reads (CBW = 1: 2.1 LDS reads per MFMA, twice
// what the LDS delivers at the MFMA rate).
// DIAG (grouped convolutions whose groups sit inside the diagonal 16x16 channel blocks, df_concept_gan.py:146): only the diagonal
// 64-channel blocks are visited (blockIdx.y; Cin block == Cout block) and an item (ci block, tap) multiplies the ONE Cout block
// with the same index: 1/8 of the dense MFMAs and half the staging.
template <int NCO, int NCI, int NT, int SA = 1, int KT = 9, int CBW = 1, bool DIAG = false, bool T16 = false>     // 16-wide blocks of (padded) Cout and Cin; NT threads
__global__ __launch_bounds__(NT, ((SA == 1 && NT == 256) ? WtOcc<NCO, NCI>::v : 1)) void wgrad_tile_kernel(const XmcConvDesc d, float* __restrict__ dwp, float* __restrict__ dbias, const WTCfg t) {
    constexpr int TH = T16 ? 16 : 8, TW = T16 ? 16 : 32;
    constexpr int KS = TH * TW / 32;                          // K steps of 32 pixels per tile
    static_assert(!T16 || SA == 1, "16 x 16 tiles: unit stride only");
    constexpr int CDP = NCO * 16, CSP = NCI * 16;
    constexpr int YS = CDP * 2 + 32, XS = CSP * 2 + 32;      // LDS row strides (bytes)
    static_assert(NCO % CBW == 0, "co blocks per wave");
    constexpr int NCG = NCO / CBW;                            // groups of co blocks
    constexpr int NS = (NT / 64) / NCG;                       // waves sharing one group of co blocks
    constexpr int MAXI = (NCI * KT + NS - 1) / NS;            // (ci block, tap) items per wave
    constexpr int YCH = CDP / 8, XCH = CSP / 8;               // 16-byte chunks per pixel
    constexpr int YIT = TH * TW * YCH / NT;                   // dy chunks per thread per tile (>= 2)
    constexpr int XIT = ((SA == 1 ? 10 * 34 : 18 * 66) * XCH + NT - 1) / NT;        // patch chunks per thread per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ydy = smem;                                // [256][YS]
    unsigned char* xp = smem + TH * TW * YS;                  // [PH*PW][XS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave % NCG, slice = wave / NCG;
    const int tpi = t.tiles_y * t.tiles_x;
    const int PH = t.PH, PW = t.PW;
    const int cd_units = d.CD / 8, cs_units = d.CS / 8;
    // channels beyond 64 are split over the grid: blockIdx.y = 64-wide Cin block, blockIdx.z = 64-wide Cout block
    const int ci0 = blockIdx.y * 64, co0 = (DIAG ? blockIdx.y : blockIdx.z) * 64;
    const int ciu0 = ci0 >> 3, cou0 = co0 >> 3;
    const u32x4* __restrict__ x16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ y16 = reinterpret_cast<const u32x4*>(d.dst);

    constexpr int ACB = DIAG ? 1 : CBW;                      // accumulator blocks per item
    // (same-box ablation, 64 x 64 blocks at batch 512: BD 1 / 2 / 3 with AF2 = 1030 / 1086 / 1014 TF/s, BD 1 without AF2 1040: the K loop
    // is no longer what a tile waits for.  Of a tile's 4.3 us, 2.4 us are the global loads' latency when nothing else runs (they
    // hide behind the K loop except ~0.5 us) and 3.75 us are K loop + LDS stores + the two barriers with the loads taken out.)
    constexpr int BD = 2;                                     // x fragments in flight ahead of the MFMAs that use them
    constexpr bool AF2 = NT == 512 && SA == 1 && !DIAG;                              // dy fragments of the next K step in a second register set
    // staging tables (below) where their ~XIT + 6 registers fit; the 4-wave variants that already sit at their register cap
    // (two workgroups per CU: the other one's MFMAs cover this one's address arithmetic) keep computing addresses per tile
    constexpr bool TAB = !(NT == 256 && SA == 1 && NCO >= 2 && NCO * NCI >= 4);
    f32x4 acc[MAXI][ACB];
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
#pragma unroll
        for (int c = 0; c < ACB; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging tables, computed ONCE: everything about a staged 16-byte unit that does not depend on the tile -- its source offset
    // from the tile's patch origin, the image borders it would cross (4 bits), whether this thread stages it at all.  Per tile
    // the loads are then unconditional (a unit outside the image reads the tile's own first pixel and is zeroed when it goes to
    // LDS): one add per load instead of two runtime divisions, four compares and a branch, which all eight waves ran in front
    // of the MFMA loop of every tile with the matrix pipe idle.
    const int ych = tid % YCH, xch = tid % XCH;               // NT % YCH == 0: a thread always holds the same channel chunk
    const bool yok = cou0 + ych < cd_units;
    const int sh = d.src_shift;
    unsigned xoff[TAB ? XIT : 1];
    unsigned xhalo = 0, xin = 0;                              // 4 halo bits per unit (XIT <= 8) / "stages this unit"
    static_assert(XIT <= 16, "halo bits");
    unsigned xhalo2 = 0;
    int xl[SA == 2 ? XIT : 1];
    // dy: unit `it` of a thread is (NT / YCH) / TW tile rows below unit 0
    static_assert((NT / YCH) % TW == 0, "dy units of a thread differ by whole tile rows");
    const unsigned yoff0 = yok ? (unsigned)((((tid / YCH) / TW) * d.MW + ((tid / YCH) % TW)) * cd_units + cou0 + ych) : 0u;
    const unsigned ystep = yok ? (unsigned)(((NT / YCH) / TW) * d.MW * cd_units) : 0u;
    const int shs = d.SH << sh, sws = d.SW << sh;
    const int row0 = t.dh0 >> sh, col0 = t.dw0 >> sh;         // patch origin at the stored resolution (floor)
    const unsigned xsafe = (unsigned)(((0 - row0) * d.SW + (0 - col0)) * cs_units);       // the tile's own first pixel, chunk 0
    if constexpr (TAB)
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int pp = (tid + it * NT) / XCH;
        const int py = pp / PW, px = pp - py * PW;
        const bool in = pp < PH * PW && ciu0 + xch < cs_units;
        xin |= in ? (1u << it) : 0u;
        xoff[it] = in ? (unsigned)(((((t.dh0 + py) >> sh) - row0) * d.SW + (((t.dw0 + px) >> sh) - col0)) * cs_units + ciu0 + xch) : xsafe;
        // outside the image: above (first tile row only), below (last tile row only), left, right
        const unsigned hb = !in ? 0u : ((t.dh0 + py < 0 ? 1u : 0u) | (t.dh0 + py >= shs - (t.tiles_y - 1) * TH * SA ? 2u : 0u) |
                                        (t.dw0 + px < 0 ? 4u : 0u) | (t.dw0 + px >= sws - (t.tiles_x - 1) * TW * SA ? 8u : 0u));
        if (it < 8) xhalo |= hb << (4 * it); else xhalo2 |= hb << (4 * (it - 8));
        if constexpr (SA == 2) xl[it] = ((py * 2 + (px & 1)) * (PW >> 1) + (px >> 1)) * XS + xch * 16;
    }

    u32x4 yv[YIT], xv[XIT];
    unsigned xzero = 0;                                       // units of the prefetched tile that lie outside the image
    auto prefetch = [&](int tile) {
        if constexpr (!TAB) {
            const int img = tile / tpi, trem = tile - img * tpi;
            const int a0 = (trem / t.tiles_x) * TH, b0 = (trem % t.tiles_x) * TW;
#pragma unroll
            for (int it = 0; it < YIT; ++it) {
                int id = tid + it * NT;
                int pix = id / YCH, ch = id - pix * YCH;
                int py = pix / TW, px = pix - py * TW;
                u32x4 z = {0, 0, 0, 0};
                yv[it] = cou0 + ch < cd_units ? y16[(((size_t)img * d.MH + a0 + py) * d.MW + b0 + px) * cd_units + cou0 + ch] : z;
            }
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                int id = tid + it * NT;
                int pp = id / XCH, ch = id - pp * XCH;
                int py = pp / PW, px = pp - py * PW;
                int sy = a0 * SA + t.dh0 + py, sx = b0 * SA + t.dw0 + px;          // coordinates at the (possibly x2-upsampled) resolution
                bool ok = pp < PH * PW && ciu0 + ch < cs_units && (unsigned)sy < (unsigned)(d.SH << d.src_shift) &&
                          (unsigned)sx < (unsigned)(d.SW << d.src_shift);
                u32x4 z = {0, 0, 0, 0};
                xv[it] = ok ? x16[(((size_t)img * d.SH + (sy >> d.src_shift)) * d.SW + (sx >> d.src_shift)) * cs_units + ciu0 + ch] : z;
            }
            return;
        }
        // (uniform, but an integer division leaves its result in a VGPR: without the readfirstlane every address below is built
        // per lane in 64 bits instead of SGPR base + 32-bit lane offset)
        const int img = __builtin_amdgcn_readfirstlane(tile / tpi), trem = tile - img * tpi;
        const int ty = __builtin_amdgcn_readfirstlane(trem / t.tiles_x), tx = trem - ty * t.tiles_x;
        const int a0 = ty * TH, b0 = tx * TW;
        const u32x4* __restrict__ yb = y16 + (((size_t)img * d.MH + a0) * d.MW + b0) * cd_units;
#pragma unroll
        for (int it = 0; it < YIT; ++it) yv[it] = yb[yoff0 + it * ystep];
        // halo depth <= tile size: a row / column of the patch is outside the image only for tiles on that border
        const unsigned border = (ty == 0 ? 1u : 0u) | (ty == t.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == t.tiles_x - 1 ? 8u : 0u);
        const long long xbase = (((long long)img * d.SH + (((a0 * SA) >> sh) + row0)) * d.SW + (((b0 * SA) >> sh) + col0)) * cs_units;
        const u32x4* __restrict__ xb = x16 + xbase;
        xzero = 0;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const unsigned hb = it < 8 ? (xhalo >> (4 * it)) & 15u : (xhalo2 >> (4 * (it - 8))) & 15u;
            const bool out = (hb & border) != 0;
            xv[it] = xb[out ? xsafe : xoff[it]];
            xzero |= out ? (1u << it) : 0u;
        }
    };

    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const XcdWalk xw = xmc_xcd_walk(t.ntiles);
    int tile = xw.first;
    if (tile < xw.end) prefetch(tile);
    const int fr = lane & 15, fg = lane >> 4;
    const int q = fr >> 2, pp4 = fr & 3;
    const int nitems = NCI * d.ntaps;
    // per-item LDS byte offset of the shifted x fragment, computed ONCE: indexing the tap tables of the kernel-argument
    // struct with a run-time tap inside the MFMA loop makes the compiler fetch them with vector memory loads
    // (rocprof: 223 VMEM reads per wave per tile instead of 10, SQ_WAIT_ANY 69 %)
    int itoff[MAXI], itib[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int item = slice + j * NS;
        int ib = 0, tap = 0;
        if (item < nitems) { ib = item / d.ntaps; tap = item - ib * d.ntaps; }
        itib[j] = ib;
        const int th = d.dh[0][tap] - t.dh0, tw = d.dw[0][tap] - t.dw0;
        itoff[j] = (SA == 1 ? th * PW + tw : (th * 2 + (tw & 1)) * (PW >> 1) + (tw >> 1)) * XS + (ib * 16) * 2;
    }
    // this wave's share of the (ci block, tap) items: j < nv (wave-uniform, fixed for the launch)
    int nv = (nitems - slice + NS - 1) / NS;
    nv = nv < 0 ? 0 : (nv > MAXI ? MAXI : nv);
    nv = __builtin_amdgcn_readfirstlane(nv);
    const unsigned char* afrag = ydy + (size_t)(4 * fg + q) * YS + (cg * CBW * 16 + 4 * pp4) * 2;
    const unsigned char* bfrag = xp + (size_t)(4 * fg + q) * XS + (4 * pp4) * 2;
    // the fragment bases the K loop uses, re-derived per tile through an opaque zero: left loop-invariant, all KS x items read
    // addresses are hoisted out of the tile loop and live across it (78 spilled registers)
    const unsigned char *afr = afrag, *bfr = bfrag;

    // K loop of one tile for a wave with NV items, branch-free and software-pipelined: the two transposing reads of item s+1 (and,
    // once per K step, the dy fragments of step r+1) are issued in front of the MFMAs of item s and pinned there.  (The loop this
    // replaces tested `item < nitems` per item: every item became a basic block of its own -- 2 reads, lgkmcnt(0), 4 MFMAs --
    // so a wave alternated between waiting on the LDS and feeding the matrix pipe.)
    auto kloop = [&](auto nvc) {
        constexpr int NV = decltype(nvc)::value;
        constexpr int NSTEP = KS * NV;
        bf16x8 af[AF2 ? 2 : 1][CBW], bq[BD + 1];
        auto rd_a = [&](int r, bf16x8* a) {
#pragma unroll
            for (int c = 0; c < CBW; ++c) {
                const unsigned char* ab = afr + (size_t)(r * 32) * YS + c * 32;
                bf16x4 alo = xmc_ds_read_tr16((ab));
                bf16x4 ahi = xmc_ds_read_tr16((ab + 16 * YS));
                a[c] = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            }
        };
        auto rd_b = [&](int r, int j) -> bf16x8 {
            const unsigned char* bb = bfr + (SA == 1 ? r * (32 / TW) * PW : r * 2 * PW) * XS + itoff[j];
            bf16x4 blo = xmc_ds_read_tr16((bb));
            bf16x4 bhi = xmc_ds_read_tr16((bb + (T16 ? PW : 16) * XS));
            return bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        };
        rd_a(0, af[0]);
#pragma unroll
        for (int k = 0; k < BD; ++k)
            if (k < NSTEP) bq[k] = rd_b(k / NV, k % NV);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const int ab = AF2 ? (r & 1) : 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int s = r * NV + j;
                // the x fragment of item s + BD goes into the ring slot item s - 1 has just released
                if (s + BD < NSTEP) bq[(s + BD) % (BD + 1)] = rd_b((s + BD) / NV, (s + BD) % NV);
                if (AF2 && j == 0 && r + 1 < KS) rd_a(r + 1, af[AF2 ? (r + 1) & 1 : 0]);
#pragma unroll
                for (int c = 0; c < CBW; ++c) acc[j][c] = XMC_MFMA_16x16x32(af[ab][c], bq[s % (BD + 1)], acc[j][c], 0, 0, 0);
                if (!AF2 && j == NV - 1 && r + 1 < KS) rd_a(r + 1, af[0]);     // single buffer: behind the step's last MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // One copy of the whole tile loop per K-loop variant, chosen once: with the variants as branches INSIDE the loop the
    // accumulators of the three paths meet in phi nodes the register allocator did not coalesce (68 extra registers, spills).
    auto tiles = [&](auto variant) {
    constexpr int VNV = decltype(variant)::value;            // items per wave known at compile time; 0 = generic loop
    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();                                      // previous tile's reads are done
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            const int id = tid + it * NT;
            u32x4 v = yv[it];
            if (!yok) v = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(ydy + (id / YCH) * YS + ych * 16) = v;
            if (dbias != nullptr && blockIdx.y == 0) {        // bias gradient: this thread always holds chunk tid % YCH
                bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += (float)h[k];
            }
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            if constexpr (!TAB) {
                const int pp = (tid + it * NT) / XCH;
                if (pp < PH * PW) *reinterpret_cast<u32x4*>(xp + pp * XS + xch * 16) = xv[it];
            } else {
                u32x4 v = xv[it];
                if ((xzero >> it) & 1) v = u32x4{0, 0, 0, 0};
                const int lo = SA == 2 ? xl[SA == 2 ? it : 0] : ((tid + it * NT) / XCH) * XS + xch * 16;
                if ((xin >> it) & 1) *reinterpret_cast<u32x4*>(xp + lo) = v;
            }
        }
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);

        // K loop: 32 pixels per step (one tile row; two rows of a 16 x 16 tile: the upper 16 pixels of a fragment are then one
        // patch row further down instead of 16 columns to the right)
        {
            int zq = 0;
            asm volatile("" : "+v"(zq));
            afr = afrag + zq; bfr = bfrag + zq;
        }
        if constexpr (VNV > 0) {
            kloop(std::integral_constant<int, VNV>{});
        } else
        for (int r = 0; r < KS; ++r) {
            // A' fragment: dy^T [co = cb*16 + lane&15][pix = r*32 + 4*fg + j (+16)]: the 32 lanes one tr16 read serves together
            // address 8 consecutive pixel rows, which the 96/160-byte strides spread over distinct banks (rows 8*fg + j would
            // put lane groups 0 and 1 on the same banks)
            bf16x8 af[CBW];
#pragma unroll
            for (int c = 0; c < (DIAG ? 0 : CBW); ++c) {
                const unsigned char* ab = ydy + (size_t)(r * 32 + 4 * fg + q) * YS + ((cg * CBW + c) * 16 + 4 * pp4) * 2;
                bf16x4 alo = xmc_ds_read_tr16((ab));
                bf16x4 ahi = xmc_ds_read_tr16((ab + 16 * YS));
                af[c] = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            }
#pragma unroll
            for (int j = 0; j < MAXI; ++j) {
                if (slice + j * NS < nitems) {                                 // wave-uniform
                    const unsigned char* bb = xp + ((SA == 1 ? r * (32 / TW) * PW : r * 2 * PW) + 4 * fg + q) * XS + itoff[j] + (4 * pp4) * 2;
                    bf16x4 blo = xmc_ds_read_tr16((bb));
                    bf16x4 bhi = xmc_ds_read_tr16((bb + (T16 ? PW : 16) * XS));
                    const bf16x8 bf = bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                    if (DIAG) {                               // the dy fragment of the Cout block with this item's index
                        const unsigned char* ab = ydy + (size_t)(r * 32 + 4 * fg + q) * YS + (itib[j] * 16 + 4 * pp4) * 2;
                        bf16x4 alo = xmc_ds_read_tr16((ab));
                        bf16x4 ahi = xmc_ds_read_tr16((ab + 16 * YS));
                        const bf16x8 a1 = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                        acc[j][0] = XMC_MFMA_16x16x32(a1, bf, acc[j][0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int c = 0; c < CBW; ++c) acc[j][c] = XMC_MFMA_16x16x32(af[c], bf, acc[j][c], 0, 0, 0);
                    }
                }
            }
        }
    }
    };
    // the pipelined K loop where its registers fit: the 8-wave 64 x 64 variants (one workgroup per CU, 256 registers per lane);
    // the occupancy-capped small-channel variants keep the generic loop and hide LDS latency behind their other workgroups
    constexpr bool PIPE = !DIAG && NT == 512 && SA == 1;
    if (PIPE && nv == MAXI) tiles(std::integral_constant<int, PIPE ? MAXI : 0>{});
    else if (PIPE && MAXI > 1 && nv == MAXI - 1) tiles(std::integral_constant<int, (PIPE && MAXI > 1) ? MAXI - 1 : 0>{});
    else tiles(std::integral_constant<int, 0>{});

    if (dbias != nullptr && blockIdx.y == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = bsum[k];
            for (int o = 32; o >= YCH; o >>= 1) v += __shfl_xor(v, o, 64);
            int ch = co0 + (lane % YCH) * 8 + k;
            if (lane < YCH && ch < d.CD) atomicAdd(&dbias[(blockIdx.x & (XMC_BIAS_REPLICAS - 1)) * d.CD + ch], v);
        }
    }
    // D[row = co][col = ci] -> one atomic per element per workgroup
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int item = slice + j * NS;
        if (item < nitems) {
            const int ib = item / d.ntaps, tap = item - ib * d.ntaps;
#pragma unroll
            for (int c = 0; c < ACB; ++c)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int co = co0 + (DIAG ? ib : cg * CBW + c) * 16 + fg * 4 + rr, ci = ci0 + ib * 16 + fr;
                if (co < d.CDw && co < co0 + CDP && ci < d.CS)
                    atomicAdd(&dwp[((size_t)d.wi[0][tap] * d.CDw + co) * d.CS + ci], acc[j][c][rr]);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Row-reuse form of the all-taps kernel for the full 3x3 tap set on 8 x 32 tiles (round 4).  The kernel above measures LDS-BOUND:
// per tile its eight waves issue 1152 transposing reads (590 KB = 2.3 us at 128 B/clk) for 1.25 us of MFMAs, because every
// (Cin block, tap) item reads its own shifted x fragment in every K step.  But K step r (tile row r) with tap row kh reads patch
// row r + kh: of the three patch rows a (Cin block, kw) GROUP needs in step r, two were already read in step r - 1.  So a wave owns
// whole groups -- the three kh taps of a (Cin block, kw) pair -- keeps their three row fragments in registers (rotating) and reads
// ONE new row fragment per group and step: 3 x fragments + CBW dy fragments for 9 * CBW MFMAs (0.56 reads per MFMA at CBW = 2,
// 0.9 above), 736 reads per tile.  Waves = NCG Cout groups x (8 / NCG) group slices, three groups = nine items per wave.
// Same staging (tables, unconditional loads, one tile prefetched into registers), same LDS images, same atomics at the end.
template <int NCO, int NCI, int NCG>
__global__ __launch_bounds__(512) void wgrad_tile_rr_kernel(const XmcConvDesc d, float* __restrict__ dwp, float* __restrict__ dbias, const WTCfg t) {
    constexpr int NT = 512, TH = 8, TW = 32, KS = 8, PH = 10, PW = 34;
    constexpr int CDP = NCO * 16, CSP = NCI * 16;
    constexpr int YS = CDP * 2 + 32, XS = CSP * 2 + 32;
    constexpr int CBW = NCO / NCG, NS = 8 / NCG;
    constexpr int GPW = NCI * 3 / NS;                          // (Cin block, kw) groups per wave
    static_assert(GPW * NS == NCI * 3 && CBW * NCG == NCO, "waves tile the (group, Cout block) grid");
    constexpr int MAXI = GPW * 3;
    constexpr int YCH = CDP / 8, XCH = CSP / 8;
    constexpr int YIT = TH * TW * YCH / NT;
    constexpr int XIT = (PH * PW * XCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ydy = smem;                                // [256][YS]
    unsigned char* xp = smem + TH * TW * YS;                  // [PH * PW][XS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave % NCG, slice = wave / NCG;
    const int tpi = t.tiles_y * t.tiles_x;
    const int cd_units = d.CD / 8, cs_units = d.CS / 8;
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    const int ciu0 = ci0 >> 3, cou0 = co0 >> 3;
    const u32x4* __restrict__ x16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ y16 = reinterpret_cast<const u32x4*>(d.dst);
    const int sh = d.src_shift;

    f32x4 acc[MAXI][CBW];
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ych = tid % YCH, xch = tid % XCH;
    const bool yok = cou0 + ych < cd_units;
    static_assert((NT / YCH) % TW == 0, "dy units of a thread differ by whole tile rows");
    const unsigned yoff0 = yok ? (unsigned)((((tid / YCH) / TW) * d.MW + ((tid / YCH) % TW)) * cd_units + cou0 + ych) : 0u;
    const unsigned ystep = yok ? (unsigned)(((NT / YCH) / TW) * d.MW * cd_units) : 0u;
    const int shs = d.SH << sh, sws = d.SW << sh;
    const int row0 = -1 >> sh, col0 = -1 >> sh;               // patch origin (tile - 1) at the stored resolution (floor)
    const unsigned xsafe = (unsigned)(((0 - row0) * d.SW + (0 - col0)) * cs_units);
    unsigned xoff[XIT];
    unsigned xhalo = 0, xin = 0;
    static_assert(XIT <= 8, "halo bits");
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int pp = (tid + it * NT) / XCH;
        const int py = pp / PW, px = pp - py * PW;
        const bool in = pp < PH * PW && ciu0 + xch < cs_units;
        xin |= in ? (1u << it) : 0u;
        xoff[it] = in ? (unsigned)(((((py - 1) >> sh) - row0) * d.SW + (((px - 1) >> sh) - col0)) * cs_units + ciu0 + xch) : xsafe;
        const unsigned hb = !in ? 0u : ((py - 1 < 0 ? 1u : 0u) | (py - 1 >= shs - (t.tiles_y - 1) * TH ? 2u : 0u) |
                                        (px - 1 < 0 ? 4u : 0u) | (px - 1 >= sws - (t.tiles_x - 1) * TW ? 8u : 0u));
        xhalo |= hb << (4 * it);
    }
    u32x4 yv[YIT], xv[XIT];
    unsigned char yb8[YIT];                                   // sign bytes of the dy units (XmcConvDesc.mask_bits: dy is a masked gradient)
    const unsigned char* __restrict__ bits8 = reinterpret_cast<const unsigned char*>(d.mask_bits);
    unsigned xzero = 0;
    auto prefetch = [&](int tile) {
        const int img = __builtin_amdgcn_readfirstlane(tile / tpi), trem = tile - img * tpi;
        const int ty = __builtin_amdgcn_readfirstlane(trem / t.tiles_x), tx = trem - ty * t.tiles_x;
        const int a0 = ty * TH, b0 = tx * TW;
        const size_t ybase = (((size_t)img * d.MH + a0) * d.MW + b0) * cd_units;
        const u32x4* __restrict__ yb = y16 + ybase;
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            yv[it] = yb[yoff0 + it * ystep];
            if (bits8) yb8[it] = bits8[ybase + yoff0 + it * ystep];
        }
        const unsigned border = (ty == 0 ? 1u : 0u) | (ty == t.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == t.tiles_x - 1 ? 8u : 0u);
        const long long xbase = (((long long)img * d.SH + ((a0 >> sh) + row0)) * d.SW + ((b0 >> sh) + col0)) * cs_units;
        const u32x4* __restrict__ xb = x16 + xbase;
        xzero = 0;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const bool out = (((xhalo >> (4 * it)) & 15u) & border) != 0;
            xv[it] = xb[out ? xsafe : xoff[it]];
            xzero |= out ? (1u << it) : 0u;
        }
    };

    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;
    const XcdWalk xw = xmc_xcd_walk(t.ntiles);
    int tile = xw.first;
    if (tile < xw.end) prefetch(tile);
    const int fr = lane & 15, fg = lane >> 4;
    const int q = fr >> 2, pp4 = fr & 3;
    // group gi of this wave: g = slice + gi * NS -> Cin block g / 3, tap column g % 3; LDS byte offset of its row-0 fragment
    int goff[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int g = slice + gi * NS;
        goff[gi] = (g % 3) * XS + (g / 3) * 32;
    }
    const unsigned char* afrag = ydy + (size_t)(4 * fg + q) * YS + (cg * CBW * 16 + 4 * pp4) * 2;
    const unsigned char* bfrag = xp + (size_t)(4 * fg + q) * XS + (4 * pp4) * 2;
    const unsigned char *afr = afrag, *bfr = bfrag;

    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();                                      // previous tile's reads are done
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            u32x4 v = yv[it];
            if (bits8) { v = xmc_apply_sign_bits(v, yb8[it]); yv[it] = v; }
            if (!yok) v = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(ydy + ((tid + it * NT) / YCH) * YS + ych * 16) = v;
        }
        if (do_bias) {
#pragma unroll
            for (int it = 0; it < YIT; ++it) {
                const bf16x8 h = __builtin_bit_cast(bf16x8, yv[it]);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += yok ? (float)h[k] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            u32x4 v = xv[it];
            if ((xzero >> it) & 1) v = u32x4{0, 0, 0, 0};
            if ((xin >> it) & 1) *reinterpret_cast<u32x4*>(xp + ((tid + it * NT) / XCH) * XS + xch * 16) = v;
        }
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        {
            int zq = 0;
            asm volatile("" : "+v"(zq));
            afr = afrag + zq; bfr = bfrag + zq;
        }
        bf16x8 af[2][CBW], xf[GPW][3];
        auto rd_a = [&](int r, bf16x8* a) {
#pragma unroll
            for (int c = 0; c < CBW; ++c) {
                const unsigned char* ab = afr + (size_t)(r * 32) * YS + c * 32;
                bf16x4 alo = xmc_ds_read_tr16((ab));
                bf16x4 ahi = xmc_ds_read_tr16((ab + 16 * YS));
                a[c] = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            }
        };
        auto rd_x = [&](int gi, int prow) -> bf16x8 {          // patch row `prow` of group gi, columns kw .. kw + 31
            const unsigned char* bb = bfr + (prow * PW) * XS + goff[gi];
            bf16x4 blo = xmc_ds_read_tr16((bb));
            bf16x4 bhi = xmc_ds_read_tr16((bb + 16 * XS));
            return bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        };
        rd_a(0, af[0]);
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            xf[gi][0] = rd_x(gi, 0);
            xf[gi][1] = rd_x(gi, 1);
        }
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            // the new patch row (r + 2) of every group and the dy fragments of step r + 1 are requested in front of the MFMAs on the
            // two rows that are already in registers (kh = 0, 1); the MFMAs on the new row (kh = 2) come last
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int gi = 0; gi < GPW; ++gi) {
                    if (kh == 0) xf[gi][(r + 2) % 3] = rd_x(gi, r + 2);
                    if (kh == 1 && gi == 0 && r + 1 < KS) rd_a(r + 1, af[(r + 1) & 1]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c)
                        acc[gi * 3 + kh][c] = XMC_MFMA_16x16x32(af[r & 1][c], xf[gi][(r + kh) % 3], acc[gi * 3 + kh][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    if (do_bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = bsum[k];
            for (int o = 32; o >= YCH; o >>= 1) v += __shfl_xor(v, o, 64);
            int ch = co0 + (lane % YCH) * 8 + k;
            if (lane < YCH && ch < d.CD) atomicAdd(&dbias[(blockIdx.x & (XMC_BIAS_REPLICAS - 1)) * d.CD + ch], v);
        }
    }
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int g = slice + gi * NS, ib = g / 3, kw = g % 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float* __restrict__ sl = dwp + (size_t)d.wi[0][kh * 3 + kw] * d.CDw * d.CS;
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int co = co0 + (cg * CBW + c) * 16 + fg * 4 + rr, ci = ci0 + ib * 16 + fr;
                    if (co < d.CDw && co < co0 + CDP && ci < d.CS) atomicAdd(&sl[(size_t)co * d.CS + ci], acc[gi * 3 + kh][c][rr]);
                }
        }
    }
}

template <int NCO, int NCI, int NCG>
int launch_wt_rr(const XmcConvDesc& d, float* dwp, float* dbias, const WTCfg& t, hipStream_t st) {
    constexpr int YS = NCO * 32 + 32, XS = NCI * 32 + 32;
    const size_t lds = (size_t)256 * YS + (size_t)10 * 34 * XS;
    if (lds > XMC_MAX_DYN_LDS) return 1;
    XMC_ALLOW_BIG_LDS((wgrad_tile_rr_kernel<NCO, NCI, NCG>));
    const int ny = (d.CS + 63) / 64, nz = (d.CD + 63) / 64;
    int gx = 256 / (ny * nz);
    if (gx < 1) gx = 1;
    if (gx > t.ntiles) gx = t.ntiles;
    hipLaunchKernelGGL((wgrad_tile_rr_kernel<NCO, NCI, NCG>), dim3(xmc_ab_grid(gx), ny, nz), dim3(512), lds, st, d, dwp, dbias, t);
    xmc_note_kernel("wgrad_tile_rr_kernel<%d, %d, %d>", NCO, NCI, NCG);
    XMC_LAUNCH_CHECK();
    return 0;
}

template <int NCO, int NCI, int NT = 256, int SA = 1, int KT = 9, int CBW = 1, bool DIAG = false, bool T16 = false>
int launch_wt(const XmcConvDesc& d, float* dwp, float* dbias, const WTCfg& t, hipStream_t st) {
    constexpr int YS = NCO * 32 + 32, XS = NCI * 32 + 32;
    size_t lds = (size_t)256 * YS + (size_t)t.PH * t.PW * XS;
    if (lds > XMC_MAX_DYN_LDS) return 1;
    XMC_ALLOW_BIG_LDS((wgrad_tile_kernel<NCO, NCI, NT, SA, KT, CBW, DIAG, T16>));
    int per_cu = (int)(160 * 1024 / lds);
    const int cap = WtOcc<NCO, NCI>::v >= 3 ? 3 : 2;
    if (per_cu > cap) per_cu = cap;
    if (NT == 512) per_cu = 1;                                // 8 waves at ~200 registers: one workgroup per CU
    const int ny = (d.CS + 63) / 64, nz = DIAG ? 1 : (d.CD + 63) / 64;
    int gx = 256 * per_cu / (ny * nz);
    if (gx < 1) gx = 1;
    if (gx > t.ntiles) gx = t.ntiles;
    hipLaunchKernelGGL((wgrad_tile_kernel<NCO, NCI, NT, SA, KT, CBW, DIAG, T16>), dim3(xmc_ab_grid(gx), ny, nz), dim3(NT), lds, st, d, dwp, dbias, t);
    xmc_note_kernel(T16 ? "wgrad_tile_kernel<%d, %d, %d, %d, %d, %d, false, true>" : DIAG ? "wgrad_tile_kernel<%d, %d, %d, %d, %d, %d, true>" : "wgrad_tile_kernel<%d, %d, %d, %d, %d, %d>", NCO, NCI, NT, SA, KT, CBW);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

static int wgrad_tile_go(const XmcConvDesc* d, float* dwp, float* dbias, void* stream, bool rr_only);

// 0 = launched, 1 = not eligible (caller falls back to the generic kernel), other = error
int xmc_conv_wgrad_tile_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream) { return wgrad_tile_go(d, dwp, dbias, stream, false); }

// weight gradient whose dy operand is masked by sign bytes while it is staged (XmcConvDesc.mask_bits): the row-reuse kernel only
extern "C" int xmc_conv_wgrad_bits(const XmcConvDesc* d, float* dwp, void* stream) {
    if (!d || !d->src || !d->dst || !dwp || !d->mask_bits) return XMC_EINVAL;
    static const bool off = xmc_debug_off("no_stage_bits");
    if (off || d->src_shift != 0 || d->SA != 1) return 1;
    return wgrad_tile_go(d, dwp, nullptr, stream, true);
}

static int wgrad_tile_go(const XmcConvDesc* d, float* dwp, float* dbias, void* stream, bool rr_only) {
    static const bool off = xmc_debug_off("no_wtile");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || d->src_shift < 0 || d->src_shift > 1) return 1;
    const bool s2 = d->SA == 2;                               // 4x4 stride-2 layers: 16 taps, no upsample, Cin <= 32 (patch size)
    if (d->SA != 1 && !(s2 && d->ntaps == 16 && d->src_shift == 0 && d->CS <= 32)) return 1;
    static const bool no_wide = xmc_debug_off("no_wt_wide");
    const bool wide = d->CD > 64 || d->CS > 64;              // 64-channel blocks over grid.y / grid.z
    if (wide && (no_wide || s2 || (d->CD > 64 && d->CD % 64) || (d->CS > 64 && d->CS % 64))) return 1;
    if (d->ntaps > (s2 ? 16 : 9) || d->ntaps < 1) return 1;
    static const bool no_t16 = xmc_debug_off("no_wt_t16");
    // 16 x 16 tiles for 3x3 layers on 16-pixel-wide maps (512 -> 512, batch 512: 1.00 ms on the row kernel, 0.60 ms = 1033 TF/s
    // here; a 1x1 layer has no tap re-use to gain from and stays on the split-K kernel)
    const bool t16 = d->MW % 32 != 0 && d->MW % 16 == 0 && d->MH % 16 == 0 && !s2 && d->ntaps >= 4 && !no_t16;
    const int TH = t16 ? 16 : 8, TW = t16 ? 16 : 32;
    if (d->MW % TW != 0 || d->MH % TH != 0) return 1;
    if (d->CD % 8 != 0 || d->CS % 8 != 0) return 1;
    WTCfg t;
    t.tiles_y = d->MH / TH; t.tiles_x = d->MW / TW; t.ntiles = d->N * t.tiles_y * t.tiles_x;
    int hmin = 127, hmax = -128, wmin = 127, wmax = -128;
    for (int k = 0; k < d->ntaps; ++k) {
        int h = d->dh[0][k], w = d->dw[0][k];
        hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax;
        wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax;
    }
    t.dh0 = hmin; t.dw0 = wmin;
    t.PH = d->SA * (TH - 1) + (hmax - hmin + 1); t.PW = d->SA * (TW - 1) + (wmax - wmin + 1);
    if (s2 ? (t.PH > 18 || t.PW > 66 || (t.PW & 1)) : t16 ? (t.PH > 18 || t.PW > 18) : (t.PH > 10 || t.PW > 34)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nco = d->CD <= 16 ? 1 : (d->CD <= 32 ? 2 : 4);
    const int nci = d->CS <= 16 ? 1 : (d->CS <= 32 ? 2 : 4);
    if (rr_only && (s2 || t16)) return 1;
    if (s2) {
        if (nco == 4 && nci == 2) return launch_wt<4, 2, 512, 2, 16, 4>(*d, dwp, dbias, t, st);
        if (nco == 2 && nci == 1) return launch_wt<2, 1, 256, 2, 16, 2>(*d, dwp, dbias, t, st);
        return 1;
    }
    if (t16) {        // the wide layers on 16-pixel-wide maps (the row kernel otherwise: 540-600 TF/s)
        if (nco == 4 && nci == 4) return launch_wt<4, 4, 512, 1, 9, 4, false, true>(*d, dwp, dbias, t, st);
        return 1;
    }
    // the full 3x3 tap set in row-major order on 8 x 32 tiles: the row-reuse form (0.56 LDS reads per MFMA instead of 0.9)
    static const bool no_rr = xmc_debug_off("no_wt_rr");
    bool std33 = d->ntaps == 9 && t.PH == 10 && t.PW == 34 && d->groups <= 1;
    for (int k = 0; k < 9 && std33; ++k) std33 = d->dh[0][k] == k / 3 - 1 && d->dw[0][k] == k % 3 - 1;
    if (!no_rr && std33) {
        if (nco == 4 && nci == 4) return launch_wt_rr<4, 4, 2>(*d, dwp, dbias, t, st);
        if (nco == 2 && nci == 4) return launch_wt_rr<2, 4, 2>(*d, dwp, dbias, t, st);
        if (nco == 4 && nci == 2) return launch_wt_rr<4, 2, 4>(*d, dwp, dbias, t, st);
    }
    if (rr_only) return 1;
#define WT_CASE(a, b) if (nco == a && nci == b) return launch_wt<a, b, 256, 1, 9, (a >= 2 ? 2 : 1)>(*d, dwp, dbias, t, st);
    // (4,4) = 64x64 channels with 4 waves needs 144 accumulator + 76 staging registers per lane and measured slower than
    // the split-K kernel (166 vs 230 TF/s); it runs with 8 waves instead (below)
    // 64 -> 32 / 32 -> 64 channels as 8 waves with the pipelined K loop (A/B: XMC_DEBUG_DISPATCH=no_wt24_8w keeps the 4-wave forms)
    static const bool no_8w = xmc_debug_off("no_wt24_8w");
    if (!no_8w && nco == 2 && nci == 4) return launch_wt<2, 4, 512, 1, 9, 2>(*d, dwp, dbias, t, st);
    if (!no_8w && nco == 4 && nci == 2) return launch_wt<4, 2, 512, 1, 9, 4>(*d, dwp, dbias, t, st);
    WT_CASE(1, 2) WT_CASE(1, 4) WT_CASE(2, 1) WT_CASE(2, 2) WT_CASE(2, 4) WT_CASE(4, 1) WT_CASE(4, 2)
    // grouped layer with its groups inside the diagonal 16x16 blocks (8 -> 8 channels per group): diagonal blocks only
    static const bool no_diag = xmc_debug_off("no_wt_diag");
    if (!no_diag && d->groups > 1 && nco == 4 && nci == 4 && d->CD == d->CS && d->CD % 64 == 0 && d->CD % d->groups == 0 &&
        16 % (d->CD / d->groups) == 0 && dbias == nullptr)
        return launch_wt<4, 4, 512, 1, 9, 4, true>(*d, dwp, dbias, t, st);
    static const bool no44 = xmc_debug_off("no_wt44");
    if (nco == 4 && nci == 4 && !no44) return launch_wt<4, 4, 512, 1, 9, 4>(*d, dwp, dbias, t, st);   // all 4 Cout blocks per wave: 0.9 LDS reads per MFMA   // 8 waves: 72 accumulator registers per lane
#undef WT_CASE
    return 1;
}
