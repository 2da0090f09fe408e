// Weight gradient of a 3x3 convolution that reads its input THROUGH a nearest x2 upsample (the generator's c1 after
// F.interpolate(scale_factor=2), df_gan.py:202,217; XmcConvDesc.src_shift == 1), at the LOW resolution.
//
// The all-taps kernel (conv_wgrad_tile.hip) treats the layer as nine taps over the 4x larger map: 36 (tap, low-res pixel)
// products per weight.  But the forward is four output-parity classes of 2x2-tap convolutions of the low-res map with pre-summed
// weights (ops.UpConvFn), so its weight gradient is 16 products over the low-res pixels.  Per dimension, with
// A(p, s) = sum_a dy[2a + p] * x[a + s]   (p = output parity, s = low-res shift):
//     dW[0] = A(0,-1) + A(1, 0)      dW[1] = A(0, 0) + A(1, 0)      dW[2] = A(0, 0) + A(1,+1)
// i.e. class p uses the two shifts s = p - 1 + u, u in {0, 1}, and product (p, u) belongs to the taps KH(p, u) = {0} / {1, 2} /
// {0, 1} / {2}.  In two dimensions: 4 classes x 4 shifts = 16 accumulators per (Cout, Cin) pair, each added into the 1, 2 or 4
// taps it belongs to when the workgroup retires: 2.25x fewer MFMAs than nine taps at the high resolution, and -- what bounds this
// family, see below -- a third of the LDS fragment reads, because the x patch is the LOW-resolution one.
//
// Structure (as conv_wgrad_tile.hip): persistent 8-wave workgroup per (64-channel Cin block, Cout block) walks tiles of 128 low-res
// pixels (4 x 32, or 8 x 16 on 16-pixel-wide maps); per tile the 2x-larger dy tile is staged DE-INTERLEAVED into four parity
// planes [class][low-res pixel][co] so that a transposing fragment read runs over consecutive low-res pixels, the x patch
// (tile + 1 halo) once; next tile prefetched into registers during the MFMAs.  Wave w owns class w & 3 and half of the
// (Cin block, shift) items: 8 items x NCO Cout blocks, all sharing the class's dy fragments: 0.75 transposing reads per MFMA
// (the nine-tap kernel: 0.9) on a quarter of the x pixels.  The nine-tap kernel measures LDS-bound (8 waves x 18 reads per
// 20 MFMAs = 2.3 us of LDS time against 1.25 us of MFMA time per tile), which is why the read count matters more than the
// MFMA count here.
// Takes over `errG.backward()`'s weight-gradient part for G_Block.c1 (train_gan.py:288; df_gan.py:217) on maps >= 16 wide.
#include "common.h"

namespace {

struct WUCfg {
    int tiles_y, tiles_x, ntiles;
    int tap_wi[9];                                            // packed-weight slice of tap kh * 3 + kw
};

template <int NCO, bool W16>
__global__ __launch_bounds__(512) void wgrad_up_kernel(const XmcConvDesc d, float* __restrict__ dwp, float* __restrict__ dbias, const WUCfg t) {
    constexpr int NT = 512, NCI = 4;
    constexpr int TLH = W16 ? 8 : 4, TLW = W16 ? 16 : 32;     // low-res tile
    constexpr int NPIX = TLH * TLW;                           // 128 low-res pixels = 4 K steps of 32
    constexpr int KS = NPIX / 32;
    constexpr int PH = TLH + 2, PW = TLW + 2;
    constexpr int CDP = NCO * 16, CSP = NCI * 16;
    constexpr int YS = CDP * 2 + 32, XS = CSP * 2 + 32;       // LDS row strides (bytes), as conv_wgrad_tile.hip
    constexpr int YCH = CDP / 8, XCH = CSP / 8;               // 16-byte chunks per pixel
    constexpr int HW = 2 * TLW;                               // dy tile: 2*TLH rows of HW hi-res pixels
    constexpr int YIT = 4 * NPIX * YCH / NT;                  // dy chunks per thread per tile
    constexpr int XIT = (PH * PW * XCH + NT - 1) / NT;        // patch chunks per thread per tile
    constexpr int MAXI = NCI * 4 / 2;                         // (Cin block, shift) items per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ydy = smem;                                // [4 classes][NPIX][YS]
    unsigned char* xp = smem + 4 * NPIX * YS;                 // [PH * PW][XS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cls = wave & 3, half = wave >> 2;
    const int cp = cls >> 1, cq = cls & 1;                    // output-row / output-column parity of this wave's class
    const int tpi = t.tiles_y * t.tiles_x;
    const int cd_units = d.CD / 8, cs_units = d.CS / 8;
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    const int ciu0 = ci0 >> 3, cou0 = co0 >> 3;
    const u32x4* __restrict__ x16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ y16 = reinterpret_cast<const u32x4*>(d.dst);

    f32x4 acc[MAXI][NCO];
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
#pragma unroll
        for (int c = 0; c < NCO; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging tables, computed once (see conv_wgrad_tile.hip): per thread the source offset of each staged unit from the tile's
    // origin, its LDS destination, and for the patch the image borders it would cross
    const int ych = tid % YCH, xch = tid % XCH;
    const bool yok = cou0 + ych < cd_units;
    // dy: unit `it` of a thread is RPI hi-res rows below unit 0 (same column), so its source offset is yoff0 + it * ystep and its
    // LDS destination ydst0 + a compile-time function of `it` (row parity -> class plane, row / 2 -> low-res row)
    constexpr int RPI = (NT / YCH) / HW;                      // hi-res rows between two units of a thread: 1, 2 or 4
    static_assert((NT / YCH) % HW == 0 && (RPI == 1 || RPI % 2 == 0), "dy units of a thread differ by whole rows");
    const int hy0 = (tid / YCH) / HW, hx0 = (tid / YCH) % HW;
    const unsigned yoff0 = yok ? (unsigned)((hy0 * d.MW + hx0) * cd_units + cou0 + ych) : 0u;
    const unsigned ystep = yok ? (unsigned)(RPI * d.MW * cd_units) : 0u;
    const int ydst0 = ((((hy0 & 1) * 2 + (hx0 & 1)) * NPIX) + (hy0 >> 1) * TLW + (hx0 >> 1)) * YS + ych * 16;
    auto ydelta = [](int it) -> int {                         // RPI == 1: row `it` -> plane (it & 1) * 2, low-res row it >> 1
        return RPI == 1 ? (((it & 1) * 2 * NPIX) + (it >> 1) * TLW) * YS : (it * (RPI / 2) * TLW) * YS;
    };
    unsigned xoff[XIT];
    unsigned xhalo = 0, xin = 0;
    static_assert(XIT <= 8, "halo bits");
    const unsigned xsafe = (unsigned)((1 * d.SW + 1) * cs_units);       // the tile's own first pixel, chunk 0 (patch origin = tile - 1)
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int pp = (tid + it * NT) / XCH;
        const int py = pp / PW, px = pp - py * PW;
        const bool in = pp < PH * PW && ciu0 + xch < cs_units;
        xin |= in ? (1u << it) : 0u;
        xoff[it] = in ? (unsigned)((py * d.SW + px) * cs_units + ciu0 + xch) : xsafe;
        // outside the image: above (first tile row only), below (last tile row only), left, right
        const unsigned hb = !in ? 0u : ((py == 0 ? 1u : 0u) | (py == PH - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PW - 1 ? 8u : 0u));
        xhalo |= hb << (4 * it);
    }

    u32x4 yv[YIT], xv[XIT];
    unsigned xzero = 0;
    auto prefetch = [&](int tile) {
        const int img = __builtin_amdgcn_readfirstlane(tile / tpi), trem = tile - img * tpi;
        const int ty = __builtin_amdgcn_readfirstlane(trem / t.tiles_x), tx = trem - ty * t.tiles_x;
        const int a0 = ty * TLH, b0 = tx * TLW;               // low-res tile origin
        const u32x4* __restrict__ yb = y16 + (((size_t)img * d.MH + 2 * a0) * d.MW + 2 * b0) * cd_units;
#pragma unroll
        for (int it = 0; it < YIT; ++it) yv[it] = yb[yoff0 + it * ystep];
        const unsigned border = (ty == 0 ? 1u : 0u) | (ty == t.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == t.tiles_x - 1 ? 8u : 0u);
        const long long xbase = (((long long)img * d.SH + (a0 - 1)) * d.SW + (b0 - 1)) * cs_units;
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
    // item j of this wave: Cin block half * 2 + j / 4, shift (u, v) = ((j >> 1) & 1, j & 1): patch pixel of low-res pixel (ay, ax) is
    // (ay + cp + u, ax + cq + v)
    const unsigned char* afrag = ydy + (size_t)(cls * NPIX + 4 * fg + q) * YS + (4 * pp4) * 2;
    const unsigned char* bfrag = xp + (size_t)((cp * PW + cq) + 4 * fg + q) * XS + (half * 2 * 16 + 4 * pp4) * 2;
    const unsigned char *afr = afrag, *bfr = bfrag;
    auto itoff = [&](int j) -> int { return (((j >> 1) & 1) * PW + (j & 1)) * XS + (j >> 2) * 32; };

    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();                                      // previous tile's reads are done
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            u32x4 v = yv[it];
            if (!yok) v = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(ydy + ydst0 + ydelta(it)) = v;
        }
        if (do_bias) {                                        // bias gradient: this thread always holds chunk tid % YCH
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
            int zq = 0;                                       // opaque zero: keeps the fragment addresses out of the tile loop's live set
            asm volatile("" : "+v"(zq));
            afr = afrag + zq; bfr = bfrag + zq;
        }
        // K loop: 32 low-res pixels per step, branch-free and software-pipelined (x fragment of item s + 2 and, once per step, the dy
        // fragments of the next step issued in front of the MFMAs of item s)
        constexpr int BD = 2, NSTEP = KS * MAXI;
        // ONE set of dy fragments (a second one does not fit beside 128 accumulators and the 48 staging registers): in the last item
        // of a step, block c's fragment of the NEXT step is requested right behind the MFMA that read block c for the last time
        bf16x8 af[NCO], bq[BD + 1];
        auto rd_a1 = [&](int r, int c) -> bf16x8 {
            const unsigned char* ab = afr + (size_t)(r * 32) * YS + c * 32;
            bf16x4 alo = xmc_ds_read_tr16((ab));
            bf16x4 ahi = xmc_ds_read_tr16((ab + 16 * YS));
            return bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        };
        auto rd_b = [&](int r, int j) -> bf16x8 {
            const unsigned char* bb = bfr + (r * (32 / TLW) * PW) * XS + itoff(j);
            bf16x4 blo = xmc_ds_read_tr16((bb));
            bf16x4 bhi = xmc_ds_read_tr16((bb + (W16 ? PW : 16) * XS));
            return bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        };
#pragma unroll
        for (int c = 0; c < NCO; ++c) af[c] = rd_a1(0, c);
#pragma unroll
        for (int k = 0; k < BD; ++k) bq[k] = rd_b(k / MAXI, k % MAXI);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
#pragma unroll
            for (int j = 0; j < MAXI; ++j) {
                const int s = r * MAXI + j;
                if (s + BD < NSTEP) bq[(s + BD) % (BD + 1)] = rd_b((s + BD) / MAXI, (s + BD) % MAXI);
#pragma unroll
                for (int c = 0; c < NCO; ++c) {
                    acc[j][c] = XMC_MFMA_16x16x32(af[c], bq[s % (BD + 1)], acc[j][c], 0, 0, 0);
                    if (j == MAXI - 1 && r + 1 < KS) af[c] = rd_a1(r + 1, c);
                }
                __builtin_amdgcn_sched_barrier(0);
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
    // accumulator (class (cp, cq), shift (u, v)) belongs to the taps KH(cp, u) x KW(cq, v): {0} / {1,2} / {0,1} / {2}
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int ib = half * 2 + (j >> 2), u = (j >> 1) & 1, v = j & 1;
        const int kh0 = cp == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), nkh = (cp == 0) == (u == 0) ? 1 : 2;
        const int kw0 = cq == 0 ? (v == 0 ? 0 : 1) : (v == 0 ? 0 : 2), nkw = (cq == 0) == (v == 0) ? 1 : 2;
        for (int kh = kh0; kh < kh0 + nkh; ++kh)
            for (int kw = kw0; kw < kw0 + nkw; ++kw) {
                float* __restrict__ slice = dwp + (size_t)t.tap_wi[kh * 3 + kw] * d.CDw * d.CS;
#pragma unroll
                for (int c = 0; c < NCO; ++c)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int co = co0 + c * 16 + fg * 4 + rr, ci = ci0 + ib * 16 + fr;
                        if (co < d.CDw && ci < d.CS) atomicAdd(&slice[(size_t)co * d.CS + ci], acc[j][c][rr]);
                    }
            }
    }
}

template <int NCO, bool W16>
int launch_wu(const XmcConvDesc& d, float* dwp, float* dbias, const WUCfg& t, hipStream_t st) {
    constexpr int YS = NCO * 32 + 32, XS = 4 * 32 + 32;
    constexpr int PH = (W16 ? 8 : 4) + 2, PW = (W16 ? 16 : 32) + 2;
    const size_t lds = (size_t)4 * 128 * YS + (size_t)PH * PW * XS;
    if (lds > XMC_MAX_DYN_LDS) return 1;
    XMC_ALLOW_BIG_LDS((wgrad_up_kernel<NCO, W16>));
    const int ny = d.CS / 64, nz = (d.CD + 63) / 64;
    int gx = 256 / (ny * nz);
    if (gx < 1) gx = 1;
    if (gx > t.ntiles) gx = t.ntiles;
    hipLaunchKernelGGL((wgrad_up_kernel<NCO, W16>), dim3(xmc_ab_grid(gx), ny, nz), dim3(512), lds, st, d, dwp, dbias, t);
    xmc_note_kernel(W16 ? "wgrad_up_kernel<%d, true>" : "wgrad_up_kernel<%d, false>", NCO);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// 0 = launched, 1 = not eligible (the caller goes on to the nine-tap kernels), other = error
int xmc_conv_wgrad_up_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream) {
    static const bool off = xmc_debug_off("no_wgrad_up");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || d->src_shift != 1 || d->SA != 1 || d->ntaps != 9 || d->groups > 1) return 1;
    if (d->MH != 2 * d->SH || d->MW != 2 * d->SW) return 1;
    if (d->CS % 64 != 0 || !(d->CD == 32 || d->CD % 64 == 0)) return 1;
    WUCfg t;
    for (int k = 0; k < 9; ++k) {                             // the 3x3 / padding-1 tap table in row-major order
        if (d->dh[0][k] != k / 3 - 1 || d->dw[0][k] != k % 3 - 1) return 1;
        t.tap_wi[k] = d->wi[0][k];
    }
    const bool w16 = d->SW % 32 != 0;
    if (w16 ? (d->SW % 16 != 0 || d->SH % 8 != 0) : (d->SH % 4 != 0)) return 1;
    t.tiles_y = d->SH / (w16 ? 8 : 4); t.tiles_x = d->SW / (w16 ? 16 : 32); t.ntiles = d->N * t.tiles_y * t.tiles_x;
    // offsets are 32-bit: (pixels of one image) * channel units must fit
    if ((long long)d->MH * d->MW * (d->CD / 8) >= (1ll << 31) || (long long)d->N * d->SH * d->SW * (d->CS / 8) >= (1ll << 40)) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->CD == 32) return w16 ? launch_wu<2, true>(*d, dwp, dbias, t, st) : launch_wu<2, false>(*d, dwp, dbias, t, st);
    return w16 ? launch_wu<4, true>(*d, dwp, dbias, t, st) : launch_wu<4, false>(*d, dwp, dbias, t, st);
}
