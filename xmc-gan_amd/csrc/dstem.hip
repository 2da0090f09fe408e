// The discriminator's stem as ONE convolution from the image (round 4).
//
// Reference: NetD.forward runs `conv_img` (3x3, 3 -> ndf, bias; df_gan.py:114,127) and hands its output to the first resD block,
// whose residual branch starts with a 4x4 stride-2 convolution WITHOUT an activation in between and whose shortcut is
// conv_s(avg_pool2d(.)) (df_gan.py:272-291).  Both are linear in the image:
//     conv_r[0](conv_img(x))          = a 6x6 stride-2 pad-2 convolution of x   with W_A = sum_mid W_0 (*) W_img
//     conv_s(avg_pool2d(conv_img(x))) = a 4x4 stride-2 pad-1 convolution of x   with W_B = W_s . (pool (*) W_img)
// (conv_img's bias becomes a constant per output channel away from the border).  The 32-channel full-resolution tensor conv_img writes -- 2.1 GB at 256x256 x 512 images, the largest
// tensor of the step, read back by two forward and three backward kernels -- never exists, and the 4x4 convolution's K = 512
// becomes K = 108.  conv_r[0] pads conv_img's OUTPUT with zeros, which the composition does not see: output pixels on the image
// border are recomputed the reference's way on 4-pixel-wide strips by the host (ops.DStemFn), everything else is exact algebra.
//
// Layouts: image [N,H,W,8] (channels 0-2 image, the rest ignored: their weights are zero); composed weights f32 [128][36 taps (ta*6+tb)][8]: rows 0-63
// = W_A (LeakyReLU'd output h1 [N,H/2,W/2,64]), rows 64-127 = W_B embedded in the 6x6 window (shortcut sc [N,H/2,W/2,64]).
//   xmc_dstem_pack    composed weights -> MFMA A-fragment order (16-bit), rows permuted so that a lane ends with 8 consecutive channels
//   xmc_dstem_fwd     h1 = lrelu(W_A * x + b_A), sc = W_B * x + b_B       (bias f32 [128]: the composed biases, constant over the interior)
//   xmc_dstem_wgrad   dW[128][36][8] += sum_pixels (dh1 | dsc) (x) patch(x), dbias[128] += sum_pixels (dh1 | dsc)
//                     (border pixels of dh1 optionally excluded)
#include "common.h"

namespace {

constexpr int kTaps = 36, kKSteps = 5;          // 36 taps x the first 4 channels of a pixel unit = 18 tap pairs of 8 K values: 4.5 MFMA K steps of 32

// one thread per (fragment, lane): fragment f = (half * 4 + j) * 5 + s holds, for row co(half, j, lane & 15) and tap pair m = 4s + (lane >> 4),
// W[co][tap 2m][0..3] | W[co][tap 2m + 1][0..3] (pairs 18, 19 of the last step: zeros)
__global__ void dstem_pack_kernel(const float* __restrict__ w, bf16x8* __restrict__ frag) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 8 * kKSteps * 64) return;
    const int lane = id & 63, f = id >> 6;
    const int s = f % kKSteps, hj = f / kKSteps, half = hj >> 2, j = hj & 3;
    const int q = lane & 15, kg = lane >> 4;
    const int co = half * 64 + (j >> 1) * 32 + (q >> 2) * 8 + (j & 1) * 4 + (q & 3);
    const int m = 4 * s + kg;
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (xmc_h16)(m < kTaps / 2 ? w[((size_t)co * kTaps + 2 * m + (c >> 2)) * 8 + (c & 3)] : 0.f);
    frag[id] = o;
}

// the shortcut's weights (rows 64..127 of the composed table: a 4x4 stride-2 pad-1 window inside the 6x6 one) as A fragments of
// v_mfma_f32_32x32x16 for the block-end kernel that recomputes the shortcut (conv_tile.hip, XmcConvDesc.sc_img): fragment (u, c), lane
// (rho = lane & 31, hh = lane >> 5) holds W[64 + co][window tap (u + 1, 2 hh + 1)][0..3] | W[64 + co][(u + 1, 2 hh + 2)][0..3] with
// co = 32 c + 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3) -- the row permutation of that kernel's own weight rows
__global__ void dstem_pack_sc_kernel(const float* __restrict__ w, bf16x8* __restrict__ frag) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 4 * 2 * 64) return;
    const int lane = id & 63, f = id >> 6, u = f >> 1, c = f & 1;
    const int rho = lane & 31, hh = lane >> 5;
    const int co = 32 * c + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (xmc_h16)w[((size_t)(64 + co) * kTaps + (u + 1) * 6 + 2 * hh + 1 + (k >> 2)) * 8 + (k & 3)];
    frag[id] = o;
}

// Persistent 8-wave workgroup, tile = 4 output rows x 32 output columns.  Wave (rp = w & 1, cq = w >> 1) owns output rows 2 rp,
// 2 rp + 1 of the tile (four 16-pixel blocks) and 32 of the 128 output channels, and keeps ITS weights -- 5 K steps x 2 row blocks
// of A fragments, 40 registers -- for the whole launch: the K loop reads only pixel fragments from LDS.  Only the first four channels
// of a pixel unit are staged (three of them are image): the source patch (12 x 68 pixels, zero outside the image) is row-major with
// 8 bytes per pixel, so the two taps (ta, tb), (ta, tb + 1), tb even, of output pixel px are the 16 contiguous bytes at column
// 2 px + tb -- ONE ds_read_b128 is a lane's eight K values (2 taps x 4 channels), consecutive output pixels are consecutive
// 16-byte slots, and the window's 36 taps are 4.5 K steps of 32 (the first version's K was tap x 8 channels: 9 steps, 5 of 8
// values zero).  Double-buffered patch, next tile prefetched into registers: one barrier per tile; <= 128 registers, so two
// workgroups share a CU and one's wait for its prefetch is the other's K loop.
// H1ONLY (the shortcut is recomputed by its consumer, XmcConvDesc.sc_img): all eight waves on h1 -- wave (row = w & 3, cq = w >> 2) owns ONE
// tile row (two 16-pixel blocks) and 32 of the 64 channels.
template <bool H1ONLY>
__global__ __launch_bounds__(512, 4) void dstem_fwd_kernel(const u32x4* __restrict__ img, const u32x4* __restrict__ frag, const float* __restrict__ bias,
                                                          bf16x8* __restrict__ h1, bf16x8* __restrict__ sc, int N, int H, int W, float slope, int ntiles) {
    constexpr int TR = 4, TC = 32, PR = 2 * TR + 4, PC = 2 * TC + 4;                      // 12 x 68 patch
    constexpr int PUNITS = PR * PC;                                                       // 816 eight-byte units
    __shared__ __attribute__((aligned(16))) u32x2 patch[2][PUNITS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NBLK = H1ONLY ? 2 : 4;         // 16-pixel blocks per wave
    const int rp = H1ONLY ? (wave & 3) : (wave & 1), cq = H1ONLY ? (wave >> 2) : (wave >> 1);
    const int p = lane & 15, g = lane >> 4;
    const int OH = H >> 1, OW = W >> 1;
    const int tiles_x = OW / TC, tpi = (OH / TR) * tiles_x;

    u32x4 afr[kKSteps][2];
#pragma unroll
    for (int s = 0; s < kKSteps; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) afr[s][j] = frag[((cq * 2 + j) * kKSteps + s) * 64 + lane];
    // this lane's tap pair of K step s is m = 4s + g = (ta, tb = 2 (m % 3)): 8-byte unit offset inside the patch relative to the output
    // pixel's own unit (row 2 py, column 2 px); the padded pairs 18, 19 re-read pair 0 (their weights are zero, the data must be finite)
    int toff[kKSteps];
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) {
        const int m = 4 * s + g, mm = m < kTaps / 2 ? m : 0;
        toff[s] = (mm / 3) * PC + 2 * (mm % 3);
    }
    const int pbase = (H1ONLY ? 2 : 4) * rp * PC + 2 * p;     // output pixel (first row of the wave, column p): block b = (row b >> 1, columns 16 (b & 1) ..)

    // staging: units u = tid, tid + 512 (816 of them)
    int urow[2], ucol[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = tid + it * 512;
        urow[it] = u / PC; ucol[it] = u - urow[it] * PC;
    }
    u32x2 pv[2];
    auto prefetch = [&](int tile) {
        const int n = tile / tpi, trem = tile - n * tpi;
        const int a0 = (trem / tiles_x) * TR, b0 = (trem % tiles_x) * TC;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int sy = 2 * a0 - 2 + urow[it], sx = 2 * b0 - 2 + ucol[it];
            const bool ok = tid + it * 512 < PUNITS && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            pv[it] = ok ? *reinterpret_cast<const u32x2*>(img + ((size_t)n * H + sy) * W + sx) : u32x2{0, 0};
        }
    };
    bf16x8* __restrict__ dst = cq < 2 ? h1 : sc;
    const float sl = cq < 2 ? slope : 1.f;
    const float* bp = bias + cq * 32 + g * 8;
    const f32x4 b0v = *reinterpret_cast<const f32x4*>(bp), b1v = *reinterpret_cast<const f32x4*>(bp + 4);
    const XcdWalk xw = xmc_xcd_walk(ntiles);
    int tile = xw.first, buf = 0;
    if (tile < xw.end) prefetch(tile);
    for (; tile < xw.end; tile += xw.step, buf ^= 1) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (tid + it * 512 < PUNITS) patch[buf][tid + it * 512] = pv[it];
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        f32x4 acc[NBLK][2];
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x2* pp = patch[buf] + pbase;
#pragma unroll
        for (int s = 0; s < kKSteps; ++s) {
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const u32x4 bf = *reinterpret_cast<const u32x4*>(pp + toff[s] + (b >> 1) * 2 * PC + (b & 1) * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[b][j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, afr[s][j]), __builtin_bit_cast(bf16x8, bf), acc[b][j], 0, 0, 0);
            }
        }
        // lane (p, g): channels cq * 32 + g * 8 + [0, 8) (row block 0: the first four, row block 1: the last four)
        const int n = tile / tpi, trem = tile - n * tpi;
        const int oy0 = (trem / tiles_x) * TR + (H1ONLY ? 1 : 2) * rp, ox0 = (trem % tiles_x) * TC + p;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const size_t pix = ((size_t)n * OH + oy0 + (b >> 1)) * OW + ox0 + (b & 1) * 16;
            bf16x8 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = acc[b][0][r] + b0v[r], v1 = acc[b][1][r] + b1v[r];
                o[r] = (xmc_h16)fmaxf(v0, v0 * sl);
                o[4 + r] = (xmc_h16)fmaxf(v1, v1 * sl);
            }
            dst[pix * 8 + (cq & 1) * 4 + g] = o;
        }
    }
}

// Weight gradient of the composed stem: dW[co][tap][c] += sum over output pixels of dy[pixel][co] * x[2 pixel + tap - 2][c], with
// dy = (dh1 | dsc).  GEMM view: M = 128 output channels, N = (tap, channel) pairs, K = pixels.  Both operands are pixel-major,
// i.e. K-strided: fragments come from transposing LDS reads (ds_read_b64_tr_b16) as in the other weight-gradient kernels.  A
// transposing read takes 8 bytes = FOUR channels of a pixel unit, and only three channels of the unit are image: a 16-wide N block
// is four consecutive taps (in window order t = ta * 6 + tb) x the unit's first four channels -- 9 blocks cover the 36 taps, every
// column but the fourth of a tap is useful (the first version's block was a tap pair x 8 channels: 18 blocks, 5 of 8 columns zero).
// Wave w owns the 16 output channels of row block w and all 9 tap quads (36 accumulator registers); K step r is output row r of
// the tile (32 pixels), whose tap row ta reads patch row 2r + ta: quad j of step r + 1 reads what quad j + 3 (12 taps = two tap
// rows on) read in step r, so a step reads 3 new quad fragments, not 9.
// Tile = 4 output rows x 32 columns; dy tile [128][128 + pad] and the 12 x 68 source patch in LDS (46 KB), next tile prefetched
// into registers; <= 128 registers, so TWO workgroups share a CU and one's wait for its prefetch is the other's K loop (the kernel
// moves 75 KB per 72 MFMAs of a wave: with one workgroup per CU a tile cost the memory latency, 16 k cycles, whatever the MFMA
// count).  One atomic per weight per workgroup at the end (channels 4-7 of dW are not written: they multiply zeros).
constexpr int kWgTR = 4;
__global__ __launch_bounds__(512, 4) void dstem_wgrad_kernel(const u32x4* __restrict__ img, const u32x4* __restrict__ dh1, const u32x4* __restrict__ dsc,
                                                         float* __restrict__ dw, float* __restrict__ dbias, int N, int H, int W, int skip_border,
                                                         int ntiles) {
    constexpr int TR = kWgTR, TC = 32, PR = 2 * TR + 4, PC = 2 * TC + 4;                  // 12 x 68 patch
    constexpr int PLANE = 48;                // 8-byte slots (channels 0-3 of a pixel) per column-parity plane: 34 used; 384 bytes = 32 banks, so the
                                             // two planes a quad's taps read (16 consecutive slots each) cover the 64 banks once per pair
    constexpr int PUNITS = PR * PC;                                                       // 1360 units
    constexpr int YS = 128 * 2 + 32;                                                      // dy row stride (bytes): 128 channels + pad
    constexpr int XIT = (PUNITS + 511) / 512;                                             // 2
    constexpr int YIT = TR * TC * 16 / 512;                                               // 4 sixteen-byte units of dy per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ydy = smem;                                // [256][YS]
    unsigned char* xp = smem + TR * TC * YS;                  // [PR][2 planes][PLANE] x 8 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OH = H >> 1, OW = W >> 1;
    const int tiles_x = OW / TC, tiles_y = OH / TR, tpi = tiles_y * tiles_x;

    f32x4 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dy staging: unit (pixel, chunk) = tid / 16 + it * 32, tid % 16; chunks 0-7 from dh1, 8-15 from dsc
    const int ych = tid & 15, ypix0 = tid >> 4;
    const u32x4* __restrict__ ysrc = ych < 8 ? dh1 : dsc;
    const int ychs = ych & 7;
    int xrow[XIT], xcol[XIT], xdst[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int u = tid + it * 512;
        xrow[it] = u / PC; xcol[it] = u - xrow[it] * PC;
        xdst[it] = ((xrow[it] * 2 + (xcol[it] & 1)) * PLANE + (xcol[it] >> 1)) * 8;
    }
    u32x4 yv[YIT];
    u32x2 xv[XIT];                                            // channels 0-3 of a source pixel (three of them image)
    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto prefetch = [&](int tile) {
        const int n = tile / tpi, trem = tile - n * tpi;
        const int a0 = (trem / tiles_x) * TR, b0 = (trem % tiles_x) * TC;
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            const int pix = ypix0 + it * 32, py = pix >> 5, px = pix & 31;
            u32x4 v = ysrc[(((size_t)n * OH + a0 + py) * OW + b0 + px) * 8 + ychs];
            // the composed weights do not hold for h1's pixels on the image border (the host recomputes those): no gradient from them
            if (skip_border && ych < 8) {
                const int oy = a0 + py, ox = b0 + px;
                if (oy == 0 || oy == OH - 1 || ox == 0 || ox == OW - 1) v = u32x4{0, 0, 0, 0};
            }
            yv[it] = v;
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int sy = 2 * a0 - 2 + xrow[it], sx = 2 * b0 - 2 + xcol[it];
            const bool ok = tid + it * 512 < PUNITS && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            xv[it] = ok ? *reinterpret_cast<const u32x2*>(img + ((size_t)n * H + sy) * W + sx) : u32x2{0, 0};
        }
    };
    const int fr = lane & 15, fg = lane >> 4;
    const int q = fr >> 2, pp4 = fr & 3;
    // A' fragment: dy^T [co = 16 wave + fr][pixel 4 fg + q (+16)]: 8 bytes = 4 channels at channel offset 4 pp4 of the wave's block
    const unsigned char* afrag = ydy + (size_t)(4 * fg + q) * YS + (wave * 16 + 4 * pp4) * 2;
    // B fragment of tap quad j: lane (q, pp4) reads channels 0-3 of the unit of output pixel 4 fg + q under tap t = 4 j + pp4 =
    // (ta, tb): patch row 2r + ta, source column 2 px + tb -> plane tb & 1, slot px + (tb >> 1)
    int boff[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int tq = 4 * j + pp4, ta = tq / 6, tb = tq - 6 * ta;
        boff[j] = ((ta * 2 + (tb & 1)) * PLANE + (tb >> 1) + 4 * fg + q) * 8;
    }
    const XcdWalk xw = xmc_xcd_walk(ntiles);
    int tile = xw.first;
    if (tile < xw.end) prefetch(tile);
    for (; tile < xw.end; tile += xw.step) {
        __syncthreads();                                      // previous tile's reads are done
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            *reinterpret_cast<u32x4*>(ydy + (size_t)(ypix0 + it * 32) * YS + ych * 16) = yv[it];
            const bf16x8 hv = __builtin_bit_cast(bf16x8, yv[it]);     // bias gradient: this thread always stages channel chunk ych
#pragma unroll
            for (int k = 0; k < 8; ++k) bsum[k] += (float)hv[k];
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (tid + it * 512 < PUNITS) *reinterpret_cast<u32x2*>(xp + xdst[it]) = xv[it];
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        int zq = 0;
        asm volatile("" : "+v"(zq));                          // opaque zero: keeps the fragment addresses out of the tile loop's live set
        const unsigned char* afr = afrag + zq;
        const unsigned char* bfr = xp + zq;
        auto rd_a = [&](int r) -> bf16x8 {
            const unsigned char* ab = afr + (size_t)(r * 32) * YS;
            bf16x4 lo = xmc_ds_read_tr16(ab), hi = xmc_ds_read_tr16(ab + 16 * YS);
            return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto rd_b = [&](int r, int j) -> bf16x8 {             // output row r (pixels 0..31), tap quad j
            const unsigned char* bb = bfr + (size_t)(2 * r * 2 * PLANE) * 8 + boff[j];
            bf16x4 lo = xmc_ds_read_tr16(bb), hi = xmc_ds_read_tr16(bb + 16 * 8);
            return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        bf16x8 bq[9];                                         // slot (j + 3 r) % 9 holds quad j of step r
#pragma unroll
        for (int j = 0; j < 6; ++j) bq[j] = rd_b(0, j);
        bf16x8 af = rd_a(0);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int j = 6; j < 9; ++j) bq[(j + 3 * r) % 9] = rd_b(r, j);      // tap rows 4, 5 of this output row are new
            bf16x8 afn = af;
            if (r + 1 < TR) afn = rd_a(r + 1);
#pragma unroll
            for (int j = 0; j < 9; ++j) acc[j] = XMC_MFMA_16x16x32(af, bq[(j + 3 * r) % 9], acc[j], 0, 0, 0);
            af = afn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // bias gradient: lanes with equal lane & 15 hold the same channel chunk; the eight waves are summed through LDS so that a workgroup
    // sends ONE atomic per channel (one per wave was 4096 atomics on each of 128 addresses in four cache lines at batch 512: ~0.5 ms of
    // serialised read-modify-writes behind a 0.2 ms kernel)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);              // [8 waves][128]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = bsum[k];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) red[wave * 128 + lane * 8 + k] = v;
    }
    __syncthreads();
    if (tid < 128) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w * 128 + tid];
        atomicAdd(&dbias[tid], v);
    }
    // D[row = co][col = (tap of the quad, channel < 4)]: lane (fr, fg) holds rows 4 fg + rr of column fr
    if ((fr & 3) < 3) {
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int co = wave * 16 + fg * 4 + rr, tap = 4 * j + (fr >> 2), c = fr & 3;
                atomicAdd(&dw[((size_t)co * kTaps + tap) * 8 + c], acc[j][rr]);
            }
    }
}

// ---- composition of the stem's weights (and its adjoint) ----------------------------------------------------------------------------
// W[o][ta*6+tb][c]      = sum_mid sum_{kh+ih=ta, kw+iw=tb} w0[o][mid][kh][kw] * wi[mid][c][ih][iw]                        o < 64
// W[64+o][ta*6+tb][c]   = sum_mid ws[o][mid] * 1/4 sum_{u+ih=ta-1, v+iw=tb-1, u,v in {0,1}} wi[mid][c][ih][iw]            1 <= ta,tb <= 4
// bias[o] = sum w0[o][mid][.][.] bi[mid];  bias[64+o] = sum ws[o][mid] bi[mid] + bs[o];  D / DB: the border tables (below)
// One thread per output element; mid = 32 conv_img channels, c < 3 image channels (the rest of the 8 stay zero).
struct ComposeArgs {
    const float *wi, *bi, *w0, *ws, *bs;                     // [32][3][3][3], [32], [64][32][4][4], [64][32], [64] or NULL
    float *W, *bias, *D, *DB;                                // [128][36][8], [128], [64][28][8], [64][8]
};
__device__ __forceinline__ float wi_at(const float* wi, int m, int c, int ih, int iw) {
    return ((unsigned)ih < 3u && (unsigned)iw < 3u) ? wi[((m * 3 + c) * 3 + ih) * 3 + iw] : 0.f;
}
// border table row t of D -> which (kh, kw) taps of w0 and which fixed (ih | iw) of wi it sums over; see compose_dstem (ops.py)
__global__ void dstem_compose_kernel(ComposeArgs a) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int NW = 128 * 36 * 8, NB = 128, ND = 64 * 28 * 8, NDB = 64 * 8;
    if (id < NW) {
        const int c = id & 7, t = (id >> 3) % 36, o = id / (36 * 8);
        const int ta = t / 6, tb = t - ta * 6;
        float s = 0.f;
        if (c < 3) {
            if (o < 64) {
                for (int m = 0; m < 32; ++m)
                    for (int kh = 0; kh < 4; ++kh)
                        for (int kw = 0; kw < 4; ++kw) s += a.w0[((o * 32 + m) * 4 + kh) * 4 + kw] * wi_at(a.wi, m, c, ta - kh, tb - kw);
            } else {
                for (int m = 0; m < 32; ++m) {
                    float p = 0.f;
                    for (int u = 0; u < 2; ++u)
                        for (int v = 0; v < 2; ++v) p += wi_at(a.wi, m, c, ta - 1 - u, tb - 1 - v);
                    s += a.ws[(o - 64) * 32 + m] * 0.25f * p;
                }
                if (ta < 1 || ta > 4 || tb < 1 || tb > 4) s = 0.f;
            }
        }
        a.W[id] = s;
    } else if (id < NW + NB) {
        const int o = id - NW;
        float s = 0.f;
        if (o < 64) {
            for (int m = 0; m < 32; ++m) {
                float q = 0.f;
                for (int k = 0; k < 16; ++k) q += a.w0[(o * 32 + m) * 16 + k];
                s += q * a.bi[m];
            }
        } else {
            for (int m = 0; m < 32; ++m) s += a.ws[(o - 64) * 32 + m] * a.bi[m];
            if (a.bs) s += a.bs[o - 64];
        }
        a.bias[o] = s;
    } else if (id < NW + NB + ND) {
        const int j = id - NW - NB, c = j & 7, t = (j >> 3) % 28, o = j / (28 * 8);
        float s = 0.f;
        if (c < 3) {
            for (int m = 0; m < 32; ++m) {
                const float* w0 = a.w0 + (o * 32 + m) * 16;
                if (t < 6) { for (int kw = 0; kw < 4; ++kw) s -= w0[0 * 4 + kw] * wi_at(a.wi, m, c, 2, t - kw); }
                else if (t < 12) { for (int kw = 0; kw < 4; ++kw) s -= w0[3 * 4 + kw] * wi_at(a.wi, m, c, 0, t - 6 - kw); }
                else if (t < 18) { for (int kh = 0; kh < 4; ++kh) s -= w0[kh * 4 + 0] * wi_at(a.wi, m, c, t - 12 - kh, 2); }
                else if (t < 24) { for (int kh = 0; kh < 4; ++kh) s -= w0[kh * 4 + 3] * wi_at(a.wi, m, c, t - 18 - kh, 0); }
                else if (t == 24) s += w0[0] * wi_at(a.wi, m, c, 2, 2);
                else if (t == 25) s += w0[3] * wi_at(a.wi, m, c, 2, 0);
                else if (t == 26) s += w0[12] * wi_at(a.wi, m, c, 0, 2);
                else s += w0[15] * wi_at(a.wi, m, c, 0, 0);
            }
        }
        a.D[j] = s;
    } else if (id < NW + NB + ND + NDB) {
        const int j = id - NW - NB - ND, e = j & 7, o = j >> 3;
        float s = 0.f;
        for (int m = 0; m < 32; ++m) {
            const float* w0 = a.w0 + (o * 32 + m) * 16;
            float q;
            if (e == 0) q = -(w0[0] + w0[1] + w0[2] + w0[3]);
            else if (e == 1) q = -(w0[12] + w0[13] + w0[14] + w0[15]);
            else if (e == 2) q = -(w0[0] + w0[4] + w0[8] + w0[12]);
            else if (e == 3) q = -(w0[3] + w0[7] + w0[11] + w0[15]);
            else if (e == 4) q = w0[0];
            else if (e == 5) q = w0[3];
            else if (e == 6) q = w0[12];
            else q = w0[15];
            s += q * a.bi[m];
        }
        a.DB[j] = s;
    }
}

// adjoint: gradients of the five parameters from the gradients of the four tables.  One thread per element of w0 / ws / bs; the
// conv_img gradients sum over the 64 output channels: one 64-thread block per element, summed by the wave.
struct ComposeBwdArgs {
    const float *wi, *bi, *w0, *ws;
    const float *dW, *dbias, *dD, *dDB;
    float *dwi, *dbi, *dw0, *dws, *dbs;                      // dbs may be NULL
};
__device__ __forceinline__ float dW_at(const float* dW, int o, int ta, int tb, int c) {
    return ((unsigned)ta < 6u && (unsigned)tb < 6u) ? dW[((size_t)o * 36 + ta * 6 + tb) * 8 + c] : 0.f;
}
__device__ __forceinline__ float dD_at(const float* dD, int o, int t0, int t, int c) {      // six-tap line t0 .. t0 + 5
    return (unsigned)t < 6u ? dD[((size_t)o * 28 + t0 + t) * 8 + c] : 0.f;
}
__global__ __launch_bounds__(64) void dstem_compose_bwd_wi_kernel(ComposeBwdArgs a) {
    // blocks 0 .. 863: dwi[m][c][ih][iw]; blocks 864 .. 895: dbi[m]; thread o = one output channel's share, summed by the wave
    const int o = threadIdx.x, b = blockIdx.x;
    float s = 0.f;
    if (b < 32 * 27) {
        const int iw = b % 3, ih = (b / 3) % 3, c = (b / 9) % 3, m = b / 27;
        const float* w0 = a.w0 + (o * 32 + m) * 16;
        for (int kh = 0; kh < 4; ++kh)
            for (int kw = 0; kw < 4; ++kw) s += dW_at(a.dW, o, kh + ih, kw + iw, c) * w0[kh * 4 + kw];
        float p = 0.f;
        for (int u = 0; u < 2; ++u)
            for (int v = 0; v < 2; ++v) p += dW_at(a.dW, 64 + o, 1 + u + ih, 1 + v + iw, c);
        s += 0.25f * p * a.ws[o * 32 + m];
        if (ih == 2) for (int kw = 0; kw < 4; ++kw) s -= dD_at(a.dD, o, 0, kw + iw, c) * w0[kw];
        if (ih == 0) for (int kw = 0; kw < 4; ++kw) s -= dD_at(a.dD, o, 6, kw + iw, c) * w0[12 + kw];
        if (iw == 2) for (int kh = 0; kh < 4; ++kh) s -= dD_at(a.dD, o, 12, kh + ih, c) * w0[kh * 4];
        if (iw == 0) for (int kh = 0; kh < 4; ++kh) s -= dD_at(a.dD, o, 18, kh + ih, c) * w0[kh * 4 + 3];
        if (ih == 2 && iw == 2) s += a.dD[((size_t)o * 28 + 24) * 8 + c] * w0[0];
        if (ih == 2 && iw == 0) s += a.dD[((size_t)o * 28 + 25) * 8 + c] * w0[3];
        if (ih == 0 && iw == 2) s += a.dD[((size_t)o * 28 + 26) * 8 + c] * w0[12];
        if (ih == 0 && iw == 0) s += a.dD[((size_t)o * 28 + 27) * 8 + c] * w0[15];
        s = wave_sum(s);
        if (o == 0) a.dwi[b] = s;
    } else {
        const int m = b - 32 * 27;
        const float* w0 = a.w0 + (o * 32 + m) * 16;
        float q = 0.f;
        for (int k = 0; k < 16; ++k) q += w0[k];
        s = a.dbias[o] * q + a.dbias[64 + o] * a.ws[o * 32 + m];
        const float* g = a.dDB + o * 8;
        s -= g[0] * (w0[0] + w0[1] + w0[2] + w0[3]) + g[1] * (w0[12] + w0[13] + w0[14] + w0[15]) + g[2] * (w0[0] + w0[4] + w0[8] + w0[12]) +
             g[3] * (w0[3] + w0[7] + w0[11] + w0[15]);
        s += g[4] * w0[0] + g[5] * w0[3] + g[6] * w0[12] + g[7] * w0[15];
        s = wave_sum(s);
        if (o == 0) a.dbi[m] = s;
    }
}

__global__ void dstem_compose_bwd_kernel(ComposeBwdArgs a) {
    constexpr int NW0 = 64 * 32 * 16, NWS = 64 * 32, NBS = 64;      // (dwi / dbi: dstem_compose_bwd_wi_kernel)
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < NW0) {                                          // dw0[o][m][kh][kw]
        const int j = id, kw = j & 3, kh = (j >> 2) & 3, m = (j >> 4) & 31, o = j >> 9;
        float s = a.bi[m] * a.dbias[o];
        const float* g = a.dDB + o * 8;
        for (int c = 0; c < 3; ++c)
            for (int ih = 0; ih < 3; ++ih)
                for (int iw = 0; iw < 3; ++iw) s += dW_at(a.dW, o, kh + ih, kw + iw, c) * a.wi[((m * 3 + c) * 3 + ih) * 3 + iw];
        if (kh == 0) {
            s -= a.bi[m] * g[0];
            for (int c = 0; c < 3; ++c)
                for (int iw = 0; iw < 3; ++iw) s -= dD_at(a.dD, o, 0, kw + iw, c) * a.wi[((m * 3 + c) * 3 + 2) * 3 + iw];
        }
        if (kh == 3) {
            s -= a.bi[m] * g[1];
            for (int c = 0; c < 3; ++c)
                for (int iw = 0; iw < 3; ++iw) s -= dD_at(a.dD, o, 6, kw + iw, c) * a.wi[((m * 3 + c) * 3 + 0) * 3 + iw];
        }
        if (kw == 0) {
            s -= a.bi[m] * g[2];
            for (int c = 0; c < 3; ++c)
                for (int ih = 0; ih < 3; ++ih) s -= dD_at(a.dD, o, 12, kh + ih, c) * a.wi[((m * 3 + c) * 3 + ih) * 3 + 2];
        }
        if (kw == 3) {
            s -= a.bi[m] * g[3];
            for (int c = 0; c < 3; ++c)
                for (int ih = 0; ih < 3; ++ih) s -= dD_at(a.dD, o, 18, kh + ih, c) * a.wi[((m * 3 + c) * 3 + ih) * 3 + 0];
        }
        const int corner = (kh == 0 && kw == 0) ? 0 : (kh == 0 && kw == 3) ? 1 : (kh == 3 && kw == 0) ? 2 : (kh == 3 && kw == 3) ? 3 : -1;
        if (corner >= 0) {
            const int ih = corner < 2 ? 2 : 0, iw = (corner & 1) ? 0 : 2;
            s += a.bi[m] * g[4 + corner];
            for (int c = 0; c < 3; ++c) s += a.dD[((size_t)o * 28 + 24 + corner) * 8 + c] * a.wi[((m * 3 + c) * 3 + ih) * 3 + iw];
        }
        a.dw0[j] = s;
    } else if (id < NW0 + NWS) {                             // dws[o][m]
        const int j = id - NW0, m = j & 31, o = j >> 5;
        float s = a.dbias[64 + o] * a.bi[m];
        for (int c = 0; c < 3; ++c)
            for (int ta = 1; ta < 5; ++ta)
                for (int tb = 1; tb < 5; ++tb) {
                    float p = 0.f;
                    for (int u = 0; u < 2; ++u)
                        for (int v = 0; v < 2; ++v) p += wi_at(a.wi, m, c, ta - 1 - u, tb - 1 - v);
                    s += a.dW[((size_t)(64 + o) * 36 + ta * 6 + tb) * 8 + c] * 0.25f * p;
                }
        a.dws[j] = s;
    } else if (id < NW0 + NWS + NBS) {
        const int o = id - NW0 - NWS;
        if (a.dbs) a.dbs[o] = a.dbias[64 + o];
    }
}

// ---- gradient of the image ------------------------------------------------------------------------------------------------------------
// dx[2a + py][2b + px][c] = sum over the three low-res offsets d = +1, 0, -1 per dimension, the 128 gradient channels o:
//      W[o][ta = py + 2 (1 - dh)][tb = px + 2 (1 - dw)][c] * dy[a + dh][b + dw][o]             dy = (dh1 | dsc)
// i.e. ONE 3x3 convolution over the low-resolution gradient map whose 12 output "channels" are (image channel c, parity class):
// the four classes share every dy fragment.  MFMA rows = c * 4 + class (12 of 16 used), K = 9 offsets x 128 channels, columns =
// low-res pixels; a lane ends with the four classes of ONE image channel of its pixel, three lanes (g = c) hold a 2x2 block of
// image pixels.  Structure of thin_out_kernel (conv_thin.hip): weights in registers (36 fragments), 8 x 16 low-res tiles with the
// 10 x 18 halo patch of both gradient tensors in LDS, next tile prefetched into registers.
// dstem_pack_t: the composed weights -> those 36 fragments: fragment (t = dh_i * 3 + dw_i, kc) holds, for row (c, class) and the
// 32 gradient channels of chunk kc, W[32 kc + k][ta][tb][c].
__global__ void dstem_pack_t_kernel(const float* __restrict__ w, bf16x8* __restrict__ frag) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 9 * 4 * 64) return;
    const int lane = id & 63, f = id >> 6, kc = f & 3, t = f >> 2;
    const int row = lane & 15, kg = lane >> 4;
    const int c = row >> 2, cls = row & 3, py = cls >> 1, px = cls & 1;
    const int dhi = t / 3, dwi = t - dhi * 3;                 // offset d = 1 - index: +1, 0, -1
    const int ta = py + 2 * dhi, tb = px + 2 * dwi;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int och = kc * 32 + kg * 8 + k;
        o[k] = (xmc_h16)(c < 3 ? w[((size_t)och * kTaps + ta * 6 + tb) * 8 + c] : 0.f);
    }
    frag[id] = o;
}

constexpr int DG_H = 8, DG_W = 16, DG_PH = DG_H + 2, DG_PW = DG_W + 2, DG_PSTR = 16 * 16 + 16;      // low-res tile, halo patch, pixel stride
__global__ __launch_bounds__(256) void dstem_dgrad_kernel(const u32x4* __restrict__ dh1, const u32x4* __restrict__ dsc, const u32x4* __restrict__ frag,
                                                         bf16x8* __restrict__ dimg, int N, int H, int W, int ntiles) {
    constexpr int NUN = DG_PH * DG_PW * 16, MAXU = (NUN + 255) / 256;          // 2880 sixteen-byte units: 12 per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[DG_PH * DG_PW * DG_PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kb = lane >> 4;
    const int OH = H >> 1, OW = W >> 1;
    const int tiles_x = OW / DG_W, tiles_y = OH / DG_H;
    bf16x8 wa[36];
    int loff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dh = 1 - t / 3, dw = 1 - t % 3;
        loff[t] = ((dh + 1) * DG_PW + dw + 1 + col) * DG_PSTR + kb * 16;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) wa[t * 4 + kc] = __builtin_bit_cast(bf16x8, frag[(t * 4 + kc) * 64 + lane]);
    }
    u32x4 pv[MAXU];
    auto prefetch = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * DG_H - 1, x0 = (tr % tiles_x) * DG_W - 1;
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int id = tid + it * 256;
            const int pp = id >> 4, ch = id & 15;
            const int py = pp / DG_PW, px = pp - py * DG_PW;
            const int sy = y0 + py, sx = x0 + px;
            const bool ok = id < NUN && (unsigned)sy < (unsigned)OH && (unsigned)sx < (unsigned)OW;
            const u32x4* src = ch < 8 ? dh1 : dsc;
            pv[it] = ok ? src[(((size_t)n * OH + sy) * OW + sx) * 8 + (ch & 7)] : u32x4{0, 0, 0, 0};
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
            if (id < NUN) *reinterpret_cast<u32x4*>(smem + (id >> 4) * DG_PSTR + (id & 15) * 16) = pv[it];
        }
        __syncthreads();
        if (tile + xw.step < xw.end) prefetch(tile + xw.step);
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * DG_H, x0 = (tr % tiles_x) * DG_W;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wave * 2 + rr;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + r * DG_PW * DG_PSTR + loff[t] + kc * 64);
                    acc = XMC_MFMA_16x16x32(wa[t * 4 + kc], b, acc, 0, 0, 0);
                }
            // lane (col, kb = c): acc[class] for image channel c of low-res pixel (y0 + r, x0 + col).  The lanes c = 1, 2 hand their
            // four values to the c = 0 lane, which writes the 2x2 block's four 16-byte pixels
            float v1[4], v2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v1[k] = __shfl(acc[k], col + 16, 64);
                v2[k] = __shfl(acc[k], col + 32, 64);
            }
            if (kb == 0) {
                const int a = y0 + r, b = x0 + col;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bf16x8 o;
                    o[0] = (xmc_h16)acc[k]; o[1] = (xmc_h16)v1[k]; o[2] = (xmc_h16)v2[k];
#pragma unroll
                    for (int z = 3; z < 8; ++z) o[z] = (xmc_h16)0.f;
                    dimg[((size_t)n * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1)] = o;
                }
            }
        }
    }
}

// the border corrections' part of the image gradient: D's taps read image row 0 / H - 1 and column 0 / W - 1 only, so only those
// image pixels receive it.  One thread per (image, line pixel): dx(line pixel) += sum over the border output pixels q that read it
// and the 64 channels of D[o][tap][c] * dh1(q)[o]; corner taps with the row lines.  Read-modify-write of dstem_dgrad_kernel's
// output, in two launches (phase 0: the two rows, phase 1: the two columns -- the corner pixels belong to both).
__global__ __launch_bounds__(256) void dstem_border_dgrad_kernel(const bf16x8* __restrict__ dh1, const float* __restrict__ D, bf16x8* __restrict__ dimg,
                                                                int N, int H, int W, int phase) {
    const int OH = H >> 1, OW = W >> 1;
    const int len = phase == 0 ? W : H, per = 2 * len;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)N * per) return;
    const int n = (int)(gid / per), i = (int)(gid - (long long)n * per);
    const int side = phase * 2 + (i >= len ? 1 : 0), pos = i >= len ? i - len : i;
    const bool rows = phase == 0;
    const int L = rows ? OW : OH;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    auto add = [&](int oy, int ox, int tap) {                 // += D[.][tap][c] . dh1(oy, ox)
        const bf16x8* g = dh1 + (((size_t)n * OH + oy) * OW + ox) * 8;
        for (int u = 0; u < 8; ++u) {
            const bf16x8 gv = g[u];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gk = (float)gv[k];
                const float* dd = D + ((size_t)(u * 8 + k) * 28 + tap) * 8;
                s0 += gk * dd[0]; s1 += gk * dd[1]; s2 += gk * dd[2];
            }
        }
    };
    // image line position pos is read by border output pixel q through tap t with 2 q - 2 + t == pos
    for (int t = (pos & 1); t < 6; t += 2) {
        const int q = (pos + 2 - t) >> 1;
        if (q < 0 || q >= L) continue;
        add(rows ? (side == 0 ? 0 : OH - 1) : q, rows ? q : (side == 2 ? 0 : OW - 1), side * 6 + t);
    }
    if (rows && (pos == 0 || pos == W - 1))                   // corner taps 24..27 read image pixels (0,0), (0,W-1), (H-1,0), (H-1,W-1)
        add(side == 0 ? 0 : OH - 1, pos == 0 ? 0 : OW - 1, 24 + side * 2 + (pos == 0 ? 0 : 1));
    const int y = rows ? (side == 0 ? 0 : H - 1) : pos, x = rows ? pos : (side == 2 ? 0 : W - 1);
    bf16x8* p = dimg + ((size_t)n * H + y) * W + x;
    bf16x8 o = *p;
    o[0] = (xmc_h16)((float)o[0] + s0); o[1] = (xmc_h16)((float)o[1] + s1); o[2] = (xmc_h16)((float)o[2] + s2);
    *p = o;
}

// ---- the border of h1 ------------------------------------------------------------------------------------------------------------
// conv_r[0] pads conv_img's output with zeros; the composed convolution instead sees conv_img evaluated one pixel outside the image.
// For an output pixel in the first row that surplus is  sum_kw W_0[kh = 0, kw] . conv_img(row -1)  and conv_img(row -1) reads image
// row 0 only (through its last tap row): a 6-tap row correction on tap row a = 2 of the composed window; likewise the last row
// (a = 3), the first / last column (b = 2 / 3) and the four corners (single taps, added back once).  The corrections are linear in
// (w0, w_img, b_img) like the composed weights themselves (ops.compose_dstem builds them):
//   D  f32 [64][28][8]: taps 0-5 top (by b), 6-11 bottom, 12-17 left (by a), 18-23 right, 24-27 corners TL TR BL BR
//   DB f32 [64][8]    : their constant terms (conv_img's bias through the dropped taps), same order
// The border kernels touch 2 (OH + OW) - 4 pixels per image -- 3 % of the map at 128 x 128:
//   dstem_border_fwd    recomputes h1 there from the image with W + D as a small MFMA GEMM, overwriting what the main kernel wrote
//   dstem_border_wgrad  dD, dDB += sum over border pixels of dh1 (x) image taps (the composed weights' own gradient takes ALL pixels)
__device__ __forceinline__ void border_pixel(int i, int OH, int OW, int& py, int& px) {      // i-th pixel of the perimeter
    if (i < OW) { py = 0; px = i; }
    else if (i < 2 * OW) { py = OH - 1; px = i - OW; }
    else { const int j = i - 2 * OW; py = 1 + (j >> 1); px = (j & 1) ? OW - 1 : 0; }
}

// MFMA fragments of the border kernel's weights: 64 "taps" (36 composed + 28 corrections) x 8 channels = 16 K steps, 64 output
// channels = 4 row blocks; fragment f = j * 16 + s, rows permuted as in dstem_pack_kernel
__global__ void dstem_border_pack_kernel(const float* __restrict__ w, const float* __restrict__ D, bf16x8* __restrict__ frag) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 4 * 16 * 64) return;
    const int lane = id & 63, f = id >> 6, s = f & 15, j = f >> 4;
    const int q = lane & 15, kg = lane >> 4;
    const int co = (j >> 1) * 32 + (q >> 2) * 8 + (j & 1) * 4 + (q & 3), t = 4 * s + kg;
    const float* src = t < 36 ? w + ((size_t)co * 36 + t) * 8 : D + ((size_t)co * 28 + t - 36) * 8;
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (xmc_h16)src[c];
    frag[id] = o;
}

// The border pixels of h1 as a small GEMM: per 16 pixels, K = 64 taps x 8 channels, the lane's 16-byte image unit straight from
// global memory into the MFMA B operand (zero where the tap does not apply to the pixel: outside the image, or a correction line of
// a side the pixel is not on).  Weights (64 KB of fragments) in LDS, staged once per workgroup; each wave walks 16-pixel blocks.
__global__ __launch_bounds__(256) void dstem_border_fwd_kernel(const u32x4* __restrict__ img, const u32x4* __restrict__ frag, const float* __restrict__ bias,
                                                              const float* __restrict__ DB, bf16x8* __restrict__ h1, int N, int H, int W, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);               // [4 blocks][16 K steps][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int id = tid; id < 4 * 16 * 64; id += 256) wl[id] = frag[id];
    __syncthreads();
    const int OH = H >> 1, OW = W >> 1, P = 2 * OW + 2 * (OH - 2);
    const long long nblk = ((long long)N * P + 15) / 16;
    const int p = lane & 15, g = lane >> 4;
    float bv[2][8], dbv[2][8][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bv[u][c] = bias[u * 32 + g * 8 + c];
#pragma unroll
            for (int e = 0; e < 8; ++e) dbv[u][c][e] = DB[(u * 32 + g * 8 + c) * 8 + e];
        }
    for (long long blk = (long long)blockIdx.x * 4 + wave; blk < nblk; blk += (long long)gridDim.x * 4) {
        const long long gid = blk * 16 + p;
        const bool live = gid < (long long)N * P;
        const int n = live ? (int)(gid / P) : 0, i = live ? (int)(gid - (long long)n * P) : 0;
        int py, px;
        border_pixel(i, OH, OW, py, px);
        const bool top = py == 0, bot = py == OH - 1, lef = px == 0, rig = px == OW - 1;
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int t = 4 * s + g;
            int sy, sx;
            if (s < 9) { sy = 2 * py - 2 + t / 6; sx = 2 * px - 2 + t % 6; }
            else {
                const int e = t - 36;                          // 0-5 top, 6-11 bottom, 12-17 left, 18-23 right, 24-27 corners
                if (e < 6) { sy = top ? 0 : -1; sx = 2 * px - 2 + e; }
                else if (e < 12) { sy = bot ? H - 1 : -1; sx = 2 * px - 2 + e - 6; }
                else if (e < 18) { sy = 2 * py - 2 + e - 12; sx = lef ? 0 : -1; }
                else if (e < 24) { sy = 2 * py - 2 + e - 18; sx = rig ? W - 1 : -1; }
                else {
                    const bool on = e == 24 ? (top && lef) : e == 25 ? (top && rig) : e == 26 ? (bot && lef) : (bot && rig);
                    sy = on ? ((e & 2) ? H - 1 : 0) : -1;
                    sx = (e & 1) ? W - 1 : 0;
                }
            }
            const bool ok = live && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            const u32x4 bf = ok ? img[((size_t)n * H + sy) * W + sx] : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, wl[(j * 16 + s) * 64 + lane]), __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
        }
        if (!live) continue;
        const float ft = top ? 1.f : 0.f, fb = bot ? 1.f : 0.f, fl = lef ? 1.f : 0.f, fr_ = rig ? 1.f : 0.f;
        bf16x8* dst = h1 + (((size_t)n * OH + py) * OW + px) * 8;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* e = dbv[u][c];
                float v = (c < 4 ? acc[2 * u][c] : acc[2 * u + 1][c - 4]) + bv[u][c];
                v += ft * e[0] + fb * e[1] + fl * e[2] + fr_ * e[3] + ft * fl * e[4] + ft * fr_ * e[5] + fb * fl * e[6] + fb * fr_ * e[7];
                o[c] = (xmc_h16)fmaxf(v, v * slope);
            }
            dst[u * 4 + g] = o;
        }
    }
}

constexpr int kBwImgs = 8;
// one workgroup per (8 images, side): the side's line of dh1 [L][64] and the image line it reads in LDS, one thread per output
__global__ __launch_bounds__(256) void dstem_border_wgrad_kernel(const u32x4* __restrict__ img, const bf16x8* __restrict__ dh1, float* __restrict__ dD,
                                                                float* __restrict__ dDB, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int side = blockIdx.y, tid = threadIdx.x;                       // side: 0 top, 1 bottom, 2 left, 3 right
    const int OH = H >> 1, OW = W >> 1;
    const bool rows = side < 2;
    const int L = rows ? OW : OH, XL = (rows ? W : H) + 4;                 // image line with two zero pixels either side
    float* gl = reinterpret_cast<float*>(smem);                           // [L][64]
    float* xl = gl + (size_t)L * 64;                                      // [XL][4]
    // kBwImgs images per workgroup, summed in registers: one atomic per table entry and workgroup (one per IMAGE was 512 atomics on
    // each address at batch 512, most of this kernel's time)
    constexpr int NA = (64 * 18 + 255) / 256;
    float accD[NA], accB = 0.f, accC[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NA; ++a) accD[a] = 0.f;
    const int n_end = min(N, ((int)blockIdx.x + 1) * kBwImgs);
    for (int n = blockIdx.x * kBwImgs; n < n_end; ++n) {
        __syncthreads();
        for (int id = tid; id < L * 8; id += 256) {
            const int i = id >> 3, ch = id & 7;
            const int py = rows ? (side == 0 ? 0 : OH - 1) : i, px = rows ? i : (side == 2 ? 0 : OW - 1);
            const bf16x8 t = dh1[(((size_t)n * OH + py) * OW + px) * 8 + ch];
#pragma unroll
            for (int k = 0; k < 8; ++k) gl[i * 64 + ch * 8 + k] = (float)t[k];
        }
        for (int id = tid; id < XL; id += 256) {
            const int s_ = id - 2;
            float a = 0.f, b = 0.f, c = 0.f;
            if (s_ >= 0 && s_ < XL - 4) {
                const int sy = rows ? (side == 0 ? 0 : H - 1) : s_, sx = rows ? s_ : (side == 2 ? 0 : W - 1);
                const bf16x8 t = __builtin_bit_cast(bf16x8, img[((size_t)n * H + sy) * W + sx]);
                a = (float)t[0]; b = (float)t[1]; c = (float)t[2];
            }
            xl[id * 4 + 0] = a; xl[id * 4 + 1] = b; xl[id * 4 + 2] = c; xl[id * 4 + 3] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NA; ++a) {                                    // (o, tap t, channel c): sum_i dh1[i][o] * x[2 i - 2 + t][c]
            const int id = tid + a * 256;
            if (id < 64 * 18) {
                const int o = id / 18, r = id - o * 18, t = r / 3, c = r - t * 3;
                float s_ = 0.f;
                for (int i = 0; i < L; ++i) s_ += gl[i * 64 + o] * xl[(2 * i + t) * 4 + c];
                accD[a] += s_;
            }
        }
        if (tid < 64) {
            float s_ = 0.f;
            for (int i = 0; i < L; ++i) s_ += gl[i * 64 + tid];
            accB += s_;
        }
        if (rows) {                                                       // corners: (first | last) pixel of the top / bottom line
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int id = tid + a * 256;
                const int which = id >> 8, o = (id >> 2) & 63, c = id & 3; // which: 0 = left end, 1 = right end
                const int i = which ? L - 1 : 0, xi = which ? XL - 3 : 2;  // image column 0 / W - 1 in the padded line
                const float g_ = gl[i * 64 + o];
                accC[a] += c < 3 ? g_ * xl[xi * 4 + c] : g_;
            }
        }
    }
    const int t0 = side * 6;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int id = tid + a * 256;
        if (id < 64 * 18) {
            const int o = id / 18, r = id - o * 18, t = r / 3, c = r - t * 3;
            atomicAdd(&dD[((size_t)o * 28 + t0 + t) * 8 + c], accD[a]);
        }
    }
    if (tid < 64) atomicAdd(&dDB[tid * 8 + side], accB);
    if (rows) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int id = tid + a * 256;
            const int which = id >> 8, o = (id >> 2) & 63, c = id & 3;
            const int tcorner = 24 + side * 2 + which;
            if (c < 3) atomicAdd(&dD[((size_t)o * 28 + tcorner) * 8 + c], accC[a]);
            else atomicAdd(&dDB[o * 8 + tcorner - 20], accC[a]);
        }
    }
}

}  // namespace

extern "C" int xmc_dstem_pack(const float* wsets, void* wfrag, void* stream) {
    if (!wsets || !wfrag) return XMC_EINVAL;
    hipLaunchKernelGGL(dstem_pack_kernel, dim3((8 * kKSteps * 64 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wsets,
                       reinterpret_cast<bf16x8*>(wfrag));
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_pack_sc(const float* wsets, void* sc_frag, void* stream) {
    if (!wsets || !sc_frag) return XMC_EINVAL;
    hipLaunchKernelGGL(dstem_pack_sc_kernel, dim3(2), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wsets, reinterpret_cast<bf16x8*>(sc_frag));
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_fwd(const void* img, const void* wfrag, const float* bias, void* h1, void* sc, int N, int H, int W, float slope,
                             void* stream) {
    if (!img || !wfrag || !bias || !h1 || N < 1) return XMC_EINVAL;
    if (H < 8 || W < 64 || H % 8 != 0 || W % 64 != 0) return XMC_ESHAPE;
    const int ntiles = N * (H / 8) * (W / 64);
    const int grid = ntiles < 512 ? ntiles : 512;
    if (sc)
        hipLaunchKernelGGL(dstem_fwd_kernel<false>, dim3(xmc_ab_grid(grid)), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const u32x4*>(img),
                           reinterpret_cast<const u32x4*>(wfrag), bias, reinterpret_cast<bf16x8*>(h1), reinterpret_cast<bf16x8*>(sc), N, H, W, slope,
                           ntiles);
    else
        hipLaunchKernelGGL(dstem_fwd_kernel<true>, dim3(xmc_ab_grid(grid)), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const u32x4*>(img),
                           reinterpret_cast<const u32x4*>(wfrag), bias, reinterpret_cast<bf16x8*>(h1), reinterpret_cast<bf16x8*>(sc), N, H, W, slope,
                           ntiles);
    xmc_note_kernel("dstem_fwd_kernel");
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_compose(const float* wi, const float* bi, const float* w0, const float* ws, const float* bs, float* W, float* bias,
                                 float* D, float* DB, void* stream) {
    if (!wi || !bi || !w0 || !ws || !W || !bias || !D || !DB) return XMC_EINVAL;
    ComposeArgs a{wi, bi, w0, ws, bs, W, bias, D, DB};
    constexpr int n = 128 * 36 * 8 + 128 + 64 * 28 * 8 + 64 * 8;
    hipLaunchKernelGGL(dstem_compose_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_compose_bwd(const float* wi, const float* bi, const float* w0, const float* ws, const float* dW, const float* dbias,
                                     const float* dD, const float* dDB, float* dwi, float* dbi, float* dw0, float* dws, float* dbs, void* stream) {
    if (!wi || !bi || !w0 || !ws || !dW || !dbias || !dD || !dDB || !dwi || !dbi || !dw0 || !dws) return XMC_EINVAL;
    ComposeBwdArgs a{wi, bi, w0, ws, dW, dbias, dD, dDB, dwi, dbi, dw0, dws, dbs};
    constexpr int n = 64 * 32 * 16 + 64 * 32 + 64;
    hipLaunchKernelGGL(dstem_compose_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipLaunchKernelGGL(dstem_compose_bwd_wi_kernel, dim3(32 * 27 + 32), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_border_fwd(const void* img, const float* w, const float* bias, const float* D, const float* DB, void* frag_scratch,
                                    void* h1, int N, int H, int W, float slope, void* stream) {
    if (!img || !w || !bias || !D || !DB || !frag_scratch || !h1 || N < 1) return XMC_EINVAL;
    if (H < 16 || W < 64 || H % 16 != 0 || W % 64 != 0) return XMC_ESHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dstem_border_pack_kernel, dim3(16), dim3(256), 0, st, w, D, reinterpret_cast<bf16x8*>(frag_scratch));
    const long long px = (long long)N * (2 * (W / 2) + 2 * (H / 2 - 2));
    const long long nblk = (px + 15) / 16;
    const int grid = (int)(nblk / 4 < 512 ? (nblk + 3) / 4 : 512);
    const size_t lds = (size_t)4 * 16 * 64 * 16;
    XMC_ALLOW_BIG_LDS(dstem_border_fwd_kernel);
    hipLaunchKernelGGL(dstem_border_fwd_kernel, dim3(grid), dim3(256), lds, st, reinterpret_cast<const u32x4*>(img),
                       reinterpret_cast<const u32x4*>(frag_scratch), bias, DB, reinterpret_cast<bf16x8*>(h1), N, H, W, slope);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_border_wgrad(const void* img, const void* dh1, float* dD, float* dDB, int N, int H, int W, void* stream) {
    if (!img || !dh1 || !dD || !dDB || N < 1) return XMC_EINVAL;
    if (H < 16 || W < 64 || H % 16 != 0 || W % 64 != 0 || H > 1024 || W > 1024) return XMC_ESHAPE;
    const int L = (H > W ? H : W) / 2, XL = (H > W ? H : W) + 4;
    const size_t lds = (size_t)L * 64 * 4 + (size_t)XL * 16;
    XMC_ALLOW_BIG_LDS(dstem_border_wgrad_kernel);
    hipLaunchKernelGGL(dstem_border_wgrad_kernel, dim3((N + kBwImgs - 1) / kBwImgs, 4), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const u32x4*>(img), reinterpret_cast<const bf16x8*>(dh1), dD, dDB, N, H, W);
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_dgrad(const void* dh1, const void* dsc, const float* wsets, const float* D, void* frag_scratch, void* dimg, int N, int H,
                               int W, void* stream) {
    if (!dh1 || !dsc || !wsets || !D || !frag_scratch || !dimg || N < 1) return XMC_EINVAL;
    if (H < 16 || W < 64 || H % 16 != 0 || W % 64 != 0) return XMC_ESHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dstem_pack_t_kernel, dim3(9), dim3(256), 0, st, wsets, reinterpret_cast<bf16x8*>(frag_scratch));
    const int ntiles = N * (H / 2 / DG_H) * (W / 2 / DG_W);
    const int grid = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL(dstem_dgrad_kernel, dim3(xmc_ab_grid(grid)), dim3(256), 0, st, reinterpret_cast<const u32x4*>(dh1), reinterpret_cast<const u32x4*>(dsc),
                       reinterpret_cast<const u32x4*>(frag_scratch), reinterpret_cast<bf16x8*>(dimg), N, H, W, ntiles);
    xmc_note_kernel("dstem_dgrad_kernel");
    for (int phase = 0; phase < 2; ++phase) {
        const long long n = (long long)N * 2 * (phase == 0 ? W : H);
        hipLaunchKernelGGL(dstem_border_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const bf16x8*>(dh1), D,
                           reinterpret_cast<bf16x8*>(dimg), N, H, W, phase);
    }
    XMC_LAUNCH_CHECK();
    return 0;
}

extern "C" int xmc_dstem_wgrad(const void* img, const void* dh1, const void* dsc, float* dw, float* dbias, int N, int H, int W,
                               int skip_border, void* stream) {
    if (!img || !dh1 || !dsc || !dw || !dbias || N < 1) return XMC_EINVAL;
    if (H < 16 || W < 64 || H % 16 != 0 || W % 64 != 0) return XMC_ESHAPE;
    const int ntiles = N * (H / (2 * kWgTR)) * (W / 64);
    const int grid = ntiles < 512 ? ntiles : 512;
    const size_t lds = (size_t)kWgTR * 32 * (128 * 2 + 32) + (size_t)(2 * kWgTR + 4) * 2 * 48 * 8;
    XMC_ALLOW_BIG_LDS(dstem_wgrad_kernel);
    hipLaunchKernelGGL(dstem_wgrad_kernel, dim3(xmc_ab_grid(grid)), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const u32x4*>(img),
                       reinterpret_cast<const u32x4*>(dh1), reinterpret_cast<const u32x4*>(dsc), dw, dbias, N, H, W, skip_border, ntiles);
    xmc_note_kernel("dstem_wgrad_kernel");
    XMC_LAUNCH_CHECK();
    return 0;
}
