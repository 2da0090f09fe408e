// Weight gradient for the wide layers (Cin % 128 == 0, Cout % 128 == 0, bf16), one KERNEL ROW of taps per workgroup.
//
// The per-tap kernel (conv_wgrad.hip) stages 32 KB (64 pixels x 128 dy channels + 64 pixels x 128 x channels) for 128 MFMAs:
// 64 bytes per CU clock at the MFMA rate, three times what a CU draws from L2 (~22 B/clk measured with s_memtime on the
// convolution kernels), so it runs load-bound at ~400 TFLOP/s and every tap re-reads both tensors (rocprof FETCH_SIZE 1.6 GB
// per dispatch for 0.27 GB tensors).  The taps of one kernel row (kh fixed, kw = 0..KW-1) read the SAME dy pixels and x
// pixels that differ by a column shift, so here one workgroup stages the dy tile and the x row segment with its halo once
// and accumulates all KW taps from shifted LDS reads: 34.5 KB per 384 MFMAs (3x3) -- 3x (4x for 4x4) less traffic per flop.
//
//   grid.x = pixel ranges (split-K, f32 atomics into the packed gradient), grid.y = 128x128 (co, ci) blocks, grid.z = kh
//   512 threads = 8 waves as 2 (co) x 4 (ci): a wave owns 64 co x 32 ci x KW taps  (96 / 128 accumulator registers)
//   K step = 64 output pixels = `nseg` row segments of `seg` = min(W, 64) pixels; the x tile holds, per segment,
//   SA*(seg-1)+KW source pixels (stride SA: 1 for the 3x3, 2 for the 4x4 stride-2 layers), zero filled outside the image;
//   with src_shift the source is read through a nearest x2 upsample (the fused upsample + 3x3 operator's weight gradient).
//   Both operands are pixel-major, fragments come from ds_read_b64_tr_b16 exactly as in conv_wgrad.hip; the tap shift is a
//   row offset in the x tile.  LDS is double buffered (one barrier per step), the next step's global loads are in flight
//   during the MFMAs.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WR_BCO = 128, WR_KP = 64, WR_NT = 512;      // BCI (ci block) is a template parameter: 128, or 64 for Cin == 64
constexpr int WR_LD = 128 + 16;                   // dy tile row stride in elements: 72 dwords, rows 0..7 land 8 banks apart
// x tile row stride: consecutive K pixels are SA rows apart, and the 32 lanes one tr16 read serves together (pixels
// 0..7) must land on 8 distinct bank octets: SA == 1: 72 dwords; SA == 2: 68 dwords (2 rows = 136 = 8 mod 64)
// (a 64-channel ci block keeps rows of 64 + 16 / 64 + 8 elements: 40 / 36 dwords, the same octet spread)
template <int SA, int BCI> struct XLd { static constexpr int v = SA == 1 ? BCI + 16 : BCI + 8; };

struct RowCfg {
    int seg, nseg, xseg, xrows;                   // segment length, segments per K step, x rows per segment / per step
    int pad;                                      // padding of the convolution (source offset of tap 0)
    int lds_bytes;
};

// The 64-channel ci block (resD block 1's 4x4 stride-2 layer, Cin = 64) has half the MFMAs per staged dy tile and runs at 660 TF/s
// against the 128-wide form's 980, its K steps waiting on the L2 -> LDS path (HBM traffic is ideal: the four kernel rows' re-reads hit
// L2).  Round 4 tried two workgroups per CU for it (74 KB of LDS each with the narrow rows): the 128-register cap that needs spills 25
// registers in the K loop, 0.83 -> 1.01 ms.  Measured and not kept; the narrow rows and the packed staging stayed.
template <int KW, int SA, int WR_BCI>
__global__ __launch_bounds__(WR_NT) void wgrad_row_kernel(const XmcConvDesc d, const RowCfg t, float* __restrict__ dwp,
                                                          float* __restrict__ dbias, int pix_per_block) {
    constexpr int NT = WR_NT, KP = WR_KP, LD = WR_LD, LDX = XLd<SA, WR_BCI>::v;
    constexpr int TM = 4, TN = WR_BCI / 64;       // 16x16 tiles per wave: 64 co x 32 ci (x 16 ci for the 64-wide ci block)
    constexpr int XCH = WR_BCI / 8;               // 16-byte chunks per x pixel row
    constexpr int XRP = NT / XCH;                 // x rows staged per pass: XCH threads per row (every thread loads, also for 64 channels)
    constexpr int XIT = ((SA * 63 + KW) + XRP - 1) / XRP + 1;         // x chunks per thread per step (upper bound over seg)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_seg[2][16];                  // per (step parity, segment): source pixel index of (n, row, col 0) or -1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int kh = blockIdx.z;
    const int nci = d.CS / WR_BCI;
    const int co0 = (blockIdx.y / nci) * WR_BCO, ci0 = (blockIdx.y % nci) * WR_BCI;
    const int MHW = d.MH * d.MW;
    const int64_t P = (int64_t)d.N * MHW;
    const int64_t p_begin = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p_end = p_begin + pix_per_block < P ? p_begin + pix_per_block : P;
    if (p_begin >= P) return;
    const int nstep = (int)((p_end - p_begin) / KP);
    const int cs_ch = d.CS >> 3, cd_ch = d.CD >> 3;
    const u32x4* __restrict__ x16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ dy16 = reinterpret_cast<const u32x4*>(d.dst);
    const int dy_bytes = KP * LD * 2, buf_bytes = dy_bytes + t.xrows * LDX * 2;

    // segment table for one step: thread s < nseg -> source pixel index of the segment's row (column 0), -1 if outside
    auto seg_entry = [&](int64_t p0, int s) -> int {
        const int64_t ps = p0 + (int64_t)s * t.seg;
        const int q = (int)(ps / d.MW);           // global output row n*MH + a
        const int n = q / d.MH, a = q - n * d.MH;
        const int sh = a * SA + kh - t.pad;       // row at the (possibly x2-upsampled: src_shift) resolution the taps address
        return (unsigned)sh < (unsigned)(d.SH << d.src_shift) ? (n * d.SH + (sh >> d.src_shift)) * d.SW : -1;
    };
    if (tid < t.nseg) {
        s_seg[0][tid] = seg_entry(p_begin, tid);
        s_seg[1][tid] = nstep > 1 ? seg_entry(p_begin + KP, tid) : -1;
    }
    // staging coordinates
    const int dch = tid & 15, drow0 = tid >> 4;   // dy: 16 chunks per pixel row, rows drow0 and drow0 + 32
    const int xch = tid % XCH, xr0 = tid / XCH;
    int xs[XIT], xj[XIT];                         // x: LDS row r = tid/XCH + XRP*it -> segment, position in segment
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int r = xr0 + XRP * it;
        xs[it] = r < t.xrows ? r / t.xseg : -1;
        xj[it] = r - (r / t.xseg) * t.xseg;
    }
    const bool do_bias = dbias != nullptr && kh == 0 && ci0 == 0;
    float bsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;

    u32x4 ro[2], ri[XIT];
    auto load_tiles = [&](int64_t p0, int par) {
        const int b0 = (int)(p0 % d.MW);          // first output column of segment 0 (0 unless W > 64)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t p = p0 + drow0 + 32 * i;
            ro[i] = dy16[(size_t)p * cd_ch + (co0 >> 3) + dch];
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            u32x4 v = {0, 0, 0, 0};
            if (xs[it] >= 0) {
                const int base = s_seg[par][xs[it]];
                const int col = (t.nseg == 1 ? b0 : 0) * SA + xj[it] - t.pad;
                if (base >= 0 && (unsigned)col < (unsigned)(d.SW << d.src_shift))
                    v = x16[(size_t)(base + (col >> d.src_shift)) * cs_ch + (ci0 >> 3) + xch];
            }
            ri[it] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* sdy = smem + buf * buf_bytes;
        unsigned char* sx = sdy + dy_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(sdy + ((drow0 + 32 * i) * LD + dch * 8) * 2) = ro[i];
            if (do_bias) {
                bf16x8 h = __builtin_bit_cast(bf16x8, ro[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += (float)h[k];
            }
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (xs[it] >= 0) *reinterpret_cast<u32x4*>(sx + ((xr0 + XRP * it) * LDX + xch * 8) * 2) = ri[it];
    };

    f32x4 acc[KW][TM][TN];
#pragma unroll
    for (int w = 0; w < KW; ++w)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[w][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing (as conv_wgrad.hip): lane 4q+pp of a 16-lane group reads pixel (32*ks + 4*fg + q [+16]),
    // columns base + 4*pp .. +3.  x rows: pixel k of the step sits at (k/seg)*xseg + SA*(k%seg) (+ kw for tap kw).
    const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, pp = fr & 3;
    int dyoff[2], xoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int k = ks * 32 + 4 * fg + q;
        dyoff[ks] = (k * LD + wm * 64 + 4 * pp) * 2;
        xoff[ks] = (((k / t.seg) * t.xseg + SA * (k % t.seg)) * LDX + wn * (WR_BCI / 4) + 4 * pp) * 2;
    }
    // pixel k+16: 16 columns on in a long segment; 16/seg segments on (same position) when segments are 16, 8 or 4 pixels long
    const int xhi = (t.seg <= 16 ? (16 / t.seg) * t.xseg : 16 * SA) * LDX * 2;

    __syncthreads();                              // segment tables
    load_tiles(p_begin, 0);
    store_tiles(0);
    __syncthreads();
    if (nstep > 1) load_tiles(p_begin + KP, 1);
    for (int step = 0; step < nstep; ++step) {
        const int buf = step & 1;
        // table for step+2 (read after this step's barrier)
        if (tid < t.nseg) s_seg[buf][tid] = step + 2 < nstep ? seg_entry(p_begin + (int64_t)(step + 2) * KP, tid) : -1;
        const unsigned char* sdy = smem + buf * buf_bytes;
        const unsigned char* sx = sdy + dy_bytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* base = sdy + dyoff[ks] + i * 32;
                bf16x4 lo = xmc_ds_read_tr16((base));
                bf16x4 hi = xmc_ds_read_tr16((base + 16 * LD * 2));
                af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int w = 0; w < KW; ++w) {
                bf16x8 bfr[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned char* base = sx + xoff[ks] + (w * LDX + j * 16) * 2;
                    bf16x4 lo = xmc_ds_read_tr16((base));
                    bf16x4 hi = xmc_ds_read_tr16((base + xhi));
                    bfr[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[w][i][j] = XMC_MFMA_16x16x32(af[i], bfr[j], acc[w][i][j], 0, 0, 0);
            }
        }
        if (step + 1 < nstep) store_tiles(buf ^ 1);
        __syncthreads();
        if (step + 2 < nstep) load_tiles(p_begin + (int64_t)(step + 2) * KP, buf);
    }

    if (do_bias) {                                // lanes with equal (lane % 16) hold the same 8 channels
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = bsum[k];
            for (int o = 32; o >= 16; o >>= 1) v += __shfl_xor(v, o, 64);
            const int ch = co0 + (lane & 15) * 8 + k;
            if (lane < 16 && ch < d.CD) atomicAdd(&dbias[(blockIdx.x & (XMC_BIAS_REPLICAS - 1)) * d.CD + ch], v);
        }
    }
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        const int twi = d.wi[0][kh * KW + w];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * 64 + i * 16 + fg * 4 + r;
                    const int ci = ci0 + wn * (WR_BCI / 4) + j * 16 + fr;
                    atomicAdd(&dwp[((size_t)twi * d.CDw + co) * d.CS + ci], acc[w][i][j][r]);
                }
    }
}

template <int KW, int SA, int WR_BCI>
int launch_row(const XmcConvDesc& d, const RowCfg& t, float* dwp, float* dbias, hipStream_t st) {
    XMC_ALLOW_BIG_LDS((wgrad_row_kernel<KW, SA, WR_BCI>));
    const int64_t P = (int64_t)d.N * d.MH * d.MW;
    const int tiles = (d.CDw / WR_BCO) * (d.CS / WR_BCI) * KW;          // KW = KH for the square kernels handled here
    // one workgroup of 8 waves per CU (accumulators: > 128 VGPRs), so a grid of more than 256 workgroups runs in two rounds: with the 3 kernel
    // rows in the grid the tile count is a multiple of 3 and rounding the split UP gave 264 / 288 workgroups (N512 8x8 512->512: 6 x 48),
    // the last 8 / 32 of them alone on the chip for a second full round.  Round down instead (5 x 48 = 240, one round).
    static const bool ceil_split = xmc_debug_off("wrow_ceil_split");
    int64_t nsplit = ceil_split ? (256 + tiles - 1) / tiles : (256 / tiles > 0 ? 256 / tiles : 1);
    const int64_t max_split = P / (4 * WR_KP) > 0 ? P / (4 * WR_KP) : 1;
    if (nsplit > max_split) nsplit = max_split;
    int64_t ppb = (P + nsplit - 1) / nsplit;
    ppb = (ppb + WR_KP - 1) / WR_KP * WR_KP;
    nsplit = (P + ppb - 1) / ppb;
    dim3 grid((unsigned)nsplit, (unsigned)((d.CDw / WR_BCO) * (d.CS / WR_BCI)), (unsigned)KW);
    hipLaunchKernelGGL((wgrad_row_kernel<KW, SA, WR_BCI>), grid, dim3(WR_NT), (size_t)t.lds_bytes, st, d, t, dwp, dbias, (int)ppb);
    xmc_note_kernel("wgrad_row_kernel<%d, %d, %d>", KW, SA, WR_BCI);
    XMC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// 0 = launched, 1 = not this kernel's case, < 0 = error
int xmc_conv_wgrad_row_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream) {
    static const bool off = xmc_debug_off("no_wrow");
    if (off) return 1;
    if (d->dtype != XMC_BF16 || (d->src_shift != 0 && d->SA != 1)) return 1;
    if ((d->CS % 128 != 0 && d->CS != 64) || d->CD % 128 != 0 || d->CDw != d->CD) return 1;
    int kw, sa;
    if (d->ntaps == 9 && d->SA == 1) { kw = 3; sa = 1; }
    else if (d->ntaps == 16 && d->SA == 2) { kw = 4; sa = 2; }
    else return 1;
    // taps must be the regular kw x kw grid, row major, with a common padding
    const int pad = -d->dh[0][0];
    if (pad < 0 || pad > 2) return 1;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->dh[0][t] != t / kw - pad || d->dw[0][t] != t % kw - pad) return 1;
    const int W = d->MW;
    static const bool no_small = xmc_debug_off("no_wrow_small");
    if (W < (no_small ? 16 : 4) || (W & (W - 1)) != 0) return 1;     // 8- and 4-pixel-wide maps: 8 / 16 row segments per K step
    const int64_t P = (int64_t)d->N * d->MH * d->MW;
    if (P % WR_KP != 0 || P < 4 * WR_KP) return 1;
    if ((int64_t)d->N * d->SH * d->SW >= (1ll << 31)) return 1;
    RowCfg t;
    t.seg = W < 64 ? W : 64;
    t.nseg = WR_KP / t.seg;
    t.xseg = sa * (t.seg - 1) + kw;
    t.xrows = t.nseg * t.xseg;
    t.pad = pad;
    const int bci = d->CS == 64 ? 64 : 128;
    t.lds_bytes = 2 * (WR_KP * WR_LD + t.xrows * (sa == 1 ? bci + 16 : bci + 8)) * 2;
    if (t.lds_bytes > XMC_MAX_DYN_LDS) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->CS == 64) return kw == 3 ? launch_row<3, 1, 64>(*d, t, dwp, dbias, st) : launch_row<4, 2, 64>(*d, t, dwp, dbias, st);
    return kw == 3 ? launch_row<3, 1, 128>(*d, t, dwp, dbias, st) : launch_row<4, 2, 128>(*d, t, dwp, dbias, st);
}
