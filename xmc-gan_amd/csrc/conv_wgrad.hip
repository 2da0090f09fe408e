// Weight-gradient convolution for gfx950:
//   dwp[t][co][ci] += sum_p dy[p][co] * x[shift_t(p)][ci]          p = (image, a, b)
// GEMM view per tap: M = co, N = ci, K = pixels.  Both operands are pixel-major in memory (NHWC),
// i.e. K-strided, so the bf16 path builds MFMA fragments with the gfx950 transposing LDS read
// ds_read_b64_tr_b16 (two reads give the 8 consecutive-k values a 16x16x32 fragment needs); the f32
// path uses one scalar LDS read per 16x16x4 operand.  Split-K over pixel ranges (grid.x) with f32
// atomic accumulation into the packed gradient buffer, which the caller zeroes.
//
// Takes over the weight-gradient halves of errD.backward()/d_loss.backward()/errG.backward()
// (train_gan.py:228,251,288) for every conv2d / Linear on the path.
#include "common.h"

extern "C" int xmc_conv_wgrad_bias(const XmcConvDesc* d, float* dwp, float* dbias, void* stream);

namespace {


template <int DT, int BCO, int BCI, int WM, int WN, int KP>
__global__ __launch_bounds__(256) void wgrad_kernel(const XmcConvDesc d, float* __restrict__ dwp, float* __restrict__ dbias, int pix_per_block) {
    constexpr int NT = 256;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int ESZ = DT == XMC_BF16 ? 2 : 4;
    constexpr int EPC = 16 / ESZ;                    // elements per 16-byte chunk
    constexpr int LDO = BCO + 16, LDI = BCI + 16;    // LDS row strides in elements
    constexpr int WTM = BCO / WM, WTN = BCI / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int CHO = BCO / EPC, CHI = BCI / EPC;  // 16-byte chunks per tile row
    using elem_t = typename std::conditional<DT == XMC_BF16, xmc_h16, float>::type;
    __shared__ __attribute__((aligned(16))) elem_t s_dy[KP * LDO];
    __shared__ __attribute__((aligned(16))) elem_t s_x[KP * LDI];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tap = blockIdx.z;
    const int nci = (d.CS + BCI - 1) / BCI;
    const int co0 = (blockIdx.y / nci) * BCO, ci0 = (blockIdx.y % nci) * BCI;
    const int MHW = d.MH * d.MW;
    const int64_t P = (int64_t)d.N * MHW;
    const int64_t p_begin = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p_end = p_begin + pix_per_block < P ? p_begin + pix_per_block : P;
    if (p_begin >= P) return;
    const int tdh = d.dh[0][tap], tdw = d.dw[0][tap], twi = d.wi[0][tap];
    const int SHv = d.SH << d.src_shift, SWv = d.SW << d.src_shift;
    const int cs_ch = d.CS / EPC;                    // chunks per source pixel
    const int cd_ch = d.CD / EPC;
    const u32x4* __restrict__ x16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ dy16 = reinterpret_cast<const u32x4*>(d.dst);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int LO = (KP * CHO + NT - 1) / NT, LI = (KP * CHI + NT - 1) / NT;
    static_assert(NT % CHO == 0 && (CHO & (CHO - 1)) == 0 && CHO <= 64, "bias reduction layout");
    const bool do_bias = dbias != nullptr && tap == 0 && ci0 == 0;
    float bsum[EPC];
#pragma unroll
    for (int k = 0; k < EPC; ++k) bsum[k] = 0.f;
    u32x4 ro[LO], ri[LI];
    auto load_tiles = [&](int64_t p0) {
#pragma unroll
        for (int i = 0; i < LO; ++i) {
            int id = tid + i * NT;
            int row = id / CHO, ch = id - row * CHO;
            int64_t p = p0 + row;
            int cch = co0 / EPC + ch;
            u32x4 z = {0, 0, 0, 0};
            bool ok = id < KP * CHO && p < p_end && cch < cd_ch;
            ro[i] = ok ? dy16[(size_t)p * cd_ch + cch] : z;
        }
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            int id = tid + i * NT;
            int row = id / CHI, ch = id - row * CHI;
            int64_t p = p0 + row;
            u32x4 z = {0, 0, 0, 0};
            bool ok = id < KP * CHI && p < p_end;
            int cch = ci0 / EPC + ch;
            ok = ok && cch < cs_ch;
            size_t off = 0;
            if (ok) {
                int n = (int)(p / MHW), rem = (int)(p - (int64_t)n * MHW);
                int a = rem / d.MW, b = rem - a * d.MW;
                int sh = a * d.SA + tdh, sw = b * d.SA + tdw;
                ok = (unsigned)sh < (unsigned)SHv && (unsigned)sw < (unsigned)SWv;
                off = (((size_t)n * d.SH + (sh >> d.src_shift)) * d.SW + (sw >> d.src_shift)) * cs_ch + cch;
            }
            ri[i] = ok ? x16[off] : z;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < LO; ++i) {
            int id = tid + i * NT;
            int row = id / CHO, ch = id - row * CHO;
            if (id < KP * CHO) *reinterpret_cast<u32x4*>(&s_dy[row * LDO + ch * EPC]) = ro[i];
            if (do_bias) {                           // bias gradient: this thread always holds chunk tid % CHO (NT % CHO == 0)
                if constexpr (DT == XMC_BF16) {
                    bf16x8 h = __builtin_bit_cast(bf16x8, ro[i]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) bsum[k] += (float)h[k];
                } else {
                    f32x4 h = __builtin_bit_cast(f32x4, ro[i]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) bsum[k] += h[k];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            int id = tid + i * NT;
            int row = id / CHI, ch = id - row * CHI;
            if (id < KP * CHI) *reinterpret_cast<u32x4*>(&s_x[row * LDI + ch * EPC]) = ri[i];
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    load_tiles(p_begin);
    for (int64_t p0 = p_begin; p0 < p_end; p0 += KP) {
        __syncthreads();                 // previous step's LDS reads are done
        store_tiles();
        __syncthreads();
        if (p0 + KP < p_end) load_tiles(p0 + KP);
        if constexpr (DT == XMC_BF16) {
          // lane 4q+pp of each 16-lane group addresses pixel row (4*fg + q [+16]), columns base + 4*pp .. +3.  (Any assignment
          // of the 32 k-values of a step to lane groups is valid as long as both operands use it.  Rows 8*fg + q would put
          // lane groups 0 and 1 -- which ds_read_b64_tr_b16 serves in the same LDS cycle -- 8 rows = 576 dwords apart, i.e.
          // on the same banks: a 2-way conflict on every read.  With 4*fg + q the 32 lanes cover rows 0..7 = 8 x 8 dwords.)
          const int q = fr >> 2, pp = fr & 3;
#pragma unroll
          for (int ks = 0; ks < KP / 32; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const xmc_h16* base = &s_dy[(ks * 32 + 4 * fg + q) * LDO + wm * WTM + i * 16 + 4 * pp];
                bf16x4 lo = xmc_ds_read_tr16((base));
                bf16x4 hi = xmc_ds_read_tr16((base + 16 * LDO));
                af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const xmc_h16* base = &s_x[(ks * 32 + 4 * fg + q) * LDI + wn * WTN + j * 16 + 4 * pp];
                bf16x4 lo = xmc_ds_read_tr16((base));
                bf16x4 hi = xmc_ds_read_tr16((base + 16 * LDI));
                bfr[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = XMC_MFMA_16x16x32(af[i], bfr[j], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
            for (int kk = 0; kk < KP / 4; ++kk) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = s_dy[(kk * 4 + fg) * LDO + wm * WTM + i * 16 + fr];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = s_x[(kk * 4 + fg) * LDI + wn * WTN + j * 16 + fr];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    if (do_bias) {                                   // lanes with equal (lane % CHO) hold the same channels
#pragma unroll
        for (int k = 0; k < EPC; ++k) {
            float v = bsum[k];
            for (int o = 32; o >= CHO; o >>= 1) v += __shfl_xor(v, o, 64);
            int ch = co0 + (lane % CHO) * EPC + k;
            if (lane < CHO && ch < d.CD) atomicAdd(&dbias[(blockIdx.x & (XMC_BIAS_REPLICAS - 1)) * d.CD + ch], v);
        }
    }
    // accumulate: D[row = co][col = ci]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int co = co0 + wm * WTM + i * 16 + fg * 4 + r;
                int ci = ci0 + wn * WTN + j * 16 + fr;
                if (co < d.CDw && ci < d.CS) atomicAdd(&dwp[((size_t)twi * d.CDw + co) * d.CS + ci], acc[i][j][r]);
            }
}

template <int DT, int BCO, int BCI, int WM, int WN, int KP>
int launch(const XmcConvDesc& d, float* dwp, float* dbias, hipStream_t st) {
    const int64_t P = (int64_t)d.N * d.MH * d.MW;
    const int nci = (d.CS + BCI - 1) / BCI;
    const int tiles = (d.CDw / BCO) * nci * d.ntaps;
    int64_t nsplit = (1024 + tiles - 1) / tiles;      // ~4 workgroups per CU; 2048 doubled the atomics for nothing (1x1 256->512 @16x16: 0.148 -> 0.111 ms)
    int64_t max_split = (P + 4 * KP - 1) / (4 * KP);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int64_t ppb = (P + nsplit - 1) / nsplit;
    ppb = (ppb + KP - 1) / KP * KP;
    nsplit = (P + ppb - 1) / ppb;
    dim3 grid((unsigned)nsplit, (unsigned)((d.CDw / BCO) * nci), (unsigned)d.ntaps);
    hipLaunchKernelGGL((wgrad_kernel<DT, BCO, BCI, WM, WN, KP>), grid, dim3(256), 0, st, d, dwp, dbias, (int)ppb);
    xmc_note_kernel("wgrad_kernel<%d, %d, %d, %d, %d, %d>", DT, BCO, BCI, WM, WN, KP);
    XMC_LAUNCH_CHECK();
    return 0;
}

template <int DT>
int dispatch(const XmcConvDesc& d, float* dwp, float* dbias, hipStream_t st) {
    const bool wide_ci = d.CS > 32;
    constexpr int KPB = DT == XMC_BF16 ? 64 : 32;     // pixels per K step for the big tiles
    // batch-sized reductions (the conditioning MLPs: 256 "pixels"): small tiles so that the few K steps are spread
    // over 64+ workgroups instead of 16
    static const bool no_small = xmc_debug_off("no_small_m");
    if (!no_small && (int64_t)d.N * d.MH * d.MW <= 1024)
        return wide_ci ? launch<DT, 32, 64, 1, 4, 32>(d, dwp, dbias, st) : launch<DT, 32, 32, 2, 2, 32>(d, dwp, dbias, st);
    if (d.CDw % 128 == 0) {
        if (d.CS % 128 == 0 && DT == XMC_BF16) return launch<DT, 128, 128, 2, 2, KPB>(d, dwp, dbias, st);
        return wide_ci ? launch<DT, 128, 64, 4, 1, KPB>(d, dwp, dbias, st) : launch<DT, 128, 32, 4, 1, KPB>(d, dwp, dbias, st);
    }
    if (d.CDw % 64 == 0) return wide_ci ? launch<DT, 64, 64, 2, 2, KPB>(d, dwp, dbias, st) : launch<DT, 64, 32, 2, 2, 32>(d, dwp, dbias, st);
    return wide_ci ? launch<DT, 32, 64, 1, 4, 32>(d, dwp, dbias, st) : launch<DT, 32, 32, 2, 2, 32>(d, dwp, dbias, st);
}

}  // namespace

int xmc_conv_wgrad_tile_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream);   // conv_wgrad_tile.hip
int xmc_conv_wgrad_row_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream);    // conv_wgrad_row.hip
int xmc_conv_wgrad_up_try(const XmcConvDesc* d, float* dwp, float* dbias, void* stream);     // conv_wgrad_up.hip

extern "C" int xmc_conv_wgrad(const XmcConvDesc* d, float* dwp, void* stream) { return xmc_conv_wgrad_bias(d, dwp, nullptr, stream); }

// same, and additionally dbias[co] += sum over all pixels of dy[.,co] (f32 [CD], zeroed by the caller) when dbias != NULL
extern "C" int xmc_conv_wgrad_bias(const XmcConvDesc* d, float* dwp, float* dbias, void* stream) {
    if (!d || !d->src || !d->dst || !dwp) return XMC_EINVAL;
    if (d->dtype != XMC_BF16 && d->dtype != XMC_F32) return XMC_EINVAL;
    const int esz = xmc_esz(d->dtype);
    if (d->ntaps < 1 || d->ntaps > XMC_MAX_TAPS) return XMC_ESHAPE;
    if ((d->CS * esz) % 16 != 0 || (d->CD * esz) % 16 != 0 || d->CDw % 32 != 0 || d->CDw < d->CD) return XMC_EALIGN;
    if (d->N < 1 || d->MH < 1 || d->MW < 1 || d->SH < 1 || d->SW < 1) return XMC_ESHAPE;
    if (d->src_shift < 0 || d->src_shift > 1 || d->SA < 1) return XMC_ESHAPE;
    if (d->mask_bits || d->sc_img) return XMC_EINVAL;           // only xmc_conv_wgrad_bits / xmc_conv_ptile_scimg honour them
    {
        int rc = xmc_conv_wgrad_up_try(d, dwp, dbias, stream);       // 3x3 through a x2 upsample: 16 low-resolution products
        if (rc != 1) return rc;
        rc = xmc_conv_wgrad_tile_try(d, dwp, dbias, stream);         // all-taps-per-tile kernel for the few-channel layers
        if (rc != 1) return rc;
        rc = xmc_conv_wgrad_row_try(d, dwp, dbias, stream);          // one kernel row of taps per workgroup for the wide layers
        if (rc != 1) return rc;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == XMC_BF16 ? dispatch<XMC_BF16>(*d, dwp, dbias, st) : dispatch<XMC_F32>(*d, dwp, dbias, st);
}
