// Implicit-GEMM convolution (forward and data-gradient) for gfx950.
//
//   D[m][n] = sum_{tap,ci} A[m][tap,ci] * W[tap][n][ci]        m = (image, a, b), n = destination channel
//
// A is never materialised: each 16-byte unit (8 bf16 / 4 f32 channels of one source pixel of one tap)
// is gathered straight from the NHWC source tensor, zero filled outside the image, into an LDS tile;
// W comes from a pre-packed [tap][n][ci] matrix so both MFMA operands are K-contiguous 16-byte
// fragments.  Work decomposition: one 256-thread workgroup (4 waves) per BM x BN output tile,
// waves arranged WM x WN, each wave a (BM/WM) x (BN/WN) patch of 16x16 MFMA tiles.
// K loop: double-buffered LDS, global loads for step k+1 issued before the MFMAs of step k and
// written to LDS after them (one barrier per step).  LDS rows are 64 B (one K sub-step) with the
// 16-byte chunk XOR-swizzled by g[(row>>2)&3] = {0,2,3,1}, which makes both the ds_write_b128
// staging pattern and the ds_read_b128 fragment pattern (lane -> row l&15, chunk l>>4) conflict
// free for the 4x16 lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
// Epilogue: accumulators -> LDS (f32) -> each thread owns 8 consecutive channels of one pixel ->
// bias / activation / alpha / residual -> one 16-byte (bf16) or 32-byte (f32) coalesced store.
//
// Takes over: F.conv2d / nn.Linear forward at df_gan.py:73-74,86,144,157-159,187-188,197,233-240,
// 273,276,280 and the input-gradient halves of errD.backward()/d_loss.backward()/errG.backward()
// (train_gan.py:228,251,288).
#include "common.h"
#include <stdlib.h>

#ifndef XMC_IGEMM_PIN
#define XMC_IGEMM_PIN 1
#endif

namespace {

template <int DT> struct Mma;
template <> struct Mma<XMC_BF16> {
    __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return XMC_MFMA_16x16x32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<XMC_F32> {
    // 16 bytes = 4 f32 per lane; instruction j contracts k = 4*(lane>>4)+j of A and B alike, so the
    // four instructions together cover the 16 k of the 64-byte row exactly once.
    __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

__device__ __forceinline__ int swz(int row) {  // {0,2,3,1}[(row>>2)&3]
    int q = (row >> 2) & 3;
    return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

template <int BM, int BN, int KSUB>
struct IgemmLds {                                    // LDS plan shared by the kernel and its launcher
    static constexpr int STAGE_U = 2 * KSUB * (BM + BN) * 4;            // 16-byte units
    static constexpr int EP_LD = BN + 4;
    // the f32 staging of the epilogue is done EP_ROWS rows at a time
    static constexpr int EP_ROWS = BN > 128 ? 64 : (BM > 128 ? 128 : BM);
    static constexpr int EP_U = (EP_ROWS * EP_LD * 4 + 15) / 16;
    static constexpr int SMEM_U = STAGE_U > EP_U ? STAGE_U : EP_U;
    static constexpr bool DYNAMIC = SMEM_U * 16 + 4 * XMC_MAX_TAPS > 64 * 1024;      // beyond the static-LDS limit
};

// One output item of the epilogue -- 8 channels `ch..ch+7` of output row m (class cls) -- from its f32 sums: bias, activation,
// residual index, shared tail.  Used by the kernel's own epilogue and by the split-K finishing pass.
template <int ODT>
__device__ __forceinline__ void igemm_epi_item(const XmcConvDesc& d, int cls, int m, int ch, float (&v)[8], float alpha, float* dacc) {
    const int MHW = d.MH * d.MW;
    const int n = m / MHW, rem = m - n * MHW;
    const int a = rem / d.MW, b = rem - a * d.MW;
    const int dph = d.dph[cls], dpw = d.dpw[cls];
    const size_t pix = ((size_t)n * d.DH + a * d.DA + dph) * d.DW + b * d.DA + dpw;
    const size_t idx8 = (pix * d.CD + ch) >> 3;
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
    const size_t ridx8 = res_index8(d, idx8, n, a * d.DA + dph, b * d.DA + dpw, a, b, ch >> 3);
    epilogue_tail<ODT>(d, idx8, ridx8, v, alpha, dacc);
}

// SK: split-K (XmcConvDesc.splitk_ws): blockIdx.y = K range `ks` of gridDim.y; the tile's f32 partial sums go to
// ws[ks][cls][m][CDw] and igemm_splitk_finish_kernel runs the epilogue.
template <int DT, int BM, int BN, int WM, int WN, int KSUB, bool SK = false>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const XmcConvDesc d) {
    constexpr int NT = 64 * WM * WN;                 // 4 waves, or 8 for the 256x256 tile
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    constexpr int RP = NT / 4;                       // tile rows staged per pass (4 threads per 64-byte row piece)
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;      // 16x16 MFMA tiles per wave
    constexpr int AL = BM * 4 / NT;                  // A chunks per thread per sub-step
    constexpr int BL = (BN * 4 + NT - 1) / NT;       // B chunks per thread per sub-step
    using Lds = IgemmLds<BM, BN, KSUB>;
    constexpr int EP_LD = Lds::EP_LD, EP_ROWS = Lds::EP_ROWS;
    extern __shared__ u32x4 smem_dyn[];
    __shared__ u32x4 smem_static[Lds::DYNAMIC ? 1 : Lds::SMEM_U];
    u32x4* const smem = Lds::DYNAMIC ? smem_dyn : smem_static;
    __shared__ int s_tap[XMC_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int cls = blockIdx.z;
    const int MHW = d.MH * d.MW;
    const int M = d.N * MHW;
    // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs, each with a private L2): XCD k walks a
    // contiguous range of M tiles and, inside it, all N tiles of one M tile back to back, so the gathered activation
    // rows (shared by the N tiles and, through the halo, by neighbouring M tiles) are re-read from that XCD's L2.
    int mt, nt;
    {
        const int MT = (M + BM - 1) / BM, NTn = d.CDw / BN;
        const int bid = blockIdx.x, nwg = MT * NTn;
        const int xcd = bid & 7, j = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;                       // bijective for any nwg
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        mt = lin / NTn; nt = lin - mt * NTn;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    constexpr int ESZ = DT == XMC_BF16 ? 2 : 4;
    const int upt = d.CS * ESZ / 16;                 // 16-byte units per tap
    const int ktot = d.ntaps * upt;                  // total units along K
    const int nsub = (ktot + 3) / 4;                 // 64-byte sub-steps
    const int nstep = (nsub + KSUB - 1) / KSUB;

    if (tid < XMC_MAX_TAPS)
        s_tap[tid] = tid < d.ntaps ? ((d.dh[cls][tid] & 0xff) | ((d.dw[cls][tid] & 0xff) << 8) | ((d.wi[cls][tid] & 0xff) << 16)) : 0;
    __syncthreads();

    // ---- per-thread staging coordinates (fixed over the K loop)
    const int c = tid & 3, r0 = tid >> 2;
    int ph[AL], pw[AL], pn[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + r0 + RP * i;
        if (m < M) {
            int n = m / MHW, rem = m - n * MHW;
            int a = rem / d.MW, b = rem - a * d.MW;
            ph[i] = a * d.SA; pw[i] = b * d.SA; pn[i] = n * d.SH;
        } else {
            ph[i] = -(1 << 20); pw[i] = 0; pn[i] = 0;
        }
    }
    const int SHv = d.SH << d.src_shift, SWv = d.SW << d.src_shift;
    const u32x4* __restrict__ src16 = reinterpret_cast<const u32x4*>(d.src);
    const u32x4* __restrict__ w16 = reinterpret_cast<const u32x4*>(d.wpk);

    // the K steps of this workgroup: all of them, or range blockIdx.y of gridDim.y
    const int s_begin = SK ? (int)((int64_t)nstep * blockIdx.y / gridDim.y) : 0;
    const int s_end = SK ? (int)((int64_t)nstep * (blockIdx.y + 1) / gridDim.y) : nstep;
    const int u0 = s_begin * KSUB * 4 + c;           // unit index u = sub*4 + c  ->  (tap, cu)
    int tap = u0 / upt, cu = u0 % upt;
    u32x4 ra[KSUB][AL], rb[KSUB][BL];

    auto load_step = [&](int step) {
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            const bool tapok = tap < d.ntaps;
            const int tv = s_tap[tapok ? tap : 0];
            const int tdh = (int8_t)(tv & 0xff), tdw = (int8_t)((tv >> 8) & 0xff), twi = (tv >> 16) & 0xff;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                int sh = ph[i] + tdh, sw = pw[i] + tdw;
                bool ok = tapok && (unsigned)sh < (unsigned)SHv && (unsigned)sw < (unsigned)SWv;
                size_t off = ((size_t)(pn[i] + (sh >> d.src_shift)) * d.SW + (sw >> d.src_shift)) * upt + cu;
                u32x4 z = {0, 0, 0, 0};
                ra[s][i] = ok ? src16[off] : z;
            }
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                int rn = r0 + RP * j;
                u32x4 z = {0, 0, 0, 0};
                bool ok = tapok && rn < BN;
                size_t off = ((size_t)twi * d.CDw + n0 + rn) * upt + cu;
                rb[s][j] = ok ? w16[off] : z;
            }
            cu += 4;
            while (cu >= upt) { cu -= upt; ++tap; }
        }
        (void)step;
    };
    auto store_step = [&](int buf) {
        u32x4* la = smem + buf * (KSUB * (BM + BN) * 4);
        u32x4* lb = la + KSUB * BM * 4;
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                int r = r0 + RP * i;
                la[(s * BM + r) * 4 + (c ^ swz(r))] = ra[s][i];
            }
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                int r = r0 + RP * j;
                if (r < BN) lb[(s * BN + r) * 4 + (c ^ swz(r))] = rb[s][j];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fc = lane >> 4;        // fragment row / 16-byte chunk
    constexpr bool pin_reads = XMC_IGEMM_PIN;
    load_step(s_begin);
    store_step(0);
    __syncthreads();
    for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        const bool more = step + 1 < s_end;
        if (more) load_step(step + 1);
        const u32x4* la = smem + buf * (KSUB * (BM + BN) * 4);
        const u32x4* lb = la + KSUB * BM * 4;
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            u32x4 af[TM], bf[TN];
            // All fragment reads of the sub-step are issued first (weights, then pixel rows in the order the MFMAs consume
            // them) and the order is pinned: left to itself the scheduler sinks each read to just before its first use and
            // every group of MFMAs then waits a full LDS round trip on lgkmcnt(0).
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int r = wn * WTN + j * 16 + fr;
                bf[j] = lb[(s * BN + r) * 4 + (fc ^ swz(r))];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int r = wm * WTM + i * 16 + fr;
                af[i] = la[(s * BM + r) * 4 + (fc ^ swz(r))];
            }
            if (pin_reads) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<DT>::run(af[i], bf[j], acc[i][j]);
                if (pin_reads) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc -> LDS (f32 [EP_ROWS][BN+4]) -> coalesced channel-vector stores, EP_ROWS rows at a time
    float* ep = reinterpret_cast<float*>(smem);
    const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
    float dacc = 0.f;
    constexpr int CPR = BN / 8;                      // 8-channel chunks per tile row
    for (int half = 0; half < BM / EP_ROWS; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * WTM + i * 16;
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
            int m = m0 + half * EP_ROWS + row, ch = n0 + cc * 8;
            if (m >= M) continue;
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8]);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(&ep[row * EP_LD + cc * 8 + 4]);
            if (SK) {                                // the raw partial sums of this K range
                f32x4* w = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.splitk_ws) +
                                                    (((size_t)blockIdx.y * gridDim.z + cls) * M + m) * d.CDw + ch);
                w[0] = e0; w[1] = e1;
                continue;
            }
            if (ch >= d.CD) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = e0[k]; v[4 + k] = e1[k]; }
            if (d.out_dtype == XMC_BF16) igemm_epi_item<XMC_BF16>(d, cls, m, ch, v, alpha, &dacc);
            else igemm_epi_item<XMC_F32>(d, cls, m, ch, v, alpha, &dacc);
        }
    }
    if (SK) return;
    if (d.dot) {                                     // one atomic per wave: the d(gamma) dot product of XmcConvDesc.dot
        dacc = wave_sum(dacc);
        if (lane == 0) atomicAdd(d.dot, dacc);
    }
}

// split-K finishing pass: out item (m, 8 channels) = epilogue(sum over the S partial tiles, in order)
constexpr int SK_FIN_ITEMS = 4;
__global__ __launch_bounds__(256) void igemm_splitk_finish_kernel(const XmcConvDesc d, int S) {
    const int cls = blockIdx.z;
    const int M = d.N * d.MH * d.MW, C8 = d.CD / 8;
    const float alpha = d.alpha_dev ? *d.alpha_dev : 1.f;
    float dacc = 0.f;
    // SK_FIN_ITEMS items per thread: the d(gamma) dot of XmcConvDesc.dot costs one atomic per WORKGROUP on a single address, and with
    // one item per thread (8192 waves at batch 512) those serialised read-modify-writes took longer than the GEMM itself
    for (int it = 0; it < SK_FIN_ITEMS; ++it) {
        const int64_t id = ((int64_t)blockIdx.x * SK_FIN_ITEMS + it) * 256 + threadIdx.x;
        if (id >= (int64_t)M * C8) break;
        const int m = (int)(id / C8), ch = (int)(id - (int64_t)m * C8) * 8;
        const float* w = reinterpret_cast<const float*>(d.splitk_ws) + ((size_t)cls * M + m) * d.CDw + ch;
        const size_t stride = (size_t)gridDim.z * M * d.CDw;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < S; ++s) {
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(w + s * stride), e1 = *reinterpret_cast<const f32x4*>(w + s * stride + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] += e0[k]; v[4 + k] += e1[k]; }
        }
        if (d.out_dtype == XMC_BF16) igemm_epi_item<XMC_BF16>(d, cls, m, ch, v, alpha, &dacc);
        else igemm_epi_item<XMC_F32>(d, cls, m, ch, v, alpha, &dacc);
    }
    if (d.dot) {
        __shared__ float sdot[4];
        dacc = wave_sum(dacc);
        if ((threadIdx.x & 63) == 0) sdot[threadIdx.x >> 6] = dacc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(d.dot, sdot[0] + sdot[1] + sdot[2] + sdot[3]);
    }
}

// K ranges of a split launch: 0 = this descriptor is not split.  16-bit operands, few output pixels and a deep K (the layers on 4x4 / 8x8
// maps).  Tile: 256x256 (8 waves; half the L2 -> LDS bytes per MAC of the 128x128 tile) when the weights allow it, with S bringing the
// launch to one workgroup per CU; else 128x128 with ~4 workgroups per CU.  Every range keeps >= 4 K steps.
struct SkPlan { int S, big; };
static SkPlan splitk_plan(const XmcConvDesc& d) {
    static const bool off = xmc_debug_off("no_igemm_splitk"), no_big = xmc_debug_off("no_igemm_splitk256");
    SkPlan p = {0, 0};
    if (off || d.dtype != XMC_BF16 || d.CDw % 64 != 0) return p;
    const int64_t M = (int64_t)d.N * d.MH * d.MW, K = (int64_t)d.ntaps * d.CS;
    // measured per layer of the benched step (tests/diag: XMC_PROF_SHAPES): K >= 4096 in one class gains 20-40 % (8x8 -> 4x4 k4s2 and
    // the 3x3 on 4x4 maps at 512 channels, the fused-upsample data gradients at 256); K = 2304 (256 channels, 36 steps) and the
    // four-class stride-2 data gradients do not
    if (M > 16384 || K < 4096 || d.nclass != 1) return p;
    // big: 1 = 256x256 tiles (8 waves), 0 = 128x128, 2 = 128x64 (the logit head's joint convolution, 768 -> 64: 32-96 tiles of 54 steps)
    const int nstep = (int)((K * 2 / 16 + 3) / 4 + 1) / 2;              // KSUB = 2 in every form
    p.big = (d.CDw % 256 == 0 && !no_big) ? 1 : (d.CDw % 128 == 0 ? 0 : 2);
    if (p.big == 1) {                             // ... unless even 8 ranges of 256x256 tiles leave a quarter of the CUs idle (batch 64)
        const int64_t t256 = (M + 255) / 256 * (d.CDw / 256);
        if (t256 * (nstep / 4 < 8 ? nstep / 4 : 8) < 192) p.big = 0;
    }
    const int BMp = p.big == 1 ? 256 : 128, BNp = p.big == 1 ? 256 : (p.big == 0 ? 128 : 64);
    const int64_t tiles = (M + BMp - 1) / BMp * (d.CDw / BNp) * d.nclass;
    int S = (int)((p.big == 1 ? 256 : 1024) / tiles);
    if (S > (p.big == 2 ? 16 : 8)) S = p.big == 2 ? 16 : 8;
    if (S > nstep / 4) S = nstep / 4;
    p.S = S >= 2 ? S : 0;
    return p;
}
static int64_t splitk_bytes(const XmcConvDesc& d, int S) { return (int64_t)S * d.nclass * d.N * d.MH * d.MW * d.CDw * 4; }

template <int BM, int BN, int WM, int WN>
int launch_splitk(const XmcConvDesc& d, int S, hipStream_t st) {
    const int64_t M = (int64_t)d.N * d.MH * d.MW;
    dim3 grid((unsigned)(((M + BM - 1) / BM) * (d.CDw / BN)), (unsigned)S, (unsigned)d.nclass);
    using Lds = IgemmLds<BM, BN, 2>;
    size_t dyn = 0;
    if (Lds::DYNAMIC) {
        dyn = (size_t)Lds::SMEM_U * 16;
        XMC_ALLOW_BIG_LDS((igemm_kernel<XMC_BF16, BM, BN, WM, WN, 2, true>));
    }
    hipLaunchKernelGGL((igemm_kernel<XMC_BF16, BM, BN, WM, WN, 2, true>), grid, dim3(64 * WM * WN), dyn, st, d);
    xmc_note_kernel("igemm_kernel<%d, %d, %d, %d, %d, %d, true>", XMC_BF16, BM, BN, WM, WN, 2);     // (SK = true: the name rocprof prints)
    XMC_LAUNCH_CHECK();
    const int64_t items = M * (d.CD / 8);
    hipLaunchKernelGGL(igemm_splitk_finish_kernel, dim3((unsigned)((items + 256 * SK_FIN_ITEMS - 1) / (256 * SK_FIN_ITEMS)), 1, (unsigned)d.nclass), dim3(256), 0, st, d, S);
    XMC_LAUNCH_CHECK();
    return 0;
}

template <int DT, int BM, int BN, int WM, int WN, int KSUB>
int launch(const XmcConvDesc& d, hipStream_t st) {
    const int64_t M = (int64_t)d.N * d.MH * d.MW;
    dim3 grid((unsigned)(((M + BM - 1) / BM) * (d.CDw / BN)), 1, (unsigned)d.nclass);
    using Lds = IgemmLds<BM, BN, KSUB>;
    size_t dyn = 0;
    if (Lds::DYNAMIC) {
        dyn = (size_t)Lds::SMEM_U * 16;
        XMC_ALLOW_BIG_LDS((igemm_kernel<DT, BM, BN, WM, WN, KSUB>));
    }
    hipLaunchKernelGGL((igemm_kernel<DT, BM, BN, WM, WN, KSUB>), grid, dim3(64 * WM * WN), dyn, st, d);
    xmc_note_kernel("igemm_kernel<%d, %d, %d, %d, %d, %d>", DT, BM, BN, WM, WN, KSUB);
    XMC_LAUNCH_CHECK();
    return 0;
}

template <int DT>
int dispatch(const XmcConvDesc& d, hipStream_t st) {
    static const int variant = xmc_debug_off("igemm128") ? 1 : 0;
    // batch-sized GEMMs (the conditioning MLPs: M = batch, K = N = 256): a 128-wide N tile leaves 4 workgroups on the
    // chip and each walks all of K alone; 32-wide tiles give 4x the workgroups and a 4x shorter critical path
    static const bool no_small = xmc_debug_off("no_small_m");
    if (DT == XMC_BF16 && d.splitk_ws) {             // few output pixels, deep K: K cut into ranges (XmcConvDesc.splitk_ws)
        const SkPlan sp = splitk_plan(d);
        if (sp.S && d.splitk_ws_bytes >= splitk_bytes(d, sp.S))
            return sp.big == 1 ? launch_splitk<256, 256, 2, 4>(d, sp.S, st) : sp.big == 0 ? launch_splitk<128, 128, 2, 2>(d, sp.S, st)
                                                                                             : launch_splitk<128, 64, 4, 1>(d, sp.S, st);
    }
    if (!no_small && (int64_t)d.N * d.MH * d.MW <= 1024) return launch<DT, 128, 32, 4, 1, 2>(d, st);
    if (d.CDw % 128 == 0) {
        const int64_t M = (int64_t)d.N * d.MH * d.MW;
        if (variant == 1) return launch<DT, 128, 128, 2, 2, 2>(d, st);
        // 256x128 tile, each wave a 128x64 patch: 1.33x fewer LDS fragment bytes per MFMA than 64x64 patches (the
        // 128x128 structure measures 600-650 TF/s against a no-global-load ceiling of ~800 TF/s: LDS-read bound)
        // 256x256 tile, 8 waves of the same 128x64 patches: the A rows are shared by twice as many columns, 32 KB instead of
        // 48 KB through the vector-memory path per 2 x (256x128x32) MACs (that path bounds this kernel)
        static const bool no_big = xmc_debug_off("no_igemm256");
        // ... whenever its tiles fill the chip once (512 -> 512 on the 8x8 maps at batch 512: 660 -> 840 TF/s; at batch 256, with
        // 128 tiles, 605 -> 480: those stay on the smaller tiles)
        const int64_t tiles256 = (M + 255) / 256 * (d.CDw / 256) * d.nclass;
        if (DT == XMC_BF16 && (M >= 256 * 256 || tiles256 >= 256) && d.CDw % 256 == 0 && !no_big) return launch<DT, 256, 256, 2, 4, 2>(d, st);
        if (DT == XMC_BF16 && M >= 256 * 256) return launch<DT, 256, 128, 2, 2, 1>(d, st);
        // few output pixels (the 4x4 / 8x8 maps at the end of D, K = 4608-8192): 128-row tiles would leave half the CUs idle
        static const bool no_m64 = xmc_debug_off("no_igemm_m64");
        // deep K on a grid of at most one workgroup per CU (the 4x4 / 8x8 maps: K = 2304-8192, 128-256 tiles): every K step is one
        // exposed memory round trip (~1.1 us measured per 64-deep step, MFMAs 0.1 us of it) and nothing else runs on the CU, so the
        // steps are made twice as deep (KSUB 4: 98 / 131 KB of LDS, which only matters when a second workgroup would have fitted)
        static const bool no_k4 = xmc_debug_off("no_igemm_ksub4");
        const bool deepk = DT == XMC_BF16 && !no_k4 && (int64_t)d.ntaps * d.CS >= 2048;
        const int64_t t128 = (M + 127) / 128 * (d.CDw / 128) * d.nclass;
        if (DT == XMC_BF16 && !no_m64 && t128 < 256) {
            if (deepk && (M + 63) / 64 * (d.CDw / 128) * d.nclass <= 256) return launch<DT, 64, 128, 2, 2, 4>(d, st);
            return launch<DT, 64, 128, 2, 2, 2>(d, st);
        }
        if (deepk && t128 <= 256) return launch<DT, 128, 128, 2, 2, 4>(d, st);
        return launch<DT, 128, 128, 2, 2, 2>(d, st);
    }
    if (d.CDw % 64 == 0) {
        // the joint convolution of the logit head (768 -> 64 on 4x4 maps, K = 6912, 32-96 workgroups): deep steps as above
        static const bool no_k4b = xmc_debug_off("no_igemm_ksub4");
        const int64_t Mb = (int64_t)d.N * d.MH * d.MW;
        if (DT == XMC_BF16 && !no_k4b && (int64_t)d.ntaps * d.CS >= 2048 && (Mb + 127) / 128 * (d.CDw / 64) * d.nclass <= 256)
            return launch<DT, 128, 64, 4, 1, 4>(d, st);
        return launch<DT, 128, 64, 4, 1, 2>(d, st);
    }
    return launch<DT, 128, 32, 4, 1, 2>(d, st);
}

}  // namespace

int xmc_conv_tile_try(const XmcConvDesc* d, void* stream);                            // conv_tile.hip
int xmc_conv_thin_try(const XmcConvDesc* d, void* stream);                            // conv_thin.hip
int xmc_conv_pw1x1_try(const XmcConvDesc* d, void* stream);                           // conv_thin.hip
int xmc_conv_wtile_try(const XmcConvDesc* d, void* stream);                           // conv_wtile.hip
int xmc_conv_wtile3_try(const XmcConvDesc* d, void* stream);                          // conv_wtile3.hip
int xmc_conv_ptile_pool_try(const XmcConvDesc* d, void* stream);                      // conv_tile.hip
int xmc_conv_ptile_slab128_try(const XmcConvDesc* d, void* stream);                   // conv_tile.hip
int xmc_conv_group_try(const XmcConvDesc* d, void* stream);                           // conv_group.hip
int xmc_conv_thin_out_try(const XmcConvDesc* d, void* stream);                        // conv_thin.hip

extern "C" int64_t xmc_conv_splitk_ws_bytes(const XmcConvDesc* d) {
    if (!d) return 0;
    const SkPlan sp = splitk_plan(*d);
    return sp.S ? splitk_bytes(*d, sp.S) : 0;
}

extern "C" int xmc_conv_igemm(const XmcConvDesc* d, void* stream) {
    if (!d || !d->src || !d->wpk || !d->dst) return XMC_EINVAL;
    if (d->dtype != XMC_BF16 && d->dtype != XMC_F32) return XMC_EINVAL;
    if (d->out_dtype != XMC_BF16 && d->out_dtype != XMC_F32) return XMC_EINVAL;
    const int esz = xmc_esz(d->dtype);
    if (d->ntaps < 1 || d->ntaps > XMC_MAX_TAPS || d->nclass < 1 || d->nclass > XMC_MAX_CLASSES) return XMC_ESHAPE;
    if ((d->CS * esz) % 16 != 0 || d->CD % 8 != 0 || d->CDw % 32 != 0 || d->CDw < d->CD) return XMC_EALIGN;
    if (d->N < 1 || d->MH < 1 || d->MW < 1 || d->SH < 1 || d->SW < 1 || d->DH < 1 || d->DW < 1) return XMC_ESHAPE;
    if (d->src_shift < 0 || d->src_shift > 1 || d->SA < 1 || d->DA < 1) return XMC_ESHAPE;
    // destination pixels must stay inside the destination tensor
    for (int z = 0; z < d->nclass; ++z)
        if ((d->MH - 1) * d->DA + d->dph[z] >= d->DH || (d->MW - 1) * d->DA + d->dpw[z] >= d->DW || d->dph[z] < 0 || d->dpw[z] < 0)
            return XMC_ESHAPE;
    if ((int64_t)d->N * d->MH * d->MW >= (1ll << 31)) return XMC_ESHAPE;
    if (d->dst_pool && (d->DA != 1 || d->nclass != 1 || (d->DH & 1) || (d->DW & 1) || d->out_dtype != d->dtype)) return XMC_ESHAPE;
    if (d->post_act != XMC_ACT_NONE && d->post_act != XMC_ACT_LRELU) return XMC_EINVAL;
    if (d->dot && !d->mask) return XMC_EINVAL;                  // the dot is <value before alpha, mask tensor>
    if (d->mask_bits) return XMC_EINVAL;                        // only xmc_conv_ptile_bits / xmc_conv_wgrad_bits apply it
    if (d->sc_img) return XMC_EINVAL;                           // only xmc_conv_ptile_scimg recomputes the residual
    if (d->wpk_lo) return XMC_EINVAL;                           // only xmc_conv_pw1x1_split multiplies a weight pair
    if (d->res_mode < 0 || d->res_mode > 2 || (d->res_mode == 2 && (d->DA != 1 || (d->DH & 1) || (d->DW & 1)))) return XMC_ESHAPE;
    static const bool no_tile = xmc_debug_off("no_tile");
    static const bool no_wt2 = xmc_debug_off("no_wtile_v2");
    static const bool no_wt3 = xmc_debug_off("no_wtile3");
    int rc = 1;
    if (!no_tile && d->dst_pool) {        // kernels that write the pooled third output from their epilogue
        rc = xmc_conv_thin_try(d, stream);
        if (rc > 0 && !no_wt3) rc = xmc_conv_wtile3_try(d, stream);
        if (rc > 0 && !no_wt2) rc = xmc_conv_wtile_try(d, stream);
        if (rc > 0) rc = xmc_conv_ptile_pool_try(d, stream);
        if (rc <= 0) return rc;
    }
    XmcConvDesc dd = *d;                  // every other kernel: plain launch, then the pool as its own pass
    dd.dst_pool = nullptr;
    if (!no_tile) {                       // streaming kernel for 8-channel sources, halo-tile kernels for unit-stride bf16 layers
        rc = xmc_conv_group_try(&dd, stream);
        if (rc > 0) rc = xmc_conv_thin_try(&dd, stream);
        if (rc > 0) rc = xmc_conv_thin_out_try(&dd, stream);
        if (rc > 0) rc = xmc_conv_pw1x1_try(&dd, stream);
        if (rc > 0) rc = xmc_conv_ptile_slab128_try(&dd, stream);
        if (rc > 0 && !no_wt3) rc = xmc_conv_wtile3_try(&dd, stream);
        if (rc > 0 && !no_wt2) rc = xmc_conv_wtile_try(&dd, stream);
        if (rc > 0) rc = xmc_conv_tile_try(&dd, stream);
        if (rc < 0) return rc;
    }
    if (rc > 0) {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        rc = dd.dtype == XMC_BF16 ? dispatch<XMC_BF16>(dd, st) : dispatch<XMC_F32>(dd, st);
        if (rc != 0) return rc;
    }
    if (d->dst_pool) return xmc_sumpool2(d->dst, d->dst_pool, d->N, d->DH, d->DW, d->CD, d->pool_scale == 0.f ? 0.25f : d->pool_scale, d->out_dtype, stream);
    return 0;
}
