// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 16-byte vector memory ops).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xmc_gan_hip.h"

// The 16-bit storage / MFMA-operand format is a build parameter of the library (one code path, two builds of these sources):
//   libxmc_gan_hip.so      bf16    (8 significant bits; the format BASELINE.json's configurations name)
//   libxmc_gan_hip_f16.so  IEEE half (11 significant bits, same MFMA rate; `-DXMC_H16_IS_F16`): the precision mode whose
//                          losses stay within 1e-3 of the f32 reference end to end
// XMC_BF16 in the C ABI means "the 16-bit format of this build" (xmc_half_format() reports which).  The typedef names below
// keep their historical bf16 spelling.
#ifdef XMC_H16_IS_F16
typedef _Float16 xmc_h16;
#define XMC_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define XMC_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define XMC_HALF_FORMAT 1
#else
typedef __bf16 xmc_h16;
#define XMC_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define XMC_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define XMC_HALF_FORMAT 0
#endif
typedef __attribute__((ext_vector_type(8))) xmc_h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) xmc_h16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short xmc_s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define XMC_LRELU 0.2f

static inline int xmc_esz(int dtype) { return dtype == XMC_BF16 ? 2 : 4; }

// name of the kernel the calling thread dispatched last (xmc_last_kernel(), used by bench.py's roofline to attribute
// its per-launch HIP-event timings to the instantiation rocprof reports)
void xmc_note_kernel(const char* fmt, ...);
void xmc_note_generic_epi(const char* kernel, int mask);
// true when `token` is listed in the XMC_DEBUG_DISPATCH environment variable (kernel A/B experiments; unset in production)
bool xmc_debug_off(const char* token);
// true in the fixed-order test mode (xmc_set_fixed_order): one workgroup per reduction target in the reductions that feed activations
bool xmc_fixed_order();
hipError_t xmc_zero_acc(void* p, size_t bytes, hipStream_t st);      // memset unless the caller promised zeros (xmc_set_prezeroed)

// HIP errors are reported as -(1000 + code): positive 1 is taken by "not this kernel's case" in the *_try dispatch chain
// (hipErrorInvalidValue == 1 once made a stale error look like "not eligible", and the next kernel in the chain ran as well).
#define XMC_LAUNCH_CHECK()                                     \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return -(1000 + (int)e__);      \
    } while (0)

// Raise a kernel's dynamic-LDS limit to what gfx950 allows.  Static __shared__ of the kernel counts against the same 160 KiB, so
// ask for a little less than all of it, and do not leave a failure behind as the thread's "last error".
#define XMC_MAX_DYN_LDS (160 * 1024 - 1024)
#define XMC_ALLOW_BIG_LDS(kernel)                                                                                              \
    do {                                                                                                                       \
        static bool once__ = false;                                                                                            \
        if (!once__) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                      XMC_MAX_DYN_LDS);                                                                        \
            (void)hipGetLastError();                                                                                           \
            once__ = true;                                                                                                     \
        }                                                                                                                      \
    } while (0)

// ds_read_b64_tr_b16 (transposing LDS read of four 16-bit elements), format-agnostic through the i16 form of the builtin
__device__ __forceinline__ bf16x4 xmc_ds_read_tr16(const void* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xmc_s16x4*)(p)));
}

__device__ __forceinline__ float lrelu_f(float v) { return v > 0.f ? v : XMC_LRELU * v; }
// tanh for results that are stored as bf16: 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp (absolute error ~1e-7, far below
// half a bf16 ulp of the result; libm's tanhf costs ~0.7 ms on the generator's 256x256 output layer).  The f32 parity mode
// keeps tanhf.
__device__ __forceinline__ float tanh_fast(float v) {
    const float e = __builtin_amdgcn_exp2f(v * 2.8853900817779268f);      // exp(2v)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float lrelu_slope(float ref) { return ref > 0.f ? 1.f : XMC_LRELU; }
// one 16-byte unit (8 channels) x LeakyReLU' from its sign byte (bit k set: channel k positive), rounded back to the 16-bit format:
// what xmc_signmask_apply stores, computed where the unit is staged (XmcConvDesc.mask_bits)
__device__ __forceinline__ u32x4 xmc_apply_sign_bits(u32x4 v, unsigned b) {
    bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float f = (float)h[k];
        h[k] = (xmc_h16)(((b >> k) & 1u) ? f : XMC_LRELU * f);
    }
    return __builtin_bit_cast(u32x4, h);
}

// 8 consecutive channels <-> 8 floats, for either storage type
template <int DT> struct Vec8;
template <> struct Vec8<XMC_BF16> {
    static constexpr int BYTES = 16;
    __device__ static __forceinline__ void load(const void* p, size_t idx8, float (&v)[8]) {
        bf16x8 t = reinterpret_cast<const bf16x8*>(p)[idx8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    __device__ static __forceinline__ void store(void* p, size_t idx8, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (xmc_h16)v[i];
        reinterpret_cast<bf16x8*>(p)[idx8] = t;
    }
};
template <> struct Vec8<XMC_F32> {
    static constexpr int BYTES = 32;
    __device__ static __forceinline__ void load(const void* p, size_t idx8, float (&v)[8]) {
        const f32x4* q = reinterpret_cast<const f32x4*>(p) + idx8 * 2;
        f32x4 a = q[0], b = q[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    __device__ static __forceinline__ void store(void* p, size_t idx8, const float (&v)[8]) {
        f32x4* q = reinterpret_cast<f32x4*>(p) + idx8 * 2;
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
        q[0] = a; q[1] = b;
    }
};

// XCD-aware walk of a persistent 1-D grid over `ntiles` tiles.  Workgroups are dealt to the chip's 8 XCDs round-robin by (linear) block
// index and each XCD has its own L2; tiles that are neighbours in the image share halo rows / columns.  With tile = blockIdx.x + k *
// gridDim.x two neighbours never meet in one L2 and every halo comes from memory again.  Here XCD x owns the contiguous tile range
// [x T8, (x+1) T8) and its gridDim.x / 8 workgroups walk it side by side.  (gridDim.x not a multiple of 8: the plain walk -- which is
// also how XMC_DEBUG_DISPATCH=no_xcd_map gets its A/B: the launchers then ask for one workgroup less, xmc_ab_grid.)
struct XcdWalk { int first, step, end; };
__device__ __forceinline__ XcdWalk xmc_xcd_walk(int ntiles) {
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
#ifdef XMC_XCD_HELPER_OFF          /* build-time A/B of the walk outside conv_tile.hip */
    return XcdWalk{b, G, ntiles};
#endif
    if ((G & 7) != 0 || G < 16) return XcdWalk{b, G, ntiles};
    const int T8 = (ntiles + 7) >> 3, base = (b & 7) * T8;
    return XcdWalk{base + (b >> 3), G >> 3, base + T8 < ntiles ? base + T8 : ntiles};
}
static inline int xmc_ab_grid(int g) {
    static const bool off = xmc_debug_off("no_xcd_map");
    return (off && g >= 16 && (g & 7) == 0) ? g - 1 : g;
}

// lane ^ 1 / lane ^ 2 exchange inside a quad as a DPP modifier of a VALU instruction: hipcc lowers __shfl_xor to ds_bpermute_b32, an
// LDS-pipe instruction with its round trip (32 of them per tile in a pooled epilogue, beside the staging traffic)
__device__ __forceinline__ float xmc_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float xmc_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm:[2,3,0,1]
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Shared tail of the convolution epilogues: second output, alpha, LeakyReLU' mask, (scaled, possibly re-indexed) residual, store.
// v holds act(acc + bias).
// (sign_bits / dot: see XmcConvDesc; `dacc` is the caller's running sum for dot, reduced and added once per workgroup)
template <int ODT>
__device__ __forceinline__ void epilogue_tail(const XmcConvDesc& d, size_t idx8, size_t ridx8, float (&v)[8], float alpha, float* dacc = nullptr) {
    if (d.sign_bits) {
        unsigned b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) b |= (v[k] > 0.f ? 1u : 0u) << k;
        reinterpret_cast<unsigned char*>(d.sign_bits)[idx8] = (unsigned char)b;
    }
    if (d.dst2) Vec8<ODT>::store(d.dst2, idx8, v);
    if (ODT == XMC_BF16 && (d.dst2 || d.round_act)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)(xmc_h16)v[k];
    }
    float mk[8];
    if (d.mask) {
        Vec8<ODT>::load(d.mask, idx8, mk);
        if (d.dot && dacc) {                       // before alpha: gamma = 0 must not lose d(gamma)
#pragma unroll
            for (int k = 0; k < 8; ++k) *dacc += v[k] * mk[k];
        }
    }
    if (d.alpha_dev) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= alpha;
    }
    if (d.mask) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= lrelu_slope(mk[k]);
    }
    if (d.res) {
        const float rs = d.res_scale == 0.f ? 1.f : d.res_scale;
        float rr[8];
        Vec8<ODT>::load(d.res, ridx8, rr);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += rs * rr[k];
    }
    if (d.post_act == XMC_ACT_LRELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = lrelu_f(v[k]);
    }
    Vec8<ODT>::store(d.dst, idx8, v);
}

// Epilogue option sets as compile-time bit masks (kernels with a template parameter EPI; -1 = read the descriptor at run time).
// A convolution epilogue is several hundred VALU instructions per wave and tile; with every option a run-time branch per 8-channel
// unit the weights-resident kernel measured 0.535 ms where the same launch with its options folded takes 0.450 (conv_tile.hip), so
// the option sets that occur in the training step get an instantiation each and everything else goes through the generic one.
constexpr int kEpiBias = 1, kEpiLrelu = 2, kEpiRound = 4, kEpiDst2 = 8, kEpiAlpha = 16, kEpiMask = 32, kEpiRes = 64, kEpiPost = 128, kEpiPool = 256,
              kEpiSign = 512, kEpiDot = 1024, kEpiScImg = 2048;     // ScImg: the residual recomputed from the image (XmcConvDesc.sc_img)
constexpr int kEpiGSum = kEpiBias | kEpiRound | kEpiAlpha | kEpiRes;                      // generator c2 + block sum (ops.GBlockEndFn)
constexpr int kEpiDKeep = kEpiLrelu | kEpiDst2 | kEpiAlpha | kEpiRes | kEpiPool;          // discriminator conv_r[2] + block end, kept for a backward
constexpr int kEpiDFwd = kEpiLrelu | kEpiRound | kEpiAlpha | kEpiRes | kEpiPool;          // ... forward only
constexpr int kEpiDLast = kEpiLrelu | kEpiDst2 | kEpiAlpha | kEpiRes;                     // ... last block of a pass (no pooled output)
constexpr int kEpiDLin = kEpiDst2 | kEpiAlpha | kEpiMask | kEpiRes;                      // the block end linearised (MA-GP: ResDBwdFn.backward)
constexpr int kEpiDKeepS = kEpiLrelu | kEpiRound | kEpiSign | kEpiAlpha | kEpiRes | kEpiPool;   // block end kept for a first-order backward: sign bits instead of the branch
constexpr int kEpiDLastS = kEpiLrelu | kEpiRound | kEpiSign | kEpiAlpha | kEpiRes;
constexpr int kEpiDgDot = kEpiMask | kEpiAlpha | kEpiDot;                                 // conv_r[2]'s data gradient with d(gamma) in its epilogue
// the mask of a descriptor, or -1 if it uses an option the masks do not describe (tanh / relu, f32 destination)
static inline int xmc_epi_mask(const XmcConvDesc& d) {
    if ((d.act != XMC_ACT_NONE && d.act != XMC_ACT_LRELU) || d.out_dtype != XMC_BF16) return -1;
    return (d.bias ? kEpiBias : 0) | (d.act == XMC_ACT_LRELU ? kEpiLrelu : 0) | ((d.round_act && !d.dst2) ? kEpiRound : 0) | (d.dst2 ? kEpiDst2 : 0) |
           (d.alpha_dev ? kEpiAlpha : 0) | (d.mask ? kEpiMask : 0) | (d.res ? kEpiRes : 0) | (d.post_act == XMC_ACT_LRELU ? kEpiPost : 0) |
           (d.dst_pool ? kEpiPool : 0) | (d.sign_bits ? kEpiSign : 0) | (d.dot ? kEpiDot : 0) | (d.sc_img ? kEpiScImg : 0);
}

// residual index (in 8-channel units) of destination pixel (n, y, x) for the three residual layouts; (a, b) is the pixel's
// position in the [MH, MW] GEMM grid
__device__ __forceinline__ size_t res_index8(const XmcConvDesc& d, size_t idx8, int n, int y, int x, int a, int b, int ch8) {
    if (d.res_mode == 0) return idx8;
    if (d.res_mode == 1) return ((size_t)(n * d.MH + a) * d.MW + b) * (d.CD >> 3) + ch8;
    return ((size_t)(n * (d.DH >> 1) + (y >> 1)) * (d.DW >> 1) + (x >> 1)) * (d.CD >> 3) + ch8;
}
