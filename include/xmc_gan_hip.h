/*
 * xmc_gan_hip.h -- C ABI of libxmc_gan_hip.so (MI355X / gfx950 kernels for the XMC-GAN G+D step).
 *
 * The reference (Eun0/XMC-GAN) has no native layer: its hot path bottoms out in stock ATen calls
 * made from the files under xmc_gan/model/ and from xmc_gan/train_gan.py.  Each entry point below therefore cites the
 * reference call site(s) whose arithmetic it takes over.  The Python host code in xmc-gan_amd/ binds
 * these with ctypes; nothing here knows about torch.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; no allocation or ownership crosses the ABI;
 *   - `stream` is a hipStream_t (void* here so the header needs no HIP include); kernels are only
 *     enqueued, never synchronised, so every call is hipGraph-capturable;
 *   - activations are NHWC, channel count a multiple of 8, element type bf16 (XMC_BF16) or
 *     f32 (XMC_F32); parameters/gradients of parameters are f32;
 *   - return value: 0 on success; XMC_EINVAL / XMC_EALIGN / XMC_ESHAPE (-1 .. -3) for a rejected argument (nothing is
 *     launched in that case); -(1000 + hipError_t) when the HIP runtime refused a launch or a memset.  Never positive.
 */
#ifndef XMC_GAN_HIP_H
#define XMC_GAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (lib.py refuses a library of another version):
 * 2: XmcConvDesc gained dst2 / dst_pool / round_act / groups, alpha applies only with alpha_dev, return codes are
 *    0 / XMC_E* / -(1000 + hipError_t); xmc_half_format() added.
 * 3: XmcConvDesc.post_act / pool_scale (an activation after the residual; the scale of the pooled third output);
 *    xmc_adam_step gained grad_scale.
 * 4: XmcConvDesc.sign_bits / dot (a discriminator block keeps sign bits; d(gamma) from a data-gradient epilogue).
 * 5: xmc_conv_pw1x1_masked_src (the sign-mask pass as a by-product of the shortcut's data gradient).
 * 6: xmc_adam_step_scaled (dynamic loss scale with a found-inf skip); xmc_gp_finish gained inv_s2.
 * 7: xmc_dstem_* (the discriminator's stem composed into one convolution from the image).
 * 8: XmcConvDesc.mask_bits, xmc_conv_ptile_bits / xmc_conv_wgrad_bits (the sign mask applied in the consumers' staging).
 * 9: XmcConvDesc.sc_img / sc_frag / sc_bias, xmc_conv_ptile_scimg, xmc_dstem_pack_sc (the stem block's shortcut recomputed from the image in
 *    its block-end kernel); xmc_dstem_fwd accepts sc == NULL.
 * 10: xmc_set_fixed_order (repeatable reductions, test mode).
 * 11: xmc_set_prezeroed (the caller hands over zero-filled accumulators; the library skips its own memsets);
 *     xmc_word_pool_fwd / _bwd (word-region attention of the repaired concept_gan.InNetG);
 *     XmcConvDesc.splitk_ws / splitk_ws_bytes, xmc_conv_splitk_ws_bytes (split-K for the layers on 4x4 / 8x8 maps);
 *     xmc_concept_query_fwd_multi / _bwd_multi (every sampler stage's sentence query in one launch);
 *     xmc_concept_head_fwd_pre / _bwd_pre, xmc_concept_outer_multi (the heads' sentence products of all stages as one GEMM / one batch product).
 * 12: XmcConvDesc.wpk_lo, XmcPackJob.lo, xmc_conv_pw1x1_split (a learned shortcut's 1x1 convolution on weights held as a 16-bit
 *     hi + lo pair: the precise trunk of the IEEE-half mode);
 *     xmc_gvec_fwd / _bwd, xmc_reasoner_fwd / _bwd, xmc_word_ctx_fwd / _bwd, xmc_word_keys_fwd / _bwd (the per-concept algebra of the
 *     word-attention generators, model/concept_gan.py). */
#define XMC_ABI_VERSION 12

/* XMC_BF16 names the 16-bit storage / MFMA-operand format THIS BUILD of the library was compiled for: bf16 in
 * libxmc_gan_hip.so, IEEE half in libxmc_gan_hip_f16.so (same sources, same entry points; xmc_half_format()). */
enum { XMC_BF16 = 0, XMC_F32 = 1 };
enum { XMC_HALF_IS_BF16 = 0, XMC_HALF_IS_F16 = 1 };
enum { XMC_ACT_NONE = 0, XMC_ACT_LRELU = 1, XMC_ACT_TANH = 2, XMC_ACT_RELU = 3 };
enum { XMC_EINVAL = -1, XMC_EALIGN = -2, XMC_ESHAPE = -3 };

#define XMC_MAX_TAPS 16
#define XMC_MAX_CLASSES 4

/*
 * Tap-table description of one implicit-GEMM convolution pass.  One descriptor covers
 *   forward   F.conv2d / nn.Linear      (df_gan.py:187-188,197,273,276,280,86,157,159; nn.Linear at 73-74,233-240,144)
 *   dgrad     d(loss)/d(input) of those  (the autograd the reference gets from errD.backward()/errG.backward(),
 *                                         train_gan.py:228,251,288)
 * Index space: m -> (n, a, b), a < MH, b < MW.  For tap t of class z the source pixel is
 * (a*SA + dh[z][t], b*SA + dw[z][t]) (zero outside [0,SH<<src_shift) x [0,SW<<src_shift), then >> src_shift:
 * a fused nearest x2 upsample, F.interpolate at df_gan.py:202), the destination pixel is
 * (a*DA + dph[z], b*DA + dpw[z]); the tap's weight slice is wpk[wi[z][t]] : [CDw][CS] (K contiguous).
 * Stride-2 dgrad uses 4 classes (output-pixel parities) of 4 taps each so no MFMA work is wasted on
 * structurally-zero taps.
 * Epilogue:  v = acc + bias[c];  v = act(v);  v = alpha*v (alpha = *alpha_dev or 1);  v += res;  store.
 */
typedef struct XmcConvDesc {
    const void* src;        /* [N,SH,SW,CS] */
    const void* wpk;        /* packed weights [nslices][CDw][CS], same dtype as src */
    void* dst;              /* [N,DH,DW,CD] */
    const float* bias;      /* [CD] or NULL */
    const void* res;        /* residual, dst layout+dtype, or NULL */
    const float* alpha_dev; /* device scalar or NULL */
    int32_t N, SH, SW, CS;
    int32_t DH, DW, CD;
    int32_t MH, MW;
    int32_t SA, DA;
    int32_t src_shift;
    int32_t ntaps, nclass;
    int32_t CDw;            /* rows per weight slice (>= CD, multiple of 32, zero padded) */
    int32_t act;            /* XMC_ACT_* */
    int32_t dtype;          /* XMC_BF16 / XMC_F32: src, wpk */
    int32_t out_dtype;      /* dst, res */
    int8_t dh[XMC_MAX_CLASSES][XMC_MAX_TAPS];
    int8_t dw[XMC_MAX_CLASSES][XMC_MAX_TAPS];
    int8_t wi[XMC_MAX_CLASSES][XMC_MAX_TAPS];
    int8_t dph[XMC_MAX_CLASSES];
    int8_t dpw[XMC_MAX_CLASSES];
    /* Epilogue extensions (all optional; a zeroed tail means "none").  Order: bias, act, alpha, mask, residual.
     *  mask     : dst layout+dtype; the result is multiplied by LeakyReLU'(mask) = (mask > 0 ? 1 : 0.2).  Lets a data
     *             gradient apply the activation mask of the layer below it (the tensor it is the gradient of).
     *  res_mode : 0 = residual has the dst layout; 1 = residual is [N,MH,MW,CD] and indexed by the GEMM row, i.e. with
     *             DA == 2 every residual pixel is added to its 2x2 block of dst (nearest upsample): the adjoint of the
     *             average pool in front of the shortcut (df_gan.py:290) folded into the data gradient of conv_r[0].
     *  res_scale: factor on the residual; 0 is read as 1. */
    const void* mask;
    float res_scale;
    int32_t res_mode;
    /* Second and third outputs of the epilogue (optional), so that a residual block is ONE pass over its last convolution
     * instead of conv + axpby + pool (df_gan.py:284-291: `shortcut + gamma * residual`, then the next block pools its input):
     *  dst2     : dst layout+dtype; receives act(acc + bias), i.e. the value BEFORE alpha / mask / residual -- the residual
     *             branch the backward pass needs (LeakyReLU' mask and d(gamma) = <dout, residual>) while dst gets the block sum.
     *  dst_pool : [N, DH/2, DW/2, CD], dst dtype; receives the 2x2 average of the final dst values (rounded to the dst dtype
     *             first, so it equals F.avg_pool2d of dst).  Needs DA == 1, one class and even DH, DW.
     *  res_mode 2: the residual is [N, DH/2, DW/2, CD] and dst pixel (y, x) reads residual pixel (y/2, x/2): nearest x2
     *             upsample of the generator block's shortcut folded into c2's epilogue (df_gan.py:200-202). */
    void* dst2;
    void* dst_pool;
    /* round_act != 0 (implied by dst2): act(acc + bias) is rounded to the dst dtype before alpha / mask / residual, so that a
     * fused block sum equals, bit for bit, the unfused sequence that stores the residual branch first. */
    int32_t round_act;
    /* groups > 1: wpk is the block-diagonal expansion of an nn.Conv2d(groups=g) weight (xmc_pack_weight_grouped); kernels that
     * know the structure skip the zero blocks (conv_group.hip), the others multiply them.  0 is read as 1. */
    int32_t groups;
    /* post_act (XMC_ACT_NONE / XMC_ACT_LRELU): an activation applied LAST, after alpha / mask / residual -- the LeakyReLU in front
     * of the generator's output convolution (df_gan.py:84-85) applied to the block sum while it is written, so that the sum itself is
     * never stored.  pool_scale: factor of the 2x2 sum written to dst_pool; 0 is read as 0.25 (average pool); 1 gives the adjoint
     * of a nearest x2 upsample (sum pool), the gradient of the half-resolution shortcut of a generator block. */
    int32_t post_act;
    float pool_scale;
    /* sign_bits (uint8 [N, DH, DW, CD/8], dst pixel order): bit r of byte u of a pixel = (act(acc + bias) > 0) for channel 8u + r -- all a
     *             backward pass needs of a LeakyReLU'd residual branch (its mask); replaces dst2 there at 1/16 of the bytes.
     * dot       : f32 device scalar; *dot += sum over the launch of (acc + bias)[e] * mask[e] (the value BEFORE alpha / mask, times the mask
     *             tensor's VALUE): with mask = the input h1 of the layer whose data gradient this is and src = dout * LeakyReLU'(branch),
     *             that is <dout, branch> = d(gamma) of `shortcut + gamma * branch` (df_gan.py:284) without the branch tensor. */
    void* sign_bits;
    float* dot;
    /* mask_bits (ABI 8; honoured by xmc_conv_ptile_bits / xmc_conv_wgrad_bits ONLY -- every other entry rejects a descriptor that sets
     *             it): sign bytes (layout of sign_bits) of the gradient operand -- `src` of a data gradient, `dst` (dy) of a weight
     *             gradient.  The kernel reads operand x LeakyReLU'(bits) (1 or 0.2, rounded to the 16-bit format as
     *             xmc_signmask_apply would store it) while it stages the operand: the masked tensor is never written. */
    const void* mask_bits;
    /* sc_img / sc_frag / sc_bias (ABI 9; honoured by xmc_conv_ptile_scimg ONLY): the block sum's residual operand is not read from
     *             `res` but recomputed per tile from the IMAGE: res = round16(W_B * img + sc_bias), the composed stem's shortcut
     *             (a 4x4 stride-2 pad-1 convolution of the [N, 2 MH, 2 MW, 8] image `sc_img`, xmc_dstem_* above).  `sc_frag`: W_B as
     *             MFMA fragments (xmc_dstem_pack_sc), `sc_bias`: f32 [64].  The shortcut tensor is never written or read. */
    const void* sc_img;
    const void* sc_frag;
    const float* sc_bias;
    /* splitk_ws / splitk_ws_bytes (ABI 11; honoured by the generic implicit-GEMM kernel only): scratch for a split-K launch.  The
     *             layers on 4x4 / 8x8 maps have K = 2304-8192 against M = 4096-16384 output pixels: 128-256 tiles, one workgroup per
     *             CU walking all of K alone, every K step an exposed memory round trip (100-400 TF/s).  With this scratch the kernel
     *             cuts K into S <= 16 ranges (S x the workgroups), each writes its f32 partial tile here ([S][class][M][CDw]) and a
     *             finishing pass sums the S partials IN ORDER and runs the ordinary epilogue -- deterministic, no atomics.
     *             xmc_conv_splitk_ws_bytes(d) says how much the descriptor wants (0: it would not split); NULL / too small = no split */
    void* splitk_ws;
    int64_t splitk_ws_bytes;
    /* wpk_lo (ABI 12; honoured by xmc_conv_pw1x1_split ONLY -- every other entry rejects a descriptor that sets it): a second packed
     *             weight tensor in wpk's layout holding round16(w - round16(w)), the part of the f32 parameter the 16-bit copy lost
     *             (xmc_pack_weight_multi with XmcPackJob.lo).  The kernel issues two MFMAs per K step into one f32 accumulator:
     *             weights at ~20 significant bits for twice the matrix work of a launch that is bound by its HBM stream. */
    const void* wpk_lo;
} XmcConvDesc;

int xmc_abi_version(void);
/* XMC_HALF_IS_BF16 or XMC_HALF_IS_F16: what a tensor of dtype code XMC_BF16 holds in this build */
int xmc_half_format(void);
/* Name (template instantiation, as rocprof prints it) of the convolution kernel the calling thread dispatched last;
 * "" before the first one.  Measurement aid for bench.py's roofline; not part of the reference's surface. */
const char* xmc_last_kernel(void);
/* test mode: the reductions that feed activations (GroupNorm statistics, the attention query gradient) run with one workgroup per
 * reduction target, i.e. in a fixed summation order; returns the previous setting (ABI 10) */
int xmc_set_fixed_order(int on);
/* Accumulators the entry points below document as "zeroed here" -- xmc_groupnorm_fwd / _bwd `ws`, xmc_attn_pool_bwd_acc `dq`,
 * xmc_global_avgpool's f32 `y` on maps of >= 256 pixels,
 * xmc_word_pool_fwd `ctx` / _bwd `dkh`, xmc_concept_query_bwd_multi `dsent`, xmc_concept_outer_multi `dX` -- are cleared with a hipMemsetAsync of their own, one more launch per call
 * (~100 per iteration of an attention-modulation generator).  on = 1: the caller promises they arrive ALREADY ZERO (the Python host
 * carves them from an arena it clears once per iteration) and the library skips those memsets.  Returns the previous setting; process-wide,
 * off by default (ABI 11) */
int xmc_set_prezeroed(int on);

/* bytes of XmcConvDesc.splitk_ws this descriptor would use (0 = the launch would not be split); fill the descriptor first (ABI 11) */
int64_t xmc_conv_splitk_ws_bytes(const XmcConvDesc* d);

/* forward / dgrad implicit GEMM on MFMA (bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_16x16x4_f32) */
int xmc_conv_igemm(const XmcConvDesc* d, void* stream);

/*
 * Weight gradient of the same convolution (train_gan.py:228,251,288 -> conv2d weight grads):
 *   dwp[t][co][ci] += sum_{n,a,b} dy[n,a,b,co] * x[n, a*SA+dh[t], b*SA+dw[t], ci]      (f32, atomically accumulated;
 * the caller zeroes dwp).  x: [N,SH,SW,CS], dy: [N,MH,MW,CD].  Uses class 0 of the tap table.
 */
int xmc_conv_wgrad(const XmcConvDesc* d /* src=x, dst=dy (read only) */, float* dwp, void* stream);
/* same launch, additionally the bias gradient (sum over all pixels of dy[.,co]) without an extra pass over dy.  To keep the
 * atomics off a single cache line, workgroups add into XMC_BIAS_REPLICAS replicas: dbias is f32 [XMC_BIAS_REPLICAS][CD],
 * zeroed by the caller, and the caller sums the replicas. */
#define XMC_BIAS_REPLICAS 16
int xmc_conv_wgrad_bias(const XmcConvDesc* d, float* dwp, float* dbias, void* stream);

/* ---- weight (un)packing between nn.Parameter layout [Co][Ci][KH][KW] f32 and kernel layouts --------------- */
/* fwd pack:  wpk[kh*KW+kw][co][ci]      (rows padded to CDw, cols to CSp, zeros)                               */
/* dgrad pack: wpk[kh*KW+kw][ci][co]     (rows padded to CSw, cols to CDp)                                       */
int xmc_pack_weight(const float* w, void* wpk, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                    int transpose /*0 fwd, 1 dgrad*/, int dtype, const int32_t* row_perm /*NULL or [Co]*/, void* stream);
/* same for an nn.Conv2d(groups=g) weight [Co][Ci/g][KH][KW] (df_concept_gan.py:146,267,546: 16 groups of 8 channels): the packed
 * matrix is the block-diagonal expansion; xmc_unpack_wgrad_grouped reads the diagonal blocks back into gw [Co][Ci/g][KH][KW] */
int xmc_pack_weight_grouped(const float* w, void* wpk, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad, int transpose,
                            int dtype, const int32_t* row_perm, int groups, void* stream);
int xmc_unpack_wgrad_grouped(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                             const float* scale_dev, const int32_t* row_perm, int accumulate, int groups, void* stream);
/* Any number of the packs above (and of xmc_pack_weight_upconv below: upconv != 0, KHW ignored) in one launch per
 * XMC_PACK_MULTI_MAX jobs: the optimizer step re-packs every cached copy of the weights it changed (ops.repack_params). */
#define XMC_PACK_MULTI_MAX 48
typedef struct XmcPackJob {
    const float* w;
    void* wpk;
    const int32_t* row_perm;
    int32_t Co, Ci, KHW, rows_pad, cols_pad, transpose, dtype, groups, upconv;
    int32_t lo;             /* ABI 12: 1 = pack round16(w - round16(w)) (XmcConvDesc.wpk_lo) instead of round16(w); 16-bit dtype only */
} XmcPackJob;
int xmc_pack_weight_multi(const XmcPackJob* jobs /* host array */, int njobs, void* stream);
/* Fused nearest-x2 upsample + 3x3 conv (F.interpolate(scale_factor=2) at df_gan.py:202 followed by the next block's c1, 187):
 * 16 slices wpk[(i*2+j)*4 + th*2+tw][...] of pre-summed weights, one 2x2-tap convolution per output parity (i,j) on the
 * LOW-resolution tensor -> 4/9 of the MACs and no materialised upsampled tensor.  Used with a 4-class tap table. */
int xmc_pack_weight_upconv(const float* w, void* wpk, int Co, int Ci, int rows_pad, int cols_pad, int transpose, int dtype,
                           void* stream);
/* grad unpack: gw[co][ci][kh][kw] (+)= scale * dwp[kh*KW+kw][co][ci] (dwp rows padded to rows_pad, cols to cols_pad) */
int xmc_unpack_wgrad(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                     const float* scale_dev, const int32_t* row_perm, int accumulate, void* stream);
/* same, and gb[c] = sum over the XMC_BIAS_REPLICAS rows of gb_replicas[r][c] (c < CD): the bias gradient xmc_conv_wgrad_bias
 * accumulated, reduced in the same launch */
int xmc_unpack_wgrad_bias(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                          const float* scale_dev, const int32_t* row_perm, int accumulate,
                          const float* gb_replicas, float* gb, int CD, void* stream);
/* same, and *dot += sum_c bias_dot[c] * (the UNSCALED bias gradient c), c < Co: the bias term of d(alpha) of a layer
 * y = alpha * (conv(x) + bias) whose output is not kept (xmc_affine2_act_bwd_dot); bias_dot and dot both NULL = the call above */
int xmc_unpack_wgrad_bias_dot(const float* dwp, float* gw, int Co, int Ci, int KH, int KW, int rows_pad, int cols_pad,
                              const float* scale_dev, const int32_t* row_perm, int accumulate,
                              const float* gb_replicas, float* gb, int CD, const float* bias_dot, float* dot, void* stream);

/* ---- layout conversion at the module boundary (NetD.forward input df_gan.py:127, NetG output df_gan.py:101) -- */
int xmc_nchw_to_nhwc8(const float* src /*[N,C,H,W] f32*/, void* dst /*[N,H,W,8]*/, int N, int C, int H, int W,
                      int dtype, void* stream);
int xmc_nhwc8_to_nchw(const void* src /*[N,H,W,8]*/, float* dst /*[N,C,H,W] f32*/, int N, int C, int H, int W,
                      int dtype, void* stream);

/* ---- pointwise / small reductions (NHWC, C % 8 == 0) ------------------------------------------------------- */
/* y = x > 0 ? x : slope*x                   (nn.LeakyReLU(0.2) df_gan.py:85,158,214-222,274,277; slope 0 = nn.ReLU 234,239) */
int xmc_lrelu(const void* x, void* y, int64_t n, float slope, int dtype, void* stream);
/* dx = ref > 0 ? dy : slope*dy             (ref may be the pre- or post-activation tensor) */
int xmc_lrelu_mask(const void* dy, const void* ref, void* dx, int64_t n, float slope, int dtype, void* stream);
/* y = tanh(x) ; dx = dy * (1 - y^2)         (nn.Tanh df_gan.py:87) */
int xmc_tanh(const void* x, void* y, int64_t n, int dtype, void* stream);
/* dx = dy * LeakyReLU'(branch), the branch given as sign bits (XmcConvDesc.sign_bits layout: byte i = the 8 elements 8i .. 8i+7) */
int xmc_signmask_apply(const void* dy, const void* bits, void* dx, int64_t n, float slope, int dtype, void* stream);
/* A 1x1 convolution (ntaps == 1, unit stride: the discriminator's learned shortcut, df_gan.py:280,286-291, or its data gradient) on the
 * streaming kernels ONLY, writing as a by-product the masked copy of its SOURCE: src_masked = src * LeakyReLU'(src_bits), bits in the
 * XmcConvDesc.sign_bits layout of the source tensor -- the backward of a block reads `dout` once for both.  Returns 1 (nothing
 * launched) when the shape is not one the streaming kernels take; the caller then uses xmc_conv_igemm + xmc_signmask_apply.  (ABI 5) */
int xmc_conv_pw1x1_masked_src(const XmcConvDesc* d, const void* src_bits, void* src_masked, float slope, void* stream);
/* The 1x1 convolution of `d` on the streaming kernels with its weights as the pair (d->wpk, d->wpk_lo) (XmcConvDesc.wpk_lo; the
 * discriminator's learned shortcuts conv_s, df_gan.py:280,286-291, in the IEEE-half mode).  Returns 1 and launches nothing when the
 * shape is not one of theirs (Cin -> Cout 64 -> 128, 128 -> 64, 64 -> 32 from registers, Cin 128 / 256 with Cout >= 128 from LDS;
 * >= 16 k pixels). */
int xmc_conv_pw1x1_split(const XmcConvDesc* d, void* stream);
/* The same masked operand WITHOUT writing it: the data gradient (weights-resident kernel, Cin / Cout <= 64, unit stride) and the 3x3
 * weight gradient (row-reuse kernel, W % 32 == 0, H % 8 == 0) apply d->mask_bits while they stage the gradient operand.  Return 1
 * when the shape is not theirs (the caller then runs xmc_signmask_apply and the plain entry), 0 on success, < 0 on error. */
int xmc_conv_ptile_bits(const XmcConvDesc* d, void* stream);
/* 64 -> 64 channel 3x3 block end (sign bits and / or pooled output) whose residual is d->sc_img's composed shortcut, recomputed per tile
 * (16 extra MFMAs per wave) instead of read from d->res (must be NULL).  1 = not this kernel's shape. */
int xmc_conv_ptile_scimg(const XmcConvDesc* d, void* stream);
int xmc_conv_wgrad_bits(const XmcConvDesc* d, float* dwp, void* stream);
int xmc_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);
/* y = a + (*alpha_dev) * b                  (shortcut + gamma*residual, df_gan.py:200,284) */
int xmc_axpby(const void* a, const void* b, const float* alpha_dev, void* y, int64_t n, int dtype, void* stream);
/* y[n,2h+i,2w+j] = a[n,h,w] + (*alpha_dev) * b[n,2h+i,2w+j]   (a is [N,H,W,C], b and y [N,2H,2W,C]) */
int xmc_axpby_up(const void* a, const void* b, const float* alpha_dev, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* same with LeakyReLU(0.2) applied to the sum (the generator's last block feeds LeakyReLU -> conv_img, df_gan.py:84-86) */
int xmc_axpby_up_lrelu(const void* a, const void* b, const float* alpha_dev, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* y = (*alpha_dev) * x */
int xmc_scale(const void* x, const float* alpha_dev, void* y, int64_t n, int dtype, void* stream);
/* *out (+)= sum(a*b)  (f32 scalar; out zeroed by caller when accumulate==0 is not desired) */
int xmc_dot(const void* a, const void* b, float* out, int64_t n, int dtype, void* stream);
/* Grouped small f32 GEMMs: all conditioning MLPs of the generator (df_gan.py:232-241, two per `affine`, 213-222: four affines
 * per G_Block) in one launch per layer and direction.  Problem g computes C[i,j] = sum_r A(i,r) * B(j,r) for i < M, j < N, r < K
 * with A(i,r) = A[i*sa_i + r*sa_r], B(j,r) = B[j*sb_j + r*sb_r] (element strides, so y = x W^T, dx = dy W and dW = dy^T x are all
 * expressible), C row-major [M,N]; then, by flags: + bias[j]; ReLU; zero where mask[i*N+j] <= 0 (ReLU' of a saved output);
 * atomicAdd into C instead of a store.  rowsum (may be NULL): rowsum[i] = sum_r A(i,r) (bias gradient riding on the weight
 * gradient).  The table is HOST memory (it is passed on in kernel arguments, 32 problems per launch); tile0 is filled in by the
 * call. */
enum { XMC_GP_BIAS = 1, XMC_GP_RELU = 2, XMC_GP_MASK = 4, XMC_GP_ATOMIC = 8 };
typedef struct XmcGemmProblem {
    const float* A;
    const float* B;
    const float* bias;
    const float* mask;
    float* C;
    float* rowsum;
    int32_t M, N, K;
    int32_t sa_i, sa_r, sb_j, sb_r;
    int32_t flags;
    int32_t tile0;
    int32_t reserved;
} XmcGemmProblem;
int xmc_gemm_group(const XmcGemmProblem* problems, int nproblems, void* stream);
/* Frozen text front end, RNN_ENCODER.forward in eval mode (reference model/encoder.py:118-153).
 * xmc_embedding_gather: nn.Embedding lookup (encoder.py:132): out[n, :] = table[ids[n], :], f32, dim % 4 == 0; ids outside
 *   [0, vocab) give a zero row (the host wrapper validates ids before the call).
 * xmc_lstm_bidir: one-layer bidirectional nn.LSTM over captions of length lens[b] (pack_padded_sequence / pad_packed_sequence,
 *   encoder.py:134-137), zero initial state (encoder.py:110-116,130).  xproj [B,T,2,4H]: x_t W_ih^T + b_ih + b_hh per direction
 *   (gate rows i,f,g,o); w_hh [2,4H,H]; words [B,2H,T] (encoder.py:140: outputs transposed; zero at t >= len);
 *   sent [B,2H] = [h_fwd(len-1), h_rev(0)] (encoder.py:142-147).  H must be 128. */
int xmc_embedding_gather(const int64_t* ids, const float* table, float* out, int64_t n_tokens, int dim, int64_t vocab, void* stream);
/* xmc_gru_bidir: the same for nn.GRU (encoder.py:99-102, TEXT.RNN_TYPE 'GRU'): xproj [B,T,2,3H] = W_i* x + b_i* (+ b_h* for the r and z
 * rows), w_hh [2,3H,H] (rows r, z, n), b_hn [2,H] (the candidate gate's hidden bias, which sits inside the product with r) */
int xmc_gru_bidir(const float* xproj, const float* w_hh, const float* b_hn, const int32_t* lens, float* words, float* sent, int B, int T,
                  int H, void* stream);
int xmc_lstm_bidir(const float* xproj, const float* w_hh, const int32_t* lens, float* words, float* sent, int B, int T, int H,
                   void* stream);
/* Spectral normalisation of a layer weight: the legacy torch.nn.utils.spectral_norm hook the reference's layer factories apply
 * when DISC.SPEC_NORM is set (model/modules.py:3,16-17,31-32).  W: f32 [R,C] row-major (the parameter weight_orig viewed as
 * [out, in*k*k]); u [R], v [C]: the hook's weight_u / weight_v buffers.  training != 0: one power iteration, IN PLACE
 * (v <- normalize(W^T u), u <- normalize(W v), normalize(x) = x / max(|x|, eps)); then sigma = u.(W v).  Leaves
 * sig[0] = 1/sigma, sig[1] = sigma on the device and, when w_eff is not NULL, w_eff = W/sigma [R,C].
 * scratch: >= R + C floats (16-byte aligned). */
int xmc_spectral_sigma(const float* W, float* u, float* v, float* scratch, float* sig, float* w_eff, int R, int C, int training,
                       float eps, void* stream);
/* gradient of L(W/sigma) w.r.t. W with u, v held constant (as the hook does): dW = g*sig[0] - <g,W>*sig[0]^2 * u v^T.
 * g: dL/d(W/sigma) [R,C]; u, v: the vectors sigma was computed with; dot: f32[1] scratch */
int xmc_spectral_bwd(const float* g, const float* W, const float* u, const float* v, const float* sig, float* dot, float* dW,
                     int R, int C, void* stream);
/* g = alpha*dy*LeakyReLU'(ref) and dot += sum(dy*ref) in one pass: backward of `shortcut + gamma*residual` (df_gan.py:284)
 * into a residual branch ending in LeakyReLU (ref = its output) together with d(gamma); dot is f32[1], zeroed by the caller */
int xmc_scale_mask_dot(const void* dy, const void* ref, const float* alpha_dev, void* g, float* dot, int64_t n, int dtype, void* stream);
/* Backward of y = a + alpha*b (up == 0; N,H,W,C = shape of dy) or y = up2(a) + alpha*b (up == 1; N,H,W,C = shape of a, dy and
 * b are [N,2H,2W,C]) in one pass: db = alpha*dy (db == NULL: not written -- the convolution gradients that consume it take alpha as
 * XmcConvDesc.alpha_dev / the unpack scale instead), da = 2x2 sum pool of dy (up only), dot += <dy,b>; dot is f32[1], zeroed by the caller */
int xmc_axpby_bwd(const void* dy, const void* b, const float* alpha_dev, void* db, void* da, float* dot,
                  int N, int H, int W, int C, int up, const void* ymask /* NULL, or y of the LeakyReLU'd form: dy *= LeakyReLU'(y) first
                  (then da is also written in the plain form: the masked dy) */, int dtype, void* stream);
/* out[c] = sum over rows of x[r][c]         (bias gradients) ; out is f32 [C], zeroed by the caller */
int xmc_colsum(const void* x, float* out, int64_t rows, int C, int dtype, void* stream);
/* 2x2 average pool (F.avg_pool2d(x,2) df_gan.py:290) and its adjoint (nearest x2 upsample * scale) */
int xmc_avgpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int xmc_upsample2(const void* x, void* y, int N, int H, int W, int C, float scale, int dtype, void* stream);
/* 2x2 sum pool * scale  (adjoint of nearest upsample, F.interpolate df_gan.py:202) == avgpool2 with scale 4*scale */
int xmc_sumpool2(const void* x, void* y, int N, int H, int W, int C, float scale, int dtype, void* stream);
/* global average over HW -> [N,C] (F.avg_pool2d(x,4) on 4x4 maps df_gan.py:165, train_gan.py:272,275) + adjoint */
int xmc_global_avgpool(const void* x, void* y, int N, int HW, int C, int dtype, int out_dtype, void* stream);
int xmc_global_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, int in_dtype, void* stream);

/*
 * DF-GAN conditional affine pair + LeakyReLU, fused (affine.forward df_gan.py:250-263 twice with
 * LeakyReLU(0.2) after each, as used at df_gan.py:213-216 and 219-222):
 *   y = lrelu( lrelu(x*g0 + b0) * g1 + b1 ),   g*,b* : f32 [N,C] (per sample, per channel)
 * backward returns dx and the four [N,C] f32 reductions (zeroed by the caller; atomically accumulated).
 */
int xmc_affine2_lrelu_fwd(const void* x, const float* g0, const float* b0, const float* g1, const float* b1,
                          void* y, int N, int HW, int C, int dtype, void* stream);
int xmc_affine2_lrelu_bwd(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                          const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                          int N, int HW, int C, int dtype, void* stream);

/* the same with the activation slope as an argument: slope 0 = ReLU, the activation of the word-attention generator's
 * conditional BatchNorm / concept modulation (concept_gan.py:419-421,446-447,496-497,508-509) */
int xmc_affine2_act_fwd(const void* x, const float* g0, const float* b0, const float* g1, const float* b1,
                        void* y, int N, int HW, int C, float slope, int dtype, void* stream);
int xmc_affine2_act_bwd(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                        const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1,
                        int N, int HW, int C, float slope, int dtype, void* stream);
/* ... with another gradient of x (dx_in, dx layout, may alias dx) added on the way out: x feeds the block's shortcut as well
 * (df_gan.py:199-200), and its gradient would otherwise meet this one in a separate add pass */
int xmc_affine2_act_bwd_acc(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                            const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1, const void* dx_in,
                            int N, int HW, int C, float slope, int dtype, void* stream);
/* ... for a consumer y -> sum + alpha * f(y) whose backward hands over dy = d loss / d f's input UNSCALED (df_gan.py:200-202:
 * `shortcut + gamma * c2(y)`): *dot += <dy, y> (y = this node's forward output, recomputed; d(alpha) up to f's bias term) and
 * dy is multiplied by *alpha_dev before use.  alpha_dev and dot both NULL = the call above. */
int xmc_affine2_act_bwd_dot(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                            const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1, const void* dx_in,
                            const float* alpha_dev, float* dot, int N, int HW, int C, float slope, int dtype, void* stream);
/* ... and dx_pool [N, H/2, W/2, C] (optional) = the 2x2 sum pool of the dx this call writes (each dx rounded as stored): the
 * gradient of the half-resolution shortcut of the generator block that produced x (the adjoint of F.interpolate(scale_factor=2),
 * df_gan.py:200-202), written in the same pass */
int xmc_affine2_act_bwd_dot_pool(const void* x, const void* dy, const float* g0, const float* b0, const float* g1,
                                 const float* b1, void* dx, float* dg0, float* db0, float* dg1, float* db1, const void* dx_in,
                                 const float* alpha_dev, float* dot, void* dx_pool, int N, int H, int W, int C, float slope,
                                 int dtype, void* stream);
/* single-stage form of the same kernels: pass g1 = b1 = NULL (and dg1 = db1 = NULL) -> y = lrelu(x*g0 + b0), the
 * concept blocks' modulation `gamma * img_embs + beta` + LeakyReLU (df_concept_gan.py:238-239,250-251) */

/*
 * Attention-modulation blocks (model/df_concept_gan.py).
 * GroupNorm over NHWC with optional fused LeakyReLU (slope < 0: none): nn.GroupNorm at 171,270-271,549-550.
 *   stats: f32 [N][G][2] (mean, rstd), written by fwd, read by bwd;  ws: f32 scratch, [N][C][2] for fwd and
 *   [N][C][2] + [N][G][2] for bwd.
 * Region attention pooling (CondConceptSampler.forward 293-299 / ConceptSampler.forward 570-578):
 *   scores[n,c,p] = scale * <q[n,c,:], key[n,p,c*pk:(c+1)*pk]>, attn = softmax over p, ctx[n,c,:] = sum_p attn * x[n,p,c*px:(c+1)*px]
 *   key [N][HW][ncon*pk], x [N][HW][ncon*px] (activation dtype); q [N][ncon][pk], ctx [N][ncon][px] f32; ncon = 16, pk = 4, px = 8.
 *   The attention weights are not materialised: fwd leaves stats [N][ncon][2] = (max_p score, sum_p exp(score - max)) from which
 *   bwd recomputes them; ws: xmc_attn_pool_ws_floats(N, HW) floats of scratch for fwd.  bwd writes dq, dkey, dx (dkey / dx:
 *   every element exactly once).
 */
int xmc_groupnorm_fwd(const void* x, const float* w, const float* b, void* y, float* stats, float* ws,
                      int N, int HW, int C, int G, float eps, float slope, int dtype, void* stream);
int xmc_groupnorm_bwd(const void* x, const void* dy, const float* w, const float* b, const float* stats, void* dx,
                      float* dw, float* db, float* ws, int N, int HW, int C, int G, float slope, int dtype, void* stream);
int64_t xmc_attn_pool_ws_floats(int N, int HW);
int xmc_attn_pool_fwd(const void* key, const float* q, const void* x, float* stats, float* ctx, float* ws, int N, int HW,
                      int ncon, int pk, int px, float scale, int dtype, void* stream);
int xmc_attn_pool_bwd(const void* key, const float* q, const void* x, const float* stats, const float* ctx, const float* dctx,
                      float* dq, void* dkey, void* dx, int N, int HW, int ncon, int pk, int px, float scale, int dtype, void* stream);
/* same, with another gradient of x (dx_in, dx layout; may alias dx) added on the way out */
int xmc_attn_pool_bwd_acc(const void* key, const float* q, const void* x, const float* stats, const float* ctx, const float* dctx,
                          float* dq, void* dkey, void* dx, const void* dx_in, int N, int HW, int ncon, int pk, int px, float scale,
                          int dtype, void* stream);

/* ---- per-concept algebra of the word-attention generators (model/concept_gan.py; C = 16 concepts, P = 4 state channels, f32) ----
 * Grouped 1x1 convolution of a per-sample vector: y[b,g,o] = bias[g,o] + sum_{i<Is} W[g,o,i] xs[b,i] + sum_{i<Ig} W[g,o,Is+i] xg[b,g,i],
 * W [G][O][Is+Ig]: the gamma / beta heads on cat(global condition, context) without the concatenation (concept_gan.py:346-371,404-418),
 * the samplers' query / value projections (545-580; Is = 0).  bwd: any of dxs / dxg / dW / dbias may be NULL. */
int xmc_gvec_fwd(const float* xs, const float* xg, const float* W, const float* bias, float* y, int B, int G, int O, int Is, int Ig, void* stream);
int xmc_gvec_bwd(const float* xs, const float* xg, const float* W, const float* dy, float* dxs, float* dxg, float* dW, float* dbias,
                 int B, int G, int O, int Is, int Ig, void* stream);
/* ConceptReasoner (concept_gan.py:632-654) on x [B,16,4]: adj = tanh(x We^T), pre = x + adj x, [BatchNorm1d(16) over (batch, state)], relu.
 * bn_w NULL: no normalisation.  training: batch statistics, running statistics updated (momentum, unbiased variance); else the running
 * ones.  y NULL: statistics only (upstream's discarded call, 432).  pre [B,16,4] and stat [32] = (mean, rstd) are kept for the backward. */
int xmc_reasoner_fwd(const float* x, const float* We, const float* bn_w, const float* bn_b, float* run_mean, float* run_var, int training,
                     float momentum, float eps, float* y, float* pre, float* stat, int B, void* stream);
int xmc_reasoner_bwd(const float* x, const float* We, const float* bn_w, const float* bn_b, const float* pre, const float* stat, int batch_stats,
                     const float* dy, float* dx, float* dWe, float* dbn_w, float* dbn_b, int B, void* stream);
/* OutConceptBlock.get_context_embs (concept_gan.py:374-394): st [B,16,4] L2-normalised over the CONCEPT axis, w [B,T,4] over the state axis,
 * cosine scores, masked_fill(pad, -inf), softmax over T (<= 32), ctx = p wd -> [B,16,4]; prob [B,16,T] is kept for the backward.  A caption
 * of padding only gives NaN, as torch.softmax does. */
int xmc_word_ctx_fwd(const float* st, const float* w, const unsigned char* pad, float* ctx, float* prob, int B, int T, void* stream);
int xmc_word_ctx_bwd(const float* st, const float* w, const float* prob, const float* dctx, float* dst, float* dw, int B, int T, void* stream);
/* CondConceptSampler's keys (concept_gan.py:566-575): kraw [B,T,64] (channel = concept * 4 + state) -> [GroupNorm(16) over (state, word)] ->
 * L2 normalisation over the state axis -> kh [B,16,T,4]; stat [B,16,2] = (mean, rstd).  dgnw / dgnb [64] are ACCUMULATED (hand over zeros). */
int xmc_word_keys_fwd(const float* kraw, const float* gnw, const float* gnb, float eps, float* kh, float* stat, int B, int T, void* stream);
int xmc_word_keys_bwd(const float* kraw, const float* gnw, const float* gnb, const float* stat, const float* dkh, float* dkraw, float* dgnw,
                      float* dgnb, int B, int T, void* stream);
/* Word-region attention pooling of the word-attention generator concept_gan.InNetG (reference model/concept_gan.py
 * CondConceptSampler.get_context_embs 532-555; the class is repaired here): every region's query attends over the caption's
 * words.  qmap [N][HW][16*4] (dtype), the grouped 1x1 query projection after its GroupNorm; kh f32 [N][16][T][4], the per-concept word
 * keys L2-normalised over the 4 state channels by the caller; pad u8 [N][T], 1 = padding word (score -inf); T <= 32.
 * ctx f32 [N][16][4] = mean over regions of sum_t softmax_t(<q/|q|, kh_t>) kh_t  (accumulated with atomics: zeroed here).
 * Backward: dq (qmap's layout and dtype) and dkh f32 [N][16][T][4] (zeroed here) from dctx f32 [N][16][4]; the attention is
 * recomputed, nothing but ctx is kept between the passes (ABI 11). */
int xmc_word_pool_fwd(const void* qmap, const float* kh, const unsigned char* pad, float* ctx, int N, int HW, int ncon, int pk, int T,
                      int dtype, void* stream);
int xmc_word_pool_bwd(const void* qmap, const float* kh, const unsigned char* pad, const float* dctx, void* dq, float* dkh, int N, int HW,
                      int ncon, int pk, int T, int dtype, void* stream);

/*
 * Contrastive head (cosine_scores + sent_loss/img_loss, train_gan.py:85-139), fused:
 *   S = normalize(A) normalize(B)^T ; loss = mean_j(-sum_i L_ij logsoftmax_col(S)_ij / np_j)
 *                                         + mean_i(-sum_j L_ij logsoftmax_row(S)_ij / np_i)
 * A,B: f32 [n,D] row-major; labels f32 [n,n] or NULL (identity); inv_num_pos f32 [n] or NULL (1).
 * ws: workspace of xmc_contrastive_ws_bytes(n,D) bytes (kept by the caller until backward has run).
 * fwd writes *loss; bwd writes dA,dB = dloss * dLoss/d{A,B}  (dloss is a device scalar).
 */
int64_t xmc_contrastive_ws_bytes(int n, int D);
int xmc_contrastive_fwd(const float* A, const float* B, const float* labels, const float* inv_num_pos,
                        int n, int D, float* loss, void* ws, void* stream);
int xmc_contrastive_bwd(const float* A, const float* B, const float* labels, const float* inv_num_pos,
                        int n, int D, const float* dloss_dev, void* ws, float* dA, float* dB, void* stream);

/* cosine_scores alone (train_gan.py:85-91), S f32 [n][n]; same workspace size as the fused head */
int xmc_cosine_scores(const float* A, const float* B, int n, int D, float* S, void* ws, void* stream);

/* hinge terms: *out = mean_i relu(1 + sign*x[i*stride])  (train_gan.py:195,204,209);
 * bwd: dx[i*stride] = (*dloss_dev)*sign/n where the hinge is active, else 0 (other elements of dx untouched) */
int xmc_hinge_fwd(const void* x, int stride, float sign, float* out, int64_t n, int dtype, void* stream);
int xmc_hinge_bwd(const void* x, int stride, float sign, const float* dloss_dev, void* dx, int64_t n, int dtype,
                  void* stream);

/* element type conversion between the activation dtypes (n % 8 == 0) */
int xmc_cast(const void* x, void* y, int64_t n, int src_dtype, int dst_dtype, void* stream);

/*
 * Multi-tensor Adam (torch.optim.Adam.step as used at train_gan.py:229,252,289; eps 1e-8, no weight decay):
 * one launch updates every tensor of `table_dev` (DEVICE array).  `chunks_dev` is a DEVICE array of
 * nchunks (tensor index, chunk index) int32 pairs, one per block, chunk = xmc_adam_chunk_elems() elements.
 * `step` is a device counter per tensor holding the number of updates already applied; the launch uses
 * step+1 for the bias corrections and then increments it, so the whole call is hipGraph-replayable.
 * `grad_scale` multiplies every gradient element as it is read (1 = torch.optim.Adam exactly; the IEEE-half mode
 * differentiates LOSS_SCALE * loss and passes 1 / LOSS_SCALE here).
 */
typedef struct XmcAdamEntry {
    float* param;
    const float* grad;
    float* m;        /* exp_avg */
    float* v;        /* exp_avg_sq */
    int32_t* step;
    int64_t n;
} XmcAdamEntry;
int xmc_adam_chunk_elems(void);
int xmc_adam_step(const XmcAdamEntry* table_dev, int ntensors, const int32_t* chunks_dev, int nchunks,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same update under a DYNAMIC loss scale (the IEEE-half mode; the reference is f32 and has none -- the rule is
 * torch.cuda.amp.GradScaler's).  scale_dev: float[2] = {scale, 1 / scale}; flags_dev: int32[4] = {a gradient of this step is
 * not finite, consecutive finite steps, that flag at the last finished step, steps skipped so far}.  `mode` is a set of phases,
 * run in this order: XMC_ADAM_CHECK raises the flag if any gradient element of the table is inf / NaN; XMC_ADAM_UPDATE reads the
 * gradients times 1 / scale and updates -- NOTHING (parameters, moments, step counters) while the flag is up; XMC_ADAM_RESCALE
 * ends an optimizer step: scale x backoff after a skipped step, x growth after `interval` finite steps in a row, flag cleared.
 * (An optimizer with several parameter groups checks all of them before it updates any.)  All on the device: capturable. */
enum { XMC_ADAM_CHECK = 1, XMC_ADAM_UPDATE = 2, XMC_ADAM_RESCALE = 4 };
int xmc_adam_step_scaled(const XmcAdamEntry* table_dev, int ntensors, const int32_t* chunks_dev, int nchunks,
                         float lr, float beta1, float beta2, float eps, float* scale_dev, int32_t* flags_dev,
                         int mode, float growth, float backoff, int interval, void* stream);

/* ---- matching-aware gradient penalty, train_gan.py:241-247:  2 * mean_b ||[d logit / d img_b, d logit / d sent_b]||_2^6 ---------
 * ss[b] += sum_k x[b][k]^2 over an f32 block [B, cols] (cols % 4 == 0; ss zeroed by the caller, blocks accumulate);
 * gp = mean_b ss[b]^3 and coef[b] = 6 ss[b]^2 / B (so that d gp / d x[b][k] = coef[b] * x[b][k]);
 * y = (*g) * coef[b] * x : the gradient of gp w.r.t. one block, scaled by the incoming gradient *g (device scalar). */
int xmc_rows_sumsq(const float* x, float* ss, int B, int64_t cols, void* stream);
int xmc_gp_finish(const float* ss, int B, float* gp, float* coef, float inv_s2, void* stream);   /* inv_s2: the blocks hold s * g; 0 or 1 = unscaled */
int xmc_rows_scale(const float* x, const float* coef, const float* g, float* y, int B, int64_t cols, void* stream);

/* ---- per-sample concept algebra of the sentence-conditioned attention-modulation block (df_concept_gan.py:213-253, 273-326) ----
 * One workgroup per sample.  Parameter-gradient buffers have the parameters' own shapes and must be ZEROED by the caller (the small
 * ones are accumulated over the batch with f32 atomics; the sentence columns of the big ones are written by one thread each).
 * E = TRAIN.NEF (<= 1024).  `hid` [B, 256]: the MLPs' layer-1 pre-activations, produced by head_fwd and consumed by head_bwd.
 * `scratch`: B*64 (query_bwd) / B*260 (head_bwd) floats of workspace.
 *  gquery: the self-attention block's query (555-569): q[b, g*4+o] = GroupNorm_4(Wq[g*4+o, :8] . q0[b, g*8 : g*8+8]), q0 = global
 *         average of the block input.
 *  head with params[10] = sent_linear.weight [4, E] (self-attention block, 471-478; NULL otherwise): the reasoner's states are
 *         re-weighted by softmax over the concepts of <Ws sent, r[g]> before the MLPs; grads[10] receives d(Ws).
 *  query: q[b, g*4+o] = GroupNorm_4(Wq[g*4+o, :] . sent[b])   (CondConceptSampler.query_gconv + gn1; gnw = gnb = NULL: no norm)
 *  head : v = value_gconv(ctx); r = ConceptReasoner(v); gamma / beta = grouped MLP([sent ; r])   (238-253, 291-326)
 *         params / grads: 11 pointers in the order value_gconv.weight [64,8], proj_edge.weight [16,4], then for gamma and for
 *         beta: layer-1 weight [128, E+4], bias [128], layer-2 weight [128, 8], bias [128]; then sent_linear.weight or NULL. */
int xmc_concept_query_fwd(const float* sent, const float* Wq, const float* gnw, const float* gnb, float* q, float* qraw,
                          int B, int E, float eps, void* stream);
int xmc_concept_query_bwd(const float* sent, const float* Wq, const float* gnw, const float* qraw, const float* dq, float* dsent,
                          float* dWq, float* dgnw, float* dgnb, float* scratch, int B, int E, float eps, void* stream);
/* The sentence queries of ALL sampler stages of a generator at once (they depend on nothing but the sentence vector; S <= 32): Wq / gnw / gnb
 * are HOST arrays of S device pointers ([64][E], [64], [64]; gnw[s] = gnb[s] = NULL: no GroupNorm); q, qraw f32 [S][B][64].
 * Backward: dq [S][B][64] -> dsent [B][E] (summed over the stages with atomics: zeroed here, xmc_set_prezeroed), dWq f32 [S][64][E] (written), dgn f32 [S][2][64] = (d gnw, d gnb)
 * (accumulated: zeroed by the caller), scratch f32 [B][S*64].  One launch forward, two backward, instead of one / two per stage (ABI 11) */
int xmc_concept_query_fwd_multi(const float* sent, const float* const* Wq, const float* const* gnw, const float* const* gnb, int S, float* q,
                                float* qraw, int B, int E, float eps, void* stream);
int xmc_concept_query_bwd_multi(const float* sent, const float* const* Wq, const float* const* gnw, int S, const float* qraw, const float* dq,
                                float* dsent, float* dWq, float* dgn, float* scratch, int B, int E, float eps, void* stream);
int xmc_concept_gquery_fwd(const float* q0, const float* Wq, const float* gnw, const float* gnb, float* q, float* qraw, int B,
                           float eps, void* stream);
int xmc_concept_gquery_bwd(const float* q0, const float* Wq, const float* gnw, const float* qraw, const float* dq, float* dq0,
                           float* dWq, float* dgnw, float* dgnb, int B, float eps, void* stream);
int xmc_concept_head_fwd(const float* ctx, const float* sent, const float* const* params, float* gamma, float* beta, float* hid,
                         int B, int E, void* stream);
/* the same with the sentence part of layer 1 computed ahead (for every stage of a generator at once, e.g. by xmc_gemm_group: it depends on
 * nothing but the sentence vector): a_pre f32 [2][B][128], a_pre[t][b][row] = sum_{i < E} W1_t[row][i] * sent[b][i] (ABI 11) */
int xmc_concept_head_fwd_pre(const float* ctx, const float* sent, const float* const* params, const float* a_pre, float* gamma, float* beta,
                             float* hid, int B, int E, void* stream);
int xmc_concept_head_bwd(const float* ctx, const float* sent, const float* hid, const float* const* params, const float* dgamma,
                         const float* dbeta, float* dctx, float* dsent, float* const* grads, float* scratch, int B, int E,
                         void* stream);
/* the same WITHOUT the batch products of layer 1's sentence columns (dW1[:, :E], and dsent's share of them): scratch[:, :256] = d of the
 * layer-1 pre-activations ([B][2*128]) is what the caller collects from every stage and hands to ONE xmc_concept_outer_multi.  grads'
 * W1 entries still receive the concept-state columns and must be zeroed; dsent is written only when params[10] (sent_linear) is set (ABI 11) */
int xmc_concept_head_bwd_pre(const float* ctx, const float* sent, const float* hid, const float* const* params, const float* dgamma,
                             const float* dbeta, float* dctx, float* dsent, float* const* grads, float* scratch, int B, int E, void* stream);
/* D [B][nW*Rp] x X [B][C]:  dW[k][r][c] = sum_b D[b][k*Rp + r] * X[b][c] for c < C (written; row pitch ldw >= C, other columns untouched),
 * dX[b][c] = sum_{k,r} D[b][k*Rp + r] * W[k][r][c] (accumulated with atomics: zeroed here, xmc_set_prezeroed).  W / dW: HOST arrays of
 * nW <= 64 device pointers; Rp even (ABI 11) */
int xmc_concept_outer_multi(const float* D, const float* X, const float* const* W, float* const* dW, int nW, int Rp, float* dX, int B, int C,
                            int ldw, void* stream);

/* ---- the discriminator's stem as one convolution from the image (csrc/dstem.hip) -------------------------------------------------
 * conv_img (df_gan.py:114,127) feeds the first resD block without an activation, so conv_r[0] o conv_img is a 6x6 stride-2 pad-2
 * convolution of the image and conv_s o avg_pool2d o conv_img a 4x4 stride-2 pad-1 one (df_gan.py:272-291).  img [N,H,W,8] in the
 * 16-bit format; composed weights f32 [128][36 taps = ta*6+tb][8] and biases f32 [128]: rows 0-63 give h1 = lrelu(.)
 * [N,H/2,W/2,64], rows 64-127 the shortcut sc [N,H/2,W/2,64].  Exact except for h1's pixels on the image border (conv_r[0] pads
 * conv_img's OUTPUT with zeros): the host recomputes those (ops.DStemBlockFn).  H % 16 == 0, W % 64 == 0.
 *  pack:  composed weights -> wfrag (40 KB, MFMA fragment order, 16-bit: 8 row blocks x 5 K steps of tap pairs x 4 channels);  fwd: h1, sc;
 *  wgrad: dw[128][36][8], dbias[128] (f32, zeroed by the caller) += sums over output pixels of (dh1 | dsc)(pixel) x patch(pixel) and
 *         of (dh1 | dsc); with skip_border the border pixels of dh1 do not contribute. */
/* the composition itself and its adjoint (parameter-sized f32 work, one thread per element): conv_img.weight [32][3][3][3] and .bias
 * [32], conv_r[0].weight [64][32][4][4], conv_s.weight [64][32] and .bias [64] (or NULL) -> the four tables; and the tables'
 * gradients -> the parameters' gradients (written, not accumulated). */
int xmc_dstem_compose(const float* wi, const float* bi, const float* w0, const float* ws, const float* bs, float* W, float* bias, float* D,
                      float* DB, void* stream);
int xmc_dstem_compose_bwd(const float* wi, const float* bi, const float* w0, const float* ws, const float* dW, const float* dbias,
                          const float* dD, const float* dDB, float* dwi, float* dbi, float* dw0, float* dws, float* dbs, void* stream);
int xmc_dstem_pack(const float* wsets, void* wfrag, void* stream);
/* rows 64..127 of the composed weights (the shortcut: its 4x4 window inside the 6x6 one) -> sc_frag (8 KB: 4 K steps x 2 row blocks of
 * 32x32x16 A fragments, rows permuted as the block-end kernel's weight rows) */
int xmc_dstem_pack_sc(const float* wsets, void* sc_frag, void* stream);
int xmc_dstem_fwd(const void* img, const void* wfrag, const float* bias, void* h1, void* sc, int N, int H, int W, float slope, void* stream);
int xmc_dstem_wgrad(const void* img, const void* dh1, const void* dsc, float* dw, float* dbias, int N, int H, int W, int skip_border,
                    void* stream);
/* h1's border pixels (conv_r[0]'s zero padding of conv_img's output): correction tables D f32 [64][28][8] (taps 0-5 first row by
 * window column, 6-11 last row, 12-17 first column by window row, 18-23 last column, 24-27 corners) and DB f32 [64][8] (their
 * constant terms), linear in the parameters like the composed weights.  border_fwd recomputes h1 on the border from the image with
 * w (f32 [128][36][8], rows 0-63 used) + D; border_wgrad: dD, dDB (zeroed by the caller) += sums over border pixels. */
int xmc_dstem_border_fwd(const void* img, const float* w, const float* bias, const float* D, const float* DB, void* frag_scratch /* 64 KB */,
                         void* h1, int N, int H, int W, float slope, void* stream);
int xmc_dstem_border_wgrad(const void* img, const void* dh1, float* dD, float* dDB, int N, int H, int W, void* stream);
/* gradient of the image: dimg [N,H,W,8] (16-bit format; channels 3-7 written as zero) = W^T (dh1 | dsc) + D^T dh1 on the border: one
 * 3x3 convolution over the low-resolution gradient map whose MFMA rows are (image channel, output-parity class), then the border
 * corrections added to image rows 0 / H-1 and columns 0 / W-1.  frag_scratch: 36 KB. */
int xmc_dstem_dgrad(const void* dh1, const void* dsc, const float* wsets, const float* D, void* frag_scratch, void* dimg, int N, int H, int W,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XMC_GAN_HIP_H */
