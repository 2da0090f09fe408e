"""ctypes binding of libxmc_gan_hip.so (C ABI: include/xmc_gan_hip.h).

The library is the product's only compute path.  Loading fails loudly (RuntimeError) when the
shared object has not been built -- there is no PyTorch/CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XMC_LIB_PATH", os.path.join(HERE, "libxmc_gan_hip.so"))   # override: kernel experiments
LIB_PATH_F16 = os.environ.get("XMC_LIB_PATH_F16", os.path.join(HERE, "libxmc_gan_hip_f16.so"))
ABI_VERSION = 12          # include/xmc_gan_hip.h XMC_ABI_VERSION; the ctypes structures below mirror that version

# BF16 is the dtype code of "the 16-bit storage format of the loaded build": bf16 in libxmc_gan_hip.so, IEEE half in
# libxmc_gan_hip_f16.so (the same sources compiled with -DXMC_H16_IS_F16; `use_variant`)
BF16, F32 = 0, 1
H16 = BF16
ACT_NONE, ACT_LRELU, ACT_TANH, ACT_RELU = 0, 1, 2, 3
MAX_TAPS, MAX_CLASSES = 16, 4

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("src", vp), ("wpk", vp), ("dst", vp), ("bias", vp), ("res", vp), ("alpha_dev", vp),
                ("N", i32), ("SH", i32), ("SW", i32), ("CS", i32),
                ("DH", i32), ("DW", i32), ("CD", i32),
                ("MH", i32), ("MW", i32), ("SA", i32), ("DA", i32), ("src_shift", i32),
                ("ntaps", i32), ("nclass", i32), ("CDw", i32), ("act", i32), ("dtype", i32), ("out_dtype", i32),
                ("dh", (C.c_int8 * MAX_TAPS) * MAX_CLASSES), ("dw", (C.c_int8 * MAX_TAPS) * MAX_CLASSES),
                ("wi", (C.c_int8 * MAX_TAPS) * MAX_CLASSES),
                ("dph", C.c_int8 * MAX_CLASSES), ("dpw", C.c_int8 * MAX_CLASSES),
                ("mask", vp), ("res_scale", f32), ("res_mode", i32), ("dst2", vp), ("dst_pool", vp), ("round_act", i32), ("groups", i32),
                ("post_act", i32), ("pool_scale", f32), ("sign_bits", vp), ("dot", vp), ("mask_bits", vp), ("sc_img", vp), ("sc_frag", vp), ("sc_bias", vp),
                ("splitk_ws", vp), ("splitk_ws_bytes", C.c_int64), ("wpk_lo", vp)]


# XmcGemmProblem as a numpy record (filled vectorised on the host, handed to xmc_gemm_group by pointer)
GEMM_PROBLEM = [("A", "u8"), ("B", "u8"), ("bias", "u8"), ("mask", "u8"), ("C", "u8"), ("rowsum", "u8"),
                ("M", "i4"), ("N", "i4"), ("K", "i4"), ("sa_i", "i4"), ("sa_r", "i4"), ("sb_j", "i4"), ("sb_r", "i4"),
                ("flags", "i4"), ("tile0", "i4"), ("reserved", "i4")]
GP_BIAS, GP_RELU, GP_MASK, GP_ATOMIC = 1, 2, 4, 8


class PackJob(C.Structure):
    _fields_ = [("w", vp), ("wpk", vp), ("row_perm", vp), ("Co", i32), ("Ci", i32), ("KHW", i32), ("rows_pad", i32),
                ("cols_pad", i32), ("transpose", i32), ("dtype", i32), ("groups", i32), ("upconv", i32), ("lo", i32)]


class AdamEntry(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("m", vp), ("v", vp), ("step", vp), ("n", i64)]


# name -> argtypes  (restype is int unless listed in _RESTYPE)
_SIGS = {
    "xmc_abi_version": [],
    "xmc_half_format": [],
    "xmc_last_kernel": [],
    "xmc_set_fixed_order": [i32],
    "xmc_set_prezeroed": [i32],
    "xmc_conv_splitk_ws_bytes": [vp],
    "xmc_conv_igemm": [C.POINTER(ConvDesc), vp],
    "xmc_conv_wgrad": [C.POINTER(ConvDesc), vp, vp],
    "xmc_conv_wgrad_bias": [C.POINTER(ConvDesc), vp, vp, vp],
    "xmc_pack_weight": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "xmc_pack_weight_grouped": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    "xmc_pack_weight_upconv": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "xmc_pack_weight_multi": [C.POINTER(PackJob), i32, vp],
    "xmc_axpby_up": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_unpack_wgrad": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp],
    "xmc_unpack_wgrad_grouped": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp],
    "xmc_unpack_wgrad_bias": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp],
    "xmc_unpack_wgrad_bias_dot": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp],
    "xmc_nchw_to_nhwc8": [vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_nhwc8_to_nchw": [vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_lrelu": [vp, vp, i64, f32, i32, vp],
    "xmc_lrelu_mask": [vp, vp, vp, i64, f32, i32, vp],
    "xmc_tanh": [vp, vp, i64, i32, vp],
    "xmc_signmask_apply": [vp, vp, vp, i64, f32, i32, vp],
    "xmc_conv_pw1x1_masked_src": [C.POINTER(ConvDesc), vp, vp, f32, vp],
    "xmc_conv_pw1x1_split": [C.POINTER(ConvDesc), vp],
    "xmc_conv_ptile_bits": [C.POINTER(ConvDesc), vp],
    "xmc_conv_ptile_scimg": [C.POINTER(ConvDesc), vp],
    "xmc_conv_wgrad_bits": [C.POINTER(ConvDesc), vp, vp],
    "xmc_tanh_bwd": [vp, vp, vp, i64, i32, vp],
    "xmc_axpby": [vp, vp, vp, vp, i64, i32, vp],
    "xmc_scale_mask_dot": [vp, vp, vp, vp, vp, i64, i32, vp],
    "xmc_axpby_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp],
    "xmc_axpby_up_lrelu": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_scale": [vp, vp, vp, i64, i32, vp],
    "xmc_dot": [vp, vp, vp, i64, i32, vp],
    "xmc_gemm_group": [vp, i32, vp],
    "xmc_embedding_gather": [vp, vp, vp, i64, i32, i64, vp],
    "xmc_lstm_bidir": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "xmc_gru_bidir": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "xmc_spectral_sigma": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "xmc_spectral_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_colsum": [vp, vp, i64, i32, i32, vp],
    "xmc_avgpool2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_upsample2": [vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "xmc_sumpool2": [vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "xmc_global_avgpool": [vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_global_avgpool_bwd": [vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_affine2_act_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "xmc_affine2_act_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "xmc_affine2_act_bwd_acc": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "xmc_affine2_act_bwd_dot": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "xmc_affine2_act_bwd_dot_pool": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "xmc_affine2_lrelu_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "xmc_affine2_lrelu_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "xmc_groupnorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp],
    "xmc_groupnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "xmc_attn_pool_ws_floats": [i32, i32],
    "xmc_attn_pool_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "xmc_attn_pool_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "xmc_attn_pool_bwd_acc": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp],
    "xmc_word_pool_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "xmc_word_pool_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "xmc_gvec_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_gvec_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "xmc_reasoner_fwd": [vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, i32, vp],
    "xmc_reasoner_bwd": [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp],
    "xmc_word_ctx_fwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_word_ctx_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_word_keys_fwd": [vp, vp, vp, f32, vp, vp, i32, i32, vp],
    "xmc_word_keys_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_contrastive_ws_bytes": [i32, i32],
    "xmc_contrastive_fwd": [vp, vp, vp, vp, i32, i32, vp, vp, vp],
    "xmc_contrastive_bwd": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp],
    "xmc_cosine_scores": [vp, vp, i32, i32, vp, vp, vp],
    "xmc_hinge_fwd": [vp, i32, f32, vp, i64, i32, vp],
    "xmc_hinge_bwd": [vp, i32, f32, vp, vp, i64, i32, vp],
    "xmc_cast": [vp, vp, i64, i32, i32, vp],
    "xmc_concept_query_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "xmc_concept_query_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "xmc_concept_query_fwd_multi": [vp, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp],
    "xmc_concept_query_bwd_multi": [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "xmc_concept_gquery_fwd": [vp, vp, vp, vp, vp, vp, i32, f32, vp],
    "xmc_concept_gquery_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp],
    "xmc_concept_head_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_concept_head_fwd_pre": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_concept_head_bwd_pre": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_concept_outer_multi": [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp],
    "xmc_concept_head_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "xmc_rows_sumsq": [vp, vp, i32, i64, vp],
    "xmc_gp_finish": [vp, i32, vp, vp, f32, vp],
    "xmc_rows_scale": [vp, vp, vp, vp, i32, i64, vp],
    "xmc_adam_chunk_elems": [],
    "xmc_adam_step": [vp, i32, vp, i32, f32, f32, f32, f32, f32, vp],
    "xmc_adam_step_scaled": [vp, i32, vp, i32, f32, f32, f32, f32, vp, vp, i32, f32, f32, i32, vp],
    "xmc_dstem_compose": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "xmc_dstem_compose_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "xmc_dstem_pack": [vp, vp, vp],
    "xmc_dstem_pack_sc": [vp, vp, vp],
    "xmc_dstem_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "xmc_dstem_wgrad": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "xmc_dstem_border_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "xmc_dstem_border_wgrad": [vp, vp, vp, vp, i32, i32, i32, vp],
    "xmc_dstem_dgrad": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
}
_RESTYPE = {"xmc_contrastive_ws_bytes": i64, "xmc_conv_splitk_ws_bytes": i64, "xmc_attn_pool_ws_floats": i64, "xmc_last_kernel": C.c_char_p}
EXPORTS = tuple(_SIGS)

_libs = {}               # variant -> loaded library
_variant = "bf16"


class XmcHipError(RuntimeError):
    pass


def use_variant(v):
    """Select the build every following call goes to: "bf16" (libxmc_gan_hip.so) or "f16" (libxmc_gan_hip_f16.so)."""
    global _variant
    assert v in ("bf16", "f16")
    _variant = v


def variant():
    return _variant


def load(v=None):
    """Load the shared library of the selected variant (once each) and set prototypes."""
    v = v or _variant
    lib = _libs.get(v)
    if lib is not None:
        return lib
    path = LIB_PATH if v == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C xmc-gan_amd/csrc` (or __graft_entry__.build()). "
            "The XMC-GAN step has no CPU/PyTorch fallback.")
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.xmc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.xmc_abi_version()}, this binding needs {ABI_VERSION} (stale build?)")
    if lib.xmc_half_format() != (0 if v == "bf16" else 1):
        raise RuntimeError(f"{path} was not built for the {v} storage format")
    # the ops package hands every accumulator the header documents as "zeroed here" over as a slice of its once-per-iteration zero arena
    lib.xmc_set_prezeroed(1)
    _libs[v] = lib
    return lib


_ECODES = {-1: "XMC_EINVAL (bad argument)", -2: "XMC_EALIGN (alignment / channel multiple)", -3: "XMC_ESHAPE (shape out of range)"}


def check(rc, what):
    if rc != 0:
        msg = _ECODES.get(rc) or (f"hipError_t {-rc - 1000}" if rc <= -1000 else f"unexpected return code {rc}")
        raise XmcHipError(f"{what} failed: {msg}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
