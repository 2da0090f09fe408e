"""xmc_gan_amd.ops: functional wrappers over the nodes.
(One of the modules ops.py was split into in round 5; `xmc_gan_amd.ops` re-exports every name.)"""
import ctypes as C
import os
import threading
import weakref
import numpy as np
import torch
from .. import lib as L
from .. import prof
from ._config import (
    _PRECISION, act_dtype, set_precision)
from ._nodes_conv import (
    AxpbyUpFn, UpConvFn)
from ._nodes_block import (
    Affine2LreluFn, AttnPoolFn, AxpbyFn, CastFn, GapFn, GroupNormFn, LreluFn, NchwToNhwc8Fn, Nhwc8ToNchwFn,
    SumPool2Fn, Up2Fn)
from ._nodes_loss import (
    ContrastiveFn, HingeFn)


# ------------------------------------------------------------------------------------------ functional sugar
def lrelu(x, slope=0.2):
    return LreluFn.apply(x, slope)


def avgpool2(x):
    return SumPool2Fn.apply(x, 0.25)


def upsample2(x):
    return Up2Fn.apply(x, 1.0)


def axpby(a, b, alpha):
    return AxpbyFn.apply(a, b, alpha)


def global_avgpool(x, out_dtype=torch.float32):
    return GapFn.apply(x, out_dtype)


def to_nhwc8(x_nchw, out=None):
    return NchwToNhwc8Fn.apply(x_nchw, act_dtype(), out)


def to_nchw(x_nhwc8, c):
    return Nhwc8ToNchwFn.apply(x_nhwc8, c)


def affine2_lrelu(x, g0, b0, g1, b1):
    return Affine2LreluFn.apply(x, g0, b0, g1, b1)


def upconv3x3(x_lo, w, b, geom):
    return UpConvFn.apply(x_lo, w, b, geom)


def axpby_up(a_lo, b_hi, alpha, lrelu=False):
    return AxpbyUpFn.apply(a_lo, b_hi, alpha, lrelu)


def affine_lrelu(x, g, b):
    return Affine2LreluFn.apply(x, g, b, None, None)


def affine_act(x, g, b, slope):
    """act(x * g[n,c] + b[n,c]); slope 0 = ReLU (the word-attention generator, concept_gan.py:421,447,497,509)"""
    return Affine2LreluFn.apply(x, g, b, None, None, slope)


def groupnorm(x, w, b, groups, slope=-1.0, eps=1e-5):
    return GroupNormFn.apply(x, w, b, groups, slope, eps)


def attn_pool(key, q, x, ncon, scale=1.0):
    return AttnPoolFn.apply(key, q, x, ncon, scale)


def hinge(logits_padded, sign):
    return HingeFn.apply(logits_padded, sign)


def contrastive(a, b, labels=None, inv_num_pos=None):
    return ContrastiveFn.apply(a, b, labels, inv_num_pos)


def cast(x, dtype):
    return CastFn.apply(x, dtype)


if _PRECISION != "bf16":          # XMC_PRECISION in the environment: select the matching build of the library
    set_precision(_PRECISION)
