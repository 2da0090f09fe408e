"""Layer factories with the reference's names and signatures (model/modules.py:13-33).

The returned layers subclass ``nn.Conv2d`` / ``nn.Linear`` so that ``module.apply(weight_init)``
(train_gan.py:65-69), ``state_dict()`` keys and ``load_state_dict`` behave exactly as upstream, but
their ``forward`` runs the MI355X implicit-GEMM kernels on NHWC activations.
"""
import torch
import torch.nn as nn

from xmc_gan_amd import ops
from xmc_gan_amd.lib import ACT_NONE


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters + gfx950 implicit-GEMM forward.  Input/output: NHWC ``[N,H,W,C]``."""

    def __init__(self, in_dim, out_dim, kernel_size, stride=1, padding=0, bias=True, act=ACT_NONE):
        super().__init__(in_dim, out_dim, kernel_size, stride, padding, bias=bias)
        self.geom = ops.ConvGeom(in_dim, out_dim, kernel_size, stride, padding)
        self.act = act

    def forward(self, x, act=None, out_dtype=None):
        return ops.conv2d(x, self.weight, self.bias, self.geom, self.act if act is None else act, out_dtype)


class HipLinear(nn.Linear):
    """nn.Linear parameters + the 1x1 path of the same kernel (f32 MFMA).  ``row_perm`` reorders the
    output features (used to emit NHWC directly from ``proj_noise``)."""

    def __init__(self, in_dim, out_dim, bias=True, act=ACT_NONE, row_perm=None):
        super().__init__(in_dim, out_dim, bias=bias)
        self.geom = ops.ConvGeom(in_dim, out_dim, 1, 1, 0, row_perm=row_perm)
        self.act = act

    def forward(self, x, act=None, out_dtype=None):
        y = ops.linear(x, self.weight, self.bias, self.geom, self.act if act is None else act, out_dtype)
        return y[:, : self.out_features] if y.shape[1] != self.out_features else y


def _no_spec_norm(spec_norm):
    if spec_norm:
        raise NotImplementedError(
            "DISC.SPEC_NORM=True (legacy torch.nn.utils.spectral_norm, modules.py:16-17) is not built yet; "
            "every shipped DF-GAN preset sets SPEC_NORM: False")


def conv2d_nxn(in_dim, out_dim, kernel_size, stride=1, padding=0, bias=True, groups=1, spec_norm=False):
    _no_spec_norm(spec_norm)
    if groups != 1:
        raise NotImplementedError("grouped convolutions are built by the attention-modulation blocks directly")
    return HipConv2d(in_dim, out_dim, kernel_size, stride, padding, bias=bias)


def linear(in_dim, out_dim, bias=True, spec_norm=False):
    _no_spec_norm(spec_norm)
    return HipLinear(in_dim, out_dim, bias=bias)


def as_nhwc(x):
    """logical NCHW tensor (any strides) -> contiguous [N,H,W,C]; free for channels-last storage."""
    return x.permute(0, 2, 3, 1).contiguous()


def as_nchw_view(y):
    """contiguous [N,H,W,C] -> logical NCHW view (no copy)."""
    return y.permute(0, 3, 1, 2)
