"""Layer factories with the reference's names and signatures (model/modules.py:13-33).

The returned layers subclass ``nn.Conv2d`` / ``nn.Linear`` so that ``module.apply(weight_init)``
(train_gan.py:65-69), ``state_dict()`` keys and ``load_state_dict`` behave exactly as upstream, but
their ``forward`` runs the MI355X implicit-GEMM kernels on NHWC activations.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from xmc_gan_amd import ops
from xmc_gan_amd.lib import ACT_NONE


class _SpectralNorm:
    """State and weight computation of ``torch.nn.utils.spectral_norm(layer)`` (model/modules.py:16-17,31-32),
    mixed into the layer instead of hung on it as a forward pre-hook.

    Matches the legacy hook's observable behaviour: ``weight`` leaves ``parameters()`` and is replaced by the
    Parameter ``weight_orig`` plus buffers ``weight_u`` / ``weight_v`` (state_dict keys bias, weight_orig,
    weight_u, weight_v), drawn in upstream's RNG order (u, then v, standard normal, L2-normalised).  ``weight``
    stays a plain tensor attribute: until the first forward it aliases ``weight_orig``'s construction-time
    storage, afterwards it is the effective weight of the latest call -- so, exactly as upstream, a
    ``weight_init`` applied after ``.cuda()`` (train_gan.py:473-478) leaves ``weight_orig`` at its default
    initialisation.  Every forward call in training mode performs one power iteration."""

    def _sn_setup(self):
        w = self._parameters.pop("weight")
        self.register_parameter("weight_orig", w)
        with torch.no_grad():
            h, wd = w.shape[0], w.numel() // w.shape[0]
            u = F.normalize(w.new_empty(h).normal_(0, 1), dim=0, eps=1e-12)
            v = F.normalize(w.new_empty(wd).normal_(0, 1), dim=0, eps=1e-12)
        object.__setattr__(self, "weight", w.data)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)

    def effective_weight(self):
        if not self.spec_norm:
            return self.weight
        w = ops.spectral_weight(self.weight_orig, self.weight_u, self.weight_v, self.training)
        object.__setattr__(self, "weight", w)
        return w


class HipConv2d(nn.Conv2d, _SpectralNorm):
    """nn.Conv2d parameters + gfx950 implicit-GEMM forward.  Input/output: NHWC ``[N,H,W,C]``."""

    def __init__(self, in_dim, out_dim, kernel_size, stride=1, padding=0, bias=True, act=ACT_NONE, spec_norm=False):
        super().__init__(in_dim, out_dim, kernel_size, stride, padding, bias=bias)
        self.geom = ops.ConvGeom(in_dim, out_dim, kernel_size, stride, padding)
        self.act = act
        self.spec_norm = bool(spec_norm)
        if self.spec_norm:
            self._sn_setup()

    def forward(self, x, act=None, out_dtype=None, want_pool=False, out=None, pair=False):
        return ops.conv2d(x, self.effective_weight(), self.bias, self.geom,
                          self.act if act is None else act, out_dtype, want_pool, out, pair)


class HipLinear(nn.Linear, _SpectralNorm):
    """nn.Linear parameters + the 1x1 path of the same kernel (f32 MFMA).  ``row_perm`` reorders the
    output features (used to emit NHWC directly from ``proj_noise``)."""

    def __init__(self, in_dim, out_dim, bias=True, act=ACT_NONE, row_perm=None, spec_norm=False):
        super().__init__(in_dim, out_dim, bias=bias)
        self.geom = ops.ConvGeom(in_dim, out_dim, 1, 1, 0, row_perm=row_perm)
        self.act = act
        self.spec_norm = bool(spec_norm)
        if self.spec_norm:
            self._sn_setup()

    def forward(self, x, act=None, out_dtype=None):
        y = ops.linear(x, self.effective_weight(), self.bias, self.geom,
                       self.act if act is None else act, out_dtype)
        return y[:, : self.out_features] if y.shape[1] != self.out_features else y


def conv2d_nxn(in_dim, out_dim, kernel_size, stride=1, padding=0, bias=True, groups=1, spec_norm=False):
    if groups != 1:
        raise NotImplementedError("grouped convolutions are built by the attention-modulation blocks directly")
    return HipConv2d(in_dim, out_dim, kernel_size, stride, padding, bias=bias, spec_norm=spec_norm)


def linear(in_dim, out_dim, bias=True, spec_norm=False):
    return HipLinear(in_dim, out_dim, bias=bias, spec_norm=spec_norm)


def as_nhwc(x):
    """logical NCHW tensor (any strides) -> contiguous [N,H,W,C]; free for channels-last storage."""
    return x.permute(0, 2, 3, 1).contiguous()


def as_nchw_view(y):
    """contiguous [N,H,W,C] -> logical NCHW view (no copy)."""
    return y.permute(0, 3, 1, 2)
