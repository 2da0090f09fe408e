"""Autograd-visible operators backed by libxmc_gan_hip.so.

Every operator is a ``torch.autograd.Function`` whose backward is itself written with operators
from this file, so the set is closed under differentiation: that is what lets the MA-GP term
(train_gan.py:231-252, ``autograd.grad(create_graph=True)`` followed by ``backward()``) run through
hand-written kernels.  Activations are contiguous NHWC tensors ``[N,H,W,C]`` (C % 8 == 0) in the
engine's activation dtype (bf16 by default, f32 in parity mode); parameters stay f32 in the
reference's ``[Co,Ci,KH,KW]`` / ``[out,in]`` layout and are packed on demand (cached).
PyTorch is used for storage, streams and the autograd graph only.
"""
from . import _config, _engine, _nodes_conv, _nodes_block, _nodes_loss, _functional

# names a module uses from a module that comes LATER in the import order (all of them inside function bodies, i.e. at call time):
# bound here, once every module is loaded
_LATE = {'_engine': [('_nodes_block', 'CastFn')], '_nodes_conv': [('_nodes_block', 'CastFn'), ('_nodes_block', 'ColSumFn'), ('_nodes_block', 'DotFn'), ('_nodes_block', 'MaskFn'), ('_nodes_block', 'ScaleFn'), ('_nodes_block', 'SumPool2Fn'), ('_nodes_block', 'TanhBwdFn'), ('_nodes_block', '_affine_bwd_raw'), ('_nodes_block', '_affine_fwd_raw'), ('_functional', 'lrelu')]}
for _m, _pairs in _LATE.items():
    for _o, _n in _pairs:
        setattr(globals()[_m], _n, getattr(globals()[_o], _n))
# the public (and test-visible) surface: every top-level name of every module
for _m in (_config, _engine, _nodes_conv, _nodes_block, _nodes_loss, _functional):
    globals().update({_k: _v for _k, _v in vars(_m).items() if not _k.startswith("__")})
del _m, _pairs, _o, _n
