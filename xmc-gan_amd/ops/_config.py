"""xmc_gan_amd.ops: configuration of the engine: precision modes and loss scales, dispatch switches, thread-local contexts, small helpers.
(One of the modules ops.py was split into in round 5; `xmc_gan_amd.ops` re-exports every name.)"""
import ctypes as C
import os
import threading
import weakref
import numpy as np
import torch
from .. import lib as L
from .. import prof


# ------------------------------------------------------------------------------------------ config
_state = threading.local()
# kernel / operator A/B switches (comma-separated tokens; unset in production).  The C dispatchers read the same variable
# (common.h: xmc_debug_off); the one host-side token is "no_fused_blocks" (blocks composed from the fine-grained Functions).
_DEBUG_DISPATCH = frozenset(t for t in os.environ.get("XMC_DEBUG_DISPATCH", "").split(",") if t)
_PRECISION = os.environ.get("XMC_PRECISION", "bf16")


def set_precision(p):
    """'bf16' (default: bf16 activations / MFMA operands, f32 accumulate, f32 parameters), 'f16' (IEEE half in the same places,
    through the f16 build of the library: same MFMA rate, 11 significant bits instead of 8 -- the mode whose losses stay within
    1e-3 of the f32 reference -- with the backward passes run on LOSS_SCALE x the loss, see `loss_scale`) or 'fp32'."""
    global _PRECISION
    assert p in ("bf16", "f16", "fp32")
    _PRECISION = p
    reset_loss_scalers()
    L.use_variant("f16" if p == "f16" else "bf16")
    bump_weights_epoch()           # packed weights of the other format / library are not ours


def precision():
    return _PRECISION


def act_dtype():
    return {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[_PRECISION]


_PRECISE = [None]           # None: on in the IEEE-half mode (the mode that promises 1e-3), off in bf16


def precise_trunk(on="query"):
    """The discriminator's shortcut path on maps of <= 8x8 pixels, COND_DNET and the learned shortcuts' weights at f32-grade precision
    (ResDFn.forward, the `split` operand of the streaming 1x1 kernels): on by default in the IEEE-half mode, off in bf16 (whose
    8-bit activations on the larger maps alone cost more than the bar, tests/diag/layer_ladder.py --fmt bf16).  `precise_trunk(True /
    False / None)` overrides / restores the default; XMC_DEBUG_DISPATCH=no_precise switches it off for A/B runs."""
    if on != "query":
        _PRECISE[0] = None if on is None else bool(on)
        return
    if "no_precise" in _DEBUG_DISPATCH or _PRECISION == "fp32":
        return False
    return _PRECISION == "f16" if _PRECISE[0] is None else _PRECISE[0]


# f16 activation gradients: a hinge / InfoNCE gradient of 1/B spread over a 256x256x32 map is ~1e-6 per element, far into
# the subnormal range of IEEE half (spacing 6e-8).  The iteration therefore differentiates scale * loss and the optimizer
# kernel divides the (f32) parameter gradients by it again (train_gan.gan_iteration, optim.HipAdam.step(scaler=)); powers of
# two, so the result does not depend on the scale while nothing leaves the format's range.  bf16 / fp32 run unscaled.
# The scale is DYNAMIC (round 4; it was a fixed 4096): one device-resident `LossScaler` per backward phase -- "D", "GP" (the
# outer backward of the matching-aware gradient penalty), "G" -- with torch.cuda.amp.GradScaler's rule: a step whose gradients
# hold an inf / NaN is skipped inside the Adam kernel (parameters, moments and step counters untouched) and halves the scale;
# `growth_interval` finite steps in a row double it.  Everything stays on the device, so the iteration remains capturable.
LOSS_SCALE_F16 = float(os.environ.get("XMC_LOSS_SCALE", 4096.0))          # initial value
LOSS_SCALE_GROWTH_INTERVAL = int(os.environ.get("XMC_LOSS_SCALE_INTERVAL", 2000))
# The INNER backward of MA-GP (d logit / d inputs, a forward quantity of the outer graph) starts from GP_INNER_SCALE x ones
# instead of ones and `grad_penalty` divides it out of the norm: d logit / d pixel of a 256x256 image is 1e-6 .. 1e-3, among
# or next to the subnormals of IEEE half.  Fixed: an overflow here means a gradient element above 65504 / 256, i.e. a diverged
# run (it would surface as a skipped GP step, and the outer scaler's back-off cannot cure it -- `train()` reports skips).
GP_INNER_SCALE_F16 = float(os.environ.get("XMC_GP_INNER_SCALE", 256.0))


class LossScaler:
    """device-resident dynamic loss scale: ``sf`` = [scale, 1/scale] (f32), ``si`` = [found-inf flag of the running step, finite
    steps in a row, the flag at the last finished step, steps skipped so far] (int32); updated by `xmc_adam_step_scaled`."""

    def __init__(self, device, init=None, growth=2.0, backoff=0.5, interval=None):
        init = LOSS_SCALE_F16 if init is None else float(init)
        self.sf = torch.tensor([init, 1.0 / init], dtype=torch.float32, device=device)
        self.si = torch.zeros(4, dtype=torch.int32, device=device)
        self.growth, self.backoff = float(growth), float(backoff)
        self.interval = LOSS_SCALE_GROWTH_INTERVAL if interval is None else int(interval)

    def scale(self, loss):
        return loss * self.sf[0]

    def stats(self):
        """host view (synchronises): current scale, whether the last step was skipped, steps skipped so far"""
        sf, si = self.sf.tolist(), self.si.tolist()
        return dict(scale=sf[0], last_step_skipped=bool(si[2]), skipped_steps=int(si[3]))


_scalers = {}


def loss_scaler(phase, device):
    """the `LossScaler` of a backward phase ("D", "GP", "G") in the IEEE-half mode, None in the other modes.  Created on first use
    (outside graph capture: the warm-up iterations come first) and kept across iterations."""
    if _PRECISION != "f16":
        return None
    key = (phase, torch.device(device).index)
    sc = _scalers.get(key)
    if sc is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("loss scaler created during graph capture; run a warm-up iteration first")
        sc = _scalers[key] = LossScaler(device)
    return sc


def reset_loss_scalers():
    _scalers.clear()


def loss_scaler_stats():
    return {f"{ph}": sc.stats() for (ph, _), sc in _scalers.items()}


def loss_scaler_state():
    """{phase: [scale, finite steps in a row, steps skipped]} of this device's scalers, for a checkpoint (one host read)"""
    return {ph: [sc.sf[0].item(), int(sc.si[1].item()), int(sc.si[3].item())] for (ph, _), sc in _scalers.items()}


def load_loss_scaler_state(state, device):
    """restore `loss_scaler_state()` (a resumed IEEE-half run continues at its scale instead of re-learning it from 4096)"""
    if _PRECISION != "f16":
        return
    for ph, (scale, good, skipped) in state.items():
        sc = _scalers[(ph, torch.device(device).index)] = LossScaler(device, init=float(scale))
        sc.si[1], sc.si[3] = int(good), int(skipped)


def gp_inner_scale():
    return GP_INNER_SCALE_F16 if _PRECISION == "f16" else 1.0


def loss_scale(phase="step"):
    """INITIAL factor of the D / G backward passes (the running value lives in `loss_scaler(phase)`); the rounding oracle of the
    tests rounds gradient tensors at this scale."""
    return LOSS_SCALE_F16 if _PRECISION == "f16" else 1.0


class composable:
    """Context: build blocks from the fine-grained differentiable Functions instead of the fused first-order block Functions
    (needed wherever the backward pass itself is differentiated: MA-GP, train_gan.py:231-252)."""

    def __enter__(self):
        self.prev = getattr(_state, "composable", False)
        _state.composable = True

    def __exit__(self, *a):
        _state.composable = self.prev


def fused_blocks():
    return not getattr(_state, "composable", False) and "no_fused_blocks" not in _DEBUG_DISPATCH


class no_wgrad:
    """Context: convolutions skip weight/bias gradients (used where the reference computes and
    then discards them, e.g. D's weight grads during the G step, train_gan.py:288 then 226-227)."""

    def __enter__(self):
        self.prev = getattr(_state, "skip_wgrad", False)
        _state.skip_wgrad = True

    def __exit__(self, *a):
        _state.skip_wgrad = self.prev


def _skip_wgrad():
    return getattr(_state, "skip_wgrad", False)


def _second_order():
    return getattr(_state, "second_order", False)


def second_order_active():
    return _second_order()


class second_order:
    """Context: forwards inside it keep what a DIFFERENTIATED backward needs (the discriminator blocks store their residual
    branch instead of its sign bits).  The MA-GP term wraps its discriminator forward in this (train_gan.py:231-247)."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = getattr(_state, "second_order", False)
        _state.second_order = bool(self.on)
        return self

    def __exit__(self, *a):
        _state.second_order = self.prev
        return False


# ------------------------------------------------------------------------------------------ helpers
def _code(dtype):
    if dtype == torch.float32:
        return L.F32
    if dtype == (torch.float16 if L.variant() == "f16" else torch.bfloat16):
        return L.H16           # "the 16-bit format of the loaded build"
    raise TypeError(f"unsupported dtype {dtype} for the {L.variant()} build of the library")


def _esz(dtype):
    return 4 if dtype == torch.float32 else 2


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("xmc_gan_amd operators run on the GPU only (no CPU fallback)")


def pad_to(n, m):
    return (n + m - 1) // m * m


def _nbytes(*ts):
    """bytes of the given tensors (None skipped): the algorithmic HBM traffic of a launch that touches each of them once"""
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def chan_pad(c, dtype):
    """stored channel count for c logical channels: a multiple of 8, except that f32 feature vectors
    whose width is already a multiple of 4 (16-byte units, e.g. the 100-d noise) are kept as they are."""
    if dtype == torch.float32 and c % 4 == 0:
        return c
    return pad_to(c, 8)


_weights_epoch = [0]


def bump_weights_epoch():
    """Invalidate EVERY cached packed weight (parameters replaced wholesale behind autograd's back)."""
    _weights_epoch[0] += 1
