"""Autograd-visible operators backed by libxmc_gan_hip.so.

Every operator is a ``torch.autograd.Function`` whose backward is itself written with operators
from this file, so the set is closed under differentiation: that is what lets the MA-GP term
(train_gan.py:231-252, ``autograd.grad(create_graph=True)`` followed by ``backward()``) run through
hand-written kernels.  Activations are contiguous NHWC tensors ``[N,H,W,C]`` (C % 8 == 0) in the
engine's activation dtype (bf16 by default, f32 in parity mode); parameters stay f32 in the
reference's ``[Co,Ci,KH,KW]`` / ``[out,in]`` layout and are packed on demand (cached).
PyTorch is used for storage, streams and the autograd graph only.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np
import torch

from . import lib as L
from . import prof

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


# ------------------------------------------------------------------------------------------ geometry
class ConvGeom:
    """Geometry of one convolution / linear layer (square kernel k, stride s, padding p)."""

    __slots__ = ("cin", "cout", "k", "s", "p", "row_perm", "_perm_dev", "groups")

    def __init__(self, cin, cout, k=1, s=1, p=0, row_perm=None, groups=1):
        assert k * k <= L.MAX_TAPS and k % s == 0
        assert cin % groups == 0 and cout % groups == 0
        self.cin, self.cout, self.k, self.s, self.p = cin, cout, k, s, p
        # groups > 1: nn.Conv2d(groups=g) weight [cout, cin/g, k, k]; the kernels see its block-diagonal expansion, which the
        # pack kernel writes and the gradient unpack kernel reads back (diagonal blocks only)
        self.groups = groups
        self.row_perm = row_perm          # optional LongTensor/list: packed output row r <- parameter row perm[r]
        self._perm_dev = None

    def out_hw(self, h, w):
        return (h + 2 * self.p - self.k) // self.s + 1, (w + 2 * self.p - self.k) // self.s + 1

    def perm_dev(self, device):
        if self.row_perm is None:
            return None
        if self._perm_dev is None or self._perm_dev.device != device:
            self._perm_dev = torch.as_tensor(self.row_perm, dtype=torch.int32, device=device).contiguous()
        return self._perm_dev


def _igemm(d, what):
    """xmc_conv_igemm, handing over the split-K scratch the descriptor asks for (the layers on 4x4 / 8x8 maps: XmcConvDesc.splitk_ws).
    The scratch is a plain caching-allocator block: stream order keeps it alive until the two launches that use it have run."""
    lib = L.load()
    nb = lib.xmc_conv_splitk_ws_bytes(C.byref(d))
    if nb > 0:
        ws = torch.empty(nb, dtype=torch.uint8, device=torch.cuda.current_device())
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), nb
    L.check(lib.xmc_conv_igemm(C.byref(d), _st()), what)


def _fill_taps(d, cls, taps):
    for t, (dh, dw, wi) in enumerate(taps):
        d.dh[cls][t] = dh
        d.dw[cls][t] = dw
        d.wi[cls][t] = wi


_pack_cache = {}          # (id(param), kind, transpose, dtype) -> _PackEntry


class _PackEntry:
    """One packed copy of a parameter.  Valid while the parameter has not changed: autograd-visible writes move
    ``w._version``; the optimizer kernels write behind autograd's back and bump ``w._xmc_epoch`` (optim.HipAdam.step), then
    re-pack every entry of the parameters they changed in ONE launch (``repack_params``) -- so after the first iteration the
    forward / backward passes find every pack valid and launch no pack kernel of their own (~100-200 launches of ~6 us per
    iteration before).  ``gepoch``: bumped when entries may point into graph-private memory (graph._capture)."""
    __slots__ = ("ref", "version", "pepoch", "gepoch", "geom", "out", "up", "transpose", "wf", "private", "born", "lo")

    def valid(self, w, geom):
        return (self.ref() is w and self.version == w._version and self.pepoch == getattr(w, "_xmc_epoch", 0) and
                self.gepoch == _weights_epoch[0] and self.geom is geom)


def _pack_shape(geom, transpose, dtype, up):
    cs_p = chan_pad(geom.cin, dtype)          # stored channels of x
    cd_p = pad_to(geom.cout, 8)               # stored channels of y
    rows, cols = (pad_to(cs_p, 32), cd_p) if transpose else (pad_to(cd_p, 32), cs_p)
    return (16 if up else geom.k * geom.k), rows, cols


def _pack_job(wf, out, geom, transpose, up, lo=False):
    j = L.PackJob()
    j.lo = int(bool(lo))            # the part of w its 16-bit copy lost: round16(w - round16(w))  (XmcConvDesc.wpk_lo)
    j.w, j.wpk = wf.data_ptr(), out.data_ptr()
    perm = None if up else geom.perm_dev(wf.device)
    j.row_perm = perm.data_ptr() if perm is not None else None
    j.Co, j.Ci, j.KHW = geom.cout, geom.cin, geom.k * geom.k
    j.rows_pad, j.cols_pad = out.shape[1], out.shape[2]
    j.transpose, j.dtype, j.groups, j.upconv = int(transpose), _code(out.dtype), (1 if up else geom.groups), int(up)
    return j


def _pack(w, geom, transpose, dtype, up=False, lo=False):
    assert not up or geom.groups == 1
    assert not lo or dtype != torch.float32
    out = torch.empty(_pack_shape(geom, transpose, dtype, up), dtype=dtype, device=w.device)
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    job = _pack_job(wf, out, geom, transpose, up, lo)
    L.check(L.load().xmc_pack_weight_multi(C.byref(job), 1, _st()), "xmc_pack_weight_multi")
    return out


def _packed_cached(w, geom, transpose, dtype, up=False, lo=False):
    """Packed copy of ``w`` ([Co,Ci,k,k] / [Co,Ci]) for the forward (transpose=0: [tap][co][ci]) or the data-gradient
    (transpose=1: [tap][ci][co]) kernel; ``up``: the 16 pre-summed 2x2-tap slices of the fused upsample convolution; ``lo``: the
    low half of a weight pair (XmcConvDesc.wpk_lo).  Cached per nn.Parameter until it changes."""
    if not isinstance(w, torch.nn.Parameter):
        return _pack(w, geom, transpose, dtype, up, lo)
    k = (id(w), "lo" if lo else bool(up), int(transpose), dtype)
    hit = _pack_cache.get(k)
    if hit is not None and hit.valid(w, geom):
        return hit.out
    e = _PackEntry()
    e.ref, e.geom, e.up, e.transpose, e.lo = weakref.ref(w), geom, bool(up), int(transpose), bool(lo)
    e.version, e.pepoch, e.gepoch = w._version, getattr(w, "_xmc_epoch", 0), _weights_epoch[0]
    wd = w.detach()
    e.wf = wd if (wd.dtype == torch.float32 and wd.is_contiguous()) else None      # None: re-packed lazily, never in bulk
    e.out = _pack(w, geom, transpose, dtype, up, lo)
    e.private = w.is_cuda and torch.cuda.is_current_stream_capturing()       # buffer lives in that graph's memory pool
    _pack_serial[0] += 1
    e.born = _pack_serial[0]
    if hit is not None and _graphs_alive[0]:
        _retired_packs.append(hit.out)        # a captured graph may still write / read the buffer it saw
    _pack_cache[k] = e
    return e.out


def _packed_upconv_cached(w, geom, transpose, dtype):
    return _packed_cached(w, geom, transpose, dtype, up=True)


_graphs_alive = [0]       # set by graph.GraphedIteration: pack buffers a capture has seen must outlive it
_retired_packs = []
_pack_serial = [0]        # counts pack entries ever created (graph.GraphedIteration: "which entries are newer than my capture?")


def end_of_capture(failed=False):
    """graph.GraphedIteration: a capture has ended.  Entries it created point into its private pool and are dropped (eager
    code must not use them); every other entry stays valid -- the captured re-pack launches keep writing the same buffers, in
    the same place of the iteration, as the eager ones.
    ``failed``: the capture raised.  Its launches were only recorded, never executed: the entries it created hold uninitialised
    memory, and a `repack_params` recorded inside it has marked OTHER entries valid for weights that were never re-packed --
    every cached copy is invalidated, and no live graph is counted."""
    if failed:
        for k in [k for k, e in _pack_cache.items() if e.private]:
            del _pack_cache[k]
        bump_weights_epoch()
        return _pack_serial[0]
    _graphs_alive[0] += 1
    for k in [k for k, e in _pack_cache.items() if e.private]:
        del _pack_cache[k]
    return _pack_serial[0]


def drop_packs_newer_than(serial):
    """graph.GraphedIteration, after a replay: the captured re-pack refreshed the copies that existed when it was captured;
    copies created since (eagerly, e.g. an evaluation pass or another dtype between two replays) were not and would pass
    `valid()` with stale contents -- drop them (their next eager use packs afresh)."""
    if _pack_serial[0] == serial:
        return serial
    for k in [k for k, e in _pack_cache.items() if e.born > serial]:
        if _graphs_alive[0]:
            _retired_packs.append(_pack_cache[k].out)
        del _pack_cache[k]
    return _pack_serial[0]


def graph_released():
    """a GraphedIteration was destroyed: buffers kept alive for its replays can go once no graph is left"""
    _graphs_alive[0] = max(0, _graphs_alive[0] - 1)
    if _graphs_alive[0] == 0:
        _retired_packs.clear()


def repack_params(params):
    """Re-pack, in one launch per XMC_PACK_MULTI_MAX copies, every cached packed copy of ``params`` (which the caller has just
    changed and whose ``_xmc_epoch`` it has bumped) into the buffers the entries already own, and mark them valid."""
    ids = {id(p): p for p in params}
    jobs, ents = [], []
    for k in [k for k, e in _pack_cache.items() if e.ref() is None]:       # the parameter is gone: drop its packed copies
        if _graphs_alive[0]:
            _retired_packs.append(_pack_cache[k].out)
        del _pack_cache[k]
    for (pid, _up, _tr, _dt), e in _pack_cache.items():
        w = ids.get(pid)
        if w is None or e.ref() is not w or e.wf is None or e.gepoch != _weights_epoch[0] or e.version != w._version:
            continue
        if e.wf.data_ptr() != w.data_ptr():
            continue
        jobs.append(_pack_job(e.wf, e.out, e.geom, e.transpose, e.up, e.lo))
        ents.append((e, w))
    if not jobs:
        return 0
    arr = (L.PackJob * len(jobs))(*jobs)
    L.check(L.load().xmc_pack_weight_multi(arr, len(jobs), _st()), "xmc_pack_weight_multi")
    for e, w in ents:
        e.pepoch = getattr(w, "_xmc_epoch", 0)
    return len(jobs)


def _upconv_fwd_raw(x, w, bias, geom, act, out_dtype):
    """conv3x3(nearest_up2(x), w) + bias on the LOW-resolution x [N,H,W,Cs] -> [N,2H,2W,Cd]: four output-parity classes,
    each a 2x2-tap convolution with pre-summed weights (4/9 of the MACs, the upsampled tensor never exists)."""
    _need_cuda(x, w)
    assert geom.k == 3 and geom.s == 1 and geom.p == 1
    N, H, W, CS = x.shape
    cd_p = pad_to(geom.cout, 8)
    wpk = _packed_upconv_cached(w, geom, 0, x.dtype)
    y = torch.empty((N, 2 * H, 2 * W, cd_p), dtype=out_dtype, device=x.device)
    d = L.ConvDesc()
    d.src, d.wpk, d.dst = x.data_ptr(), wpk.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.N, d.SH, d.SW, d.CS = N, H, W, CS
    d.DH, d.DW, d.CD = 2 * H, 2 * W, cd_p
    d.MH, d.MW, d.SA, d.DA, d.src_shift = H, W, 1, 2, 0
    d.ntaps, d.nclass, d.CDw = 4, 4, wpk.shape[1]
    d.act, d.dtype, d.out_dtype = act, _code(x.dtype), _code(out_dtype)
    for i in range(2):
        for j in range(2):
            cls = i * 2 + j
            _fill_taps(d, cls, [(i - 1 + th, j - 1 + tw, cls * 4 + th * 2 + tw) for th in range(2) for tw in range(2)])
            d.dph[cls], d.dpw[cls] = i, j
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * 4 * H * W * geom.cout * geom.cin * 9,
                     f"upconv-fwd {x.dtype} N{N} {2 * H}x{2 * W} {geom.cin}->{geom.cout} k3s1", _nbytes(x, wpk, y)):
        _igemm(d, "xmc_conv_igemm(upconv fwd)")
    return y


def _upconv_dgrad_raw(dy, w, geom, in_dtype):
    """gradient of _upconv_fwd_raw w.r.t. its low-resolution input: a 4x4-tap stride-2 gather over dy with the same
    pre-summed weight slices (transposed)."""
    _need_cuda(dy, w)
    N, OH, OW, CDy = dy.shape
    H, W = OH // 2, OW // 2
    cs_p = chan_pad(geom.cin, in_dtype)
    wpk = _packed_upconv_cached(w, geom, 1, dy.dtype)
    dx = torch.empty((N, H, W, cs_p), dtype=in_dtype, device=dy.device)
    d = L.ConvDesc()
    d.src, d.wpk, d.dst = dy.data_ptr(), wpk.data_ptr(), dx.data_ptr()
    d.N, d.SH, d.SW, d.CS = N, OH, OW, CDy
    d.DH, d.DW, d.CD = H, W, cs_p
    d.MH, d.MW, d.SA, d.DA, d.src_shift = H, W, 2, 1, 0
    d.ntaps, d.nclass, d.CDw = 16, 1, wpk.shape[1]
    d.act, d.dtype, d.out_dtype = L.ACT_NONE, _code(dy.dtype), _code(in_dtype)
    taps = []
    for i in range(2):
        for th in range(2):
            ro = i - 2 * (i - 1 + th)
            for j in range(2):
                for tw in range(2):
                    co = j - 2 * (j - 1 + tw)
                    taps.append((ro, co, (i * 2 + j) * 4 + th * 2 + tw))
    _fill_taps(d, 0, taps)
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * geom.cin * 9,
                     f"upconv-dgrad {dy.dtype} N{N} {OH}x{OW} {geom.cin}->{geom.cout} k3s1", _nbytes(dy, wpk, dx)):
        _igemm(d, "xmc_conv_igemm(upconv dgrad)")
    return dx


def _conv_fwd_raw(x, w, bias, geom, act, out_dtype, res=None, alpha=None, up=False, res_mode=0, want2=False, want_pool=False,
                  round_act=False, mask=None, out=None, post_act=L.ACT_NONE, want_sign=False, sc_img=None, w_lo=False):
    """y = act(conv(x, w) + bias) [*alpha] [+ res]; x [N,H,W,Cs]. ``up``: x is read through a fused nearest x2.
    ``res_mode`` 2: res is [N,OH/2,OW/2,C] and read through a nearest x2.  ``want2``: also return act(conv + bias) itself (the
    branch value before alpha / res);  ``want_pool``: also return avg_pool2d(y, 2).  Extras are appended: (y[, y2][, ypool]).
    ``sc_img`` = (image [N,2 OH,2 OW,8], sc_frag, sc_bias): the residual is the composed stem's shortcut, recomputed from the image inside the
    kernel (XmcConvDesc.sc_img, xmc_conv_ptile_scimg) -- returns None when the kernel declines the shape."""
    _need_cuda(x, w)
    N, H, W, CS = x.shape
    sh = 1 if up else 0
    Hv, Wv = H << sh, W << sh
    OH, OW = geom.out_hw(Hv, Wv)
    cd_p = pad_to(geom.cout, 8)
    assert CS == chan_pad(geom.cin, x.dtype), (CS, geom.cin)
    wpk = _packed_cached(w, geom, 0, x.dtype, lo=w_lo)      # (w_lo: the low half of the weight pair, PairConvFn)
    if out is None:
        y = torch.empty((N, OH, OW, cd_p), dtype=out_dtype, device=x.device)
    else:                          # caller-provided destination
        assert tuple(out.shape) == (N, OH, OW, cd_p) and out.dtype == out_dtype and out.is_contiguous() and out.device == x.device
        y = out
    d = L.ConvDesc()
    d.src, d.wpk, d.dst = x.data_ptr(), wpk.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.alpha_dev = alpha.data_ptr() if alpha is not None else None
    d.N, d.SH, d.SW, d.CS = N, H, W, CS
    d.DH, d.DW, d.CD = OH, OW, cd_p
    d.MH, d.MW, d.SA, d.DA, d.src_shift = OH, OW, geom.s, 1, sh
    d.ntaps, d.nclass, d.CDw = geom.k * geom.k, 1, wpk.shape[1]
    d.act, d.dtype, d.out_dtype = act, _code(x.dtype), _code(out_dtype)
    d.res_mode, d.round_act = res_mode, int(bool(round_act))
    d.groups = geom.groups
    d.post_act = post_act           # applied last, to the sum with the residual (XmcConvDesc.post_act)
    _fill_taps(d, 0, [(kh - geom.p, kw - geom.p, kh * geom.k + kw) for kh in range(geom.k) for kw in range(geom.k)])
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= cd_p, "bias must be f32 and padded to the stored channels"
    if res is not None:
        want = (N, OH // 2, OW // 2, cd_p) if res_mode == 2 else tuple(y.shape)
        assert tuple(res.shape) == want and res.dtype == out_dtype and res.is_contiguous(), (res.shape, want)
    if mask is not None:      # y *= LeakyReLU'(mask) after alpha, before the residual (the linearised form of a LeakyReLU layer)
        assert mask.shape == y.shape and mask.dtype == out_dtype and mask.is_contiguous()
        d.mask = mask.data_ptr()
    outs = [y]
    if want2:
        y2 = torch.empty_like(y)
        d.dst2 = y2.data_ptr()
        outs.append(y2)
    if want_sign:                 # sign bits of act(conv + bias), one byte per 8-channel unit (XmcConvDesc.sign_bits)
        assert not want2
        bits = torch.empty((N, OH, OW, cd_p // 8), dtype=torch.uint8, device=x.device)
        d.sign_bits = bits.data_ptr()
        outs.append(bits)
    if want_pool:
        assert OH % 2 == 0 and OW % 2 == 0 and out_dtype == x.dtype
        yp = torch.empty((N, OH // 2, OW // 2, cd_p), dtype=out_dtype, device=x.device)
        d.dst_pool = yp.data_ptr()
        outs.append(yp)
    if sc_img is not None:
        img, frag, sbias = sc_img
        assert res is None and tuple(img.shape) == (N, 2 * OH, 2 * OW, 8) and img.dtype == x.dtype and img.is_contiguous()
        d.sc_img, d.sc_frag, d.sc_bias = img.data_ptr(), frag.data_ptr(), sbias.data_ptr()
        with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * (geom.cin * geom.k * geom.k + 48),
                         f"fwd+sc {x.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(x, wpk, img, *outs)):
            rc = L.load().xmc_conv_ptile_scimg(C.byref(d), _st())
        if rc == 1:
            return None
        L.check(rc, "xmc_conv_ptile_scimg")
        return y if len(outs) == 1 else tuple(outs)
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * geom.cin * geom.k * geom.k,
                     f"fwd {x.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(x, wpk, res, mask, *outs)):
        _igemm(d, "xmc_conv_igemm(fwd)")
    return y if len(outs) == 1 else tuple(outs)


def _conv1x1_pair_raw(x, w, bias, geom, out_dtype):
    """y = conv1x1(x, w) + bias with w at f32 grade on 16-bit activations: `xmc_conv_pw1x1_split` (weights as the 16-bit pair
    round16(w) + round16(w - round16(w)), XmcConvDesc.wpk_lo), else the exact-f32 MFMA kernel on the widened input."""
    _need_cuda(x, w)
    assert geom.k == 1 and geom.s == 1 and geom.p == 0 and geom.groups == 1
    N, H, W, CS = x.shape
    cd_p = pad_to(geom.cout, 8)
    if "no_pw1x1_split" not in _DEBUG_DISPATCH:
        wpk, wlo = _packed_cached(w, geom, 0, x.dtype), _packed_cached(w, geom, 0, x.dtype, lo=True)
        y = torch.empty((N, H, W, cd_p), dtype=out_dtype, device=x.device)
        d = L.ConvDesc()
        d.src, d.wpk, d.wpk_lo, d.dst = x.data_ptr(), wpk.data_ptr(), wlo.data_ptr(), y.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.N, d.SH, d.SW, d.CS = N, H, W, CS
        d.DH, d.DW, d.CD = H, W, cd_p
        d.MH, d.MW, d.SA, d.DA, d.src_shift = H, W, 1, 1, 0
        d.ntaps, d.nclass, d.CDw = 1, 1, wpk.shape[1]
        d.act, d.dtype, d.out_dtype = L.ACT_NONE, _code(x.dtype), _code(out_dtype)
        _fill_taps(d, 0, [(0, 0, 0)])
        with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * H * W * geom.cout * geom.cin,
                         f"fwd-pair {x.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k1s1", _nbytes(x, wpk, wlo, y)):
            rc = L.load().xmc_conv_pw1x1_split(C.byref(d), _st())
        if rc == 0:
            return y
        if rc != 1:
            L.check(rc, "xmc_conv_pw1x1_split")
    y32 = _conv_fwd_raw(CastFn.apply(x, torch.float32), w, bias, geom, L.ACT_NONE, torch.float32)
    return CastFn.apply(y32, out_dtype)


class _StagedMask:
    """A gradient that is read as ``dy x LeakyReLU'(bits)`` (bits: sign bytes in dy's layout) by kernels that apply the mask while
    they stage the operand (XmcConvDesc.mask_bits: xmc_conv_ptile_bits, xmc_conv_wgrad_bits).  A consumer whose shape those kernels
    do not take asks for ``materialised()``: one mask pass (xmc_signmask_apply), shared by the consumers of this object."""

    def __init__(self, dy, bits):
        assert bits.dtype == torch.uint8 and bits.numel() * 8 == dy.numel() and dy.is_contiguous()
        self.dy, self.bits, self._full = dy, bits, None

    def materialised(self):
        if self._full is None:
            self._full = torch.empty_like(self.dy)
            L.call("xmc_signmask_apply", _p(self.dy), _p(self.bits), _p(self._full), self.dy.numel(), 0.2, _code(self.dy.dtype), _st())
        return self._full


def _conv_dgrad_raw(dy, w, geom, in_hw, in_dtype, mask=None, res=None, res_rows=False, res_scale=1.0, alpha=None, want_sumpool=False,
                    dot=None, src_bits=None):
    """dx [N,H,W,cin_p] from dy [N,OH,OW,cout_p].  Epilogue options: ``mask`` (dx layout): dx *= LeakyReLU'(mask);
    ``res``: dx += res_scale * res, with ``res_rows`` the residual is [N,H/s,W/s,cin_p] and every pixel of it is added to its
    s x s block of dx (s == 2: the adjoint of avg_pool2d, df_gan.py:290).  ``want_sumpool`` (stride 1): returns (dx, 2x2 sum pool
    of dx) -- the adjoint of a nearest x2 upsample, the gradient of a generator block's half-resolution shortcut."""
    staged = dy if isinstance(dy, _StagedMask) else None        # dy x LeakyReLU'(bits), masked where the kernel stages it
    if staged is not None:
        dy = staged.dy
    _need_cuda(dy, w)
    N, OH, OW, CDy = dy.shape
    H, W = in_hw
    assert CDy == pad_to(geom.cout, 8)
    cs_p = chan_pad(geom.cin, in_dtype)
    wpk = _packed_cached(w, geom, 1, dy.dtype)
    if cs_p % 8:
        raise RuntimeError("dgrad destination needs a channel count that is a multiple of 8")
    dx = torch.empty((N, H, W, cs_p), dtype=in_dtype, device=dy.device)
    d = L.ConvDesc()
    d.src, d.wpk, d.dst = dy.data_ptr(), wpk.data_ptr(), dx.data_ptr()
    d.N, d.SH, d.SW, d.CS = N, OH, OW, CDy
    d.DH, d.DW, d.CD = H, W, cs_p
    s, k, p = geom.s, geom.k, geom.p
    d.MH, d.MW, d.SA, d.DA, d.src_shift = H // s, W // s, 1, s, 0
    d.nclass, d.CDw = s * s, wpk.shape[1]
    d.act, d.dtype, d.out_dtype = L.ACT_NONE, _code(dy.dtype), _code(in_dtype)
    d.groups = geom.groups
    assert H % s == 0 and W % s == 0
    ntaps = None
    for ph in range(s):
        for pw in range(s):
            cls = ph * s + pw
            taps = [((ph + p - kh) // s, (pw + p - kw) // s, kh * k + kw)
                    for kh in range(k) if (ph + p - kh) % s == 0
                    for kw in range(k) if (pw + p - kw) % s == 0]
            assert ntaps in (None, len(taps))
            ntaps = len(taps)
            _fill_taps(d, cls, taps)
            d.dph[cls], d.dpw[cls] = ph, pw
    d.ntaps = ntaps
    if alpha is not None:         # dx = alpha * dgrad(dy) (f32 device scalar, applied to the accumulator)
        d.alpha_dev = alpha.data_ptr()
    if mask is not None:
        assert mask.shape == dx.shape and mask.dtype == dx.dtype and mask.is_contiguous()
        d.mask = mask.data_ptr()
    if dot is not None:           # dot += <dgrad(dy) before alpha, mask values> (XmcConvDesc.dot), f32 [1], accumulated
        assert mask is not None and dot.dtype == torch.float32 and dot.numel() == 1
        d.dot = dot.data_ptr()
    if res is not None:
        want = (N, H // s, W // s, cs_p) if res_rows else tuple(dx.shape)
        assert tuple(res.shape) == want and res.dtype == dx.dtype and res.is_contiguous(), (res.shape, want)
        assert not res_rows or s == 2
        d.res, d.res_mode, d.res_scale = res.data_ptr(), 1 if res_rows else 0, float(res_scale)
    dxp = None
    if want_sumpool:
        assert s == 1 and H % 2 == 0 and W % 2 == 0 and in_dtype == dy.dtype
        dxp = torch.empty((N, H // 2, W // 2, cs_p), dtype=in_dtype, device=dy.device)
        d.dst_pool, d.pool_scale = dxp.data_ptr(), 1.0
    if src_bits is not None:
        # ``src_bits`` (sign bytes in dy's layout): also return dy x LeakyReLU'(bits).  The streaming 1x1 kernels write it while they
        # read dy (xmc_conv_pw1x1_masked_src); any other shape runs the data gradient and the mask pass separately.
        assert k == 1 and s == 1 and not want_sumpool and src_bits.dtype == torch.uint8 and src_bits.numel() * 8 == dy.numel()
        dym = torch.empty_like(dy)
        with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * geom.cin,
                         f"dgrad+srcmask {dy.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k1s1", _nbytes(dy, wpk, dx, dym)):
            rc = L.load().xmc_conv_pw1x1_masked_src(C.byref(d), _p(src_bits), _p(dym), 0.2, _st())
            if rc == 1:
                _igemm(d, "xmc_conv_igemm(dgrad)")
                L.call("xmc_signmask_apply", _p(dy), _p(src_bits), _p(dym), dy.numel(), 0.2, _code(dy.dtype), _st())
            else:
                L.check(rc, "xmc_conv_pw1x1_masked_src")
        return dx, dym
    if staged is not None:
        d.mask_bits = staged.bits.data_ptr()
        with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * geom.cin * geom.k * geom.k,
                         f"dgrad+bits {dy.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(dy, wpk, dx, mask, res, dxp)):
            rc = L.load().xmc_conv_ptile_bits(C.byref(d), _st())
        if rc == 0:
            return (dx, dxp) if want_sumpool else dx
        if rc != 1:
            L.check(rc, "xmc_conv_ptile_bits")
        d.mask_bits, d.src = None, staged.materialised().data_ptr()
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * OH * OW * geom.cout * geom.cin * geom.k * geom.k,
                     f"dgrad {dy.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(dy, wpk, dx, mask, res, dxp)):
        _igemm(d, "xmc_conv_igemm(dgrad)")
    return (dx, dxp) if want_sumpool else dx


class _ZeroArena:
    """Zero-filled f32 scratch for the weight-gradient kernels (packed dW accumulators, bias replicas), which accumulate with
    atomics and need zeros.  One memset per iteration (`new_iteration()`) instead of one fill launch per buffer -- ~400 launches
    per G+D iteration.  Sized by the previous iteration's demand; anything beyond falls back to torch.zeros."""

    def __init__(self):
        self.buf, self.off, self.need, self.retired, self.active = {}, {}, {}, [], set()

    def new_iteration(self, device):
        key = (device.type, device.index)
        need = self.need.get(key, 0)
        buf = self.buf.get(key)
        # grow only outside stream capture (a buffer allocated inside a capture would live in that graph's private pool), and
        # keep outgrown buffers alive: an earlier captured graph may still memset / accumulate into them on replay
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if need and (buf is None or buf.numel() < need) and not capturing:
            if buf is not None:
                self.retired.append(buf)
            buf = self.buf[key] = torch.empty(int(need * 1.05) + 1024, dtype=torch.float32, device=device)
        elif buf is not None and need and need < buf.numel() // 2 and not capturing and not _graphs_alive[0]:
            # demand fell well below the block (another model, a smaller batch): do not keep zeroing the large one every iteration
            buf = self.buf[key] = torch.empty(int(need * 1.05) + 1024, dtype=torch.float32, device=device)
        if buf is not None:
            buf.zero_()
        self.off[key], self.need[key] = 0, 0
        self.active.add(key)

    def end_iteration(self, device):
        """Slices are handed out, and demand is recorded, only between `new_iteration` and `end_iteration`: a forward-only loop
        (evaluation, sampling) never calls `new_iteration`, so its requests would pile up as `need` and size -- and re-zero, every
        iteration afterwards -- a block for the whole evaluation epoch."""
        self.active.discard((device.type, device.index))

    def zeros(self, shape, device):
        key = (device.type, device.index)
        n = 1
        for s_ in shape:
            n *= s_
        n4 = (n + 63) // 64 * 64        # 256-byte granules: a 16-byte scalar between two packed-dW accumulators shifted every later buffer
        # off its cache-line alignment and cost the iteration 0.45-0.6 ms (round 3, same-box A/B)
        if key not in self.active:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        self.need[key] = self.need.get(key, 0) + n4
        buf, off = self.buf.get(key), self.off.get(key, 0)
        if buf is None or off + n4 > buf.numel() or key not in self.off:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        self.off[key] = off + n4
        return buf[off:off + n].view(shape)


_arena = _ZeroArena()


class _EscapeBlocks:
    """Zero-filled f32 accumulators that LEAVE the kernels as tensors the caller may keep: forward outputs of `DotFn` / `ColSumFn`,
    parameter gradients (a block's d(gamma), the concept heads' weight gradients) -- autograd's AccumulateGrad adopts such a
    tensor as ``p.grad`` without copying it.  They must not alias memory that `new_iteration()` re-zeroes and re-issues, so they
    are slices of a block that is allocated FRESH once per iteration (one fill) and never recycled: what a slice's owner holds
    stays valid for as long as it holds it (``zero_grad(set_to_none=False)``, gradient accumulation over iterations, a caller
    keeping ``.grad``)."""

    GRANULE, MIN_BLOCK = 64, 1 << 14

    def __init__(self):
        self.buf, self.off, self.need, self.last = {}, {}, {}, {}

    def new_iteration(self, device):
        key = (device.type, device.index)
        self.buf.pop(key, None)                                  # owners of its slices keep the storage alive
        self.last[key], self.need[key] = self.need.get(key, 0), 0

    def zeros(self, shape, device):
        key = (device.type, device.index)
        n = 1
        for s_ in shape:
            n *= s_
        n4 = (n + self.GRANULE - 1) // self.GRANULE * self.GRANULE
        self.need[key] = self.need.get(key, 0) + n4
        buf, off = self.buf.get(key), self.off.get(key, 0)
        if buf is None or off + n4 > buf.numel():
            # a fresh block sized by what the previous iteration asked for in total (the attention-modulation generators' head
            # gradients are 100-300 KB each: with 64 KB blocks every one of their 48 requests was a block, i.e. a fill launch, of its own)
            want = max(self.MIN_BLOCK, n4, self.last.get(key, 0) - self.need[key] + n4)
            buf = self.buf[key] = torch.zeros(want, dtype=torch.float32, device=device)
            off = 0
        self.off[key] = off + n4
        return buf[off:off + n].view(shape)


_escape = _EscapeBlocks()


def _zeros_f32(shape, device):
    """zero-filled f32 scratch for an accumulator the kernels add into (dot products, per-channel sums, small parameter gradients):
    a slice of the per-iteration arena (one memset per iteration) instead of one fill launch each -- ~100 launches per iteration in
    the headline configuration, ~420 with the attention-modulation generators.  Valid until the next `new_iteration()`."""
    if isinstance(shape, int):
        shape = (shape,)
    if "no_arena_scalars" in _DEBUG_DISPATCH:
        return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    return _arena.zeros(tuple(shape), torch.device(device))


def _zeros_f32_out(shape, device):
    """zero-filled f32 accumulator whose tensor escapes to the caller (`_EscapeBlocks`)"""
    if isinstance(shape, int):
        shape = (shape,)
    if "no_arena_scalars" in _DEBUG_DISPATCH:
        return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    return _escape.zeros(tuple(shape), torch.device(device))


def new_iteration(device):
    """Call once at the start of a training iteration (before any backward): re-zeroes the weight-gradient scratch arena."""
    _arena.new_iteration(torch.device(device))
    _escape.new_iteration(torch.device(device))
    _pooled_grads.clear()


def end_iteration(device):
    """Call at the end of a training iteration: until the next `new_iteration` accumulators come from torch.zeros and record no demand."""
    _arena.end_iteration(torch.device(device))


# By-products handed from one backward node to the next: {(data_ptr, shape, dtype) of a gradient tensor: (the tensor, its 2x2 sum
# pool)}.  The node that WRITES the gradient of a generator block's output (the next block's affine backward) can pool it in the
# same pass; the node that CONSUMES it (GBlockEndFn.backward, which needs the pooled tensor as the gradient of the half-resolution
# shortcut) takes the entry instead of launching a pooling pass.  The entry HOLDS the gradient tensor, so its address cannot be
# re-issued while the entry exists (the key is unique by construction, not by allocator behaviour), and the consumer checks that
# what autograd handed it is that very storage, unmodified (a hook, a second consumer or a cast gives it another tensor: then the
# pooling pass runs).  Entries nobody took die with the iteration (new_iteration).
_pooled_grads = {}


def _pool_key(t):
    return (t.data_ptr(), tuple(t.shape), t.dtype)


def _pooled_put(dx, dxp):
    _pooled_grads[_pool_key(dx)] = (dx, dx._version, dxp)


def _pooled_take(dz):
    ent = _pooled_grads.pop(_pool_key(dz), None)
    if ent is None:
        return None
    dx, ver, dxp = ent
    same = dx.untyped_storage().data_ptr() == dz.untyped_storage().data_ptr() and dx.stride() == dz.stride() and dx._version == ver
    return dxp if same else None


def _conv_wgrad_raw(x, dy, geom, scale=None, up=False, want_bias=False, bias_dot=None, dot=None):
    """gw [Co,Ci,k,k] f32 from x [N,H,W,cs_p], dy [N,OH,OW,cd_p] (and the bias gradient [cd_p] f32 from the same launch).
    ``bias_dot`` (f32 [>= cout]) / ``dot`` (f32 [1]): dot += <bias_dot, unscaled bias gradient> (xmc_unpack_wgrad_bias_dot)."""
    staged = dy if isinstance(dy, _StagedMask) else None        # dy x LeakyReLU'(bits), masked where the kernel stages it
    if staged is not None:
        dy = staged.dy
    _need_cuda(x, dy)
    N, H, W, CS = x.shape
    _, OH, OW, CDy = dy.shape
    assert x.dtype == dy.dtype, (x.dtype, dy.dtype)
    rows = pad_to(CDy, 32)
    dwp = _arena.zeros((geom.k * geom.k, rows, CS), x.device)
    gb = _arena.zeros((16, CDy), x.device) if want_bias else None     # XMC_BIAS_REPLICAS
    d = L.ConvDesc()
    d.src, d.dst = x.data_ptr(), dy.data_ptr()
    d.N, d.SH, d.SW, d.CS = N, H, W, CS
    d.DH, d.DW, d.CD = OH, OW, CDy
    d.MH, d.MW, d.SA, d.DA, d.src_shift = OH, OW, geom.s, 1, 1 if up else 0
    d.ntaps, d.nclass, d.CDw = geom.k * geom.k, 1, rows
    d.dtype, d.out_dtype = _code(x.dtype), L.F32
    d.groups = geom.groups
    _fill_taps(d, 0, [(kh - geom.p, kw - geom.p, kh * geom.k + kw) for kh in range(geom.k) for kw in range(geom.k)])
    rc = 1
    if staged is not None and not want_bias:
        d.mask_bits = staged.bits.data_ptr()
        with prof.launch("wgrad_kernel (conv weight gradient, MFMA + split-K atomics)", 2.0 * N * OH * OW * geom.cout * geom.cin * geom.k * geom.k,
                         f"wgrad+bits {x.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(x, dy, dwp)):
            rc = L.load().xmc_conv_wgrad_bits(C.byref(d), _p(dwp), _st())
        if rc not in (0, 1):
            L.check(rc, "xmc_conv_wgrad_bits")
        d.mask_bits = None
    if rc == 1:
        if staged is not None:
            d.dst = staged.materialised().data_ptr()
        with prof.launch("wgrad_kernel (conv weight gradient, MFMA + split-K atomics)", 2.0 * N * OH * OW * geom.cout * geom.cin * geom.k * geom.k,
                         f"wgrad {x.dtype} N{N} {H}x{W} {geom.cin}->{geom.cout} k{geom.k}s{geom.s}", _nbytes(x, dy, dwp)):
            L.check(L.load().xmc_conv_wgrad_bias(C.byref(d), _p(dwp), _p(gb), _st()), "xmc_conv_wgrad")
    gw = torch.empty((geom.cout, geom.cin // geom.groups, geom.k, geom.k), dtype=torch.float32, device=x.device)
    if want_bias:
        assert geom.groups == 1
        gbs = torch.empty(CDy, dtype=torch.float32, device=x.device)
        assert (bias_dot is None) == (dot is None)
        L.call("xmc_unpack_wgrad_bias_dot", _p(dwp), _p(gw), geom.cout, geom.cin, geom.k, geom.k, rows, CS, _p(scale),
               _p(geom.perm_dev(x.device)), 0, _p(gb), _p(gbs), CDy, _p(bias_dot), _p(dot), _st())
        return gw, gbs
    assert bias_dot is None
    L.call("xmc_unpack_wgrad_grouped", _p(dwp), _p(gw), geom.cout, geom.cin, geom.k, geom.k, rows, CS, _p(scale),
           _p(geom.perm_dev(x.device)), 0, geom.groups, _st())
    return gw


# ------------------------------------------------------------------------------------------ discriminator stem (csrc/dstem.hip)
def compose_dstem(w_img, b_img, w0, ws, bs):
    """conv_img followed (without an activation) by the first resD block's conv_r[0] and by its pooled 1x1 shortcut, as ONE set of
    convolution weights on the image (df_gan.py:114,127,272-291; derivation in csrc/dstem.hip):
        W[0:64]   = sum_mid w0 (*) w_img                 6x6, stride 2, pad 2    (the residual branch's first convolution)
        W[64:128] = ws . (2x2 box / 4 (*) w_img)         4x4, stride 2, pad 1, embedded in the 6x6 window (the shortcut)
    plus the corrections of the residual branch on the image border, where conv_r[0] pads conv_img's OUTPUT with zeros: what the
    dropped taps (kh = 0 in the first output row, kh = 3 in the last, kw = 0 / 3 in the first / last column; corners added back
    once) contributed through the one image row / column conv_img reads from outside.
    returns (W f32 [128,36,8], bias f32 [128], D f32 [64,28,8], DB f32 [64,8]) -- layouts in include/xmc_gan_hip.h.  Parameter-sized
    f32 algebra.  THIS function is the readable statement of it in differentiable torch ops and the reference the tests hold the
    product's own launches against: `xmc_dstem_compose` (forward) and `xmc_dstem_compose_bwd` (its adjoint: the gradients of these
    tables, from `xmc_dstem_wgrad` / `xmc_dstem_border_wgrad`, back to the five parameters)."""
    F = torch.nn.functional
    co, mid = w0.shape[0], w0.shape[1]
    w0, b_img = w0.float(), b_img.float()
    wie = F.pad(w_img.float(), (0, 0, 0, 0, 0, 8 - w_img.shape[1]))                       # [mid, 8, 3, 3]
    wa = F.conv_transpose2d(w0, wie)                                                        # [co, 8, 6, 6]: full correlation over mid
    box = torch.full((1, 1, 2, 2), 0.25, dtype=torch.float32, device=w_img.device)
    wp = F.conv_transpose2d(wie.reshape(mid * 8, 1, 3, 3), box).reshape(mid, 8, 4, 4)     # avg_pool2d o conv_img
    wb = torch.einsum("om,mcab->ocab", ws.float()[:, :, 0, 0], wp)
    w = torch.cat((wa, F.pad(wb, (1, 1, 1, 1)))).permute(0, 2, 3, 1).reshape(2 * co, 36, 8)
    ba = torch.einsum("omhw,m->o", w0, b_img)
    bb = ws.float()[:, :, 0, 0] @ b_img + (bs.float() if bs is not None else 0.0)
    ct1 = F.conv_transpose1d
    lines = [-ct1(w0[:, :, 0, :], wie[:, :, 2, :]), -ct1(w0[:, :, 3, :], wie[:, :, 0, :]),          # first / last row, by window column
             -ct1(w0[:, :, :, 0], wie[:, :, :, 2]), -ct1(w0[:, :, :, 3], wie[:, :, :, 0])]          # first / last column, by window row
    corners = [(0, 0, 2, 2), (0, 3, 2, 0), (3, 0, 0, 2), (3, 3, 0, 0)]                              # (kh, kw, ih, iw) of TL TR BL BR
    D = torch.cat([t.permute(0, 2, 1) for t in lines] + [(w0[:, :, kh, kw] @ wie[:, :, ih, iw]).unsqueeze(1) for kh, kw, ih, iw in corners], 1)
    DB = torch.stack([-(w0[:, :, 0, :].sum(2) @ b_img), -(w0[:, :, 3, :].sum(2) @ b_img), -(w0[:, :, :, 0].sum(2) @ b_img),
                      -(w0[:, :, :, 3].sum(2) @ b_img)] + [w0[:, :, kh, kw] @ b_img for kh, kw, _, _ in corners], 1)
    return w.contiguous(), torch.cat((ba, bb)).contiguous(), D.contiguous(), DB.contiguous()


def _dstem_compose_raw(w_img, b_img, w0, ws, bs):
    """`compose_dstem` as one launch (xmc_dstem_compose): the four f32 tables of the composed stem"""
    _need_cuda(w_img, w0)
    dev = w_img.device
    assert tuple(w_img.shape) == (32, 3, 3, 3) and tuple(w0.shape) == (64, 32, 4, 4) and tuple(ws.shape[:2]) == (64, 32)
    flat = torch.empty(128 * 36 * 8 + 128 + 64 * 28 * 8 + 64 * 8, dtype=torch.float32, device=dev)
    W, b = flat[:36864].view(128, 36, 8), flat[36864:36992]
    D, DB = flat[36992:36992 + 14336].view(64, 28, 8), flat[36992 + 14336:].view(64, 8)
    f = lambda t: None if t is None else t.detach().float().contiguous()
    wi_, bi_, w0_, ws_, bs_ = f(w_img), f(b_img), f(w0), f(ws), f(bs)
    L.call("xmc_dstem_compose", _p(wi_), _p(bi_), _p(w0_), _p(ws_), _p(bs_), _p(W), _p(b), _p(D), _p(DB), _st())
    return W, b, D, DB


def _dstem_compose_bwd_raw(w_img, b_img, w0, ws, bs, dW, dbias, dD, dDB):
    """gradients of (w_img, b_img, w0, ws, bs) from the gradients of the four tables (xmc_dstem_compose_bwd)"""
    f = lambda t: t.detach().float().contiguous()
    wi_, bi_, w0_, ws_ = f(w_img), f(b_img), f(w0), f(ws)
    dwi, dbi, dw0, dws = (torch.empty_like(t) for t in (wi_, bi_, w0_, ws_))
    dbs = torch.empty(64, dtype=torch.float32, device=wi_.device) if bs is not None else None
    L.call("xmc_dstem_compose_bwd", _p(wi_), _p(bi_), _p(w0_), _p(ws_), _p(dW.contiguous()), _p(dbias.contiguous()), _p(dD.contiguous()),
           _p(dDB.contiguous()), _p(dwi), _p(dbi), _p(dw0), _p(dws), _p(dbs), _st())
    return dwi, dbi, dw0, dws, dbs


def _dstem_fwd_raw(xin, wsets, bias, slope=0.2, want_sc=True):
    """h1 = lrelu(W_A * x + b_A) [N,H/2,W/2,64], sc = W_B * x + b_B [N,H/2,W/2,64] from the image xin [N,H,W,8] (border pixels of h1
    are the composition's, not the reference's: see DStemBlockFn).  ``want_sc`` False: sc is None (its consumer recomputes it)."""
    _need_cuda(xin, wsets)
    N, H, W, _ = xin.shape
    wfrag = torch.empty(8 * 5 * 64 * 8, dtype=xin.dtype, device=xin.device)         # 8 row blocks x 5 K steps of MFMA A fragments
    L.call("xmc_dstem_pack", _p(wsets), _p(wfrag), _st())
    h1 = torch.empty((N, H // 2, W // 2, 64), dtype=xin.dtype, device=xin.device)
    sc = torch.empty_like(h1) if want_sc else None
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * (H // 2) * (W // 2) * 64 * (3 * 36 + (3 * 16 if want_sc else 0)),
                     f"dstem-fwd {xin.dtype} N{N} {H}x{W} 3->64{'+64' if want_sc else ''} k6s2", _nbytes(xin, h1, sc)):
        L.call("xmc_dstem_fwd", _p(xin), _p(wfrag), _p(bias), _p(h1), _p(sc), N, H, W, float(slope), _st())
    return h1, sc


def _dstem_sc_operands(xin, wsets, bias):
    """(image, sc_frag, sc_bias) for `_conv_fwd_raw(sc_img=...)`: the shortcut rows of the composed table as the block-end kernel's MFMA fragments"""
    frag = torch.empty(4 * 2 * 64 * 8, dtype=xin.dtype, device=xin.device)
    L.call("xmc_dstem_pack_sc", _p(wsets), _p(frag), _st())
    return xin, frag, bias[64:128]


def _dstem_border_fwd_raw(xin, wsets, bias, D, DB, h1, slope=0.2):
    """overwrite h1's border pixels with the reference's values (composed weights + border corrections, f32 arithmetic)"""
    N, H, W, _ = xin.shape
    wt = torch.empty(64 * 1024, dtype=torch.uint8, device=xin.device)              # the kernel's MFMA fragments of W + D (16-bit)
    L.call("xmc_dstem_border_fwd", _p(xin), _p(wsets), _p(bias), _p(D), _p(DB), _p(wt), _p(h1), N, H, W, float(slope), _st())


def _dstem_dgrad_raw(dh1, dsc, wsets, D, H, W):
    """gradient of the image [N,H,W,8] from (d h1 in front of its LeakyReLU, d shortcut): the adjoint of the composed stem"""
    _need_cuda(dh1, dsc)
    N = dh1.shape[0]
    dh1, dsc = dh1.contiguous(), dsc.contiguous()
    frag = torch.empty(36 * 1024, dtype=torch.uint8, device=dh1.device)
    dimg = torch.empty((N, H, W, 8), dtype=dh1.dtype, device=dh1.device)
    with prof.launch("igemm_kernel (conv fwd+dgrad, MFMA implicit GEMM)", 2.0 * N * (H // 2) * (W // 2) * 64 * (3 * 36 + 3 * 16),
                     f"dstem-dgrad {dh1.dtype} N{N} {H}x{W} 64+64->3 k6s2", _nbytes(dh1, dsc, dimg)):
        L.call("xmc_dstem_dgrad", _p(dh1), _p(dsc), _p(wsets), _p(D), _p(frag), _p(dimg), N, H, W, _st())
    return dimg


def _dstem_wgrad_raw(xin, dh1, dsc, skip_border=False, border=True):
    """gradients of the composed weights / biases and (``border``) of the border corrections:
    (dW f32 [128,36,8], dbias f32 [128], dD f32 [64,28,8], dDB f32 [64,8])"""
    _need_cuda(xin, dh1, dsc)
    N, H, W, _ = xin.shape
    dh1, dsc = dh1.contiguous(), dsc.contiguous()
    n_w, n_d = 128 * 36 * 8, 64 * 28 * 8
    flat = _arena.zeros((n_w + 128 + n_d + 64 * 8,), xin.device)
    dw, db = flat[:n_w].view(128, 36, 8), flat[n_w:n_w + 128]
    dD, dDB = flat[n_w + 128:n_w + 128 + n_d].view(64, 28, 8), flat[n_w + 128 + n_d:].view(64, 8)
    with prof.launch("wgrad_kernel (conv weight gradient, MFMA + split-K atomics)", 2.0 * N * (H // 2) * (W // 2) * 64 * (3 * 36 + 3 * 16),
                     f"dstem-wgrad {xin.dtype} N{N} {H}x{W} 3->64+64 k6s2", _nbytes(xin, dh1, dsc)):
        L.call("xmc_dstem_wgrad", _p(xin), _p(dh1), _p(dsc), _p(dw), _p(db), N, H, W, 1 if skip_border else 0, _st())
    if border:
        L.call("xmc_dstem_border_wgrad", _p(xin), _p(dh1), _p(dD), _p(dDB), N, H, W, _st())
    return dw, db, dD, dDB


# ------------------------------------------------------------------------------------------ conv / linear
class ConvFn(torch.autograd.Function):
    """y = act(conv2d(x, w) + b).  F.conv2d / nn.Linear call sites: df_gan.py:73-74,86,144,157-159,187-188,
    197,233-240,273,276,280."""

    @staticmethod
    def forward(ctx, x, w, b, geom, act, out_dtype, want_pool=False, out=None, pair=False):
        """``want_pool``: returns (y, avg_pool2d(y, 2)); the pooled tensor is a by-product for the consumer's shortcut branch
        (written from the epilogue where the kernel can) and carries no gradient of its own.  ``out``: destination tensor.
        ``pair``: a 1x1 layer of the precise trunk -- forward on the weights' hi + lo pair (`_conv1x1_pair_raw`), backward as ever."""
        x = x.contiguous()
        # `out` is written behind autograd's back (no version bump): it must be a tensor no earlier node has saved
        assert out is None or out._version == 0, "ConvFn(out=): the destination must be a fresh tensor"
        bp = None
        if b is not None:
            cd_p = pad_to(geom.cout, 8)
            bp = b.detach().float()
            if geom.row_perm is not None:
                bp = bp.index_select(0, geom.perm_dev(b.device).long())
            if bp.numel() < cd_p:
                bp = torch.nn.functional.pad(bp, (0, cd_p - bp.numel()))
            bp = bp.contiguous()
        if pair and x.dtype != torch.float32 and act == L.ACT_NONE and not want_pool and out is None:
            y = _conv1x1_pair_raw(x, w, bp, geom, out_dtype)
        else:
            y = _conv_fwd_raw(x, w, bp, geom, act, out_dtype, want_pool=want_pool, out=out)
        yp = None
        if want_pool:
            y, yp = y
        ctx.geom, ctx.act, ctx.has_b = geom, act, b is not None
        ctx.save_for_backward(x, w, y if act != L.ACT_NONE else None)
        if want_pool:
            ctx.mark_non_differentiable(yp)
            ctx.set_materialize_grads(False)      # no zero-filled gradient tensor for the pooled by-product on every backward
            return y, yp
        return y

    @staticmethod
    def backward(ctx, dy, _dyp=None):
        if dy is None:
            return None, None, None, None, None, None, None, None, None
        x, w, y = ctx.saved_tensors
        geom = ctx.geom
        dy = dy.contiguous()
        if ctx.act in (L.ACT_LRELU, L.ACT_RELU):
            dy = MaskFn.apply(dy, y, 0.2 if ctx.act == L.ACT_LRELU else 0.0)
        elif ctx.act == L.ACT_TANH:
            dy = TanhBwdFn.apply(dy, y)
        if dy.dtype != x.dtype:
            dy = CastFn.apply(dy, x.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ConvDgradFn.apply(dy, w, geom, (x.shape[1], x.shape[2]), x.dtype)
        if not _skip_wgrad():
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                if want_b:                      # bias gradient rides on the weight-gradient launch
                    dw, db = ConvWgradBiasFn.apply(x, dy, geom)
                    dw = dw.view(w.shape)
                else:
                    dw = ConvWgradFn.apply(x, dy, geom).view(w.shape)
            elif want_b:
                db = ColSumFn.apply(dy)
            if db is not None:
                if geom.row_perm is not None:
                    db = torch.zeros_like(db).index_copy(0, geom.perm_dev(db.device).long(), db)
                db = db[: geom.cout]
        return dx, dw, db, None, None, None, None, None, None


class PairConvFn(torch.autograd.Function):
    """y = act(conv2d(x, w)) for an f32 x [N,H,W,C] at f32 GRADE on the 16-bit matrix pipeline (the precise trunk's COND_DNET,
    df_gan.py:157-159,170-175): both operands as 16-bit pairs, x = xh + xl, w = wh + wl (`_packed_cached(lo=True)`), and the three
    products that matter, conv(xh, wh) + conv(xl, wh) + conv(xh, wl) (the fourth is 2^-22 of the result), as three launches of the
    16-bit kernel that accumulate in an f32 destination; the activation runs on the last one (XmcConvDesc.post_act).  Against the
    exact-f32 MFMA kernel (1/16 of the rate): 0.77 -> ~0.3 ms per iteration for joint_conv.0.  The backward is the 16-bit layer's
    (xh, wh): gradients keep the bars of the 16-bit modes."""

    @staticmethod
    def forward(ctx, x, w, geom, act):
        dt = act_dtype()
        assert x.dtype == torch.float32 and dt != torch.float32
        x = x.contiguous()
        xh = CastFn.apply(x, dt)
        xl = CastFn.apply(x - xh.float(), dt)
        y = _conv_fwd_raw(xh, w, None, geom, L.ACT_NONE, torch.float32)
        y = _conv_fwd_raw(xl, w, None, geom, L.ACT_NONE, torch.float32, res=y)
        y = _conv_fwd_raw(xh, w, None, geom, L.ACT_NONE, torch.float32, res=y, w_lo=True, post_act=act)
        ctx.geom, ctx.act = geom, act
        ctx.save_for_backward(xh, w, y if act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None
        xh, w, y = ctx.saved_tensors
        geom = ctx.geom
        dy = dy.contiguous()
        if ctx.act in (L.ACT_LRELU, L.ACT_RELU):
            dy = MaskFn.apply(dy, y, 0.2 if ctx.act == L.ACT_LRELU else 0.0)
        dy = CastFn.apply(dy, xh.dtype)
        dx = CastFn.apply(ConvDgradFn.apply(dy, w, geom, (xh.shape[1], xh.shape[2]), xh.dtype), torch.float32) if ctx.needs_input_grad[0] else None
        dw = ConvWgradFn.apply(xh, dy, geom).view(w.shape) if (ctx.needs_input_grad[1] and not _skip_wgrad()) else None
        return dx, dw, None, None


def pair_conv2d(x, w, geom, act=L.ACT_NONE):
    return PairConvFn.apply(x, w, geom, act)


class ConvDgradFn(torch.autograd.Function):
    """dx of ConvFn (a transposed convolution); linear in dy and in w."""

    @staticmethod
    def forward(ctx, dy, w, geom, in_hw, in_dtype):
        dy = dy.contiguous()
        dx = _conv_dgrad_raw(dy, w, geom, in_hw, in_dtype)
        ctx.geom = geom
        ctx.save_for_backward(dy, w)
        return dx

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        geom = ctx.geom
        g = g.contiguous()
        if g.dtype != dy.dtype:
            g = CastFn.apply(g, dy.dtype)
        ddy = dw = None
        if ctx.needs_input_grad[0]:
            ddy = ConvFn.apply(g, w, None, geom, L.ACT_NONE, dy.dtype)
        if ctx.needs_input_grad[1] and not _skip_wgrad():
            dw = ConvWgradFn.apply(g, dy, geom).view(w.shape)
        return ddy, dw, None, None, None


class ConvWgradFn(torch.autograd.Function):
    """dw of ConvFn; bilinear in (x, dy)."""

    @staticmethod
    def forward(ctx, x, dy, geom):
        x, dy = x.contiguous(), dy.contiguous()
        gw = _conv_wgrad_raw(x, dy, geom)
        ctx.geom = geom
        ctx.save_for_backward(x, dy)
        return gw

    @staticmethod
    def backward(ctx, ggw):
        x, dy = ctx.saved_tensors
        geom = ctx.geom
        ggw = ggw.contiguous().view(geom.cout, geom.cin // geom.groups, geom.k, geom.k)
        dx = ddy = None
        if ctx.needs_input_grad[0]:
            dx = ConvDgradFn.apply(dy, ggw, geom, (x.shape[1], x.shape[2]), x.dtype)
        if ctx.needs_input_grad[1]:
            ddy = ConvFn.apply(x, ggw, None, geom, L.ACT_NONE, dy.dtype)
        return dx, ddy, None


class ConvWgradBiasFn(torch.autograd.Function):
    """ConvWgradFn that also returns the bias gradient (column sums of dy) from the same kernel launch."""

    @staticmethod
    def forward(ctx, x, dy, geom):
        x, dy = x.contiguous(), dy.contiguous()
        gw, gb = _conv_wgrad_raw(x, dy, geom, want_bias=True)
        ctx.geom = geom
        ctx.save_for_backward(x, dy)
        ctx.mark_non_differentiable(gb)
        return gw, gb

    @staticmethod
    def backward(ctx, ggw, _ggb):
        x, dy = ctx.saved_tensors
        geom = ctx.geom
        ggw = ggw.contiguous().view(geom.cout, geom.cin // geom.groups, geom.k, geom.k)
        dx = ddy = None
        if ctx.needs_input_grad[0]:
            dx = ConvDgradFn.apply(dy, ggw, geom, (x.shape[1], x.shape[2]), x.dtype)
        if ctx.needs_input_grad[1]:
            ddy = ConvFn.apply(x, ggw, None, geom, L.ACT_NONE, dy.dtype)
        return dx, ddy, None


class UpConvFn(torch.autograd.Function):
    """conv3x3(F.interpolate(x, scale_factor=2), w) + b as ONE operator on the low-resolution tensor
    (df_gan.py:202 of block i followed by c1 of block i+1, 187/217).  First-order only (generator path)."""

    @staticmethod
    def forward(ctx, x, w, b, geom):
        x = x.contiguous()
        bp = None
        if b is not None:
            cd_p = pad_to(geom.cout, 8)
            bp = b.detach().float()
            if bp.numel() < cd_p:
                bp = torch.nn.functional.pad(bp, (0, cd_p - bp.numel()))
            bp = bp.contiguous()
        y = _upconv_fwd_raw(x, w, bp, geom, L.ACT_NONE, x.dtype)
        ctx.geom, ctx.has_b = geom, b is not None
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        geom = ctx.geom
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _upconv_dgrad_raw(dy, w, geom, x.dtype)
        if not _skip_wgrad():
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                # the weight gradient is taken w.r.t. the original 3x3 taps: wgrad kernel reading x through the x2 upsample
                r = _conv_wgrad_raw(x, dy, geom, up=True, want_bias=want_b)
                dw, db = (r if want_b else (r, None))
                dw = dw.view(w.shape)
            elif want_b:
                db = ColSumFn.apply(dy)
            if db is not None:
                db = db[: geom.cout]
        return dx, dw, db, None


def _axpby_bwd_fused(dy, b, alpha, up, ymask=None, want_db=True):
    """(da, db, dalpha) of a + alpha*b / up2(a) + alpha*b from one pass over dy and b (not differentiable again).
    ``ymask``: the forward applied LeakyReLU to the sum; dy is multiplied by LeakyReLU'(y) first.  ``want_db`` False: alpha*dy
    is not written (db is None); the caller hands alpha to the consumers of db instead."""
    dy = dy.contiguous()
    N, OH, OW, Cc = dy.shape
    H, W = (OH // 2, OW // 2) if up else (OH, OW)
    al = alpha.detach().reshape(-1).float()
    db = torch.empty_like(dy) if want_db else None
    da = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device) if (up or ymask is not None) else None
    dot = _zeros_f32_out(1, dy.device)
    L.call("xmc_axpby_bwd", _p(dy), _p(b), _p(al), _p(db), _p(da), _p(dot), N, H, W, Cc, 1 if up else 0, _p(ymask), _code(dy.dtype), _st())
    return da, db, dot.reshape(alpha.shape).to(alpha.dtype)


class AxpbyUpFn(torch.autograd.Function):
    """up2(a) + alpha*b without materialising up2(a): the block output `upsample(shortcut) + gamma*residual`."""

    @staticmethod
    def forward(ctx, a, b, alpha, lrelu=False):
        a, b = a.contiguous(), b.contiguous()
        N, H, W, Cc = a.shape
        assert b.shape == (N, 2 * H, 2 * W, Cc)
        al = alpha.detach().reshape(-1).float()
        y = torch.empty_like(b)
        L.call("xmc_axpby_up_lrelu" if lrelu else "xmc_axpby_up", _p(a), _p(b), _p(al), _p(y), N, H, W, Cc, _code(a.dtype), _st())
        ctx.lrelu = lrelu
        ctx.save_for_backward(b, alpha, y if lrelu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, alpha, y = ctx.saved_tensors
        if not torch.is_grad_enabled() and all(ctx.needs_input_grad[:3]) and fused_blocks():
            return _axpby_bwd_fused(dy, b, alpha, up=True, ymask=y) + (None,)      # first-order: one pass over dy and b
        if ctx.lrelu:
            dy = MaskFn.apply(dy.contiguous(), y, 0.2)
        da = SumPool2Fn.apply(dy, 1.0) if ctx.needs_input_grad[0] else None
        db = ScaleFn.apply(dy, alpha) if ctx.needs_input_grad[1] else None
        dal = DotFn.apply(dy, b).reshape(alpha.shape) if ctx.needs_input_grad[2] else None
        return da, db, dal, None


class ConvAxpbyUpFn(torch.autograd.Function):
    """The end of a generator block as ONE pass: up2(shortcut) + gamma * (conv3x3(h) + b)  (df_gan.py:197-202: c2, the block
    sum, F.interpolate of the previous block's output folded in as a half-resolution residual read, XmcConvDesc.res_mode 2).
    The convolution output itself (needed for d(gamma)) is the epilogue's second output.  First order only (generator path)."""

    @staticmethod
    def forward(ctx, h, w, b, geom, sc_lo, gamma):
        h, sc_lo = h.contiguous(), sc_lo.contiguous()
        bp = None
        if b is not None:
            cd_p = pad_to(geom.cout, 8)
            bp = b.detach().float()
            if bp.numel() < cd_p:
                bp = torch.nn.functional.pad(bp, (0, cd_p - bp.numel()))
            bp = bp.contiguous()
        al = gamma.detach().reshape(-1).float()
        y, res = _conv_fwd_raw(h, w, bp, geom, L.ACT_NONE, h.dtype, res=sc_lo, alpha=al, res_mode=2, want2=True)
        ctx.geom, ctx.has_b = geom, b is not None
        ctx.save_for_backward(h, w, res, gamma)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        h, w, res, gamma = ctx.saved_tensors
        geom = ctx.geom
        # one pass over dy and res; gamma*dy itself is never written: the data gradient applies gamma to its accumulator, the weight
        # (and bias) gradient in its unpack
        dy = dy.contiguous()
        al = gamma.detach().reshape(-1).float()
        dsc, _, dgamma = _axpby_bwd_fused(dy, res, gamma, up=True, want_db=False)
        dh = _conv_dgrad_raw(dy, w, geom, (h.shape[1], h.shape[2]), h.dtype, alpha=al) if ctx.needs_input_grad[0] else None
        dw = db = None
        if not _skip_wgrad():
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                r = _conv_wgrad_raw(h, dy, geom, scale=al, want_bias=want_b)
                dw, db = (r if want_b else (r, None))
                dw = dw.view(w.shape)
            elif want_b:
                db = ColSumFn.apply(dy) * al
            if db is not None:
                db = db[: geom.cout]
        return dh, dw, db, None, (dsc if ctx.needs_input_grad[4] else None), (dgamma if ctx.needs_input_grad[5] else None)


def conv_axpby_up(h, w, b, geom, sc_lo, gamma):
    return ConvAxpbyUpFn.apply(h, w, b, geom, sc_lo, gamma)


def _bias_padded(b, geom):
    if b is None:
        return None
    bp = b.detach().float()
    cd_p = pad_to(geom.cout, 8)
    if bp.numel() < cd_p:
        bp = torch.nn.functional.pad(bp, (0, cd_p - bp.numel()))
    return bp.contiguous()


class GBlockEndFn(torch.autograd.Function):
    """The second half of a generator block as ONE first-order node (df_gan.py:199-202,219-224):
        h1 -> affine2, LeakyReLU, affine3, LeakyReLU -> c2 -> up2(shortcut) + gamma * (.)
    and, for the last block, the network's tail as well (df_gan.py:84-88): -> LeakyReLU -> conv_out -> tanh.
    What the node buys over its parts (Affine2LreluFn, ConvAxpbyUpFn / AxpbyUpFn, ConvFn):
      * c2's own output is never stored.  d(gamma) = <dout, c2(h2) + b2> = <c2^T dout, h2> + <b2, colsum(dout)>: the first term
        is accumulated by the affine backward kernel, which recomputes h2 anyway and receives c2^T dout UNSCALED (it applies
        gamma itself, so gamma = 0 -- the reference's initial value -- loses nothing); the second rides on the unpack of c2's
        bias gradient.  One hi-res write in the forward and one hi-res read in the backward less per block.
      * last block: the block sum is written once, already through the tail's LeakyReLU (XmcConvDesc.post_act), and its
        LeakyReLU' mask is applied in the epilogue of conv_out's data gradient: the sum, gamma * dout and the masked gradient
        are not separate passes over the largest tensor of the generator (256 x 256 x 32 per image)."""

    @staticmethod
    def forward(ctx, h1, g0, b0, g1, b1, w2, b2, geom2, sc_lo, gamma, w_out=None, b_out=None, geom_out=None, nhwc_dst=None):
        h1, sc_lo = h1.contiguous(), sc_lo.contiguous()
        ps = [t.contiguous().float() for t in (g0, b0, g1, b1)]
        h2 = _affine_fwd_raw(h1, ps, 0.2)
        al = gamma.detach().reshape(-1).float()
        tail = w_out is not None
        # round_act: c2's output is rounded to the storage format before it enters the sum, as when it was stored (same rounding
        # points as the unfused sequence and as the quantisation-aware oracle)
        y = _conv_fwd_raw(h2, w2, _bias_padded(b2, geom2), geom2, L.ACT_NONE, h2.dtype, res=sc_lo, alpha=al, res_mode=2,
                          round_act=True, post_act=L.ACT_LRELU if tail else L.ACT_NONE)
        img = None
        if tail:
            assert nhwc_dst is None or nhwc_dst._version == 0, "GBlockEndFn(nhwc_dst=): the destination must be a fresh tensor"
            img = _conv_fwd_raw(y, w_out, _bias_padded(b_out, geom_out), geom_out, L.ACT_TANH, y.dtype, out=nhwc_dst)
        ctx.geom2, ctx.geom_out, ctx.tail = geom2, geom_out, tail
        ctx.has_b2, ctx.has_bo = b2 is not None, b_out is not None
        ctx.save_for_backward(h1, *ps, h2, w2, b2, gamma, y if tail else None, w_out, img)
        return img if tail else y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        h1, g0, b0, g1, b1, h2, w2, b2, gamma, y, w_out, img = ctx.saved_tensors
        geom2, geom_out = ctx.geom2, ctx.geom_out
        dt = h1.dtype
        N, H, W, _ = h2.shape
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        skip_w = _skip_wgrad()
        dw_out = db_out = None
        if ctx.tail:
            dpre = torch.empty_like(dy)
            L.call("xmc_tanh_bwd", _p(dy), _p(img), _p(dpre), dy.numel(), _code(dt), _st())
            if not skip_w:
                r = _conv_wgrad_raw(y, dpre, geom_out, want_bias=ctx.has_bo)
                dw_out, db_out = r if ctx.has_bo else (r, None)
                dw_out = dw_out.view(w_out.shape)
                if db_out is not None:
                    db_out = db_out[: geom_out.cout]
            # gradient of the block SUM: conv_out's data gradient times LeakyReLU'(y) (sign(y) = sign(sum)), in its epilogue
            # ... and its 2x2 sums, the gradient of the half-resolution shortcut, as the epilogue's pooled output
            dz, dsc = _conv_dgrad_raw(dpre, w_out, geom_out, (H, W), dt, mask=y, want_sumpool=True)
        else:
            dz = dy
            dsc = _pooled_take(dz)          # written by the producer of dy in the pass that wrote dy
            if dsc is None:
                dsc = torch.empty((N, H // 2, W // 2, dz.shape[3]), dtype=dt, device=dz.device)
                L.call("xmc_sumpool2", _p(dz), _p(dsc), N, H, W, dz.shape[3], 1.0, _code(dt), _st())
        al = gamma.detach().reshape(-1).float()
        dot = _zeros_f32_out(1, dz.device)
        dw2 = db2 = None
        if ctx.has_b2:
            # (the bias term of d(gamma) rides on the unpack of the bias gradient, so that launch runs even when the weight gradients are
            # skipped)
            bdot = _bias_padded(b2, geom2)
            dw2, db2 = _conv_wgrad_raw(h2, dz, geom2, scale=al, want_bias=True, bias_dot=bdot, dot=dot)
            dw2, db2 = dw2.view(w2.shape), db2[: geom2.cout]
        elif not skip_w:
            dw2 = _conv_wgrad_raw(h2, dz, geom2, scale=al).view(w2.shape)
        dh2u = _conv_dgrad_raw(dz, w2, geom2, (H, W), dt)                         # c2^T dz, NOT yet times gamma
        dh1, red = _affine_bwd_raw(h1, dh2u, (g0, b0, g1, b1), 0.2, alpha=al, dot=dot)
        dgamma = dot.reshape(gamma.shape).to(gamma.dtype)
        if skip_w:
            dw2 = db2 = None
        return (dh1, red[0], red[1], red[2], red[3], dw2, db2, None, dsc, dgamma, dw_out, db_out, None, None)


def g_block_end(h1, mod4, c2w, c2b, geom2, sc_lo, gamma, tail=None, nhwc_dst=None):
    """``tail``: (conv_out weight, bias, geometry) for the last block -> the tanh image in the engine layout."""
    if tail is None:
        return GBlockEndFn.apply(h1, *mod4, c2w, c2b, geom2, sc_lo, gamma)
    return GBlockEndFn.apply(h1, *mod4, c2w, c2b, geom2, sc_lo, gamma, tail[0], tail[1], tail[2], nhwc_dst)


def conv2d(x, w, b, geom, act=L.ACT_NONE, out_dtype=None, want_pool=False, out=None, pair=False):
    return ConvFn.apply(x, w, b, geom, act, out_dtype or x.dtype, want_pool, out, pair)


def linear(x, w, b, geom, act=L.ACT_NONE, out_dtype=None):
    """x [B,K] -> [B,cout_p] through the 1x1 path (nn.Linear: df_gan.py:73-74,144,233-240)."""
    y = ConvFn.apply(x.contiguous().view(x.shape[0], 1, 1, x.shape[1]), w, b, geom, act, out_dtype or x.dtype)
    return y.view(x.shape[0], -1)


# ------------------------------------------------------------------------------------------ conditioning MLP bank
def _gemm_group(tab):
    L.call("xmc_gemm_group", C.c_void_p(tab.ctypes.data), len(tab), _st())


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    return off


class CondMLPBankFn(torch.autograd.Function):
    """All G conditioning MLPs  y_g = Linear2_g(ReLU(Linear1_g(c)))  (df_gan.py:232-241; two per `affine`, four affines per
    G_Block, every block reads the same sentence embedding c) as grouped GEMM launches: 2 forward (x ceil(G/32) kernel-argument
    chunks), 3 backward (+1 when c needs a gradient) instead of 2 / 4-5 launches PER MLP.

    apply(c, w1_0, b1_0, w2_0, b2_0, w1_1, ...) -> (y_0 [B,C_0], y_1, ...), all f32.  Once differentiable (the generator is
    never differentiated twice: MA-GP is a discriminator-only term, train_gan.py:232-252)."""

    @staticmethod
    def forward(ctx, c, *params):
        assert len(params) % 4 == 0 and c.dtype == torch.float32 and c.is_cuda
        c = c.contiguous()
        G = len(params) // 4
        w1, b1, w2, b2 = params[0::4], params[1::4], params[2::4], params[3::4]
        B, K = c.shape
        Hd = w1[0].shape[0]
        for g in range(G):
            assert w1[g].shape == (Hd, K) and w2[g].shape[1] == Hd and w1[g].is_contiguous() and w2[g].is_contiguous()
            assert w1[g].dtype == torch.float32 and w2[g].dtype == torch.float32
        Cs = np.array([w.shape[0] for w in w2], dtype=np.int64)
        yoff = _offsets(Cs * B)
        h = torch.empty(G, B, Hd, dtype=torch.float32, device=c.device)
        y = torch.empty(int(yoff[-1]), dtype=torch.float32, device=c.device)
        ptr = lambda ts: np.array([t.data_ptr() for t in ts], dtype=np.uint64)
        pw1, pb1, pw2, pb2 = ptr(w1), ptr(b1), ptr(w2), ptr(b2)
        ph = (h.data_ptr() + np.arange(G, dtype=np.int64) * (B * Hd * 4)).astype(np.uint64)
        py = (y.data_ptr() + yoff[:-1] * 4).astype(np.uint64)
        t = np.zeros(G, dtype=L.GEMM_PROBLEM)            # h_g = relu(c W1_g^T + b1_g)
        t["A"], t["B"], t["bias"], t["C"] = c.data_ptr(), pw1, pb1, ph
        t["M"], t["N"], t["K"] = B, Hd, K
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = K, 1, K, 1
        t["flags"] = L.GP_BIAS | L.GP_RELU
        _gemm_group(t)
        t = np.zeros(G, dtype=L.GEMM_PROBLEM)            # y_g = h_g W2_g^T + b2_g
        t["A"], t["B"], t["bias"], t["C"] = ph, pw2, pb2, py
        t["M"], t["N"], t["K"] = B, Cs, Hd
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = Hd, 1, Hd, 1
        t["flags"] = L.GP_BIAS
        _gemm_group(t)
        ctx.save_for_backward(c, h, *w1, *w2)
        ctx.dims = (G, B, K, Hd, Cs, yoff)
        outs = tuple(y[int(yoff[g]):int(yoff[g + 1])].view(B, int(Cs[g])) for g in range(G))
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *dys):
        G, B, K, Hd, Cs, yoff = ctx.dims
        saved = ctx.saved_tensors
        c, h, w1, w2 = saved[0], saved[1], saved[2:2 + G], saved[2 + G:2 + 2 * G]
        dev = c.device
        dys = [torch.zeros(B, int(Cs[g]), dtype=torch.float32, device=dev) if d is None else d.contiguous().float()
               for g, d in enumerate(dys)]
        ptr = lambda ts: np.array([t.data_ptr() for t in ts], dtype=np.uint64)
        pdy, pw1, pw2 = ptr(dys), ptr(w1), ptr(w2)
        steps = np.arange(G, dtype=np.int64)
        ph = (h.data_ptr() + steps * (B * Hd * 4)).astype(np.uint64)
        dh = torch.empty(G, B, Hd, dtype=torch.float32, device=dev)
        pdh = (dh.data_ptr() + steps * (B * Hd * 4)).astype(np.uint64)
        w2off = _offsets(Cs * Hd)
        dw2 = torch.empty(int(w2off[-1]), dtype=torch.float32, device=dev)
        boff = _offsets(Cs)
        db2 = torch.empty(int(boff[-1]), dtype=torch.float32, device=dev)
        dw1 = torch.empty(G, Hd, K, dtype=torch.float32, device=dev)
        db1 = torch.empty(G, Hd, dtype=torch.float32, device=dev)
        t = np.zeros(G, dtype=L.GEMM_PROBLEM)            # dh_g = (dy_g W2_g) * relu'(h_g)
        t["A"], t["B"], t["mask"], t["C"] = pdy, pw2, ph, pdh
        t["M"], t["N"], t["K"] = B, Hd, Cs
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = Cs, 1, 1, Hd
        t["flags"] = L.GP_MASK
        _gemm_group(t)
        t = np.zeros(G, dtype=L.GEMM_PROBLEM)            # dW2_g = dy_g^T h_g, db2_g = colsum(dy_g)
        t["A"], t["B"] = pdy, ph
        t["C"] = (dw2.data_ptr() + w2off[:-1] * 4).astype(np.uint64)
        t["rowsum"] = (db2.data_ptr() + boff[:-1] * 4).astype(np.uint64)
        t["M"], t["N"], t["K"] = Cs, Hd, B
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = 1, Cs, 1, Hd
        _gemm_group(t)
        t = np.zeros(G, dtype=L.GEMM_PROBLEM)            # dW1_g = dh_g^T c, db1_g = colsum(dh_g)
        t["A"], t["B"] = pdh, c.data_ptr()
        t["C"] = (dw1.data_ptr() + steps * (Hd * K * 4)).astype(np.uint64)
        t["rowsum"] = (db1.data_ptr() + steps * (Hd * 4)).astype(np.uint64)
        t["M"], t["N"], t["K"] = Hd, K, B
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = 1, Hd, 1, K
        _gemm_group(t)
        dc = None
        if ctx.needs_input_grad[0]:                      # dc = sum_g dh_g W1_g
            dc = _zeros_f32((B, K), dev)
            t = np.zeros(G, dtype=L.GEMM_PROBLEM)
            t["A"], t["B"], t["C"] = pdh, pw1, dc.data_ptr()
            t["M"], t["N"], t["K"] = B, K, Hd
            t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = Hd, 1, 1, K
            t["flags"] = L.GP_ATOMIC
            _gemm_group(t)
        grads = [dc]
        for g in range(G):
            Cg = int(Cs[g])
            grads += [dw1[g], db1[g], dw2[int(w2off[g]):int(w2off[g + 1])].view(Cg, Hd), db2[int(boff[g]):int(boff[g + 1])]]
        return tuple(grads)


def cond_mlp_bank(c, mlps):
    """mlps: sequence of (w1, b1, w2, b2) parameter tuples -> tuple of f32 [B, C_g] outputs."""
    if not c.is_cuda:
        raise RuntimeError("xmc_gan_amd.ops.cond_mlp_bank: CPU tensors are not supported (no CPU fallback)")
    flat = [t for m in mlps for t in m]
    return CondMLPBankFn.apply(c.float(), *flat)


# ------------------------------------------------------------------------------------------ text front end
def embedding(ids, table):
    """nn.Embedding lookup (encoder.py:132), forward only (the encoder is frozen, train_gan.py:466-468).
    ids: int64 [...] on the device; table f32 [V, D] with D % 4 == 0.  Returns f32 [..., D]."""
    if not table.is_cuda or not ids.is_cuda:
        raise RuntimeError("xmc_gan_amd.ops.embedding: CPU tensors are not supported (no CPU fallback)")
    assert ids.dtype == torch.int64 and table.dtype == torch.float32 and table.dim() == 2
    ids = ids.contiguous()
    table = table.detach().contiguous()
    out = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=table.device)
    L.call("xmc_embedding_gather", _p(ids), _p(table), _p(out), ids.numel(), table.shape[1], table.shape[0], _st())
    return out


def lstm_bidir(xproj, w_hh, lens, T):
    """One-layer bidirectional LSTM recurrence over length-packed sequences (encoder.py:134-147), forward only.
    xproj f32 [B,T,2,4H] (input projections + both biases), w_hh f32 [2,4H,H], lens int32 [B].
    Returns words [B,2H,T] (zero at t >= len) and sent [B,2H] = [h_fwd(len-1), h_rev(0)]."""
    if not xproj.is_cuda:
        raise RuntimeError("xmc_gan_amd.ops.lstm_bidir: CPU tensors are not supported (no CPU fallback)")
    B, H = xproj.shape[0], w_hh.shape[2]
    assert xproj.dtype == torch.float32 and xproj.is_contiguous() and xproj.shape == (B, T, 2, 4 * H)
    assert w_hh.dtype == torch.float32 and w_hh.is_contiguous() and w_hh.shape == (2, 4 * H, H)
    assert lens.dtype == torch.int32 and lens.is_contiguous() and lens.numel() == B
    words = torch.empty(B, 2 * H, T, dtype=torch.float32, device=xproj.device)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=xproj.device)
    L.call("xmc_lstm_bidir", _p(xproj), _p(w_hh), _p(lens), _p(words), _p(sent), B, T, H, _st())
    return words, sent


def gru_bidir(xproj, w_hh, b_hn, lens, T):
    """One-layer bidirectional GRU recurrence over length-packed sequences (encoder.py:99-102,134-147 with RNN_TYPE 'GRU'),
    forward only.  xproj f32 [B,T,2,3H] (W_i* x + b_i*, plus b_h* for the r and z rows), w_hh f32 [2,3H,H], b_hn f32 [2,H],
    lens int32 [B].  Returns words [B,2H,T] (zero at t >= len) and sent [B,2H] = [h_fwd(len-1), h_rev(0)]."""
    if not xproj.is_cuda:
        raise RuntimeError("xmc_gan_amd.ops.gru_bidir: CPU tensors are not supported (no CPU fallback)")
    B, H = xproj.shape[0], w_hh.shape[2]
    assert xproj.dtype == torch.float32 and xproj.is_contiguous() and xproj.shape == (B, T, 2, 3 * H)
    assert w_hh.dtype == torch.float32 and w_hh.is_contiguous() and w_hh.shape == (2, 3 * H, H)
    assert b_hn.dtype == torch.float32 and b_hn.is_contiguous() and b_hn.shape == (2, H)
    assert lens.dtype == torch.int32 and lens.is_contiguous() and lens.numel() == B
    words = torch.empty(B, 2 * H, T, dtype=torch.float32, device=xproj.device)
    sent = torch.empty(B, 2 * H, dtype=torch.float32, device=xproj.device)
    L.call("xmc_gru_bidir", _p(xproj), _p(w_hh), _p(b_hn), _p(lens), _p(words), _p(sent), B, T, H, _st())
    return words, sent


class SpectralNormFn(torch.autograd.Function):
    """W / sigma(W) as the legacy ``torch.nn.utils.spectral_norm`` hook computes it (reference model/modules.py:3,16-17,
    31-32): in training mode ONE power iteration updates ``u`` [R] / ``v`` [C] in place (v <- normalize(W^T u),
    u <- normalize(W v)), then sigma = u . (W v) with u, v constants.  W is ``w.view(R, -1)``, f32.

    Differentiable once in W (dW = g/sigma - <g,W>/sigma^2 u v^T).  That is all the iteration ever asks for: W reaches
    the discriminator only through W/sigma, so even the MA-GP second-order pass crosses this node exactly once, with
    the gradient w.r.t. W/sigma that the (twice differentiable) convolution Functions produce."""

    @staticmethod
    def forward(ctx, w, u, v, training, eps):
        assert w.dtype == torch.float32 and u.dtype == torch.float32 and v.dtype == torch.float32
        w = w.contiguous()
        R = w.shape[0]
        C = w.numel() // R
        assert u.numel() == R and v.numel() == C and u.is_contiguous() and v.is_contiguous()
        scratch = torch.empty(C + R + 4, dtype=torch.float32, device=w.device)
        sig = torch.empty(2, dtype=torch.float32, device=w.device)
        y = torch.empty_like(w)
        L.call("xmc_spectral_sigma", _p(w), _p(u), _p(v), _p(scratch), _p(sig), _p(y), R, C, int(bool(training)), float(eps),
               _st())
        ctx.mark_non_differentiable(u, v)
        ctx.save_for_backward(w, u.clone(), v.clone(), sig)
        ctx.dims = (R, C)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w, u, v, sig = ctx.saved_tensors
        R, C = ctx.dims
        g = g.contiguous().float()
        dw = torch.empty_like(w)
        dot = torch.empty(1, dtype=torch.float32, device=w.device)
        L.call("xmc_spectral_bwd", _p(g), _p(w), _p(u), _p(v), _p(sig), _p(dot), _p(dw), R, C, _st())
        return dw, None, None, None, None


def spectral_weight(w, u, v, training, eps=1e-12):
    """effective weight of a spectrally normalised layer; updates the u / v buffers in place when ``training``."""
    if not w.is_cuda:
        raise RuntimeError("xmc_gan_amd.ops.spectral_weight: CPU tensors are not supported (no CPU fallback)")
    return SpectralNormFn.apply(w, u, v, training, eps)


# ------------------------------------------------------------------------------------------ pointwise
class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        x = x.contiguous()
        y = torch.empty_like(x, dtype=dtype)
        L.call("xmc_cast", _p(x), _p(y), x.numel(), _code(x.dtype), _code(dtype), _st())
        return y

    @staticmethod
    def backward(ctx, dy):
        return CastFn.apply(dy, ctx.src), None


class MaskFn(torch.autograd.Function):
    """ref > 0 ? dy : slope*dy  (derivative of LeakyReLU/ReLU applied to dy; linear in dy)."""

    @staticmethod
    def forward(ctx, dy, ref, slope):
        dy = dy.contiguous()
        if dy.dtype != ref.dtype:
            dy = dy.to(ref.dtype)
        out = torch.empty_like(dy)
        L.call("xmc_lrelu_mask", _p(dy), _p(ref), _p(out), dy.numel(), float(slope), _code(dy.dtype), _st())
        ctx.slope = slope
        ctx.save_for_backward(ref)
        return out

    @staticmethod
    def backward(ctx, g):
        (ref,) = ctx.saved_tensors
        return MaskFn.apply(g, ref, ctx.slope), None, None


class LreluFn(torch.autograd.Function):
    """nn.LeakyReLU(0.2) (df_gan.py:85,158,214-222,274,277); slope 0 gives nn.ReLU."""

    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.call("xmc_lrelu", _p(x), _p(y), x.numel(), float(slope), _code(x.dtype), _st())
        ctx.slope = slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return MaskFn.apply(dy, y, ctx.slope), None


class TanhBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, y):
        dy = dy.contiguous()
        if dy.dtype != y.dtype:
            dy = dy.to(y.dtype)
        out = torch.empty_like(dy)
        L.call("xmc_tanh_bwd", _p(dy), _p(y), _p(out), dy.numel(), _code(dy.dtype), _st())
        return out

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("second derivative through tanh is not on the XMC-GAN path")


class ScaleFn(torch.autograd.Function):
    """alpha * x with alpha a device scalar (f32 tensor with one element)."""

    @staticmethod
    def forward(ctx, x, alpha):
        x = x.contiguous()
        a = alpha.detach().reshape(-1).float()
        y = torch.empty_like(x)
        L.call("xmc_scale", _p(x), _p(a), _p(y), x.numel(), _code(x.dtype), _st())
        ctx.save_for_backward(x, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dx = ScaleFn.apply(dy, alpha) if ctx.needs_input_grad[0] else None
        da = DotFn.apply(dy, x).reshape(alpha.shape) if ctx.needs_input_grad[1] else None
        return dx, da


class DotFn(torch.autograd.Function):
    """sum(a*b) -> f32 [1]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        if a.dtype != b.dtype:
            b = b.to(a.dtype)
        out = _zeros_f32_out(1, a.device)
        L.call("xmc_dot", _p(a), _p(b), _p(out), a.numel(), _code(a.dtype), _st())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = ScaleFn.apply(b, g) if ctx.needs_input_grad[0] else None
        db = ScaleFn.apply(a, g) if ctx.needs_input_grad[1] else None
        return da, db


class AxpbyFn(torch.autograd.Function):
    """a + alpha*b  (shortcut + gamma*residual, df_gan.py:200,284)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = a.contiguous(), b.contiguous()
        al = alpha.detach().reshape(-1).float()
        y = torch.empty_like(a)
        L.call("xmc_axpby", _p(a), _p(b), _p(al), _p(y), a.numel(), _code(a.dtype), _st())
        ctx.save_for_backward(b, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, alpha = ctx.saved_tensors
        if not torch.is_grad_enabled() and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and fused_blocks():
            _, db, dal = _axpby_bwd_fused(dy, b, alpha, up=False)
            return (dy if ctx.needs_input_grad[0] else None), db, dal
        da = dy if ctx.needs_input_grad[0] else None
        db = ScaleFn.apply(dy, alpha) if ctx.needs_input_grad[1] else None
        dal = DotFn.apply(dy, b).reshape(alpha.shape) if ctx.needs_input_grad[2] else None
        return da, db, dal


class ResDFn(torch.autograd.Function):
    """One discriminator block, `shortcut(x) + gamma * residual(x)` (df_gan.py:269-291), as a single first-order node:
    forward = the same launches as the composed block; backward fuses what autograd would run as separate passes:
      * gamma*dout, the LeakyReLU mask of the residual output and d(gamma) = <dout, res> in one kernel (7 tensor passes -> 3),
      * the LeakyReLU mask of conv_r[0]'s output in the epilogue of conv_r[2]'s data gradient,
      * the shortcut's gradient (adjoint of the average pool: x0.25, nearest x2) as the row-indexed residual of conv_r[0]'s
        data gradient, so neither the upsampled tensor nor the sum of the two branches is written separately."""

    @staticmethod
    def forward(ctx, x, w0, w2, ws, bs, gamma, g0, g2, gs, xp_hint=None, want_pool=False):
        """``xp_hint``: avg_pool2d(x, 2) if the producer of x already wrote it (the previous block's third output);
        ``want_pool``: return (out, avg_pool2d(out, 2)) -- the pooled tensor is a by-product for the NEXT block's shortcut and
        carries no gradient of its own (that block returns the full gradient of its input, pool path included).
        The backward is ResDBwdFn, itself a differentiable node (MA-GP)."""
        x = x.contiguous()
        dt = x.dtype
        N, H, W, _ = x.shape
        xp32 = None
        if xp_hint is not None and xp_hint.dtype == torch.float32 and dt != torch.float32:
            xp32 = xp_hint                        # the previous block ran on the precise trunk (below): its pooled sum in f32
            xp = None
        elif xp_hint is not None:
            xp = xp_hint
        else:
            xp = torch.empty((N, H // 2, W // 2, x.shape[3]), dtype=dt, device=x.device)
            L.call("xmc_sumpool2", _p(x), _p(xp), N, H, W, x.shape[3], 0.25, _code(dt), _st())
        bp = None
        if ws is not None and bs is not None:
            bp = bs.detach().float()
            cd_p = pad_to(gs.cout, 8)
            if bp.numel() < cd_p:
                bp = torch.nn.functional.pad(bp, (0, cd_p - bp.numel()))
            bp = bp.contiguous()
        al = gamma.detach().reshape(-1).float()
        # PRECISE TRUNK (round 5, the IEEE-half mode; DESIGN 5.1, tests/diag/layer_ladder.py).  With the reference's small block gammas
        # the logits are a function of the SHORTCUT path image -> [pool -> conv_s -> block sum] x depth -> COND_DNET: the residual
        # branches enter times gamma.  The per-layer ladder puts 60 % of the logit vector's rounding error on the last two blocks'
        # shortcut / block-sum / pooled tensors and on the head, all on maps of <= 8x8 pixels -- 1 % of the discriminator's bytes.  On
        # those maps the shortcut (conv_s in exact-f32 MFMA on the f32 pooled input), the block sum (f32 destination and f32 residual
        # of the gather kernel, 16-bit MFMA operands) and the pooled by-product stay f32; the last block hands COND_DNET an f32 map.
        if precise_trunk() and dt != torch.float32 and H // 2 <= 8 and not _second_order():
            if xp32 is None:
                xp32 = CastFn.apply(xp, torch.float32)
            if xp is None and ws is not None:     # the backward's 16-bit operand of conv_s's weight gradient
                xp = CastFn.apply(xp32, dt)
            sc32 = _conv_fwd_raw(xp32, ws, bp, gs, L.ACT_NONE, torch.float32) if ws is not None else xp32
            h1 = _conv_fwd_raw(x, w0, None, g0, L.ACT_LRELU, dt)
            keep = any(ctx.needs_input_grad[:6])
            r = _conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, torch.float32, res=sc32, alpha=al, want_sign=keep)
            out32, bits = r if keep else (r, None)
            ctx.geoms, ctx.learned, ctx.has_bs = (g0, g2, gs), ws is not None, bs is not None
            ctx.save_for_backward(x, xp if ws is not None else None, h1, bits, w0, w2, ws, gamma)
            if not want_pool:
                return out32                      # the last block: COND_DNET reads the f32 map
            outp = torch.empty((N, out32.shape[1] // 2, out32.shape[2] // 2, out32.shape[3]), dtype=torch.float32, device=x.device)
            L.call("xmc_sumpool2", _p(out32), _p(outp), N, out32.shape[1], out32.shape[2], out32.shape[3], 0.25, L.F32, _st())
            out = CastFn.apply(out32, dt)         # conv_r[0] of the next block reads 16-bit operands (a residual-branch input)
            ctx.mark_non_differentiable(outp)
            ctx.set_materialize_grads(False)
            return out, outp
        if xp is None:
            xp = CastFn.apply(xp32, dt)
        if ws is not None:
            # precise trunk, larger maps: the learned shortcut's WEIGHTS at f32 grade (a rounding error shared by every sample and
            # pixel: 40 % of the generated-image logits' error in the ladder) -- the streaming 1x1 kernels on a hi + lo weight pair
            # (two MFMAs per K step of a launch that is bound by its HBM stream), the exact-f32 kernel where they decline (few pixels)
            sc = _conv1x1_pair_raw(xp, ws, bp, gs, dt) if (precise_trunk() and dt != torch.float32 and not _second_order()) else None
            if sc is None:
                sc = _conv_fwd_raw(xp, ws, bp, gs, L.ACT_NONE, dt)
        else:
            sc = xp
        h1 = _conv_fwd_raw(x, w0, None, g0, L.ACT_LRELU, dt)
        # conv_r[2], LeakyReLU, `shortcut + gamma * residual` (df_gan.py:276-277,284) and the next block's pool in ONE pass: the
        # residual branch itself is kept (second output) only when a backward pass will ask for it
        keep = any(ctx.needs_input_grad[:6])
        pool_ok = want_pool and res_pool_ok(h1, g2)
        # what the backward needs of the residual branch: its LeakyReLU' mask -- the SIGN bits, 1/16 of the tensor -- unless the
        # backward itself will be differentiated (MA-GP: ops.second_order()), whose linearised forward needs the values
        bits_mode = keep and not _second_order() and "no_sign_bits" not in _DEBUG_DISPATCH
        r = _conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, res=sc, alpha=al, want2=keep and not bits_mode, want_sign=bits_mode,
                          want_pool=pool_ok, round_act=True)
        r = r if isinstance(r, tuple) else (r,)
        out = r[0]
        res = r[1] if keep else None              # the branch (bf16 tensor) or its sign bits (uint8 [N,H,W,C/8])
        outp = r[-1] if pool_ok else None
        ctx.geoms = (g0, g2, gs)
        ctx.learned = ws is not None
        ctx.has_bs = bs is not None
        ctx.save_for_backward(x, xp if ws is not None else None, h1, res, w0, w2, ws, gamma)
        if want_pool:
            if outp is None:
                outp = torch.empty((N, out.shape[1] // 2, out.shape[2] // 2, out.shape[3]), dtype=dt, device=x.device)
                L.call("xmc_sumpool2", _p(out), _p(outp), N, out.shape[1], out.shape[2], out.shape[3], 0.25, _code(dt), _st())
            ctx.mark_non_differentiable(outp)
            ctx.set_materialize_grads(False)      # no zero-filled gradient tensor for the pooled by-product on every backward
            return out, outp
        return out

    @staticmethod
    def backward(ctx, dout, _doutp=None):
        if dout is None:
            return (None,) * 11
        x, xp, h1, res, w0, w2, ws, gamma = ctx.saved_tensors
        need = tuple(bool(v) for v in ctx.needs_input_grad[:6])
        if res is not None and res.dtype == torch.uint8 and torch.is_grad_enabled():      # create_graph=True
            raise RuntimeError("ResDFn: this block kept only the sign bits of its residual branch; wrap the forward in "
                               "ops.second_order() to differentiate its backward (the MA-GP pattern)")
        outs = ResDBwdFn.apply(dout, x, xp, h1, res, w0, w2, ws, gamma, ctx.geoms, ctx.learned, ctx.has_bs, need, _skip_wgrad())
        return tuple(outs) + (None, None, None, None, None)


class fixed_order:
    """Context manager: the reductions that feed activations (GroupNorm statistics, the attention query gradient) in a fixed summation
    order (xmc_set_fixed_order: one workgroup per reduction target).  A test mode -- it costs those launches their parallelism -- that
    makes an iteration of the attention-modulation generators repeatable, so that their gradient tests need not budget for run-to-run
    spread."""

    def __enter__(self):
        self.was = L.load().xmc_set_fixed_order(1)
        return self

    def __exit__(self, *a):
        L.load().xmc_set_fixed_order(self.was)
        return False


def debug_switch(token):
    """True when `token` is listed in XMC_DEBUG_DISPATCH (A/B experiments; unset in production)"""
    return token in _DEBUG_DISPATCH


def dstem_eligible(xin, c_img, c_sc, c_out):
    """the composed-stem path (DStemBlockFn) takes 16-bit images whose size tiles (H % 16 == 0, W % 64 == 0) at the widths the
    kernel is built for (conv_img: 3 -> 32, first block: 32 -> 64 with its learned shortcut)"""
    return (xin.is_cuda and xin.dtype != torch.float32 and xin.shape[3] == 8 and xin.shape[1] % 16 == 0 and xin.shape[2] % 64 == 0 and
            c_img == 32 and c_out == 64 and c_sc and "no_dstem" not in _DEBUG_DISPATCH)


class DStemBlockFn(torch.autograd.Function):
    """conv_img and the first discriminator block (df_gan.py:114,127,269-291) as one first-order node on the COMPOSED stem
    (csrc/dstem.hip, `compose_dstem`): the image goes straight to h1 = lrelu(conv_r[0](conv_img(x))) and to the shortcut
    conv_s(avg_pool2d(conv_img(x))); conv_img's 32-channel full-resolution output and its pooled copy are never written, and the
    backward needs neither them nor their gradients -- the weight gradients of conv_img, conv_r[0] and conv_s come from ONE
    weight-gradient launch on the image (gradients of the composed weights; a second, tiny one for the border corrections)
    through autograd on the composition.
    The rest of the block is ResDFn's: conv_r[2] + LeakyReLU + block sum (+ sign bits, + pooled output) in one launch, its data
    gradient with the LeakyReLU' mask of h1 and d(gamma) in the epilogue.  The gradient of the image, where asked for (the G step's
    pass over the generated batch), is the adjoint of the composed stem: one launch on the low-resolution gradients."""

    @staticmethod
    def forward(ctx, xin, w_img, b_img, w0, w2, ws, bs, gamma, g_img, g0, g2, gs, want_pool=False):
        xin = xin.contiguous()
        dt = xin.dtype
        N, H, W, _ = xin.shape
        OH, OW = H // 2, W // 2
        wsets, bias, D, DB = _dstem_compose_raw(w_img, b_img, w0, ws, bs)
        al = gamma.detach().reshape(-1).float()
        keep = any(ctx.needs_input_grad[:8])
        pool_ok = want_pool and H % 4 == 0 and W % 4 == 0
        # the shortcut (0.54 GB per 256 images, written here and read once by the block end) is recomputed from the image inside the
        # block-end kernel where that kernel takes the shape: 16 more MFMAs per wave and tile on an 18 x 66 pixel image patch
        # MA-GP (ops.second_order()): the backward of this node is differentiated again (DStemBwdFn), whose linearised forward needs the
        # residual branch's VALUES as its LeakyReLU' mask operand -- kept instead of the sign bits, with the shortcut as a tensor
        so2 = _second_order()
        fuse_sc = pool_ok and OH % 8 == 0 and OW % 32 == 0 and not so2 and "no_scimg" not in _DEBUG_DISPATCH      # (its two epilogue sets write the pooled output)
        h1, sc = _dstem_fwd_raw(xin, wsets, bias, want_sc=not fuse_sc)
        _dstem_border_fwd_raw(xin, wsets, bias, D, DB, h1)          # conv_r[0]'s zero padding of conv_img's output: 3 % of the pixels
        assert pool_ok == (want_pool and res_pool_ok(h1, g2))
        r = None
        if fuse_sc:
            r = _conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, alpha=al, want_sign=keep, want_pool=pool_ok, round_act=True,
                              sc_img=_dstem_sc_operands(xin, wsets, bias))
            if r is None:
                # the kernel's own conditions are tighter than the predicate above (its tile plan, the LDS limit, its two epilogue sets,
                # the A/B switches of tests/diag/ab.sh): write the shortcut after all and take the residual form
                _, sc = _dstem_fwd_raw(xin, wsets, bias, want_sc=True)
        if r is None:
            r = _conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, res=sc, alpha=al, want_sign=keep and not so2, want2=keep and so2,
                              want_pool=pool_ok, round_act=True)
        r = r if isinstance(r, tuple) else (r,)
        out = r[0]
        bits = r[1] if keep else None             # sign bytes, or (second order) the branch itself
        outp = r[-1] if pool_ok else None
        ctx.geoms = (g_img, g0, g2, gs)
        ctx.has_bs = bs is not None
        ctx.save_for_backward(xin, h1, bits, w_img, b_img, w0, w2, ws, bs, gamma, wsets, D)
        if want_pool:
            if outp is None:
                outp = torch.empty((N, out.shape[1] // 2, out.shape[2] // 2, out.shape[3]), dtype=dt, device=xin.device)
                L.call("xmc_sumpool2", _p(out), _p(outp), N, out.shape[1], out.shape[2], out.shape[3], 0.25, _code(dt), _st())
            ctx.mark_non_differentiable(outp)
            ctx.set_materialize_grads(False)
            return out, outp
        return out

    @staticmethod
    def backward(ctx, dout, _doutp=None):
        nin = 13
        if dout is None:
            return (None,) * nin
        xin, h1, bits, w_img, b_img, w0, w2, ws, bs, gamma, wsets, D = ctx.saved_tensors
        if bits is not None and bits.dtype != torch.uint8:
            # second-order form: the first-order backward as a node of its own
            need = tuple(bool(v) for v in ctx.needs_input_grad[:8])
            outs = DStemBwdFn.apply(dout, xin, h1, bits, w_img, b_img, w0, w2, ws, bs, gamma, ctx.geoms, need, _skip_wgrad())
            return tuple(outs) + (None,) * 5
        if torch.is_grad_enabled():
            raise RuntimeError("DStemBlockFn: this block kept only the sign bits of its residual branch; wrap the forward in "
                               "ops.second_order() to differentiate its backward (the MA-GP pattern)")
        with torch.no_grad():
            return DStemBlockFn._backward_bits(ctx, dout)

    @staticmethod
    def _backward_bits(ctx, dout):
        nin = 13
        xin, h1, bits, w_img, b_img, w0, w2, ws, bs, gamma, wsets, D = ctx.saved_tensors
        g_img, g0, g2, gs = ctx.geoms
        dt = xin.dtype
        N, H, W, _ = xin.shape
        OH, OW = H // 2, W // 2
        dout = dout.contiguous()
        if dout.dtype != dt:
            dout = dout.to(dt)
        skip_w = _skip_wgrad()
        al = gamma.detach().reshape(-1).float()
        dgam = _zeros_f32_out(1, xin.device)
        # residual branch, as ResDBwdFn on sign bits: gr = s * dout, d(gamma) from the data gradient's epilogue
        need_x = ctx.needs_input_grad[0]
        # (gr is never written: both of its consumers apply the sign bytes while they stage dout -- _StagedMask)
        gr = _StagedMask(dout, bits)
        dw2 = _conv_wgrad_raw(h1, gr, g2, scale=al).view(w2.shape) if (ctx.needs_input_grad[4] and not skip_w) else None
        gh = _conv_dgrad_raw(gr, w2, g2, (OH, OW), dt, mask=h1, alpha=al, dot=dgam)          # d h1 in front of its LeakyReLU
        dgamma = dgam.reshape(gamma.shape).to(gamma.dtype) if ctx.needs_input_grad[7] else None
        dx = None
        if need_x:
            # the gradient of the IMAGE (the G step's pass over the generated batch): the adjoint of the composed stem, one launch
            # on the low-resolution gradients (+ the border corrections); "dstem_old_dgrad": the un-composed transposed chain
            if "dstem_old_dgrad" in _DEBUG_DISPATCH:
                dxp = _conv_dgrad_raw(dout, ws, gs, (OH, OW), dt)
                dci = _conv_dgrad_raw(gh, w0, g0, (H, W), dt, res=dxp, res_rows=True, res_scale=0.25)
                dx = _conv_dgrad_raw(dci, w_img, g_img, (H, W), dt)
            else:
                dx = _dstem_dgrad_raw(gh, dout, wsets, D, H, W)
        if skip_w or not any(ctx.needs_input_grad[1:7]):
            return (dx, None, None, None, dw2, None, None, dgamma) + (None,) * 5
        # gradients of the composed weights (every pixel) and of the border corrections (border pixels of h1), then back through the
        # composition to the five parameters
        tabs = _dstem_wgrad_raw(xin, gh, dout)
        dwi, dbi, dw0, dws, dbs = _dstem_compose_bwd_raw(w_img, b_img, w0, ws, bs, *tabs)
        dwi, dw0, dws = dwi.view(w_img.shape), dw0.view(w0.shape), dws.view(ws.shape)
        return (dx, dwi.to(w_img.dtype), dbi.to(b_img.dtype), dw0.to(w0.dtype), dw2, dws.to(ws.dtype),
                None if dbs is None else dbs.to(bs.dtype), dgamma) + (None,) * 5


class DStemBwdFn(torch.autograd.Function):
    """First-order backward of DStemBlockFn as a node of its own (MA-GP: the penalty is a function of this node's dx; ResDBwdFn is the
    same idea for the later blocks).  With A = the composed residual-branch stem (6x6 stride 2 + border corrections), B = the composed
    shortcut, C2 = conv_r[2], m1 = LeakyReLU'(h1), m2 = LeakyReLU'(branch):
        forward:   gr = gamma m2 * dout,  gh = m1 * C2^T gr,  dx = A^T gh + B^T dout          (xmc_dstem_dgrad + border)
                   parameter gradients as in DStemBlockFn (tables from xmc_dstem_wgrad, through the composition's adjoint)
        backward for g = dL/d(dx) -- the linearised forward of the block applied to g, on the SAME stem kernels:
                   A g, B g   = xmc_dstem_fwd / _border_fwd on g with zero biases and slope 1
                   v = m1 * A g;  d(dout) = B g + gamma m2 * C2 v;  d(gamma) = <m2 * dout, C2 v>;  d(w2) = wgrad(v, gr)
                   d(tables) = xmc_dstem_wgrad(image := g, gh, dout) with the bias entries dropped (dx has no bias term), then the
                   composition's adjoint to conv_img / conv_r[0] / conv_s.
    Only d(dx) is differentiated again."""

    @staticmethod
    def forward(ctx, dout, xin, h1, res, w_img, b_img, w0, w2, ws, bs, gamma, geoms, need, skip_w):
        g_img, g0, g2, gs = geoms
        ctx.set_materialize_grads(False)
        ctx.dout_dtype = dout.dtype
        dt = xin.dtype
        N, H, W, _ = xin.shape
        OH, OW = H // 2, W // 2
        dout = dout.contiguous()
        if dout.dtype != dt:
            dout = dout.to(dt)
        al = gamma.detach().reshape(-1).float()
        dgam = _zeros_f32_out(1, xin.device)
        wsets, bias, D, DB = _dstem_compose_raw(w_img, b_img, w0, ws, bs)
        gr = torch.empty_like(res)
        L.call("xmc_scale_mask_dot", _p(dout), _p(res), _p(al), _p(gr), _p(dgam), res.numel(), _code(dt), _st())
        dw2 = _conv_wgrad_raw(h1, gr, g2).view(w2.shape) if (need[4] and not skip_w) else None
        gh = _conv_dgrad_raw(gr, w2, g2, (OH, OW), dt, mask=h1)                          # includes LeakyReLU'(h1)
        dx = _dstem_dgrad_raw(gh, dout, wsets, D, H, W) if need[0] else None
        dwi = dbi = dw0 = dws = dbs = None
        if not skip_w and any(need[1:4] + need[5:7]):
            tabs = _dstem_wgrad_raw(xin, gh, dout)
            dwi, dbi, dw0, dws, dbs = _dstem_compose_bwd_raw(w_img, b_img, w0, ws, bs, *tabs)
            dwi, dw0, dws = dwi.view(w_img.shape).to(w_img.dtype), dw0.view(w0.shape).to(w0.dtype), dws.view(ws.shape).to(ws.dtype)
            dbi = dbi.to(b_img.dtype)
            dbs = None if dbs is None else dbs.to(bs.dtype)
        dgamma = dgam.reshape(gamma.shape).to(gamma.dtype) if need[7] else None
        ctx.geoms = geoms
        ctx.save_for_backward(dout, h1, res, w_img, b_img, w0, w2, ws, bs, gamma, gr, gh, wsets, D)
        return dx, dwi, dbi, dw0, dw2, dws, dbs, dgamma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, *g_params):
        if any(t is not None for t in g_params):
            raise NotImplementedError("DStemBwdFn: only d(dx) is differentiated again (the MA-GP penalty)")
        nin = 14
        if g is None:
            return (None,) * nin
        dout, h1, res, w_img, b_img, w0, w2, ws, bs, gamma, gr, gh, wsets, D = ctx.saved_tensors
        g_img, g0, g2, gs = ctx.geoms
        dt = h1.dtype
        g = g.contiguous()
        if g.dtype != dt:
            g = g.to(dt)
        skip_w = _skip_wgrad()
        al = gamma.detach().reshape(-1).float()
        zb = torch.zeros(128, dtype=torch.float32, device=g.device)
        zdb = torch.zeros(64 * 8, dtype=torch.float32, device=g.device)
        ag, bg = _dstem_fwd_raw(g, wsets, zb, slope=1.0)                                 # A g (interior form), B g
        _dstem_border_fwd_raw(g, wsets, zb, D, zdb, ag, slope=1.0)                      # ... border pixels of A g
        v = torch.empty_like(ag)
        L.call("xmc_lrelu_mask", _p(ag), _p(h1), _p(v), ag.numel(), 0.2, _code(dt), _st())      # m1 * A g
        ddout, c2 = _conv_fwd_raw(v, w2, None, g2, L.ACT_NONE, dt, res=bg, alpha=al, mask=res, want2=True)      # c2 = C2 v
        dgamma = None
        if ctx.needs_input_grad[10]:
            u = torch.empty_like(c2)
            L.call("xmc_lrelu_mask", _p(c2), _p(res), _p(u), c2.numel(), 0.2, _code(dt), _st())
            dg = _zeros_f32_out(1, g.device)
            L.call("xmc_dot", _p(dout), _p(u), _p(dg), u.numel(), _code(dt), _st())
            dgamma = dg.reshape(gamma.shape).to(gamma.dtype)
        dwi = dbi = dw0 = dw2 = dws = dbs = None
        if not skip_w:
            if ctx.needs_input_grad[7]:
                dw2 = _conv_wgrad_raw(v, gr, g2).view(w2.shape)
            if any(ctx.needs_input_grad[4:7]) or any(ctx.needs_input_grad[8:10]):
                dW, dB, dD, dDB = _dstem_wgrad_raw(g, gh, dout)
                dB.zero_()
                dDB.zero_()                     # dx = A^T gh + B^T dout carries no bias term
                dwi, dbi, dw0, dws, dbs = _dstem_compose_bwd_raw(w_img, b_img, w0, ws, bs, dW, dB, dD, dDB)
                dwi, dw0, dws = dwi.view(w_img.shape).to(w_img.dtype), dw0.view(w0.shape).to(w0.dtype), dws.view(ws.shape).to(ws.dtype)
                dbi = dbi.to(b_img.dtype)
                dbs = None if dbs is None else dbs.to(bs.dtype)
        return (ddout.to(ctx.dout_dtype) if ctx.needs_input_grad[0] else None, None, None, None, dwi, dbi, dw0, dw2, dws, dbs, dgamma,
                None, None, None)


class ResDBwdFn(torch.autograd.Function):
    """First-order backward of ResDFn as a node of its own, so that it can be differentiated again (MA-GP, train_gan.py:231-252:
    the penalty is a function of d(logit)/d(image), i.e. of this node's dx).  forward = the fused backward of the block:
      * gamma*dout, the LeakyReLU mask of the residual output and d(gamma) = <dout, res> in one kernel (7 tensor passes -> 3),
      * the LeakyReLU mask of conv_r[0]'s output in the epilogue of conv_r[2]'s data gradient,
      * the shortcut's gradient (adjoint of the average pool: x0.25, nearest x2) as the row-indexed residual of conv_r[0]'s
        data gradient, so neither the upsampled tensor nor the sum of the two branches is written separately.
    With the masks m1 = LeakyReLU'(h1), m2 = LeakyReLU'(res) (piecewise constant: no gradient flows into the activations, as in
    autograd's own leaky_relu double backward) the node is LINEAR in dout:
        dx = Pool^T Ws^T dout + C0^T (m1 * C2^T (gamma m2 * dout))
    so its backward for an incoming g = dL/d(dx) is the linearised FORWARD of the block applied to g -- the same fused launches
    as the forward, masks in place of the activations -- plus three weight gradients:
        d(dout)  = Ws Pool g + gamma m2 * C2 (m1 * C0 g)           d(gamma) = <m2 * dout, C2 (m1 * C0 g)>
        d(w0) = wgrad(x = g, dy = gh)     d(w2) = wgrad(x = m1 * C0 g, dy = gr)     d(ws) = wgrad(x = Pool g, dy = dout)
    (gh, gr: the data gradients this node computed on the way).  The composed block (ops.composable()) computes the same
    quantities from ~25 fine-grained nodes; both forms are tested against each other."""

    @staticmethod
    def forward(ctx, dout, x, xp, h1, res, w0, w2, ws, gamma, geoms, learned, has_bs, need, skip_w):
        g0, g2, gs = geoms
        ctx.set_materialize_grads(False)       # gradients of outputs nothing depends on arrive as None, not as zeros
        ctx.dout_dtype = dout.dtype
        dout = dout.contiguous()
        dt = x.dtype
        if dout.dtype != dt:
            dout = dout.to(dt)
        # residual branch: g2 = gamma * dout * LeakyReLU'(res), d(gamma) = <dout, res>
        al = gamma.detach().reshape(-1).float()
        dgam = _zeros_f32_out(1, x.device)
        if res.dtype == torch.uint8:
            # `res` holds only the branch's sign bits.  With s = LeakyReLU'(branch) and branch = s * C2 h1:
            #   <dout, branch> = <s * dout, C2 h1> = <C2^T (s * dout), h1>
            # so the data gradient of conv_r[2] runs on the UNSCALED s * dout, accumulates the dot with h1 -- the tensor it reads as
            # its LeakyReLU' mask anyway -- before it applies gamma, and the weight gradient takes gamma as its scale.
            # s * dout comes out of the kernel that streams dout for the shortcut's data gradient where there is one (a learned 1x1
            # shortcut whose input gradient is needed), else from the mask pass
            gr = dxp_early = None
            if learned and need[0] and "no_pw1x1_masked_src" not in _DEBUG_DISPATCH:
                dxp_early, gr = _conv_dgrad_raw(dout, ws, gs, (xp.shape[1], xp.shape[2]), dt, src_bits=res)
            if gr is None:
                gr = torch.empty_like(dout)
                L.call("xmc_signmask_apply", _p(dout), _p(res), _p(gr), dout.numel(), 0.2, _code(dt), _st())
            dw2 = _conv_wgrad_raw(h1, gr, g2, scale=al).view(w2.shape) if (need[2] and not skip_w) else None
            gh = _conv_dgrad_raw(gr, w2, g2, (h1.shape[1], h1.shape[2]), dt, mask=h1, alpha=al, dot=dgam)
        else:
            dxp_early = None
            gr = torch.empty_like(res)
            L.call("xmc_scale_mask_dot", _p(dout), _p(res), _p(al), _p(gr), _p(dgam), res.numel(), _code(dt), _st())
            dw2 = _conv_wgrad_raw(h1, gr, g2).view(w2.shape) if (need[2] and not skip_w) else None
            gh = _conv_dgrad_raw(gr, w2, g2, (h1.shape[1], h1.shape[2]), dt, mask=h1)        # includes LeakyReLU'(h1)
        dw0 = _conv_wgrad_raw(x, gh, g0).view(w0.shape) if (need[1] and not skip_w) else None
        # shortcut branch
        dws = dbs = None
        if learned:
            if need[3] and not skip_w:
                if has_bs and need[4]:
                    dws, dbs = _conv_wgrad_raw(xp, dout, gs, want_bias=True)
                    dbs = dbs[: gs.cout]
                else:
                    dws = _conv_wgrad_raw(xp, dout, gs)
                dws = dws.view(ws.shape)
            dxp = dxp_early if dxp_early is not None else (_conv_dgrad_raw(dout, ws, gs, (xp.shape[1], xp.shape[2]), dt) if need[0] else None)
        else:
            dxp = dout
        dx = None
        if need[0]:
            dx = _conv_dgrad_raw(gh, w0, g0, (x.shape[1], x.shape[2]), dt, res=dxp, res_rows=True, res_scale=0.25)
        dgamma = dgam.reshape(gamma.shape).to(gamma.dtype) if need[5] else None
        ctx.geoms, ctx.learned = geoms, learned
        ctx.save_for_backward(dout, h1, res, w0, w2, ws, gamma, gr, gh)
        return dx, dw0, dw2, dws, dbs, dgamma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, g_dw0=None, g_dw2=None, g_dws=None, g_dbs=None, g_dgamma=None):
        if any(t is not None for t in (g_dw0, g_dw2, g_dws, g_dbs, g_dgamma)):
            raise NotImplementedError("ResDBwdFn: only d(dx) is differentiated again (the MA-GP penalty); use ops.composable() "
                                      "for second derivatives through the weight gradients")
        nin = 14
        if g is None:
            return (None,) * nin
        dout, h1, res, w0, w2, ws, gamma, gr, gh = ctx.saved_tensors
        g0, g2, gs = ctx.geoms
        dt = h1.dtype
        g = g.contiguous()
        if g.dtype != dt:
            g = g.to(dt)
        N, H, W, Cx = g.shape
        skip_w = _skip_wgrad()
        al = gamma.detach().reshape(-1).float()
        gp = torch.empty((N, H // 2, W // 2, Cx), dtype=dt, device=g.device)           # Pool g
        L.call("xmc_sumpool2", _p(g), _p(gp), N, H, W, Cx, 0.25, _code(dt), _st())
        sc = _conv_fwd_raw(gp, ws, None, gs, L.ACT_NONE, dt) if ctx.learned else gp     # Ws Pool g (the bias does not enter dx)
        v = _conv_fwd_raw(g, w0, None, g0, L.ACT_NONE, dt, mask=h1)                     # m1 * C0 g
        ddout, c2 = _conv_fwd_raw(v, w2, None, g2, L.ACT_NONE, dt, res=sc, alpha=al, mask=res, want2=True)   # c2 = C2 v
        dgamma = None
        if ctx.needs_input_grad[8]:
            u = torch.empty_like(c2)
            L.call("xmc_lrelu_mask", _p(c2), _p(res), _p(u), c2.numel(), 0.2, _code(dt), _st())
            dg = _zeros_f32_out(1, g.device)
            L.call("xmc_dot", _p(dout), _p(u), _p(dg), u.numel(), _code(dt), _st())
            dgamma = dg.reshape(gamma.shape).to(gamma.dtype)
        dw0 = dw2 = dws = None
        if not skip_w:
            if ctx.needs_input_grad[5]:
                dw0 = _conv_wgrad_raw(g, gh, g0).view(w0.shape)
            if ctx.needs_input_grad[6]:
                dw2 = _conv_wgrad_raw(v, gr, g2).view(w2.shape)
            if ctx.learned and ctx.needs_input_grad[7]:
                dws = _conv_wgrad_raw(gp, dout, gs).view(ws.shape)
        return (ddout.to(ctx.dout_dtype) if ctx.needs_input_grad[0] else None, None, None, None, None, dw0, dw2, dws, dgamma,
                None, None, None, None, None)


def res_pool_ok(h1, g2):
    """the pooled third output needs an even-sized map (the C side falls back to its own pool pass where the kernel cannot)"""
    OH, OW = g2.out_hw(h1.shape[1], h1.shape[2])
    return OH % 2 == 0 and OW % 2 == 0


class ColSumFn(torch.autograd.Function):
    """sum over all pixels -> f32 [C]  (bias gradients)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        Cc = x.shape[-1]
        out = _zeros_f32_out(Cc, x.device)
        L.call("xmc_colsum", _p(x), _p(out), x.numel() // Cc, Cc, _code(x.dtype), _st())
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).expand(ctx.shape).contiguous()


class SumPool2Fn(torch.autograd.Function):
    """scale * (2x2 sum pool).  scale=0.25: F.avg_pool2d(x, 2) (df_gan.py:290); adjoint of Up2Fn."""

    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.call("xmc_sumpool2", _p(x), _p(y), N, H, W, Cc, float(scale), _code(x.dtype), _st())
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        return Up2Fn.apply(dy, ctx.scale), None


class Up2Fn(torch.autograd.Function):
    """scale * nearest x2 upsample.  scale=1: F.interpolate(scale_factor=2) (df_gan.py:202)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        N, H, W, Cc = x.shape
        y = torch.empty((N, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
        L.call("xmc_upsample2", _p(x), _p(y), N, H, W, Cc, float(scale), _code(x.dtype), _st())
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        return SumPool2Fn.apply(dy, ctx.scale), None


class GapFn(torch.autograd.Function):
    """mean over all pixels of an [N,H,W,C] map -> [N,C] (F.avg_pool2d(x,4) on 4x4: df_gan.py:165, train_gan.py:272,275)."""

    @staticmethod
    def forward(ctx, x, out_dtype):
        x = x.contiguous()
        N, H, W, Cc = x.shape
        # (an f32 result is accumulated with atomics on big maps: handed over zero-filled, lib.load() has told the library so)
        y = _zeros_f32_out((N, Cc), x.device) if out_dtype == torch.float32 else torch.empty((N, Cc), dtype=out_dtype, device=x.device)
        L.call("xmc_global_avgpool", _p(x), _p(y), N, H * W, Cc, _code(x.dtype), _code(out_dtype), _st())
        ctx.hw, ctx.dtype = (H, W), x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        return GapBwdFn.apply(dy, ctx.hw, ctx.dtype), None


class GapBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, hw, dtype):
        dy = dy.contiguous()
        N, Cc = dy.shape
        dx = torch.empty((N, hw[0], hw[1], Cc), dtype=dtype, device=dy.device)
        L.call("xmc_global_avgpool_bwd", _p(dy), _p(dx), N, hw[0] * hw[1], Cc, _code(dtype), _code(dy.dtype), _st())
        ctx.in_dtype = dy.dtype
        return dx

    @staticmethod
    def backward(ctx, g):
        return GapFn.apply(g, ctx.in_dtype), None, None


class NchwToNhwc8Fn(torch.autograd.Function):
    """[N,C<=8,H,W] f32 (module boundary, df_gan.py:127) -> [N,H,W,8] activation dtype, zero padded."""

    @staticmethod
    def forward(ctx, x, dtype, out=None):
        _need_cuda(x)
        x = x.contiguous().float()
        N, Cc, H, W = x.shape
        if out is None:
            y = torch.empty((N, H, W, 8), dtype=dtype, device=x.device)
        else:                      # caller-provided destination (e.g. one half of the discriminator's 2B input); written
            # behind autograd's back (no version bump), so it must be a tensor no earlier node has saved
            assert tuple(out.shape) == (N, H, W, 8) and out.dtype == dtype and out.is_contiguous() and out._version == 0
            y = out
        L.call("xmc_nchw_to_nhwc8", _p(x), _p(y), N, Cc, H, W, _code(dtype), _st())
        ctx.c = Cc
        return y

    @staticmethod
    def backward(ctx, dy):
        return Nhwc8ToNchwFn.apply(dy, ctx.c), None, None


class Nhwc8ToNchwFn(torch.autograd.Function):
    """[N,H,W,8] -> [N,C,H,W] f32 (the image NetG returns, df_gan.py:101-103)."""

    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        N, H, W, _ = x.shape
        y = torch.empty((N, c, H, W), dtype=torch.float32, device=x.device)
        L.call("xmc_nhwc8_to_nchw", _p(x), _p(y), N, c, H, W, _code(x.dtype), _st())
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        return NchwToNhwc8Fn.apply(dy, ctx.dtype), None


def _affine_fwd_raw(x, ps, slope):
    """ps: (g0, b0) or (g0, b0, g1, b1), contiguous f32 [N, C]"""
    N, H, W, Cc = x.shape
    for t in ps:
        assert t.shape == (N, Cc), (t.shape, (N, Cc))
    y = torch.empty_like(x)
    ptrs = [_p(t) for t in ps] + ([] if len(ps) == 4 else [None, None])
    L.call("xmc_affine2_act_fwd", _p(x), *ptrs, _p(y), N, H * W, Cc, float(slope), _code(x.dtype), _st())
    return y


def _affine_bwd_raw(x, dy, ps, slope, dx_acc=None, alpha=None, dot=None, want_sumpool=False):
    """-> dx, red [len(ps), N, C] (the gradients of ps).  ``dx_acc``: another gradient of x, added on the way out.
    ``alpha`` / ``dot`` (f32 [1] each): dy is the UNSCALED gradient from a consumer `sum + alpha * f(y)`: dot += <dy, y>, dy *= alpha
    (xmc_affine2_act_bwd_dot).  ``want_sumpool``: -> dx, red, 2x2 sum pool of dx (same pass)."""
    N, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    if dx_acc is not None:
        dx_acc = dx_acc.contiguous()
        assert dx_acc.shape == x.shape and dx_acc.dtype == x.dtype
    nred = len(ps)
    red = _zeros_f32((nred, N, Cc), x.device)
    ptrs = [_p(t) for t in ps] + ([] if nred == 4 else [None, None])
    rptrs = [_p(red[i]) for i in range(nred)] + ([] if nred == 4 else [None, None])
    assert (alpha is None) == (dot is None)
    dxp = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device) if want_sumpool else None
    L.call("xmc_affine2_act_bwd_dot_pool", _p(x), _p(dy), *ptrs, _p(dx), *rptrs, _p(dx_acc), _p(alpha), _p(dot), _p(dxp), N, H, W, Cc,
           float(slope), _code(x.dtype), _st())
    return (dx, red, dxp) if want_sumpool else (dx, red)


class Affine2LreluFn(torch.autograd.Function):
    """lrelu(lrelu(x*g0+b0)*g1+b1) with per-sample, per-channel f32 g/b [N,C] -- two DF-GAN `affine` modules each
    followed by LeakyReLU(0.2) (df_gan.py:213-216 / 219-222, affine.forward 250-263).  With g1 = b1 = None it is the
    single modulation lrelu(x*g0+b0) of the concept blocks (df_concept_gan.py:238-239)."""

    @staticmethod
    def forward(ctx, x, g0, b0, g1, b1, slope=0.2):
        x = x.contiguous()
        two = g1 is not None
        ps = [t.contiguous().float() for t in ((g0, b0, g1, b1) if two else (g0, b0))]
        y = _affine_fwd_raw(x, ps, slope)
        ctx.two, ctx.slope = two, float(slope)
        ctx.save_for_backward(x, *ps)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, *ps = ctx.saved_tensors
        dx, red = _affine_bwd_raw(x, dy.contiguous(), ps, ctx.slope)
        if ctx.two:
            return dx, red[0], red[1], red[2], red[3], None
        return dx, red[0], red[1], None, None, None


def _gn_fwd_raw(x, wf, bf, groups, slope, eps):
    N, H, W, Cc = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((N, groups, 2), dtype=torch.float32, device=x.device)
    ws = _zeros_f32((N, Cc, 2), x.device)          # accumulators arrive zero (xmc_set_prezeroed): no memset launch per call
    L.call("xmc_groupnorm_fwd", _p(x), _p(wf), _p(bf), _p(y), _p(stats), _p(ws), N, H * W, Cc, groups, float(eps),
           float(slope), _code(x.dtype), _st())
    return y, stats


def _gn_bwd_raw(x, dy, wf, bf, stats, groups, slope):
    N, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(wf), torch.empty_like(bf)
    ws = _zeros_f32(N * Cc * 2 + N * groups * 2, x.device)
    L.call("xmc_groupnorm_bwd", _p(x), _p(dy), _p(wf), _p(bf), _p(stats), _p(dx), _p(dw), _p(db), _p(ws), N, H * W, Cc,
           groups, float(slope), _code(x.dtype), _st())
    return dx, dw, db


class Affine2LreluSkipFn(torch.autograd.Function):
    """Affine2LreluFn for an input that also feeds the block's shortcut (df_gan.py:199-200): returns (h, x) -- the second output
    IS x, for the shortcut branch to consume -- so that both gradients of x arrive at this node and are summed inside the
    affine backward kernel instead of in a framework add pass over the block input."""

    @staticmethod
    def forward(ctx, x, g0, b0, g1, b1, pool_grad=False):
        """``pool_grad``: x came out of a GBlockEndFn -- the backward pools its dx for that node (ops._pooled_grads)"""
        x = x.contiguous()
        ps = [t.contiguous().float() for t in (g0, b0, g1, b1)]
        y = _affine_fwd_raw(x, ps, 0.2)
        ctx.set_materialize_grads(False)
        ctx.pool = bool(pool_grad)
        ctx.save_for_backward(x, *ps)
        return y, x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, dskip):
        x, *ps = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None, None, None
        if ctx.pool and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[3] // 8 in (1, 2, 4, 8, 16, 32):
            # x is the output of the previous generator block, whose backward needs the 2x2 sum pool of this gradient
            dx, red, dxp = _affine_bwd_raw(x, dy.contiguous(), ps, 0.2, dx_acc=dskip, want_sumpool=True)
            _pooled_put(dx, dxp)
        else:
            dx, red = _affine_bwd_raw(x, dy.contiguous(), ps, 0.2, dx_acc=dskip)
        return dx, red[0], red[1], red[2], red[3], None


def affine2_lrelu_skip(x, g0, b0, g1, b1, pool_grad=False):
    return Affine2LreluSkipFn.apply(x, g0, b0, g1, b1, pool_grad)


class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm over NHWC (df_concept_gan.py:171,270-271,549-550) with optional fused LeakyReLU (slope >= 0)."""

    @staticmethod
    def forward(ctx, x, w, b, groups, slope, eps):
        x = x.contiguous()
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        y, stats = _gn_fwd_raw(x, wf, bf, groups, slope, eps)
        ctx.groups, ctx.slope = groups, slope
        ctx.save_for_backward(x, wf, bf, stats)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, wf, bf, stats = ctx.saved_tensors
        dx, dw, db = _gn_bwd_raw(x, dy.contiguous(), wf, bf, stats, ctx.groups, ctx.slope)
        return dx, dw, db, None, None, None


class BatchNormTrainFn(torch.autograd.Function):
    """nn.BatchNorm2d in training mode over NHWC (concept_gan.py:467-468,499-500,507-508): per-channel statistics over
    the whole batch = the GroupNorm kernels with one channel per group on the batch viewed as ONE sample of N*H*W pixels.
    Returns (y, stats) with stats f32 [C,2] = (batch mean, rstd) for the caller's running-statistics update."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        N, H, W, Cc = x.shape
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        y = torch.empty_like(x)
        stats = torch.empty((1, Cc, 2), dtype=torch.float32, device=x.device)
        ws = _zeros_f32((1, Cc, 2), x.device)
        L.call("xmc_groupnorm_fwd", _p(x), _p(wf), _p(bf), _p(y), _p(stats), _p(ws), 1, N * H * W, Cc, Cc, float(eps), -1.0,
               _code(x.dtype), _st())
        ctx.save_for_backward(x, wf, bf, stats)
        out_stats = stats.view(Cc, 2)
        ctx.mark_non_differentiable(out_stats)
        return y, out_stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, _dstats):
        x, wf, bf, stats = ctx.saved_tensors
        dy = dy.contiguous()
        N, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        dw, db = torch.empty_like(wf), torch.empty_like(bf)
        ws = _zeros_f32(Cc * 2 + Cc * 2, x.device)
        L.call("xmc_groupnorm_bwd", _p(x), _p(dy), _p(wf), _p(bf), _p(stats), _p(dx), _p(dw), _p(db), _p(ws), 1, N * H * W, Cc,
               Cc, -1.0, _code(x.dtype), _st())
        return dx, dw, db, None


def batchnorm_train(x, w, b, eps=1e-5):
    return BatchNormTrainFn.apply(x, w, b, eps)


def _attn_fwd_raw(key, q, x, ncon, scale):
    N, H, W, CK = key.shape
    pk, px = CK // ncon, x.shape[3] // ncon
    stats = torch.empty((N, ncon, 2), dtype=torch.float32, device=x.device)     # (max, sum of exp): the weights are recomputed
    out = torch.empty((N, ncon, px), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(L.load().xmc_attn_pool_ws_floats(N, H * W)), dtype=torch.float32, device=x.device)
    L.call("xmc_attn_pool_fwd", _p(key), _p(q), _p(x), _p(stats), _p(out), _p(ws), N, H * W, ncon, pk, px, float(scale),
           _code(x.dtype), _st())
    return out, stats


def _attn_bwd_raw(key, q, x, stats, out, dctx, ncon, scale, dx_acc=None):
    """``dx_acc``: another gradient of x; the kernel adds it on the way out and the sum is written IN PLACE into it."""
    N, H, W, CK = key.shape
    pk, px = CK // ncon, x.shape[3] // ncon
    dq = _zeros_f32_out(tuple(q.shape), q.device)
    dkey = torch.empty_like(key)
    dx = torch.empty_like(x) if dx_acc is None else dx_acc
    assert dx.shape == x.shape and dx.dtype == x.dtype and dx.is_contiguous()
    L.call("xmc_attn_pool_bwd_acc", _p(key), _p(q), _p(x), _p(stats), _p(out), _p(dctx), _p(dq), _p(dkey), _p(dx), _p(dx_acc),
           N, H * W, ncon, pk, px, float(scale), _code(x.dtype), _st())
    return dkey, dq, dx


class WordRegionPoolFn(torch.autograd.Function):
    """Word-region attention of the repaired concept_gan.InNetG (concept_gan.py:532-555): qmap NHWC [B,H,W,64] (the query projection of
    the map), kh f32 [B,16,T,4] (per-concept word keys, L2-normalised over the last axis), pad bool [B,T] (True = padding) ->
    ctx f32 [B,16,4], the mean over the regions of each region's attention-weighted key sum.  One pass forward, one backward
    (csrc/word_attention.hip); the [B,16,HW,T] attention tensor of the reference is never formed."""

    @staticmethod
    def forward(ctx, qmap, kh, pad):
        qmap, kh = qmap.contiguous(), kh.contiguous().float()
        _need_cuda(qmap, kh)
        B, H, W, Cq = qmap.shape
        T = kh.shape[2]
        assert Cq == 64 and kh.shape == (B, 16, T, 4) and pad.shape == (B, T), (qmap.shape, kh.shape, pad.shape)
        padu = pad.to(torch.uint8).contiguous()
        out = _zeros_f32_out((B, 16, 4), qmap.device)
        L.call("xmc_word_pool_fwd", _p(qmap), _p(kh), _p(padu), _p(out), B, H * W, 16, 4, T, _code(qmap.dtype), _st())
        ctx.save_for_backward(qmap, kh, padu)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dctx):
        qmap, kh, padu = ctx.saved_tensors
        B, H, W, _ = qmap.shape
        T = kh.shape[2]
        dq = torch.empty_like(qmap)
        dkh = _zeros_f32_out((B, 16, T, 4), qmap.device)
        # (the query map's gradient is an activation gradient: in the half mode it carries the backward's loss scale like every
        # other one, and dctx arrives scaled already)
        L.call("xmc_word_pool_bwd", _p(qmap), _p(kh), _p(padu), _p(dctx.contiguous().float()), _p(dq), _p(dkh), B, H * W, 16, 4, T,
               _code(qmap.dtype), _st())
        return dq, dkh, None


def word_region_pool(qmap, kh, pad):
    return WordRegionPoolFn.apply(qmap, kh, pad)


class AttnPoolFn(torch.autograd.Function):
    """Region attention of the concept samplers (df_concept_gan.py:293-299, 570-578): per (sample, concept) softmax over
    H*W of scale*<q, key>, then the attention-weighted sum of x.  key [N,H,W,ncon*pk], x [N,H,W,ncon*px], q f32 [N,ncon,pk]
    -> f32 [N,ncon,px]."""

    @staticmethod
    def forward(ctx, key, q, x, ncon, scale):
        key, x = key.contiguous(), x.contiguous()
        q = q.contiguous().float()
        out, stats = _attn_fwd_raw(key, q, x, ncon, scale)
        ctx.ncon, ctx.scale = ncon, scale
        ctx.save_for_backward(key, q, x, stats, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dctx):
        key, q, x, stats, out = ctx.saved_tensors
        dkey, dq, dx = _attn_bwd_raw(key, q, x, stats, out, dctx.contiguous().float(), ctx.ncon, ctx.scale)
        return dkey, dq, dx, None, None


# ------------------------------------------------------------------------------------------ losses
class HingeFn(torch.autograd.Function):
    """mean(relu(1 + sign*logit)) over the first channel of a padded [B,...,8] logit tensor
    (train_gan.py:195 sign=-1, 204/209 sign=+1)."""

    @staticmethod
    def forward(ctx, logits, sign):
        logits = logits.contiguous()
        n = logits.numel() // logits.shape[-1]
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        L.call("xmc_hinge_fwd", _p(logits), logits.shape[-1], float(sign), _p(out), n, _code(logits.dtype), _st())
        ctx.sign = sign
        ctx.save_for_backward(logits)
        return out.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        n = logits.numel() // logits.shape[-1]
        dx = torch.zeros_like(logits)
        gg = g.reshape(1).float().contiguous()
        L.call("xmc_hinge_bwd", _p(logits), logits.shape[-1], float(ctx.sign), _p(gg), _p(dx), n, _code(logits.dtype), _st())
        return dx, None


class ContrastiveFn(torch.autograd.Function):
    """Symmetric InfoNCE on cosine similarities, no temperature (cosine_scores + sent_loss/img_loss,
    train_gan.py:85-139).  a,b: [n,D] f32;  labels: None (identity) or f32 [n,n];  inv_num_pos: None or f32 [n]."""

    @staticmethod
    def forward(ctx, a, b, labels, inv_num_pos):
        a, b = a.contiguous().float(), b.contiguous().float()
        n, D = a.shape
        ws = torch.empty(L.load().xmc_contrastive_ws_bytes(n, D), dtype=torch.uint8, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        L.call("xmc_contrastive_fwd", _p(a), _p(b), _p(labels), _p(inv_num_pos), n, D, _p(loss), _p(ws), _st())
        ctx.save_for_backward(a, b, labels, inv_num_pos, ws)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b, labels, inv_num_pos, ws = ctx.saved_tensors
        n, D = a.shape
        da, db = torch.empty_like(a), torch.empty_like(b)
        gg = g.reshape(1).float().contiguous()
        L.call("xmc_contrastive_bwd", _p(a), _p(b), _p(labels), _p(inv_num_pos), n, D, _p(gg), _p(ws), _p(da), _p(db), _st())
        return da, db, None, None


class ConceptQueryFn(torch.autograd.Function):
    """Sentence query of CondConceptSampler (df_concept_gan.py:273-286): grouped 1x1 on the sentence vector that every concept
    receives + GroupNorm over each concept's 4 state values.  sent f32 [B,E], wq [64,E,1,1] -> q f32 [B,16,4]."""

    @staticmethod
    def forward(ctx, sent, wq, gnw, gnb, eps):
        sent = sent.contiguous().float()
        _need_cuda(sent, wq)
        B, E = sent.shape
        w = wq.detach().contiguous().float().view(64, E)
        q = torch.empty(B, 64, dtype=torch.float32, device=sent.device)
        qraw = torch.empty_like(q)
        L.call("xmc_concept_query_fwd", _p(sent), _p(w), _p(gnw), _p(gnb), _p(q), _p(qraw), B, E, float(eps), _st())
        ctx.eps, ctx.wshape = float(eps), tuple(wq.shape)
        ctx.save_for_backward(sent, w, gnw, qraw)
        return q.view(B, 16, 4)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dq):
        sent, w, gnw, qraw = ctx.saved_tensors
        B, E = sent.shape
        dq = dq.contiguous().float()
        dsent = torch.empty_like(sent)
        flat = _zeros_f32_out(64 * E + 128, sent.device)      # one slice for all accumulators
        dw = flat[:64 * E].view(64, E)
        dgw = flat[64 * E:64 * E + 64] if gnw is not None else None
        dgb = flat[64 * E + 64:] if gnw is not None else None
        scratch = torch.empty(B, 64, dtype=torch.float32, device=sent.device)
        L.call("xmc_concept_query_bwd", _p(sent), _p(w), _p(gnw), _p(qraw), _p(dq), _p(dsent), _p(dw), _p(dgw), _p(dgb), _p(scratch),
               B, E, ctx.eps, _st())
        return dsent, dw.view(ctx.wshape), dgw, dgb, None


class ConceptQueryAllFn(torch.autograd.Function):
    """ConceptQueryFn for EVERY sampler stage of a generator at once: the sentence queries depend on nothing but the sentence vector, so the
    24-28 per-stage launches (and, backward, as many pairs of launches) are one (two).  apply(sent, eps, wq_0, gnw_0, gnb_0, wq_1, ...) ->
    (q_0, q_1, ...) each f32 [B,16,4] (gnw_s / gnb_s None: no GroupNorm)."""

    @staticmethod
    def _tabs(ws, gws, gbs=None):
        mk = lambda ts: (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        return (mk(ws), mk(gws)) + ((mk(gbs),) if gbs is not None else ())

    @staticmethod
    def forward(ctx, sent, eps, *params):
        sent = sent.contiguous().float()
        S = len(params) // 3
        assert 1 <= S <= 32 and len(params) == 3 * S
        B, E = sent.shape
        ws = [params[3 * s].detach().contiguous().float().view(64, E) for s in range(S)]
        gws = [None if params[3 * s + 1] is None else params[3 * s + 1].detach().contiguous().float() for s in range(S)]
        gbs = [None if params[3 * s + 2] is None else params[3 * s + 2].detach().contiguous().float() for s in range(S)]
        _need_cuda(sent, *ws)
        q = torch.empty(S, B, 64, dtype=torch.float32, device=sent.device)
        qraw = torch.empty_like(q)
        tw, tg, tb = ConceptQueryAllFn._tabs(ws, gws, gbs)
        L.call("xmc_concept_query_fwd_multi", _p(sent), tw, tg, tb, S, _p(q), _p(qraw), B, E, float(eps), _st())
        ctx.eps, ctx.S, ctx.wshapes, ctx.has_gn = float(eps), S, [tuple(params[3 * s].shape) for s in range(S)], [g is not None for g in gws]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(sent, qraw, *ws, *[g for g in gws if g is not None])
        return tuple(q[s].view(B, 16, 4) for s in range(S))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *dqs):
        sent, qraw, *rest = ctx.saved_tensors
        S = ctx.S
        ws, gl = rest[:S], list(rest[S:])
        gws = [gl.pop(0) if h else None for h in ctx.has_gn]
        B, E = sent.shape
        dq = torch.stack([torch.zeros(B, 64, dtype=torch.float32, device=sent.device) if d is None else d.reshape(B, 64).float() for d in dqs])
        dsent = _zeros_f32_out(tuple(sent.shape), sent.device)          # accumulated with atomics (arrives zero: xmc_set_prezeroed)
        flat = _zeros_f32_out(S * 64 * E + S * 128, sent.device)
        dw, dgn = flat[:S * 64 * E].view(S, 64, E), flat[S * 64 * E:].view(S, 2, 64)
        scratch = torch.empty(B, S * 64, dtype=torch.float32, device=sent.device)
        tw, tg = ConceptQueryAllFn._tabs(ws, gws)
        L.call("xmc_concept_query_bwd_multi", _p(sent), tw, tg, S, _p(qraw), _p(dq.contiguous()), _p(dsent), _p(dw), _p(dgn), _p(scratch),
               B, E, ctx.eps, _st())
        out = [dsent, None]
        for s in range(S):
            out += [dw[s].view(ctx.wshapes[s]), dgn[s, 0] if ctx.has_gn[s] else None, dgn[s, 1] if ctx.has_gn[s] else None]
        return tuple(out)


def concept_query_all(sent, stages, eps=1e-5):
    """stages: [(wq, gnw | None, gnb | None), ...] in any order -> the list of their queries"""
    flat = [t for st in stages for t in st]
    return list(ConceptQueryAllFn.apply(sent, eps, *flat))


class ConceptGQueryFn(torch.autograd.Function):
    """Query of the self-attention sampler (df_concept_gan.py:555-569): grouped 1x1 (8 -> 4 per concept) on the globally
    averaged block input + GroupNorm over each concept's 4 values.  q0 f32 [B,128], wq [64,8,1,1] -> q f32 [B,16,4]."""

    @staticmethod
    def forward(ctx, q0, wq, gnw, gnb, eps):
        q0 = q0.contiguous().float()
        _need_cuda(q0, wq)
        B = q0.shape[0]
        assert q0.shape[1] == 128 and wq.numel() == 64 * 8
        w = wq.detach().contiguous().float().view(64, 8)
        q = torch.empty(B, 64, dtype=torch.float32, device=q0.device)
        qraw = torch.empty_like(q)
        L.call("xmc_concept_gquery_fwd", _p(q0), _p(w), _p(gnw), _p(gnb), _p(q), _p(qraw), B, float(eps), _st())
        ctx.eps, ctx.wshape = float(eps), tuple(wq.shape)
        ctx.save_for_backward(q0, w, gnw, qraw)
        return q.view(B, 16, 4)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dq):
        q0, w, gnw, qraw = ctx.saved_tensors
        B = q0.shape[0]
        dq = dq.contiguous().float()
        dq0 = torch.empty_like(q0)
        flat = _zeros_f32_out(64 * 8 + 128, q0.device)
        dw = flat[:512].view(64, 8)
        dgw = flat[512:576] if gnw is not None else None
        dgb = flat[576:] if gnw is not None else None
        L.call("xmc_concept_gquery_bwd", _p(q0), _p(w), _p(gnw), _p(qraw), _p(dq), _p(dq0), _p(dw), _p(dgw), _p(dgb), B, ctx.eps, _st())
        return dq0, dw.view(ctx.wshape), dgw, dgb, None


def _head_fwd_raw(pooled, sent, ps, a_pre=None):
    """``a_pre`` f32 [2,B,128]: the sentence part of the two MLPs' first layer, computed ahead (head_sentence_products)"""
    B, E = sent.shape
    tab = (C.c_void_p * 11)(*[p_.data_ptr() for p_ in ps])          # tab[10] stays NULL without sent_linear
    gamma = torch.empty(B, 128, dtype=torch.float32, device=sent.device)
    beta = torch.empty_like(gamma)
    hid = torch.empty(B, 256, dtype=torch.float32, device=sent.device)
    if a_pre is not None:
        assert a_pre.dtype == torch.float32 and tuple(a_pre.shape) == (2, B, 128) and a_pre.is_contiguous()
        L.call("xmc_concept_head_fwd_pre", _p(pooled), _p(sent), tab, _p(a_pre), _p(gamma), _p(beta), _p(hid), B, E, _st())
    else:
        L.call("xmc_concept_head_fwd", _p(pooled), _p(sent), tab, _p(gamma), _p(beta), _p(hid), B, E, _st())
    return gamma, beta, hid


class _HeadHoist:
    """What the stage nodes and HeadSentProductsFn share during one backward: the heads' layer-1 weight gradients of ALL stages, f32
    [S, 2, 128, E + 4], zero-filled.  A stage's backward accumulates its concept-state columns and nothing else into its slice; the
    hoisted node, which autograd runs after every stage that used its outputs, writes the sentence columns and hands the slices out as the
    gradients of the W1 parameters -- so neither side's contribution goes through a framework add."""

    def __init__(self, S, E, device):
        self.S, self.E, self.device, self.dw1 = S, E, device, None

    def grads(self):
        if self.dw1 is None:
            self.dw1 = _zeros_f32_out((self.S, 2, 128, self.E + 4), self.device)
        return self.dw1


class HeadSentProductsFn(torch.autograd.Function):
    """The sentence part of layer 1 of the gamma / beta heads for EVERY stage of a generator at once (df_concept_gan.py:238-253: the heads'
    grouped 1x1 over [sentence ; concept state], whose sentence columns see the same vector in every stage).
    apply(hoist, sent, W1_gamma_0, W1_beta_0, W1_gamma_1, ...) -> (A_0, A_1, ...), A_s f32 [2, B, 128] = sent @ W1_t[:, :E].T: one grouped
    GEMM forward; backward ONE batch product over all stages (xmc_concept_outer_multi) from the d(pre-activation) every stage's node returns
    as the gradient of its A_s, instead of one per stage."""

    @staticmethod
    def forward(ctx, hoist, sent, *w1):
        sent = sent.contiguous().float()
        B, E = sent.shape
        S = len(w1) // 2
        ws = [w.detach() for w in w1]
        assert len(w1) == 2 * S and all(w.is_contiguous() and w.dtype == torch.float32 and w.numel() == 128 * (E + 4) for w in ws)
        _need_cuda(sent, *ws)
        out = torch.empty(S, 2, B, 128, dtype=torch.float32, device=sent.device)
        t = np.zeros(2 * S, dtype=L.GEMM_PROBLEM)
        t["A"], t["B"] = sent.data_ptr(), np.array([w.data_ptr() for w in ws], dtype=np.uint64)
        t["C"] = (out.data_ptr() + np.arange(2 * S, dtype=np.int64) * (B * 128 * 4)).astype(np.uint64)
        t["M"], t["N"], t["K"] = B, 128, E
        t["sa_i"], t["sa_r"], t["sb_j"], t["sb_r"] = E, 1, E + 4, 1
        _gemm_group(t)
        ctx.hoist, ctx.shapes = hoist, [tuple(w.shape) for w in w1]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(sent, *ws)
        return tuple(out[s] for s in range(S))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *dA):
        sent, *ws = ctx.saved_tensors
        B, E = sent.shape
        S = len(ws) // 2
        # stage s hands back d A_s as a [2, B, 128] view of its [B, 256] d(pre-activation): row layout [B][S * 256] for the batch product
        D = torch.cat([torch.zeros(B, 256, dtype=torch.float32, device=sent.device) if g is None else g.permute(1, 0, 2).reshape(B, 256)
                       for g in dA], dim=1).contiguous()
        dw1 = ctx.hoist.grads()
        ctx.hoist.dw1 = None                      # the next backward starts from a fresh zero buffer
        dsent = _zeros_f32_out(tuple(sent.shape), sent.device)          # accumulated with atomics (arrives zero: xmc_set_prezeroed)
        mk = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
        L.call("xmc_concept_outer_multi", _p(D), _p(sent), mk([w.data_ptr() for w in ws]),
               mk([dw1[k // 2, k % 2].data_ptr() for k in range(2 * S)]), 2 * S, 128, _p(dsent), B, E, E + 4, _st())
        return (None, dsent) + tuple(dw1[k // 2, k % 2].view(ctx.shapes[k]) for k in range(2 * S))


def head_sentence_products(sent, w1s):
    """w1s = [(W1_gamma, W1_beta), ...] -> ([A_s f32 [2,B,128]], hoist): see HeadSentProductsFn / _HeadHoist"""
    hoist = _HeadHoist(len(w1s), sent.shape[1], sent.device)
    return list(HeadSentProductsFn.apply(hoist, sent, *[w for pair in w1s for w in pair])), hoist


def _head_bwd_raw(pooled, sent, hid, ps, dgamma, dbeta, hoist=None):
    """``hoist`` = (_HeadHoist, stage index): the batch products of layer 1's sentence columns are left to HeadSentProductsFn.backward;
    returns (dpooled, dsent | None, grads with None for the two W1 tensors, d(pre-activation) [B,256])."""
    B, E = sent.shape
    dpooled, dsent = torch.empty_like(pooled), torch.empty_like(sent)
    sizes = [(p_.numel() + 3) // 4 * 4 for p_ in ps]                      # 16-byte aligned slices of ONE zero-filled buffer
    flat = _zeros_f32_out(sum(sizes), sent.device)
    grads, off = [], 0
    for p_, n_ in zip(ps, sizes):
        grads.append(flat[off:off + p_.numel()].view(p_.shape))
        off += n_
    tab = (C.c_void_p * 11)(*[p_.data_ptr() for p_ in ps])
    scratch = torch.empty(B, 260, dtype=torch.float32, device=sent.device)
    if hoist is not None:
        H, k = hoist
        own = [g_.data_ptr() for g_ in grads]
        own[2], own[6] = H.grads()[k, 0].data_ptr(), H.grads()[k, 1].data_ptr()       # the stage's concept-state columns land in the shared buffer
        L.call("xmc_concept_head_bwd_pre", _p(pooled), _p(sent), _p(hid), tab, _p(dgamma), _p(dbeta), _p(dpooled), _p(dsent),
               (C.c_void_p * 11)(*own), _p(scratch), B, E, _st())
        grads[2] = grads[6] = None
        # (scratch holds d(pre-activation) as [B][256] followed by the [B][4] of the sent_linear term: not a [B, 260] matrix)
        return dpooled, (dsent if len(ps) > 10 else None), grads, scratch.view(-1)[:B * 256].view(B, 256)
    gtab = (C.c_void_p * 11)(*[g_.data_ptr() for g_ in grads])
    L.call("xmc_concept_head_bwd", _p(pooled), _p(sent), _p(hid), tab, _p(dgamma), _p(dbeta), _p(dpooled), _p(dsent), gtab,
           _p(scratch), B, E, _st())
    return dpooled, dsent, grads


class ConceptHeadFn(torch.autograd.Function):
    """Everything between the region attention and the channel modulation of one sampler stage of the attention-modulation
    blocks (df_concept_gan.py:238-253, 291-326; 471-478 for the self-attention block): value projection, ConceptReasoner,
    [sentence -> concept softmax re-weighting], the gamma and beta grouped MLPs on [sentence ; concept state].
    pooled f32 [B,16,8], sent f32 [B,E], ten parameters (+ sent_linear.weight) -> (gamma, beta) f32 [B,128]."""

    @staticmethod
    def forward(ctx, pooled, sent, *params):
        assert len(params) in (10, 11)
        pooled, sent = pooled.contiguous().float(), sent.contiguous().float()
        _need_cuda(pooled, sent)
        ps = [p_.detach().contiguous().float() for p_ in params]
        gamma, beta, hid = _head_fwd_raw(pooled, sent, ps)
        ctx.shapes = [tuple(p_.shape) for p_ in params]
        ctx.save_for_backward(pooled, sent, hid, *ps)
        return gamma, beta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dgamma, dbeta):
        pooled, sent, hid, *ps = ctx.saved_tensors
        dpooled, dsent, grads = _head_bwd_raw(pooled, sent, hid, ps, dgamma.contiguous().float(), dbeta.contiguous().float())
        return (dpooled, dsent) + tuple(g_.view(sh) for g_, sh in zip(grads, ctx.shapes))


def concept_query(sent, wq, gnw=None, gnb=None, eps=1e-5):
    return ConceptQueryFn.apply(sent, wq, gnw, gnb, eps)


class ConceptStageFn(torch.autograd.Function):
    """One sampler stage of an attention-modulation block as ONE autograd node (df_concept_gan.py:238-253 / 443-478 with the
    sampler 287-302 / 570-581):  key = key_gconv(x) [-> GroupNorm];  pooled = region attention(key, q, x);
    (gamma, beta) = concept head(pooled, sent);  y = lrelu(gamma * x + beta).
    x has three consumers (key projection, attention values, modulation).  As separate nodes their three gradients met in two
    framework add passes per stage; here the modulation's gradient is the buffer the attention backward accumulates into, and
    the key projection's data gradient takes that buffer as its residual: no add pass, no framework kernel in the stage."""

    @staticmethod
    def forward(ctx, x, q, sent, wk, gnw, gnb, geom, ncon, scale, eps, a_pre, hoist, *params):
        x = x.contiguous()
        q, sent = q.contiguous().float(), sent.contiguous().float()
        _need_cuda(x, q, sent)
        key = _conv_fwd_raw(x, wk, None, geom, L.ACT_NONE, x.dtype)
        gn = gnw is not None
        if gn:
            gwf, gbf = gnw.detach().float().contiguous(), gnb.detach().float().contiguous()
            keyn, gstats = _gn_fwd_raw(key, gwf, gbf, ncon, -1.0, eps)
        else:
            gwf = gbf = gstats = None
            keyn = key
        pooled, astats = _attn_fwd_raw(keyn, q, x, ncon, scale)
        ps = [p_.detach().contiguous().float() for p_ in params]
        gamma, beta, hid = _head_fwd_raw(pooled, sent, ps, a_pre)
        y = _affine_fwd_raw(x, [gamma, beta], 0.2)
        ctx.geom, ctx.ncon, ctx.scale, ctx.gn = geom, ncon, scale, gn
        ctx.hoist = hoist if a_pre is not None else None
        ctx.shapes = [tuple(p_.shape) for p_ in params]
        ctx.save_for_backward(x, q, sent, wk, key if gn else None, keyn, gwf, gbf, gstats, astats, pooled, hid, gamma, beta, *ps)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, q, sent, wk, key, keyn, gwf, gbf, gstats, astats, pooled, hid, gamma, beta, *ps = ctx.saved_tensors
        geom = ctx.geom
        dx, red = _affine_bwd_raw(x, dy.contiguous(), [gamma, beta], 0.2)
        da_pre = None
        if ctx.hoist is not None:
            dpooled, dsent, grads, da = _head_bwd_raw(pooled, sent, hid, ps, red[0], red[1], hoist=ctx.hoist)
            da_pre = da.view(da.shape[0], 2, 128).permute(1, 0, 2)            # d A_s = d(pre-activation), in A_s's [2, B, 128] indexing
        else:
            dpooled, dsent, grads = _head_bwd_raw(pooled, sent, hid, ps, red[0], red[1])
        dkeyn, dq, dx = _attn_bwd_raw(keyn, q, x, astats, pooled, dpooled, ctx.ncon, ctx.scale, dx_acc=dx)
        dgw = dgb = None
        if ctx.gn:
            dkey, dgw, dgb = _gn_bwd_raw(key, dkeyn, gwf, gbf, gstats, ctx.ncon, -1.0)
        else:
            dkey = dkeyn
        dxt = _conv_dgrad_raw(dkey, wk, geom, (x.shape[1], x.shape[2]), x.dtype, res=dx)
        dwk = None
        if ctx.needs_input_grad[3] and not _skip_wgrad():
            dwk = _conv_wgrad_raw(x, dkey, geom).view(wk.shape)
        return (dxt, dq.view(q.shape), dsent, dwk, dgw, dgb, None, None, None, None, da_pre, None) + \
            tuple(None if g_ is None else g_.view(sh) for g_, sh in zip(grads, ctx.shapes))


def concept_stage(x, q, sent, wk, gnw, gnb, geom, ncon, scale, head_params, eps=1e-5, a_pre=None, hoist=None):
    """``a_pre`` / ``hoist`` = (A_s, (_HeadHoist, s)) from head_sentence_products: layer 1's sentence products, computed for all stages at once"""
    return ConceptStageFn.apply(x, q, sent, wk, gnw, gnb, geom, ncon, scale, eps, a_pre, hoist, *head_params)


def concept_gquery(q0, wq, gnw=None, gnb=None, eps=1e-5):
    return ConceptGQueryFn.apply(q0, wq, gnw, gnb, eps)


def concept_head(pooled, sent, params):
    return ConceptHeadFn.apply(pooled, sent, *params)


# ------------------------------------------------------------------------------------------ per-concept algebra of the word-attention generators
# (csrc/concept_word.hip; model/concept_gan.py).  First-order nodes (the generator path is never differentiated twice).
def _f32c(t):
    return None if t is None else t.detach().contiguous().float()


class GroupedVecFn(torch.autograd.Function):
    """y[b,g,o] = bias[g,o] + W[g,o,:Is] . xs[b] + W[g,o,Is:] . xg[b,g]: a grouped 1x1 convolution of a per-sample vector whose
    groups share the first Is inputs (the gamma / beta heads on cat(global condition, context), concept_gan.py:346-371,404-418;
    Is = 0: the samplers' query / value projections).  xs [B,Is] or None, xg [B,G,Ig] or None, W [G*O, Is+Ig(,1,1)], bias [G*O] or None."""

    @staticmethod
    def forward(ctx, xs, xg, W, bias, G):
        xs_, xg_, W_ = _f32c(xs), _f32c(xg), _f32c(W).view(W.shape[0], -1)
        B = (xs_ if xs_ is not None else xg_).shape[0]
        Is, Ig = (0 if xs_ is None else xs_.shape[1]), (0 if xg_ is None else xg_.shape[2])
        O = W_.shape[0] // G
        assert W_.shape[1] == Is + Ig and W_.shape[0] == G * O
        _need_cuda(W_, xs_, xg_)
        y = torch.empty((B, G, O), dtype=torch.float32, device=W_.device)
        L.call("xmc_gvec_fwd", _p(xs_), _p(xg_), _p(W_), _p(_f32c(bias)), _p(y), B, G, O, Is, Ig, _st())
        ctx.dims = (B, G, O, Is, Ig)
        ctx.wshape = W.shape
        ctx.save_for_backward(xs_, xg_, W_)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        xs_, xg_, W_ = ctx.saved_tensors
        B, G, O, Is, Ig = ctx.dims
        dy = dy.contiguous().float()
        need = ctx.needs_input_grad
        dxs = torch.empty_like(xs_) if (need[0] and Is) else None
        dxg = torch.empty_like(xg_) if (need[1] and Ig) else None
        dW = torch.empty_like(W_) if need[2] else None
        db = torch.empty(G * O, dtype=torch.float32, device=dy.device) if need[3] else None
        L.call("xmc_gvec_bwd", _p(xs_), _p(xg_), _p(W_), _p(dy), _p(dxs), _p(dxg), _p(dW), _p(db), B, G, O, Is, Ig, _st())
        return dxs, dxg, (None if dW is None else dW.view(ctx.wshape)), db, None


def grouped_vec(xs, xg, W, bias, groups):
    return GroupedVecFn.apply(xs, xg, W, bias, groups)


class ReasonerFn(torch.autograd.Function):
    """concept_gan.ConceptReasoner (632-654): relu(BatchNorm1d(x + tanh(x We^T) x)) on x [B,16,4]; ``bn`` = (weight, bias, running_mean,
    running_var, training, momentum, eps) or None.  Training mode updates the running statistics in place like nn.BatchNorm1d."""

    @staticmethod
    def forward(ctx, x, We, bn_w, bn_b, run_mean, run_var, training, momentum, eps):
        x_, We_ = _f32c(x), _f32c(We)
        B = x_.shape[0]
        assert tuple(x_.shape[1:]) == (16, 4) and tuple(We_.shape) == (16, 4)
        _need_cuda(x_, We_)
        y, pre = torch.empty_like(x_), torch.empty_like(x_)
        stat = torch.empty(32, dtype=torch.float32, device=x_.device)
        L.call("xmc_reasoner_fwd", _p(x_), _p(We_), _p(_f32c(bn_w)), _p(_f32c(bn_b)), _p(run_mean), _p(run_var), int(bool(training)),
               float(momentum), float(eps), _p(y), _p(pre), _p(stat), B, _st())
        ctx.batch_stats = bool(training) and bn_w is not None
        ctx.save_for_backward(x_, We_, _f32c(bn_w), _f32c(bn_b), pre, stat)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x_, We_, bw, bb, pre, stat = ctx.saved_tensors
        B = x_.shape[0]
        dy = dy.contiguous().float()
        need = ctx.needs_input_grad
        dx = torch.empty_like(x_) if need[0] else None
        dWe = torch.empty_like(We_) if need[1] else None
        dbw = torch.empty(16, dtype=torch.float32, device=dy.device) if (bw is not None and need[2]) else None
        dbb = torch.empty(16, dtype=torch.float32, device=dy.device) if (bw is not None and need[3]) else None
        L.call("xmc_reasoner_bwd", _p(x_), _p(We_), _p(bw), _p(bb), _p(pre), _p(stat), int(ctx.batch_stats), _p(dy), _p(dx), _p(dWe), _p(dbw),
               _p(dbb), B, _st())
        return dx, dWe, dbw, dbb, None, None, None, None, None


def reasoner(x, We, bn=None):
    if bn is None:
        return ReasonerFn.apply(x, We, None, None, None, None, False, 0.0, 1e-5)
    if bn.training:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    return ReasonerFn.apply(x, We, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps)


def reasoner_stats_only(x, We, bn):
    """upstream's discarded reasoner call (concept_gan.py:432): all that survives is the BatchNorm1d running-statistics update"""
    x_, We_ = _f32c(x), _f32c(We)
    pre = torch.empty_like(x_)
    L.call("xmc_reasoner_fwd", _p(x_), _p(We_), _p(_f32c(bn.weight)), _p(_f32c(bn.bias)), _p(bn.running_mean), _p(bn.running_var), 1,
           float(bn.momentum), float(bn.eps), None, _p(pre), None, x_.shape[0], _st())
    with torch.no_grad():
        bn.num_batches_tracked += 1


class WordContextFn(torch.autograd.Function):
    """OutConceptBlock.get_context_embs (concept_gan.py:374-394): state [B,16,4], words [B,T,4], mask [B,T] (True = padding) -> [B,16,4]."""

    @staticmethod
    def forward(ctx, state, words, mask):
        st_, w_ = _f32c(state), _f32c(words)
        B, T = w_.shape[0], w_.shape[1]
        assert tuple(st_.shape) == (B, 16, 4) and w_.shape[2] == 4
        _need_cuda(st_, w_)
        pad = mask.to(torch.uint8).contiguous()
        out = torch.empty_like(st_)
        prob = torch.empty((B, 16, T), dtype=torch.float32, device=st_.device)
        L.call("xmc_word_ctx_fwd", _p(st_), _p(w_), _p(pad), _p(out), _p(prob), B, T, _st())
        ctx.save_for_backward(st_, w_, prob)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dctx):
        st_, w_, prob = ctx.saved_tensors
        B, T = w_.shape[0], w_.shape[1]
        dctx = dctx.contiguous().float()
        dst = torch.empty_like(st_) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w_) if ctx.needs_input_grad[1] else None
        L.call("xmc_word_ctx_bwd", _p(st_), _p(w_), _p(prob), _p(dctx), _p(dst), _p(dw), B, T, _st())
        return dst, dw, None


def word_context(state, words, mask):
    return WordContextFn.apply(state, words, mask)


class WordKeysFn(torch.autograd.Function):
    """CondConceptSampler's keys (concept_gan.py:566-575): kraw [B,T,64] -> [GroupNorm(16, 64) over (state, word)] -> normalised over the
    state axis -> [B,16,T,4]."""

    @staticmethod
    def forward(ctx, kraw, gnw, gnb, eps):
        k_ = _f32c(kraw)
        B, T = k_.shape[0], k_.shape[1]
        assert k_.shape[2] == 64
        _need_cuda(k_)
        kh = torch.empty((B, 16, T, 4), dtype=torch.float32, device=k_.device)
        stat = torch.empty((B, 16, 2), dtype=torch.float32, device=k_.device) if gnw is not None else None
        L.call("xmc_word_keys_fwd", _p(k_), _p(_f32c(gnw)), _p(_f32c(gnb)), float(eps), _p(kh), _p(stat), B, T, _st())
        ctx.save_for_backward(k_, _f32c(gnw), _f32c(gnb), stat)
        return kh

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dkh):
        k_, gw, gb, stat = ctx.saved_tensors
        B, T = k_.shape[0], k_.shape[1]
        dkh = dkh.contiguous().float()
        dk = torch.empty_like(k_)
        dgw = _zeros_f32_out(64, k_.device) if (gw is not None and ctx.needs_input_grad[1]) else None
        dgb = _zeros_f32_out(64, k_.device) if (gw is not None and ctx.needs_input_grad[2]) else None
        L.call("xmc_word_keys_bwd", _p(k_), _p(gw), _p(gb), _p(stat), _p(dkh), _p(dk), _p(dgw), _p(dgb), B, T, _st())
        return dk, dgw, dgb, None


def word_keys(kraw, gnw=None, gnb=None, eps=1e-5):
    return WordKeysFn.apply(kraw, gnw, gnb, eps)


class GradPenaltyFn(torch.autograd.Function):
    """mean_b ||[g0_b, g1_b, ...]||_2^6 over f32 gradient blocks [B, ...] (train_gan.py:241-247: cat, **2, sum, sqrt, **6,
    mean) in two passes over the data: per-sample sums of squares, then -- in the backward -- one scaled copy per block.
    Once differentiable, which is what `d_loss.backward()` asks of it: the blocks are themselves outputs of
    `autograd.grad(create_graph=True)`, so the gradients returned here continue into the second-order graph of D."""

    @staticmethod
    def forward(ctx, inner_scale, *blocks):
        B = blocks[0].shape[0]
        flat = []
        for g in blocks:
            g = g.contiguous().float().reshape(B, -1)
            if g.shape[1] % 4:
                g = torch.nn.functional.pad(g, (0, 4 - g.shape[1] % 4))
            flat.append(g)
        _need_cuda(*flat)
        ss = _zeros_f32(B, flat[0].device)
        for g in flat:
            L.call("xmc_rows_sumsq", _p(g), _p(ss), B, g.shape[1], _st())
        gp = torch.empty(1, dtype=torch.float32, device=ss.device)
        coef = torch.empty(B, dtype=torch.float32, device=ss.device)
        L.call("xmc_gp_finish", _p(ss), B, _p(gp), _p(coef), 1.0 / float(inner_scale) ** 2, _st())
        ctx.shapes = [tuple(b.shape) for b in blocks]
        ctx.dtypes = [b.dtype for b in blocks]
        ctx.save_for_backward(coef, *flat)
        return gp.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dgp):
        coef, *flat = ctx.saved_tensors
        B = coef.numel()
        gdev = dgp.reshape(1).float().contiguous()
        outs = []
        for g, shp, dt in zip(flat, ctx.shapes, ctx.dtypes):
            y = torch.empty_like(g)
            L.call("xmc_rows_scale", _p(g), _p(coef), _p(gdev), _p(y), B, g.shape[1], _st())
            n = 1
            for v in shp[1:]:
                n *= v
            outs.append(y[:, :n].reshape(shp).to(dt))
        return (None, *outs)


def grad_penalty(*blocks, inner_scale=1.0):
    """``inner_scale``: the blocks hold inner_scale x the gradients (see `gp_inner_scale`)"""
    return GradPenaltyFn.apply(float(inner_scale), *blocks)


def cosine_scores(a, b):
    """normalize(a) @ normalize(b).T -> f32 [n,n] (train_gan.py:85-91); not differentiable (label construction only)."""
    a, b = a.detach().contiguous().float(), b.detach().contiguous().float()
    _need_cuda(a, b)
    n, D = a.shape
    ws = torch.empty(L.load().xmc_contrastive_ws_bytes(n, D), dtype=torch.uint8, device=a.device)
    s = torch.empty((n, n), dtype=torch.float32, device=a.device)
    L.call("xmc_cosine_scores", _p(a), _p(b), n, D, _p(s), _p(ws), _st())
    return s


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
