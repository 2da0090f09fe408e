"""xmc_gan_amd.ops: layer geometry, packed-weight cache, the raw convolution launches (forward, data gradient, weight gradient, composed stem) and the zero-filled scratch pools.
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
    _DEBUG_DISPATCH, _code, _nbytes, _need_cuda, _p, _st, _weights_epoch, bump_weights_epoch, chan_pad, pad_to)


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
