"""xmc_gan_amd.ops: autograd nodes of the convolution / linear layers, the generator block end, the conditioning-MLP bank and the text front end.
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
    _code, _p, _skip_wgrad, _st, act_dtype, fused_blocks, pad_to)
from ._engine import (
    _conv1x1_pair_raw, _conv_dgrad_raw, _conv_fwd_raw, _conv_wgrad_raw, _pooled_take, _upconv_dgrad_raw,
    _upconv_fwd_raw, _zeros_f32, _zeros_f32_out)


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
