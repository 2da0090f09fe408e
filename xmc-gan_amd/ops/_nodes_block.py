"""xmc_gan_amd.ops: pointwise nodes, the discriminator blocks (first- and second-order), pooling, affine modulation, GroupNorm / BatchNorm, region attention.
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
    _DEBUG_DISPATCH, _code, _need_cuda, _p, _second_order, _skip_wgrad, _st, fused_blocks, pad_to, precise_trunk)
from ._engine import (
    _StagedMask, _conv1x1_pair_raw, _conv_dgrad_raw, _conv_fwd_raw, _conv_wgrad_raw, _dstem_border_fwd_raw,
    _dstem_compose_bwd_raw, _dstem_compose_raw, _dstem_dgrad_raw, _dstem_fwd_raw, _dstem_sc_operands,
    _dstem_wgrad_raw, _pooled_put, _zeros_f32, _zeros_f32_out)
from ._nodes_conv import (
    _axpby_bwd_fused)


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
