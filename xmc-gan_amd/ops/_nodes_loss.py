"""xmc_gan_amd.ops: losses (hinge, contrastive head), the concept algebra of the attention-modulation and word-attention generators, the gradient penalty.
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
    _code, _need_cuda, _p, _skip_wgrad, _st)
from ._engine import (
    _conv_dgrad_raw, _conv_fwd_raw, _conv_wgrad_raw, _zeros_f32, _zeros_f32_out)
from ._nodes_conv import (
    _gemm_group)
from ._nodes_block import (
    _affine_bwd_raw, _affine_fwd_raw, _attn_bwd_raw, _attn_fwd_raw, _gn_bwd_raw, _gn_fwd_raw)


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
