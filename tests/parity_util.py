"""Shared helpers for the GPU parity tests and __graft_entry__.smoke(): build the product modules from a cfg
preset, load the oracle's synthetic parameters, run the product step and compare with the CPU oracle.
(Test infrastructure: imports the oracle.)"""
import os

import numpy as np
import torch

import xmc_ref as X
from golden_util import CFG_DIR

DEV = "cuda"


def setup_cfg(yml, **over):
    from xmc_gan.config import gan
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, yml))
    for dotted, v in over.items():
        node = gan.cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return gan.cfg, X.Hyper.from_cfg(gan.cfg)


def build_product(h, PG, PD, eps=1e-8):
    import xmc_gan.train_gan as tg
    from xmc_gan_amd.optim import HipAdam
    netG = tg._GEN_ARCH[h.gen](tg.cfg).to(DEV)
    netD = tg._DISC_ARCH["DF_DISC"](tg.cfg, is_disc=True).to(DEV)
    netG.load_state_dict(PG, strict=True)
    netD.load_state_dict(PD, strict=True)
    optG = HipAdam(netG.parameters(), lr=h.g_lr, betas=h.g_betas, eps=eps)
    optD = HipAdam(netD.parameters(), lr=h.d_lr, betas=h.d_betas, eps=eps)
    return netG, netD, optG, optD


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def mean_abs_err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().mean().item()


class GradTap:
    """Captures the gradients an optimizer is about to apply (clone at .step() time)."""

    def __init__(self, opt, named_params):
        self.named = list(named_params)
        self.records = []
        orig = opt.step

        def step(*a, **k):
            gs = float(k.get("grad_scale", 1.0))
            if k.get("scaler") is not None:           # IEEE-half mode: the backward ran on the scaler's scale x the loss
                gs = float(k["scaler"].sf[1])
            self.records.append({n: (None if p.grad is None else p.grad.detach().float().cpu().clone() * gs) for n, p in self.named})
            return orig(*a, **k)

        opt.step = step


def run_product_steps(h, PG, PD, batches, eps=1e-8):
    """Run len(batches) iterations of the product; returns (netG, netD, list of loss dicts, grad taps)."""
    import xmc_gan.train_gan as tg
    netG, netD, optG, optD = build_product(h, PG, PD, eps)
    tapG, tapD = GradTap(optG, netG.named_parameters()), GradTap(optD, netD.named_parameters())
    state, outs = {}, []
    for b in batches:
        o = tg.gan_iteration(netG, netD, optG, optD, b["imgs"].to(DEV), b["sent_embs"].to(DEV), b["words_embs"].to(DEV),
                             b["mask"].to(DEV), b["noise"].to(DEV), state)
        outs.append({k: (v.float().cpu() if torch.is_tensor(v) else v) for k, v in o.items()})
    torch.cuda.synchronize()
    return netG, netD, outs, tapG, tapD


def run_oracle_steps(h, PG, PD, batches, eps=1e-8, quant=False, fmt=None):
    """`quant`: the oracle's quantisation-aware mode (rounding to the 16-bit format `fmt`, default bf16, at the engine's
    storage points; f32 arithmetic)."""
    with X.quant(quant, fmt=fmt or torch.bfloat16):
        return _run_oracle_steps(h, PG, PD, batches, eps)


def _run_oracle_steps(h, PG, PD, batches, eps):
    PG = {k: v.clone() for k, v in PG.items()}
    PD = {k: v.clone() for k, v in PD.items()}
    optG, optD = X.AdamState(h.g_lr, h.g_betas, eps), X.AdamState(h.d_lr, h.d_betas, eps)
    outs, it = [], 0
    for b in batches:
        it += 1
        o = X.train_step(PG, PD, optG, optD, h, b, it_count=it)
        if it % h.n_critic == 0:
            it = 0
        outs.append(o)
    return PG, PD, outs


LOSS_KEYS = ("errD_real", "errD_fake", "errD_mismatch", "ds_loss", "errD", "d_loss_gp", "errG_fake", "gs_loss",
             "disc_loss", "errG")


def compare_losses(prod, orac, rtol, atol, after_gp=1.0):
    """``after_gp``: factor on the bar of the generator-step losses when a gradient-penalty Adam step precedes them (MA-GP on)"""
    worst = 0.0
    used_l = dict(loss=0.0, loss_name="")
    for k in LOSS_KEYS:
        if k in orac:
            assert k in prod, f"product did not report {k}"
            p, o = float(prod[k]), float(orac[k])
            r = rtol * (6.0 if k == "d_loss_gp" else 1.0)      # 6th power of a norm: relative error x6
            if "d_loss_gp" in orac and k in ("errG_fake", "gs_loss", "disc_loss", "errG"):
                r *= after_gp
            err = abs(p - o) / (abs(o) + atol / r)
            worst = max(worst, err / (6.0 if k == "d_loss_gp" else 1.0))
            frac = abs(p - o) / (r * abs(o) + atol)
            if frac > used_l["loss"]:
                used_l["loss"], used_l["loss_name"] = frac, k
            assert abs(p - o) <= r * abs(o) + atol, f"{k}: product {p} vs oracle {o}"
    USED.append(("losses", used_l))
    return worst


def compare_grads(tap_rec, oracle_grads, rtol, name="", floor=1e-5, agg_rtol=None, loose=None):
    """Per-tensor relative L2 error of every gradient; None-ness must match.  Parameters with <= 4 elements (the block
    gammas: d/dgamma = <dout, residual>, and conv_out's 3-channel bias: heavily cancelling sums over every pixel) are
    compared on the scale of the largest such gradient in the same backward instead of their own magnitude.
    `loose` = (predicate on the parameter name, factor): per-tensor tolerance x factor for the tensors it selects (they still
    count in the aggregate)."""
    worst = 0.0
    num2 = den2 = 0.0
    used = dict(tensor=0.0, tensor_name="", loose=0.0, loose_name="", agg=0.0)     # fraction of each bar this call used (USED, below)
    sc_scale = max([go.abs().max().item() for go in oracle_grads.values() if go is not None and go.numel() <= 4] + [0.0])
    # gradients that are structurally zero (e.g. the bias of the GroupNorm applied to attention KEYS: a per-concept constant
    # added to every score leaves the softmax unchanged) come out as rounding noise on both sides: compare those on the
    # scale of the largest gradient norm of this backward instead of their own
    big = max([go.norm().item() for go in oracle_grads.values() if go is not None] + [0.0])
    for n, go in oracle_grads.items():
        gp = tap_rec.get(n)
        assert (gp is None) == (go is None), f"{name}{n}: None-ness differs (product {gp is None}, oracle {go is None})"
        if go is None:
            continue
        if go.numel() <= 4:      # gammas and the 3-channel conv_out bias: sums over every pixel with heavy cancellation
            err = (gp - go).abs().max().item() / max(sc_scale, 1e-12)
        else:
            den = max(go.norm().item(), floor * big)
            err = (gp - go).norm().item() / den if den > 0 else gp.norm().item()
        worst = max(worst, err)
        num2 += (gp - go).double().pow(2).sum().item()
        den2 += go.double().pow(2).sum().item()
        is_loose = loose is not None and loose[0](n)
        tol = rtol * ((loose[1](n) if callable(loose[1]) else loose[1]) if is_loose else 1.0)
        key = "loose" if is_loose else "tensor"
        if err / tol > used[key]:
            used[key], used[key + "_name"] = err / tol, n
        assert err <= tol, f"{name}{n}: rel error {err:.3e} > {tol}"
    if agg_rtol is not None:      # all gradient tensors of this backward taken as one vector
        agg = (num2 / max(den2, 1e-300)) ** 0.5
        used["agg"] = agg / agg_rtol
        assert agg <= agg_rtol, f"{name}aggregate gradient error {agg:.3e} > {agg_rtol}"
    USED.append((name.strip(), used))
    return worst


# what fraction of its bar every compare_grads / compare_losses call used (the tests print the maxima: a bar is ratcheted to <= 1.5 x
# the largest measured value, i.e. no call should stay below 0.67 for long without the bar coming down)
USED = []


def used_summary(clear=True):
    out = {}
    for name, u in USED:
        o = out.setdefault(name, dict(tensor=0.0, tensor_name="", loose=0.0, loose_name="", agg=0.0, loss=0.0, loss_name=""))
        for k in ("tensor", "loose", "loss"):
            if u.get(k, 0.0) > o[k]:
                o[k], o[k + "_name"] = u[k], u.get(k + "_name", "")
        o["agg"] = max(o["agg"], u.get("agg", 0.0))
    if clear:
        USED.clear()
    return out


def concept_quant_walk(kind, over=None, batch=3):
    """Teacher-forced walk over the storage points of an attention-modulation generator (kind "in" / "out") in the engine's CURRENT
    16-bit mode: every stage of the product reads the quantisation-aware ORACLE's (already rounded) input of that stage and its output is
    compared with the oracle's value at the same point.  Returns [(block, site, bit-equal fraction, relative L2 error)]; the
    key projection and its GroupNorm live inside the stage node and are covered through the stage output."""
    import xmc_ref as X
    from xmc_gan_amd import ops
    from xmc_gan_amd.lib import ACT_LRELU
    yml = {"in": "concept_in_df_gan_damsm_nomagp.yml", "out": "concept_out_df_gan_sbert_damsm_nomagp.yml"}[kind]
    cfg, h = setup_cfg(yml, **(over or {"TRAIN.NCH": 8}))
    PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
    b = X.synth_batch(h, batch, seed=200, words_len=cfg.TEXT.MAX_LENGTH)
    netG, _, _, _ = build_product(h, PG, PD)
    fmt = torch.bfloat16 if ops.precision() == "bf16" else torch.float16
    log, q0_, q1_ = [], X.q, X.q_

    def rec_q(x, site=None):
        y = q0_(x, site)
        log.append((site, y.detach()))
        return y

    X.q = X.q_ = rec_q
    try:
        with torch.no_grad(), X.quant(True, fmt=fmt):
            X.gen_forward(PG, h, b["noise"], b["sent_embs"])
    finally:
        X.q, X.q_ = q0_, q1_

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous().to(DEV, fmt)

    rows = []

    def cmp(blk_i, what, got_nhwc, want):
        got = got_nhwc.permute(0, 3, 1, 2).float().cpu()[:, : want.size(1)]
        rows.append((blk_i, what, (got == want).float().mean().item(), float((got - want).norm() / want.norm())))

    sent = netG.proj_sent(b["sent_embs"].to(DEV)).float()
    it = iter(log)
    site, cur = next(it)
    assert site == "g.stem", site
    cur = cur.view(cur.size(0), 8 * h.nch, 4, 4)
    with torch.no_grad():
        for i, blk in enumerate(netG.upblocks):
            xin = cur
            for cname in ("concept1", "concept2"):
                cb = getattr(blk, cname)
                site, want = next(it); assert site == "g.c.split", site
                cmp(i, cname + ".split_conv", cb.split_conv(nhwc(cur), act=ACT_LRELU), want); cur = want
                site, want = next(it); assert site == "g.c.trans", site
                cmp(i, cname + ".trans_gconv", cb.trans_gconv(nhwc(cur)), want); cur = want
                site, want = next(it); assert site == "g.c.trunk", site
                cmp(i, cname + ".trunk", ops.groupnorm(nhwc(cur), cb.gn.weight, cb.gn.bias, cb.cardinality, slope=0.2) if cb.normalize
                    else ops.lrelu(nhwc(cur)), want); cur = want
                for j in (1, 2):
                    samp, reas = getattr(cb, f"concept_sampler{j}"), getattr(cb, f"concept_reasoner{j}")
                    gm, bm = getattr(cb, f"gamma{j}_gconv"), getattr(cb, f"beta{j}_gconv")
                    site, _ = next(it); assert site == "g.c.key", site
                    if cb.normalize:
                        site, _ = next(it); assert site == "g.c.keyn", site
                    site, want = next(it); assert site == "g.c.mod", site
                    x = nhwc(cur)
                    hp = (samp.value_gconv.weight, reas.proj_edge.weight, gm[0].weight, gm[0].bias, gm[2].weight, gm[2].bias,
                          bm[0].weight, bm[0].bias, bm[2].weight, bm[2].bias)
                    gn1 = (samp.gn1.weight, samp.gn1.bias) if samp.normalize else (None, None)
                    if kind == "in":
                        y = samp.stage(x, ops.concept_query(sent, samp.query_gconv.weight, *gn1), 1.0, sent, hp)
                    else:
                        qv = ops.concept_gquery(ops.global_avgpool(x).view(x.size(0), -1), samp.query_gconv.weight, *gn1)
                        y = samp.stage(x, qv, samp._scale, sent, hp + (getattr(cb, f"sent_linear{j}").weight,))
                    cmp(i, f"{cname}.stage{j}", y, want); cur = want
                site, want = next(it); assert site in ("g.c.out1", "g.c.out2"), site
                conv = blk.conv_out1 if cname == "concept1" else blk.conv_out2
                cmp(i, cname + "->" + site[4:], conv(nhwc(cur), act=ACT_LRELU) if cname == "concept1" else conv(nhwc(cur)), want); cur = want
            r2 = cur
            if blk.learnable_sc:
                site, sc = next(it); assert site == "g.sc", site
                cmp(i, "c_sc", blk.c_sc(nhwc(xin)), sc)
            else:
                sc = xin
            site, want = next(it); assert site == "g.sum", site
            cmp(i, "sum", ops.axpby(nhwc(sc), nhwc(r2), blk.gamma), want)
            cur = torch.nn.functional.interpolate(want, scale_factor=2) if blk.upsample else want
    return rows
