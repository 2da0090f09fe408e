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
    for k in LOSS_KEYS:
        if k in orac:
            assert k in prod, f"product did not report {k}"
            p, o = float(prod[k]), float(orac[k])
            r = rtol * (6.0 if k == "d_loss_gp" else 1.0)      # 6th power of a norm: relative error x6
            if "d_loss_gp" in orac and k in ("errG_fake", "gs_loss", "disc_loss", "errG"):
                r *= after_gp
            err = abs(p - o) / (abs(o) + atol / r)
            worst = max(worst, err / (6.0 if k == "d_loss_gp" else 1.0))
            assert abs(p - o) <= r * abs(o) + atol, f"{k}: product {p} vs oracle {o}"
    return worst


def compare_grads(tap_rec, oracle_grads, rtol, name="", floor=1e-5, agg_rtol=None, loose=None):
    """Per-tensor relative L2 error of every gradient; None-ness must match.  Parameters with <= 4 elements (the block
    gammas: d/dgamma = <dout, residual>, and conv_out's 3-channel bias: heavily cancelling sums over every pixel) are
    compared on the scale of the largest such gradient in the same backward instead of their own magnitude.
    `loose` = (predicate on the parameter name, factor): per-tensor tolerance x factor for the tensors it selects (they still
    count in the aggregate)."""
    worst = 0.0
    num2 = den2 = 0.0
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
        tol = rtol * ((loose[1](n) if callable(loose[1]) else loose[1]) if loose is not None and loose[0](n) else 1.0)
        assert err <= tol, f"{name}{n}: rel error {err:.3e} > {tol}"
    if agg_rtol is not None:      # all gradient tensors of this backward taken as one vector
        agg = (num2 / max(den2, 1e-300)) ** 0.5
        assert agg <= agg_rtol, f"{name}aggregate gradient error {agg:.3e} > {agg_rtol}"
    return worst
