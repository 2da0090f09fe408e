"""The three precision modes of the engine against north_star's bar -- "logits/loss within 1e-3 rel of the CPU reference".

  fp32  exact-f32 MFMA, f32 storage            meets it (tests/test_models_gpu.py, measured ~1e-6); 1/16 of the bf16 MFMA rate
  bf16  bf16 storage + MFMA operands           cannot: the ladder (tests/diag/quant_ladder.py) shows the error is
                                               the 8-bit significand of EVERY convolution operand, not a few stored tensors
  f16   IEEE-half storage + MFMA operands      the same kernels compiled for the 11-bit format (libxmc_gan_hip_f16.so), same
                                               MFMA rate: THIS is the mode that is both fast and inside 1e-3 on the losses

Checked here at the real widths (NCH=32, 64x64, batch 8: BASELINE configuration 1/2's network) on two parameterisations:
the parity tests' synthetic one (block gammas 0.25-0.75: worst case) and the reference's own start of training (Kaiming weights,
zero biases, block gammas 0.1 as bench.py sets them)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import xmc_ref as X
    from xmc_gan_amd import ops
    from parity_util import (DEV, build_product, compare_grads, compare_losses, rel_err, run_oracle_steps, run_product_steps,
                             setup_cfg)


def _params(h, kind, seed):
    if kind == "synth":
        return X.synth_params(X.gen_shapes(h), 5 + seed), X.synth_params(X.netd_shapes(h), 6 + seed)
    return X.ref_init_params(X.gen_shapes(h), 5 + seed, 0.1), X.ref_init_params(X.netd_shapes(h), 6 + seed, 0.1)


def _logits(h, PG, PD, b):
    netG, netD, _, _ = build_product(h, PG, PD)
    with torch.no_grad():
        ps = netG.proj_sent(b["sent_embs"].to(DEV))
        return netD.COND_DNET(netD(b["imgs"].to(DEV)), sent_embs=ps)[0].float().cpu()


@pytest.fixture(autouse=True)
def _restore_precision():
    yield
    ops.set_precision("bf16")


@pytest.mark.parametrize("kind", ["synth", "ref"])
@pytest.mark.parametrize("yml", ["df_gan_damsm_nomagp.yml", "df_gan_damsm.yml"])
def test_f16_mode_losses_within_1e_3_of_f32_reference(yml, kind):
    """first G+D iteration in the IEEE-half mode vs the plain f32 oracle: every loss scalar (D step, MA-GP, G step) and the
    real-pair logit vector within 1e-3 relative (north_star); against the oracle rounding to half where the engine stores a
    tensor (kernel error proper) 5e-4."""
    ops.set_precision("f16")
    cfg, h = setup_cfg(yml)
    PG, PD = _params(h, kind, 0)
    batches = [X.synth_batch(h, 8, seed=200, words_len=cfg.TEXT.MAX_LENGTH)]
    _, _, o = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
    _, _, oq = run_oracle_steps(h, PG, PD, batches, eps=1e-3, quant=True, fmt=torch.float16)
    _, _, p, tapG, tapD = run_product_steps(h, PG, PD, batches, eps=1e-3)
    # D-step losses (and every loss of the headline configuration): 1e-3.  With MA-GP the generator step runs against a discriminator
    # that has just taken the penalty's Adam step.  Rounds 1-3 ran the penalty's INNER backward unscaled (gradients among the
    # subnormals of the format; G-step losses 1.4e-3, bar 3e-3); since round 4 it runs on ops.gp_inner_scale() x ones and the outer
    # backward on its own dynamic scale.
    dkeys = ("errD_real", "errD_fake", "errD_mismatch", "ds_loss", "errD")
    sel = lambda d_, keys: {k: v for k, v in d_.items() if k in keys}
    wl = compare_losses(sel(p[0], dkeys), sel(o[0], dkeys), 1e-3, 1e-4)
    if h.magp:
        # the penalty 2 mean(||d logit / d inputs||^6) is evaluated on the discriminator AFTER its Adam step (beta1 = 0: a weight whose
        # gradient sits inside rounding noise of zero moves by up to 2 lr the other way: `final D weights: worst element 2 lr from the
        # oracle's` in tests/diag/precise_magp_probe.py) and raises the norm to the 6th power.  Five seeds on one box, relative error of
        # the penalty, with / without the precise trunk: 6.6e-3 / 2.2e-3, 3e-6 / 7e-6, 7e-5 / 2.3e-4, 4.6e-4 / 1.8e-3, 4.6e-3 / 4.3e-3 --
        # seed noise of a few 1e-3 in either mode, i.e. ~1e-3 on the norm itself: the bar is 1.5e-3 on the norm (9e-3 on its 6th power)
        wl = max(wl, compare_losses(sel(p[0], ("d_loss_gp",)), sel(o[0], ("d_loss_gp",)), 1.5e-3, 1e-4))
    gkeys = ("errG_fake", "gs_loss", "disc_loss", "errG")
    # G-step losses: 1e-3 in the headline configuration (measured 3.4e-4 .. 4.6e-4).  With MA-GP they are evaluated on a discriminator
    # that has taken the PENALTY's Adam step, and that step is sensitive at the last bit: swapping the kernel of ONE layer's data gradient
    # for an equally exact one (tests/diag/s128_swap_*.py: 0.09 % of its elements differ, by one ulp) moves errG_fake from 1.8e-4 to
    # 1.9e-3 (tests/diag/magp_swap_sensitivity.sh) -- so no implementation can promise 1e-3 there; the bar is 3e-3, as in rounds 1-3
    wl = max(wl, compare_losses(sel(p[0], gkeys), sel(o[0], gkeys), 3e-3 if h.magp else 1e-3, 1e-4))
    # (with MA-GP the G-step losses follow a discriminator that has taken the penalty's Adam step: the rounding points of its double
    # backward are only approximately the engine's, and the f32 atomics of the weight gradients move them from run to run:
    # 2.5e-4 .. 8e-4 against the rounding oracle, while the bar against the PLAIN oracle above is the 1e-3 that matters)
    wq = compare_losses(p[0], oq[0], 3e-3 if h.magp else 5e-4, 1e-4)
    lg = rel_err(_logits(h, PG, PD, batches[0]), o[0]["logit_real"])
    assert lg <= 1e-3, lg                                                              # measured 4.4e-4 (synth) / 6.6e-4 (ref)
    # gradients: against the half-rounding oracle (same storage points): single tensors 0.1, all tensors of a backward as one vector
    # D 1e-2, G 3e-2 (measured 1.3e-2; x2 behind the penalty step)
    gd = compare_grads(tapD.records[0], oq[0]["grads_D"], 0.1, "f16 D ", 2e-2, 1e-2)
    gg = compare_grads(tapG.records[0], oq[0]["grads_G"], 0.1, "f16 G ", 2e-2, 6e-2 if h.magp else 3e-2)
    print(f"\n[f16 {yml} {kind}] losses vs f32 oracle {wl:.2e}, vs half-rounding oracle {wq:.2e}; logits vs f32 {lg:.2e}; "
          f"grads vs half-rounding oracle D {gd:.2e} G {gg:.2e}")


@pytest.mark.parametrize("kind", ["synth", "ref"])
def test_bf16_mode_loss_error_is_the_format_not_the_kernels(kind):
    """the benched bf16 mode on the same inputs: inside 1e-2 of the bf16-rounding oracle (kernels right), NOT inside 1e-3 of the
    f32 oracle (the format's floor; recorded, with the ladder) -- the figure bench.py prints as `parity`."""
    ops.set_precision("bf16")
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml")
    PG, PD = _params(h, kind, 0)
    batches = [X.synth_batch(h, 8, seed=200, words_len=cfg.TEXT.MAX_LENGTH)]
    _, _, o = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
    _, _, oq = run_oracle_steps(h, PG, PD, batches, eps=1e-3, quant=True)
    _, _, p, _, _ = run_product_steps(h, PG, PD, batches, eps=1e-3)
    wq = compare_losses(p[0], oq[0], 1e-2, 2e-3)
    wl = compare_losses(p[0], o[0], 5e-2, 1e-2)
    print(f"\n[bf16 {kind}] losses vs f32 oracle {wl:.2e} (format floor), vs bf16-rounding oracle {wq:.2e} (kernels)")


def _logits_on(netG, netD, b, img):
    with torch.no_grad():
        ps = netG.proj_sent(b["sent_embs"].to(DEV))
        return netD.COND_DNET(netD(img.to(DEV)), sent_embs=ps)[0].float().cpu()


# (mode, bar on the losses, bar on the logit vectors, bar on the losses against the ROUNDING oracle = kernel error proper) at the
# benched image size; the first two against the PLAIN f32 oracle.
# Measured (round 4, parameters of seed 5/6): fp32 4.8e-7 / 1.8e-6; f16 losses 6.2e-4, logits 2.8e-4 (real) / 1.25e-3 (generated),
# kernel error 1.9e-4; bf16 6.3e-3, 3.5e-3 / 1.4e-2, kernel error 5.8e-4.  bench.py's own leg (parameters of seed 1/2) has f16 at
# 2.2e-4 / 5.1e-4 / 5.7e-4 -- inside the bar on every count -- and bf16 at 2.2e-3 / 2.7e-3 / 3.8e-3: at this depth the half mode sits
# AT the bar on the generated-image logit vector (seed-dependent, 0.6 .. 1.3e-3) and inside it on everything else.
# (with the composed discriminator stem the same seed reads losses 1.4e-3 -- errD_fake, on the generated images of an untrained
# generator -- while bench.py's seed stays at 2.3e-4 / 5.8e-4 / 6.3e-4: the half mode's loss error at this depth is 2e-4 .. 1.4e-3
# by seed, i.e. ON the 1e-3 bar, not safely inside it; fp32 is the mode that is.)
# Round 5: the half mode runs the discriminator's PRECISE TRUNK (ops.precise_trunk): its bars are north_star's 1e-3 on every
# count (measured: see test_f16_logits_inside_the_bar_over_seeds below).
FULLSIZE_BARS = {"fp32": (1e-3, 1e-3, None), "f16": (1e-3, 1e-3, 1e-3), "bf16": (2e-2, 4e-2, 2e-3)}


@pytest.mark.parametrize("size,kind", [(64, "ref"), (256, "ref"), (64, "synth"), (256, "synth")])
def test_f16_logits_inside_the_bar_over_seeds(size, kind):
    """The figure `bench.py` prints as `parity.logit_rel`, over FIVE parameter / batch seeds instead of one: the logit vectors of
    D + COND_DNET in the IEEE-half mode on the real images and on the ORACLE's generated images against the f32 CPU oracle,
    NCH = 32, batch 8, at 64 and at 256 pixels.  North_star's bar (1e-3 relative) with the precise trunk: every seed inside it.
    The per-layer ladder (tests/diag/layer_ladder.py, profiles/r05_layer_ladder_*.txt) predicts rms 0.9e-4 .. 3.6e-4 / max 6.3e-4 on
    the reference's initialisation and max 8.7e-4 on the synthetic worst-case parameters (block gammas 0.25 .. 0.75, where the
    residual branches' own rounding -- not on the precise path -- carries 2.6e-4 .. 4e-4)."""
    ops.set_precision("f16")
    assert ops.precise_trunk()
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": size})
    worst, sq = [0.0, 0.0], [0.0, 0.0]
    for seed in range(5):
        PG, PD = _params(h, kind, seed)
        b = X.synth_batch(h, 8, seed=300 + seed, words_len=cfg.TEXT.MAX_LENGTH)
        with torch.no_grad():
            fake = X.gen_forward(PG, h, b["noise"], b["sent_embs"], words_embs=b["words_embs"], mask=b["mask"])
            ps = X.proj_sent(PG, b["sent_embs"])
            ref = [X.cond_dnet(PD, h, X.netd_forward(PD, h, im), ps)[0].flatten() for im in (b["imgs"], fake)]
        netG, netD, _, _ = build_product(h, PG, PD)
        for j, im in enumerate((b["imgs"], fake)):
            e = rel_err(_logits_on(netG, netD, b, im).flatten(), ref[j])
            worst[j], sq[j] = max(worst[j], e), sq[j] + e * e
        del netG, netD
    print(f"\n[f16 precise trunk {size}x{size} b8 NCH32 {kind}, 5 seeds] logit vectors vs the f32 oracle: real rms {(sq[0] / 5) ** 0.5:.2e} "
          f"max {worst[0]:.2e}; generated rms {(sq[1] / 5) ** 0.5:.2e} max {worst[1]:.2e}")
    assert max(worst) <= 1e-3, worst


def test_precise_trunk_is_what_brings_the_logits_inside():
    """the same measurement with the precise trunk switched off (the f16 mode of rounds 3-4): larger by the factor the ladder
    predicts (2-4x), i.e. the switch -- not a change of seeds -- is what moved the figure"""
    ops.set_precision("f16")
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": 64})
    e = {True: 0.0, False: 0.0}
    for seed in range(3):
        PG, PD = _params(h, "ref", seed)
        b = X.synth_batch(h, 8, seed=300 + seed, words_len=cfg.TEXT.MAX_LENGTH)
        with torch.no_grad():
            ps = X.proj_sent(PG, b["sent_embs"])
            ref = X.cond_dnet(PD, h, X.netd_forward(PD, h, b["imgs"]), ps)[0].flatten()
        for on in (True, False):
            ops.precise_trunk(on)
            try:
                netG, netD, _, _ = build_product(h, PG, PD)
                e[on] += rel_err(_logits_on(netG, netD, b, b["imgs"]).flatten(), ref) ** 2
                del netG, netD
            finally:
                ops.precise_trunk(None)
    on, off = (e[True] / 3) ** 0.5, (e[False] / 3) ** 0.5
    print(f"\n[f16 64x64 ref-init, 3 seeds] real-image logit vector vs f32 oracle: precise trunk {on:.2e}, without {off:.2e}")
    assert on <= 4e-4 and off >= 1.5 * on, (on, off)


@pytest.mark.parametrize("mode", ["fp32", "f16", "bf16"])
def test_parity_at_the_benched_image_size(mode):
    """ONE G+D iteration at 256x256, batch 8, NCH=32 (the benched network; `bench.py`'s `parity` object is this measurement) in each
    precision mode against the f32 CPU oracle: every loss scalar and the relative L2 error of the real / generated logit vectors
    (the discriminator evaluated on the oracle's generated image, so that it is a statement about D on identical inputs)."""
    ops.set_precision(mode)
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": 256})
    PG, PD = _params(h, "ref", 0)
    batches = [X.synth_batch(h, 8, seed=300, words_len=cfg.TEXT.MAX_LENGTH)]
    _, _, o = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
    netG, netD, _, _ = build_product(h, PG, PD)
    lr = rel_err(_logits_on(netG, netD, batches[0], batches[0]["imgs"]), o[0]["logit_real"])
    lf = rel_err(_logits_on(netG, netD, batches[0], o[0]["fake"]), o[0]["logit_fake"])
    del netG, netD
    _, _, p, _, _ = run_product_steps(h, PG, PD, batches, eps=1e-3)
    lbar, gbar, kbar = FULLSIZE_BARS[mode]
    wl = compare_losses(p[0], o[0], lbar, 1e-4)
    wq = None
    if kbar is not None:
        _, _, oq = run_oracle_steps(h, PG, PD, batches, eps=1e-3, quant=True, fmt=torch.float16 if mode == "f16" else torch.bfloat16)
        wq = compare_losses(p[0], oq[0], kbar, 1e-4)
    print(f"\n[{mode} 256x256 b8 NCH32 ref-init] losses vs f32 oracle {wl:.2e}; logits real {lr:.2e} fake-on-oracle-image {lf:.2e}"
          + (f"; losses vs the rounding oracle {wq:.2e}" if wq is not None else ""))
    assert max(lr, lf) <= gbar, (lr, lf)


def test_dynamic_loss_scale_skips_a_step_with_non_finite_gradients():
    """`xmc_adam_step_scaled` (the IEEE-half mode's optimizer step): gradients are read times 1 / scale; one inf / NaN anywhere in
    ANY parameter group skips the whole step on the device -- parameters, moments and step counters bit-identical -- and halves the
    scale; `interval` finite steps in a row double it.  Checked against torch.optim.Adam on the unscaled gradients."""
    from xmc_gan_amd.optim import HipAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, device=DEV)) for n in (1000, 37, 4096 * 5 + 3)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = HipAdam([dict(params=ps[:2]), dict(params=ps[2:], lr=3e-4)], lr=1e-3, betas=(0.0, 0.9))
    topt = torch.optim.Adam([dict(params=ref[:2]), dict(params=ref[2:], lr=3e-4)], lr=1e-3, betas=(0.0, 0.9))
    sc = ops.LossScaler(DEV, init=1024.0, interval=3)
    snap = lambda: [p.detach().clone() for p in ps]
    n_ok = 0
    for it, bad in enumerate([None, 2, None, 0, None, None, None, None]):
        g = [torch.randn_like(p) for p in ps]
        scale = float(sc.sf[0])
        for p, r, gi in zip(ps, ref, g):
            p.grad = gi * scale
            r.grad = gi.clone()
        if bad is not None:
            ps[bad].grad.view(-1)[5] = float("inf") if it % 2 else float("nan")
        before, steps_before = snap(), [int(opt.state[p]["step"]) if opt.state[p] else 0 for p in ps]
        opt.step(scaler=sc)
        st = sc.stats()
        if bad is not None:
            assert st["last_step_skipped"] and st["scale"] == scale * 0.5
            assert all(torch.equal(a, b) for a, b in zip(before, snap()))
            assert [int(opt.state[p]["step"]) for p in ps] == steps_before
            n_ok = 0
        else:
            topt.step()
            n_ok += 1
            assert not st["last_step_skipped"]
            assert st["scale"] == (scale * 2 if n_ok % 3 == 0 else scale), (it, st, scale)
            for p, r in zip(ps, ref):
                assert torch.allclose(p, r, rtol=1e-5, atol=1e-7)
    assert sc.stats()["skipped_steps"] == 2


def test_f16_build_is_a_separate_library_with_the_same_abi():
    from xmc_gan_amd import lib as L
    a, b = L.load("bf16"), L.load("f16")
    assert a is not b and a.xmc_half_format() == 0 and b.xmc_half_format() == 1
    assert a.xmc_abi_version() == b.xmc_abi_version() == L.ABI_VERSION
    ops.set_precision("f16")
    assert ops.act_dtype() == torch.float16 and ops.loss_scale() > 1 and ops.gp_inner_scale() > 1
    assert ops.loss_scaler("D", DEV) is ops.loss_scaler("D", DEV) and ops.loss_scaler("D", DEV) is not ops.loss_scaler("G", DEV)
    with pytest.raises(TypeError):
        ops._code(torch.bfloat16)
    ops.set_precision("bf16")
    assert ops.act_dtype() == torch.bfloat16 and ops.loss_scale() == 1.0 and ops.gp_inner_scale() == 1.0
    assert ops.loss_scaler("D", DEV) is None
