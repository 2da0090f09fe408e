"""BASELINE config 4 at its full size (256x256, 256 images, NCH=32) through a size-independent property: neither network
couples samples of a batch (no BatchNorm on the DF path), so the first samples of a full-batch forward must equal the forward
of just those samples -- which in turn is pinned to the CPU oracle.  Exercises the launch geometry of every forward kernel at
16.7 M output pixels (grid sizes, persistent tile loops, 32-bit index headroom) where the oracle itself would take hours."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import xmc_ref as X
    from xmc_gan_amd import ops
    from parity_util import DEV, build_product, mean_abs_err, rel_err, setup_cfg


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_batch_forward_equals_small_batch_forward_equals_oracle(mode):
    ops.set_precision(mode)
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": 256})
    assert h.nch == 32
    B, k = 256, 2
    PG, PD = X.synth_params(X.gen_shapes(h), 3), X.synth_params(X.netd_shapes(h), 4)
    b = X.synth_batch(h, B, seed=77, words_len=cfg.TEXT.MAX_LENGTH)
    netG, netD, _, _ = build_product(h, PG, PD)
    dev = {key: v.to(DEV) for key, v in b.items()}
    with torch.no_grad():
        fake = netG(noise=dev["noise"], sent_embs=dev["sent_embs"], words_embs=dev["words_embs"], mask=dev["mask"])
        feat = netD(dev["imgs"])
        ps = netG.proj_sent(dev["sent_embs"])
        logit = netD.COND_DNET(feat, sent_embs=ps)[0]
        fake_k = netG(noise=dev["noise"][:k], sent_embs=dev["sent_embs"][:k], words_embs=dev["words_embs"][:k], mask=dev["mask"][:k])
        feat_k = netD(dev["imgs"][:k])
        logit_k = netD.COND_DNET(feat_k, sent_embs=ps[:k])[0]
        # a slice from the far end of the batch: the last workgroups of every launch
        feat_end = netD(dev["imgs"][B - k:])
    assert fake.shape == (B, 3, 256, 256) and feat.shape == (B, 16 * 32, 4, 4) and torch.isfinite(fake).all()
    tight = 1e-5 if mode == "fp32" else 2e-2          # different batch sizes may take different kernels (other summation order)
    assert mean_abs_err(fake[:k], fake_k) < tight and rel_err(feat[:k], feat_k) < tight and rel_err(logit[:k], logit_k) < 2 * tight
    assert rel_err(feat[B - k:], feat_end) < tight
    # ... and those k samples against the CPU oracle (a few seconds at this size)
    with torch.no_grad():
        fake_o = X.gen_forward(PG, h, b["noise"][:k], b["sent_embs"][:k])
        feat_o = X.netd_forward(PD, h, b["imgs"][:k])
        logit_o = X.cond_dnet(PD, h, feat_o, X.proj_sent(PG, b["sent_embs"][:k]))[0]
    t = 1e-3 if mode == "fp32" else 3e-2
    assert mean_abs_err(fake[:k], fake_o) < t, mean_abs_err(fake[:k], fake_o)
    assert rel_err(feat[:k], feat_o) < t, rel_err(feat[:k], feat_o)
    assert rel_err(logit[:k], logit_o) < 2 * t, rel_err(logit[:k], logit_o)


def test_full_size_iteration_runs_and_is_reproducible():
    """one complete G+D iteration at the bench configuration, twice from the same state: finite losses that repeat.  The only
    run-to-run freedom is the order of f32 atomics in the weight-gradient kernels: the D-step losses do not see it at all, the
    G-step losses see it through the D update they follow (last-bit differences in D's weights: 1e-3 relative allowed,
    2e-5 measured)."""
    import xmc_gan.train_gan as tg
    ops.set_precision("bf16")
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": 256})
    PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
    b = {key: v.to(DEV) for key, v in X.synth_batch(h, 256, seed=9, words_len=cfg.TEXT.MAX_LENGTH).items()}
    outs = []
    for _ in range(2):
        models = build_product(h, PG, PD)
        o = tg.gan_iteration(*models, b["imgs"], b["sent_embs"], b["words_embs"], b["mask"], b["noise"], {})
        outs.append({key: float(v) for key, v in o.items() if torch.is_tensor(v) and v.numel() == 1})
    assert {"errD", "errG", "ds_loss", "gs_loss", "disc_loss"} <= set(outs[0])
    for key, v in outs[0].items():
        assert v == v and abs(v) < 1e6, key
        g_step = key in ("errG", "errG_fake", "gs_loss", "disc_loss")
        assert abs(v - outs[1][key]) <= (1e-3 if g_step else 1e-6) * abs(v) + 1e-6, (key, v, outs[1][key])


def _d_backward(netD, netG, imgs, sent, r_feat):
    """D forward + backward under a loss LINEAR in the feature map (sum(feat * r) + sum(logit)): returns (d loss / d image,
    {name: weight gradient}) -- linear so that gradients of batch slices add up exactly."""
    x = imgs.clone().requires_grad_()
    feat = netD(x)
    logit = netD.COND_DNET(feat, sent_embs=sent)[0]
    loss = (feat.float() * r_feat).sum() + logit.float().sum()
    netD.zero_grad()
    loss.backward()
    return x.grad.detach().clone(), {n: p.grad.detach().clone() for n, p in netD.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("size,batch,mode", [(256, 256, "bf16"), (64, 64, "bf16"), (64, 64, "fp32")])
def test_full_size_backward_equals_sum_of_batch_slices(size, batch, mode):
    """BASELINE config 4 (256 px, 256 images) and config 2 (64 px, 64 images) at their REAL batch through the discriminator's
    backward pass, by a size-independent property: the network does not couple samples, so under a loss linear in its outputs
      * d loss / d image of a sample slice of the full-batch backward == the backward of just that slice, and
      * every weight gradient of the full batch == the sum of the weight gradients of its two halves.
    The full batch takes the large-launch kernels (>= 65 536-pixel tile thresholds, streaming 1x1 kernels, persistent tile
    loops over all 256 CUs); the slices are the sizes the oracle-pinned iteration tests run at."""
    ops.set_precision(mode)
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": size})
    assert h.nch == 32
    PG, PD = X.synth_params(X.gen_shapes(h), 3), X.synth_params(X.netd_shapes(h), 4)
    b = X.synth_batch(h, batch, seed=31, words_len=cfg.TEXT.MAX_LENGTH)
    netG, netD, _, _ = build_product(h, PG, PD)
    imgs = b["imgs"].to(DEV)
    with torch.no_grad():
        sent = netG.proj_sent(b["sent_embs"].to(DEV))
    g = torch.Generator().manual_seed(5)
    r = torch.randn(batch, 16 * 32, 4, 4, generator=g).to(DEV)
    dx, dw = _d_backward(netD, netG, imgs, sent, r)
    hb = batch // 2
    dx0, dw0 = _d_backward(netD, netG, imgs[:hb], sent[:hb], r[:hb])
    dx1, dw1 = _d_backward(netD, netG, imgs[hb:], sent[hb:], r[hb:])
    k = 8 if size == 64 else 2
    dxe, _ = _d_backward(netD, netG, imgs[batch - k:], sent[batch - k:], r[batch - k:])
    # data gradients: different batch sizes may take different kernels.  At 256 px the full batch runs the 8x8 maps on conv_wtile3's
    # multi-image tiles and the slices on the gather kernel: operator by operator they agree to 1e-5 .. 1e-4 (f32 summation order,
    # tests/test_ops_gpu.py::test_multi_image_tiles_equal_the_small_batch_kernels), which moves ~1e-4 of the later pre-activations
    # across LeakyReLU's kink; each flipped mask element changes its gradient by 0.8 |g|, so the image gradient differs by
    # ~sqrt(1e-4) = 1e-2 in relative L2 (measured 1.2e-2 .. 1.5e-2).  Where both sides take the same kernels the difference is 0.
    t = 1e-5 if mode == "fp32" else 3e-2
    print(f"\n[{size}px b{batch} {mode}] halves {rel_err(dx[:hb], dx0):.2e} {rel_err(dx[hb:], dx1):.2e}  last {k}: {rel_err(dx[batch - k:], dxe):.2e}")
    assert rel_err(dx[:hb], dx0) < t and rel_err(dx[hb:], dx1) < t, (rel_err(dx[:hb], dx0), rel_err(dx[hb:], dx1))
    assert rel_err(dx[batch - k:], dxe) < t, rel_err(dx[batch - k:], dxe)
    assert set(dw) == set(dw0) == set(dw1)
    worst = 0.0
    for n in dw:
        e = rel_err(dw[n], dw0[n] + dw1[n])
        worst = max(worst, e)
        # f32 accumulation of identical bf16 / f32 products in another order (atomics): 1e-4; bf16 mode re-rounds the
        # intermediate gradient tensors when the batch changes kernels: 1e-2
        assert e < (1e-4 if mode == "fp32" else 3e-2), (n, e)
    print(f"\n[{size}px b{batch} {mode}] dgrad slices {rel_err(dx[:hb], dx0):.1e}, wgrad full vs sum of halves worst {worst:.1e}")


@pytest.mark.parametrize("yml,size,batch,nsl,mode", [
    ("df_gan_damsm_nomagp.yml", 64, 64, 8, "fp32"), ("df_gan_damsm_nomagp.yml", 64, 64, 8, "bf16"),                         # config 2
    ("concept_in_df_gan_damsm_nomagp.yml", 128, 64, 8, "fp32"), ("concept_in_df_gan_damsm_nomagp.yml", 128, 64, 8, "bf16"),  # config 3
    ("df_gan_damsm_nomagp.yml", 256, 256, 2, "bf16")])                                                                        # configs 4 / 5
def test_generator_full_batch_equals_slices(yml, size, batch, nsl, mode):
    """The generators of BASELINE configs 2, 3 (word-region attention modulation, 128 px, the real width NCH = 32) and 4 / 5 at their
    REAL batch: forward and weight gradients of the full batch == those of its batch slices (8 slices of 8 samples -- the batch
    size the oracle-pinned iteration tests run at -- or, at 256 images of 256 px, the two halves).  The generators do not couple
    the samples of a batch (GroupNorm is per sample), the loss is linear in the image, and kernel dispatch is size-dependent:
    the full batch takes the persistent / multi-image / large-launch kernels, the slices the small-launch ones."""
    ops.set_precision(mode)
    cfg, h = setup_cfg(yml, **{"IMG.SIZE": size})
    assert h.nch == 32
    PG, PD = X.synth_params(X.gen_shapes(h), 3), X.synth_params(X.netd_shapes(h), 4)
    b = {k_: v.to(DEV) for k_, v in X.synth_batch(h, batch, seed=32, words_len=cfg.TEXT.MAX_LENGTH).items()}
    netG, _, _, _ = build_product(h, PG, PD)
    r = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(6)).to(DEV)

    def run(sl):
        netG.zero_grad()
        ops.new_iteration(DEV)
        img = netG(noise=b["noise"][sl], sent_embs=b["sent_embs"][sl], words_embs=b["words_embs"][sl], mask=b["mask"][sl])
        (img * r[sl]).sum().backward()
        return img.detach().clone(), {n: p.grad.detach().clone() for n, p in netG.named_parameters() if p.grad is not None}

    img, gw = run(slice(0, batch))
    acc = None
    t = 1e-5 if mode == "fp32" else 2e-2
    per = batch // nsl
    worst_img = 0.0
    for i in range(nsl):
        im_i, gw_i = run(slice(per * i, per * (i + 1)))
        worst_img = max(worst_img, mean_abs_err(img[per * i:per * (i + 1)], im_i))
        acc = gw_i if acc is None else {n: acc[n] + gw_i[n] for n in acc}
    assert worst_img < t, worst_img
    assert set(gw) == set(acc)
    # per tensor, relative to its own norm or (tensors that are zero up to rounding: a GroupNorm bias in front of a softmax shifts
    # every logit alike and has NO gradient) to a floor of the largest tensor norm, as compare_grads does
    big = max(acc[n].norm().item() for n in acc)
    fl = 1e-4 if mode == "fp32" else 2e-2
    # (parameters with <= 4 elements -- the block gammas: <dout, residual>, one cancelling sum over every pixel -- on the scale of the largest
    # such gradient, as parity_util.compare_grads does: one of them near zero otherwise turns f32 summation order into 2e-3 .. 8e-3 run to run)
    small = max([acc[n].abs().max().item() for n in acc if acc[n].numel() <= 4] + [0.0])
    errs = {n: ((gw[n] - acc[n]).norm() / max(acc[n].norm().item(), fl * big, small if acc[n].numel() <= 4 else 0.0)).item() for n in gw}
    worst = max(errs, key=errs.get)
    tot = rel_err(torch.cat([gw[n].flatten() for n in gw]), torch.cat([acc[n].flatten() for n in gw]))
    print(f"\n[G {yml} {size}px b{batch} {mode}] image {worst_img:.1e}; weight gradients full vs sum of {nsl} slices: worst {errs[worst]:.1e} "
          f"({worst}), all tensors as one vector {tot:.1e}")
    # fp32: f32 accumulation in another order (atomics, tile order).  The attention-modulation generator's query / key parameters
    # carry a heavily cancelling sum over 16 384 regions (tests/test_models_gpu.py): their f32 run-to-run spread is 1e-3 .. 5e-3
    wbar = (2e-4 if h.gen == "DF_GEN" else 5e-3) if mode == "fp32" else 3e-2        # measured 5e-6 / 5.6e-4 .. 2.4e-3 (run to run: f32 atomics in cancelling sums) / 8e-6
    if mode == "bf16" and h.gen != "DF_GEN":
        wbar = 0.15         # (round 5 ratchet: 0.3 -> 0.15; measured 0.088 .. 0.13 by box, vector 1.5e-2 .. 1.6e-2) per tensor: a gross-error guard for those cancelling sums at 8-bit storage; the vector bound is the check
    assert errs[worst] < wbar, (worst, errs[worst])
    assert tot < (1e-3 if mode == "fp32" else 2.4e-2), tot          # (bf16: 3e-2 -> 2.4e-2, measured 1.6e-2 on config 3, 1.7e-4 on DF_GEN)
