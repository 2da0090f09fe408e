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
