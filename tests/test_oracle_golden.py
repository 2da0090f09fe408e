"""Pin the CPU oracle (oracle/xmc_ref.py) to the reference's own outputs (tests/golden/*.npz,
produced by oracle/make_golden.py from the reference running on CPU)."""
import numpy as np
import pytest
import torch

import xmc_ref as X
from golden_util import fixtures, hyper_for, load, stats

torch.set_num_threads(4)


def _close(a, b, rtol=2e-4, atol=2e-5):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", fixtures("fwd_"))
def test_forward_matches_reference(name):
    fx = load(name)
    h, cfg = hyper_for(fx)
    seed, batch = int(fx["seed"]), int(fx["batch"])
    # state_dict key/shape parity
    tab = lambda s: sorted(f"{k}:{','.join(map(str, v))}" for k, v in s.items())
    assert tab(X.gen_shapes(h)) == sorted(map(str, fx["g_keys"]))
    assert tab(X.netd_shapes(h)) == sorted(map(str, fx["d_keys"]))
    PG = X.synth_params(X.gen_shapes(h), seed)
    PD = X.synth_params(X.netd_shapes(h), seed + 1)
    b = X.synth_batch(h, batch, seed=seed + 50, words_len=cfg["TEXT"]["MAX_LENGTH"])
    with torch.no_grad():
        fake = X.gen_forward(PG, h, b["noise"], b["sent_embs"], words_embs=b["words_embs"], mask=b["mask"])
        psent = b["sent_embs"] if h.seperate else X.proj_sent(PG, b["sent_embs"])
        feat = X.netd_forward(PD, h, b["imgs"])
        logit, img_emb, txt_emb = X.cond_dnet(PD, h, feat, psent)
        feat_f = X.netd_forward(PD, h, fake)
    _close(fake, fx["fake"])
    _close(psent, fx["psent"])
    _close(feat, fx["feat"], atol=2e-4)
    _close(logit, fx["logit"], atol=2e-4)
    _close(img_emb, fx["img_emb"], atol=2e-4)
    _close(txt_emb, fx["txt_emb"])
    _close(stats(feat_f), fx["feat_fake_stats"], rtol=1e-3, atol=1e-3)
    if "d_buf_names" in fx.files:                 # spectral norm: power-iteration buffers after the same sequence of forward calls
        for j, n in enumerate(fx["d_buf_names"]):
            _close(stats(PD[str(n)]), fx["d_buf_after"][j], rtol=1e-3, atol=1e-4)
    if "g_buf_names" in fx.files:                 # BatchNorm generators: running statistics after one training-mode forward
        assert len(fx["g_buf_names"]) == sum(k.endswith((".running_mean", ".running_var", ".num_batches_tracked", ".norm"))
                                              for k in PG)
        for j, n in enumerate(fx["g_buf_names"]):
            _close(stats(PG[str(n)]), fx["g_buf_after"][j], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", fixtures("rnn_"))
def test_rnn_encoder_matches_reference(name):
    """frozen text front end (encoder.py:73-153): state_dict layout and outputs of the reference's RNN_ENCODER."""
    fx = load(name)
    over = dict(kv.split("=") for kv in map(str, fx["over"]))
    T = int(over.get("TEXT.MAX_LENGTH", 20))
    V = int(over.get("TEXT.VOCA_SIZE", 27297))
    shapes = X.rnn_encoder_shapes(V, 256, rnn_type=over.get("TEXT.RNN_TYPE", "LSTM"))
    tab = lambda s: sorted(f"{k}:{','.join(map(str, v))}" for k, v in s.items())
    assert tab(shapes) == sorted(map(str, fx["keys"]))
    P = X.synth_rnn_params(shapes, int(fx["seed"]))
    caps, lens = X.synth_captions(int(fx["batch"]), T, V, int(fx["seed"]) + 1)
    assert np.array_equal(caps.numpy(), fx["caps"]) and np.array_equal(lens.numpy(), fx["lens"])
    assert int(lens[0]) == T and int(lens[1]) == 1                    # full-length and single-token captions are covered
    words, sent, mask = X.rnn_encoder(P, caps, lens, T)
    assert np.array_equal(mask.numpy(), fx["mask"])
    _close(words, fx["words"], rtol=1e-5, atol=1e-6)
    _close(sent, fx["sent"], rtol=1e-5, atol=1e-6)
    assert (fx["words"][1, :, 1:] == 0).all()                        # padded positions are exactly zero


def test_labels_and_contrastive_losses():
    fx = load("labels.npz")
    sent, a, b = (torch.from_numpy(fx[k]) for k in ("sent", "a", "b"))
    _close(X.cosine_scores(a, b), fx["scores"], atol=1e-6)
    for tag, bg, sg in (("local", False, 0.0), ("adaptive", True, 0.0), ("smooth", True, 0.5)):
        labels = X.make_labels(a.size(0), sent, bg, sg)
        # the 0/1/weight pattern comes from exact comparisons -> must be identical
        assert np.array_equal(labels.numpy(), fx[f"labels_{tag}"]), tag
        if bg:
            assert (labels.numpy() != np.eye(a.size(0), dtype=np.float32)).any()
        for nm in ("sent_loss", "img_loss"):
            _close(X.contrastive_loss(a, b, labels, bg, sg).item(), fx[f"{nm}_{tag}"].item(), rtol=1e-6, atol=1e-6)


def _expected_scalars(h, out, it):
    seq = [("hinge", out["errD_real"]), ("hinge", out["errD_fake"])]
    if h.rmis:
        seq.append(("hinge", out["errD_mismatch"]))
    if h.enc_sent:
        seq.append(("sent_loss", out["ds_loss"]))
    seq.append(("backward", out["errD"]))
    if h.magp:
        seq.append(("backward", 2.0 * out["d_loss_gp"]))
    if it % h.n_critic == 0:
        if h.enc_sent:
            seq.append(("sent_loss", out["gs_loss"]))
        if h.enc_disc:
            seq.append(("img_loss", out["disc_loss"]))
        seq.append(("backward", out["errG"]))
    return seq


@pytest.mark.parametrize("name", fixtures("step_"))
def test_train_step_matches_reference_loop(name):
    fx = load(name)
    h, cfg = hyper_for(fx)
    seed, batch, steps = int(fx["seed"]), int(fx["batch"]), int(fx["steps"])
    PG = X.synth_params(X.gen_shapes(h), seed)
    PD = X.synth_params(X.netd_shapes(h), seed + 1)
    optG = X.AdamState(h.g_lr, h.g_betas)
    optD = X.AdamState(h.d_lr, h.d_betas)
    g_names = [str(s) for s in fx["g_names"]]
    d_names = [str(s) for s in fx["d_names"]]
    seq, opt_i, it = [], 0, 0
    for s in range(steps):
        b = X.synth_batch(h, batch, seed=seed + 100 + s, words_len=cfg["TEXT"]["MAX_LENGTH"])
        b["noise"] = torch.from_numpy(fx["noises"][s])
        it += 1
        out = X.train_step(PG, PD, optG, optD, h, b, it_count=it)
        seq += _expected_scalars(h, out, it)
        if s == 0:
            _close(out["fake"], fx["fake0"])
        if s == steps - 1:
            _close(stats(out["fake"]), fx["fake_last_stats"], rtol=2e-3, atol=2e-3)
        did_g = it % h.n_critic == 0
        for tag, key, names in (("D", "grads_D", d_names), ("D", "grads_GP", d_names), ("G", "grads_G", g_names)):
            if key not in out or (key == "grads_G" and not did_g):
                continue
            assert str(fx[f"opt{opt_i}_tag"]) == tag
            none_ref = fx[f"opt{opt_i}_none"]
            st_ref = fx[f"opt{opt_i}_stats"]
            for j, n in enumerate(names):
                g = out[key].get(n)
                assert (g is None) == bool(none_ref[j]), f"{name} opt{opt_i} {n}: None-ness differs"
                if g is not None:
                    _close(stats(g), st_ref[j], rtol=5e-3, atol=1e-4 * max(1.0, abs(st_ref[j][1])))
            opt_i += 1
        if did_g:
            it = 0
    assert opt_i == int(fx["n_opt"])
    assert [n for n, _ in seq] == [str(s) for s in fx["scal_names"]]
    _close([v for _, v in seq], fx["scal_vals"], rtol=2e-4, atol=2e-4)
    # parameters after the last optimizer step (Adam with beta1=0 moves each weight by ~lr*sign(g),
    # so compare with an lr-scaled tolerance)
    for names, P, ref, lr in ((g_names, PG, fx["g_final"], h.g_lr), (d_names, PD, fx["d_final"], h.d_lr)):
        for j, n in enumerate(names):
            st = stats(P[n])
            tol = 4 * steps * lr
            assert abs(st[0] - ref[j][0]) <= tol * P[n].numel() * 0.02 + 1e-5, n
            assert np.allclose(st[2:], ref[j][2:], atol=tol), n
    if "d_buf_names" in fx.files:                 # spectral norm: u / v after every forward call of the loop
        for j, n in enumerate(fx["d_buf_names"]):
            _close(stats(PD[str(n)]), fx["d_buf_final"][j], rtol=5e-3, atol=2e-3)
    if "g_buf_names" in fx.files:                 # BatchNorm generators: running statistics after the loop
        for j, n in enumerate(fx["g_buf_names"]):
            _close(stats(PG[str(n)]), fx["g_buf_final"][j], rtol=5e-3, atol=2e-3)


@pytest.mark.parametrize("gen", ["DF_GEN", "CONCEPT_IN_DF_GEN", "CONCEPT_OUT_DF_GEN"])
def test_rounding_mode_is_off_by_default_and_visits_every_site(gen):
    """`X.quant` (the oracle's 16-bit storage mode, used only by the GPU parity tests): off, a forward is the plain f32 restatement the
    goldens pin; on, every rounding site tagged for the generator family is reached and each value it returns is a bf16 number."""
    h = X.Hyper(img_size=64, nch=8, gen=gen)
    PG = X.synth_params(X.gen_shapes(h), 5)
    b = X.synth_batch(h, 2, seed=1)
    seen, q0 = {}, X.q

    def rec(x, site=None, *a, **k):
        y = q0(x, site, *a, **k)
        seen.setdefault(site, []).append(y.detach())
        return y

    with torch.no_grad():
        plain = X.gen_forward(PG, h, b["noise"], b["sent_embs"])
        X.q = X.q_ = rec
        try:
            again = X.gen_forward(PG, h, b["noise"], b["sent_embs"])           # hooks in place, mode off: identity
            assert torch.equal(plain, again)
            seen.clear()
            with X.quant(True):
                rounded = X.gen_forward(PG, h, b["noise"], b["sent_embs"])
        finally:
            X.q = X.q_ = q0
    assert not torch.equal(plain, rounded) and (plain - rounded).abs().mean() < 2e-2
    # (DF_GEN's restatement folds the tail LeakyReLU into the last block's stored tensor, as the engine does: no g.act there)
    want = {"g.stem", "g.sum", "g.img"} | ({"g.aff", "g.c1", "g.c2", "g.sc"} if gen == "DF_GEN" else
                                            {s for s in X.QUANT_SITES if s.startswith("g.c.")} | {"g.sc", "g.act"})
    assert want <= set(seen), want - set(seen)
    for site, vals in seen.items():
        for v in vals:
            assert torch.equal(v, v.to(torch.bfloat16).float()), site
