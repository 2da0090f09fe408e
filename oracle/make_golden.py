"""Generate tests/golden/*.npz by running the REFERENCE on CPU (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/make_golden.py [--out tests/golden]

Each fixture holds only data: the seeds/shapes needed to re-synthesise inputs and parameters
(``oracle.xmc_ref.synth_params`` / ``synth_batch`` are deterministic and order-independent) plus
the reference's outputs.  No reference source text is stored.

Fixtures
  fwd_<name>.npz     reference NetG / NetD / COND_DNET forward outputs + state_dict key/shape table
  labels.npz         reference make_labels / sent_loss / img_loss on fixed embeddings (all b_global modes)
  step_<name>.npz    the reference's real train() loop for a few iterations: every loss scalar,
                     per-parameter gradient statistics at each optimizer step (incl. which grads are
                     None), per-parameter statistics after the last step, first fake image.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as RH  # noqa: E402
import xmc_ref as X      # noqa: E402

torch.set_num_threads(8)


def stats(t):
    """(sum, abs-sum, first, middle, last) of a tensor as float64."""
    f = t.detach().double().flatten()
    n = f.numel()
    return np.array([f.sum().item(), f.abs().sum().item(), f[0].item(), f[n // 2].item(), f[-1].item()])


def repaired_word_in_netg(cfg, M):
    """The reference's concept_gan.InNetG with the two run-time patches that make it runnable (oracle/xmc_ref.py
    word_in_netg_forward states them): every word-region sampler is re-created BY THE REFERENCE'S OWN CLASS with cond_dim = NEF --
    the channel count the forward feeds it (concept_gan.py:570-573) instead of noise_dim + nef (137, 183) -- and every InConceptBlock
    gets the `upsample` attribute its forward reads (222) from the enclosing block.  Everything else that runs is upstream code."""
    cg = M.concept_gan
    net = cg.InNetG(cfg)
    for blk in net.upblocks:
        if isinstance(blk, cg.ICAttnResBlockUp):
            cb = blk.concept1
            cb.upsample = blk.upsample
            for j in (1, 2):
                setattr(cb, f"concept_sampler{j}", cg.CondConceptSampler(
                    cardinality=cb.cardinality, bottleneck_width=8, state_dim=4, cond_dim=cfg.TRAIN.NEF, normalize=cb.normalize))
    return net


def build_ref(cfg, M, seed):
    """Reference models with deterministic synthetic parameters (strict load pins keys+shapes)."""
    h = X.Hyper.from_cfg(cfg)
    gen_cls = {"DF_GEN": M.df_gan.NetG, "CONCEPT_IN_DF_GEN": M.df_concept_gan.InNetG,
               "CONCEPT_OUT_DF_GEN": M.df_concept_gan.OutNetG,
               # un-wired upstream (its registry entry is commented out, train_gan.py:31,44): the class itself is importable
               "CONCEPT_OUTATTN_GEN": M.concept_gan.OutNetG,
               # broken upstream; runs with two patches applied to the reference's objects
               "CONCEPT_INATTN_GEN": lambda c: repaired_word_in_netg(c, M)}[cfg.GEN.ENCODER_NAME]
    netG = gen_cls(cfg)
    netD = M.df_gan.NetD(cfg, is_disc=True)
    PG = X.synth_params(X.gen_shapes(h), seed)
    PD = X.synth_params(X.netd_shapes(h), seed + 1)
    netG.load_state_dict(PG, strict=True)
    netD.load_state_dict(PD, strict=True)
    return h, netG, netD


def shape_table(sd):
    return np.array([f"{k}:{','.join(map(str, v.shape))}" for k, v in sd.items()])


def golden_forward(name, yml, batch, seed, out_dir, **over):
    cfg = RH.load_cfg(yml, **over)
    M = RH.modules()
    h, netG, netD = build_ref(cfg, M, seed)
    b = X.synth_batch(h, batch, seed=seed + 50, words_len=cfg.TEXT.MAX_LENGTH)
    with torch.no_grad():
        fake = netG(noise=b["noise"], sent_embs=b["sent_embs"], words_embs=b["words_embs"], mask=b["mask"])
        psent = b["sent_embs"] if cfg.DISC.SEPERATE else netG.proj_sent(b["sent_embs"])
        feat = netD(b["imgs"])
        logit, img_emb, txt_emb = netD.COND_DNET(feat, sent_embs=psent)
        feat_f = netD(fake)
    np.savez_compressed(
        os.path.join(out_dir, f"fwd_{name}.npz"),
        yml=yml, over=np.array([f"{k}={v}" for k, v in over.items()]), batch=batch, seed=seed,
        g_keys=shape_table(netG.state_dict()), d_keys=shape_table(netD.state_dict()),
        fake=fake.numpy(), feat=feat.numpy(), feat_fake_stats=stats(feat_f), logit=logit.numpy(),
        img_emb=img_emb.numpy(), txt_emb=txt_emb.numpy(), psent=psent.numpy(),
        # spectral norm: the power-iteration buffers after these calls (each forward call in training mode updates them)
        # BatchNorm generators: running statistics after this one training-mode forward
        g_buf_names=np.array([n for n, _ in netG.named_buffers()]),
        g_buf_after=np.stack([stats(b_) for _, b_ in netG.named_buffers()]) if len(list(netG.named_buffers())) else np.zeros((0, 5)),
        d_buf_names=np.array([n for n, _ in netD.named_buffers()]),
        d_buf_after=np.stack([stats(b_) for _, b_ in netD.named_buffers()]) if len(list(netD.named_buffers())) else np.zeros((0, 5)))
    print(f"fwd_{name}: fake {tuple(fake.shape)} feat {tuple(feat.shape)}")


def golden_labels(out_dir):
    cfg = RH.load_cfg("df_gan_damsm.yml")
    tg = RH.modules().train_gan
    g = torch.Generator().manual_seed(7)
    B, Dm = 12, 48
    base = torch.randn(4, Dm, generator=g)
    sent = base.repeat(3, 1) + 0.35 * torch.randn(B, Dm, generator=g)   # clusters -> some global positives
    a = torch.randn(B, Dm, generator=g)
    b = a * 0.5 + torch.randn(B, Dm, generator=g)
    rec = dict(sent=sent.numpy(), a=a.numpy(), b=b.numpy())
    for tag, bg, sg in (("local", False, 0.0), ("adaptive", True, 0.0), ("smooth", True, 0.5)):
        cfg.TRAIN.SMOOTH.GLOBAL = sg
        labels = tg.make_labels(B, sent, bg)
        rec[f"labels_{tag}"] = labels.numpy()
        rec[f"sent_loss_{tag}"] = tg.sent_loss(a, b, labels, bg).item()
        rec[f"img_loss_{tag}"] = tg.img_loss(a, b, labels, bg).item()
    rec["scores"] = tg.cosine_scores(a, b).numpy()
    np.savez_compressed(os.path.join(out_dir, "labels.npz"), **rec)
    print("labels: ok", {k: v for k, v in rec.items() if "loss" in k})


class _RecAdam(torch.optim.Adam):
    """torch.optim.Adam that records gradient statistics just before each step."""

    def __init__(self, named, log, tag, **kw):
        self._names = [n for n, _ in named]
        self._log, self._tag = log, tag
        super().__init__([p for _, p in named], **kw)

    def step(self, closure=None):
        rec = {}
        for n, p in zip(self._names, self.param_groups[0]["params"]):
            rec[n] = None if p.grad is None else stats(p.grad)
        self._log.append((self._tag, rec))
        return super().step(closure)


def golden_step(name, yml, batch, steps, seed, out_dir, **over):
    cfg = RH.load_cfg(yml, **over)
    cfg.TRAIN.MAX_EPOCH = 1
    cfg.TRAIN.LOG_INTERVAL = 10 ** 9
    cfg.TEXT.TYPE = "SENT"            # skips index_to_sent(train_set.i2w, ...) at train_gan.py:149
    M = RH.modules()
    tg = M.train_gan
    h, netG, netD = build_ref(cfg, M, seed)
    T = cfg.TEXT.MAX_LENGTH
    batches = [X.synth_batch(h, batch, seed=seed + 100 + i, words_len=T) for i in range(steps)]

    # loader yields (imgs, [(caps, cap_lens)], keys); first next(it) is consumed for the fixed batch
    def loader_items():
        return [(b["imgs"], [((i,), torch.full((batch,), T))], None) for i, b in enumerate(batches)]

    class Loader:
        def __iter__(self):
            return iter(loader_items())

        def __len__(self):
            return steps

    def text_encoder(caps, cap_lens):
        b = batches[caps[0]]
        return b["words_embs"], b["sent_embs"], b["mask"]

    log = []
    scal = []
    optG = _RecAdam(list(netG.named_parameters()), log, "G", lr=cfg.TRAIN.OPT.G_LR,
                    betas=(cfg.TRAIN.OPT.G_BETA1, cfg.TRAIN.OPT.G_BETA2))
    optD = _RecAdam(list(netD.named_parameters()), log, "D", lr=cfg.TRAIN.OPT.D_LR,
                    betas=(cfg.TRAIN.OPT.D_BETA1, cfg.TRAIN.OPT.D_BETA2))

    # record every loss the loop computes: the contrastive helpers by wrapping, the scalars
    # that get .backward() by hooking Tensor.backward, the hinge terms by wrapping F.relu
    orig_sent, orig_img = tg.sent_loss, tg.img_loss
    tg.sent_loss = lambda **k: (lambda v: (scal.append(("sent_loss", v.item())), v)[1])(orig_sent(**k))
    tg.img_loss = lambda **k: (lambda v: (scal.append(("img_loss", v.item())), v)[1])(orig_img(**k))
    orig_bwd = torch.Tensor.backward

    def rec_bwd(self, *a, **k):
        scal.append(("backward", self.item()))
        return orig_bwd(self, *a, **k)

    torch.Tensor.backward = rec_bwd
    orig_relu = tg.F.relu
    relu_ns = types.SimpleNamespace(**{k: getattr(tg.F, k) for k in dir(tg.F) if not k.startswith("__")})

    def rec_relu(x, inplace=False):
        y = orig_relu(x, inplace=inplace)
        scal.append(("hinge", y.mean().item()))
        return y

    relu_ns.relu = rec_relu
    tg.F = relu_ns
    fakes = []
    orig_G_fwd = netG.forward
    netG.forward = lambda *a, **k: (lambda y: (fakes.append(y.detach().clone()), y)[1])(orig_G_fwd(*a, **k))

    tmp = tempfile.mkdtemp()
    tg.img_dir = tmp
    tg.args = types.SimpleNamespace(log_type="tb")
    tg.writer = types.SimpleNamespace(add_scalar=lambda *a, **k: None)
    tg.netD = netD
    logger = types.SimpleNamespace(info=lambda *a, **k: None)

    torch.manual_seed(seed)
    noise_rng_check = torch.get_rng_state()
    try:
        tg.train(train_loader=Loader(), test_loader=None, state_epoch=0, text_encoder=text_encoder,
                 netG=netG, netD=netD, optimizerG=optG, optimizerD=optD, logger=logger, model_dir=tmp)
    finally:
        torch.Tensor.backward = orig_bwd
        tg.sent_loss, tg.img_loss = orig_sent, orig_img
        import torch.nn.functional as realF
        tg.F = realF

    # the noise sequence the loop drew from the global CPU generator (train_gan.py:156,197)
    torch.set_rng_state(noise_rng_check)
    fixed_noise = torch.randn(batch, cfg.TRAIN.NOISE_DIM)
    noises = [torch.randn(batch, cfg.TRAIN.NOISE_DIM) for _ in range(steps)]

    rec = dict(yml=yml, over=np.array([f"{k}={v}" for k, v in over.items()]), batch=batch, steps=steps,
               seed=seed, noises=torch.stack(noises).numpy(), fake0=fakes[0].numpy(),
               fake_last_stats=stats(fakes[steps - 1]),
               scal_names=np.array([s[0] for s in scal]), scal_vals=np.array([s[1] for s in scal]))
    for i, (tag, r) in enumerate(log):
        names = list(r.keys())
        rec[f"opt{i}_tag"] = tag
        rec[f"opt{i}_none"] = np.array([r[n] is None for n in names])
        rec[f"opt{i}_stats"] = np.stack([np.zeros(5) if r[n] is None else r[n] for n in names])
    rec["n_opt"] = len(log)
    rec["g_names"] = np.array([n for n, _ in netG.named_parameters()])
    rec["d_names"] = np.array([n for n, _ in netD.named_parameters()])
    rec["g_final"] = np.stack([stats(p) for _, p in netG.named_parameters()])
    rec["d_final"] = np.stack([stats(p) for _, p in netD.named_parameters()])
    rec["g_buf_names"] = np.array([n for n, _ in netG.named_buffers()])
    rec["g_buf_final"] = np.stack([stats(b_) for _, b_ in netG.named_buffers()]) if len(list(netG.named_buffers())) else np.zeros((0, 5))
    rec["d_buf_names"] = np.array([n for n, _ in netD.named_buffers()])
    rec["d_buf_final"] = np.stack([stats(b_) for _, b_ in netD.named_buffers()]) if len(list(netD.named_buffers())) else np.zeros((0, 5))
    np.savez_compressed(os.path.join(out_dir, f"step_{name}.npz"), **rec)
    print(f"step_{name}: {len(log)} optimizer steps, scalars:",
          [(n, round(v, 4)) for n, v in scal[: 12]])


def golden_rnn_encoder(name, yml, batch, seed, out_dir, **over):
    """The reference's frozen RNN_ENCODER (encoder.py:73-153) in eval mode on synthetic parameters and captions."""
    cfg = RH.load_cfg(yml, **over)
    M = RH.modules()
    enc = M.encoder.RNN_ENCODER(cfg)
    shapes = X.rnn_encoder_shapes(cfg.TEXT.VOCA_SIZE, cfg.TEXT.EMBEDDING_DIM, rnn_type=cfg.TEXT.RNN_TYPE)
    enc.load_state_dict(X.synth_rnn_params(shapes, seed), strict=True)
    enc.eval()
    caps, lens = X.synth_captions(batch, cfg.TEXT.MAX_LENGTH, cfg.TEXT.VOCA_SIZE, seed + 1)
    with torch.no_grad():
        words, sent, mask = enc(caps, lens)
    np.savez_compressed(os.path.join(out_dir, f"rnn_{name}.npz"), yml=yml, batch=batch, seed=seed,
                        over=np.array([f"{k}={v}" for k, v in over.items()]),
                        keys=shape_table(enc.state_dict()), caps=caps.numpy(), lens=lens.numpy(),
                        words=words.numpy(), sent=sent.numpy(), mask=mask.numpy())
    print(f"rnn_{name}: words {tuple(words.shape)} sent {tuple(sent.shape)}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--only", default="", help="regenerate only the fixtures whose name contains this substring")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.only:                                    # filter: wrap the generators
        g = globals()
        for fn in ("golden_forward", "golden_step", "golden_rnn_encoder"):
            g[fn] = (lambda f: (lambda name, *args, **kw: f(name, *args, **kw) if a.only in name else None))(g[fn])
        g["golden_labels"] = (lambda f: (lambda *args, **kw: f(*args, **kw) if a.only in "labels" else None))(g["golden_labels"])
    N8 = {"TRAIN.NCH": 8}
    golden_labels(a.out)
    golden_forward("df64_nch32", "df_gan_damsm.yml", 2, 11, a.out)
    golden_forward("df128_nch8", "df_gan_damsm.yml", 2, 12, a.out, **{"IMG.SIZE": 128, **N8})
    golden_forward("df256_nch8", "df_gan_sbert_damsm_nomagp.yml", 2, 13, a.out, **{"IMG.SIZE": 256, **N8})
    golden_forward("sep64_nch8", "df_gan_sbert_seperate.yml", 2, 14, a.out, **N8)
    # df_gan_sbert.yml / concept_in_df_gan_sbert.yml cannot run in the reference itself: with E=768 != NEF and
    # neither IMG_MATCH/SENT_MATCH/SEPERATE, D_GET_LOGITS sizes joint_conv for text_dim (df_gan.py:152-157)
    # while train() feeds it netG.proj_sent(sent) of width NEF (train_gan.py:191) -> channel mismatch.
    golden_forward("nomatch64_nch8", "concept_in_df_gan.yml", 2, 15, a.out, **N8)
    golden_forward("cin64_nch8", "concept_in_df_gan_damsm_nomagp.yml", 2, 16, a.out, **N8)
    golden_forward("cin128_nch8", "concept_in_df_gan_sbert_n2_damsm.yml", 1, 17, a.out, **{"IMG.SIZE": 128, **N8})
    golden_forward("cout64_nch8", "concept_out_df_gan_sbert_damsm_nomagp.yml", 2, 18, a.out, **N8)
    golden_forward("cout64_nonorm", "concept_out_df_gan_sbert_damsm_nomagp.yml", 2, 19, a.out,
                   **{"GEN.NORMALIZE": False, **N8})
    golden_step("df64_magp", "df_gan_damsm.yml", 4, 2, 21, a.out, **N8)
    golden_step("df64_nomagp", "df_gan_damsm_nomagp.yml", 4, 2, 22, a.out, **N8)
    golden_step("df64_sep", "df_gan_sbert_seperate.yml", 3, 2, 23, a.out, **N8)
    golden_step("df64_ncrit2", "concept_in_df_gan_sbert_n2_damsm.yml", 3, 2, 24, a.out, **N8)
    golden_step("cout64", "concept_out_df_gan_sbert_damsm_nomagp.yml", 3, 1, 25, a.out, **N8)
    golden_step("df128_nomagp", "df_gan_damsm_nomagp.yml", 2, 1, 26, a.out, **{"IMG.SIZE": 128, **N8})
    # DISC.SPEC_NORM: True (config/gan.py:62 default): weight_orig / weight_u / weight_v keys, one power
    # iteration per forward call
    SN = {"DISC.SPEC_NORM": True, **N8}
    golden_forward("sn64_nch8", "df_gan_damsm.yml", 2, 31, a.out, **SN)
    golden_step("sn64_nomagp", "df_gan_damsm_nomagp.yml", 4, 2, 32, a.out, **SN)
    golden_step("sn64_magp", "df_gan_damsm.yml", 3, 1, 33, a.out, **SN)
    # word-attention generator concept_gan.OutNetG: BatchNorm blocks + word<->concept attention with mask
    WG = {"GEN.ENCODER_NAME": "CONCEPT_OUTATTN_GEN", **N8}
    golden_forward("wordg64_nch8", "df_gan_damsm_nomagp.yml", 3, 51, a.out, **WG)
    golden_forward("wordg128_nonorm", "df_gan_sbert_damsm_nomagp.yml", 2, 52, a.out, **{"IMG.SIZE": 128, "GEN.NORMALIZE": False, **WG})
    golden_step("wordg64", "df_gan_damsm_nomagp.yml", 4, 2, 53, a.out, **WG)
    # the word-REGION attention generator concept_gan.InNetG, repaired: repaired_word_in_netg above
    WI = {"GEN.ENCODER_NAME": "CONCEPT_INATTN_GEN", **N8}
    golden_forward("wordin64_nch8", "df_gan_damsm_nomagp.yml", 3, 61, a.out, **WI)
    golden_forward("wordin128_nonorm", "df_gan_sbert_damsm_nomagp.yml", 2, 62, a.out, **{"IMG.SIZE": 128, "GEN.NORMALIZE": False, **WI})
    golden_step("wordin64", "df_gan_damsm_nomagp.yml", 4, 2, 63, a.out, **WI)
    # frozen text front end: embedding + bidirectional LSTM over packed captions
    golden_rnn_encoder("enc_damsm", "df_gan_damsm.yml", 6, 41, a.out)
    golden_rnn_encoder("enc_len12", "df_gan_damsm.yml", 5, 42, a.out, **{"TEXT.MAX_LENGTH": 12, "TEXT.VOCA_SIZE": 500})
    # the GRU branch of the same encoder (encoder.py:99-102; no shipped preset selects it)
    golden_rnn_encoder("enc_gru", "df_gan_damsm.yml", 5, 43, a.out, **{"TEXT.RNN_TYPE": "GRU", "TEXT.VOCA_SIZE": 400})


if __name__ == "__main__":
    main()
