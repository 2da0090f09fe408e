"""Build-container cross-check for bench.py's `cpu_baseline`: the reference's own `train()` and this repo's CPU
restatement (`oracle/xmc_ref.train_step`, `cpu_baseline.kind = "port"`) timed side by side on BASELINE config 1
(64x64, batch 8, df_gan_damsm.yml: MA-GP on), same thread count.  TEST INFRASTRUCTURE; needs /root/reference, so it runs in
the build container only.      python oracle/time_reference_vs_port.py [--steps 6 --threads 8]"""
import argparse
import os
import sys
import tempfile
import time
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as RH  # noqa: E402
import xmc_ref as X      # noqa: E402
import make_golden as MG  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    yml, batch = "df_gan_damsm.yml", 8
    cfg = RH.load_cfg(yml)
    cfg.TRAIN.MAX_EPOCH = 1
    cfg.TRAIN.LOG_INTERVAL = 10 ** 9
    cfg.TEXT.TYPE = "SENT"
    M = RH.modules()
    tg = M.train_gan
    h, netG, netD = MG.build_ref(cfg, M, 7)
    T = cfg.TEXT.MAX_LENGTH
    batches = [X.synth_batch(h, batch, seed=500 + i, words_len=T) for i in range(a.steps + 1)]
    times = []

    class Loader:
        def __iter__(self):
            for i, b in enumerate(batches):
                times.append(time.perf_counter())
                yield (b["imgs"], [((i,), torch.full((batch,), T))], None)

        def __len__(self):
            return len(batches)

    enc = lambda caps, lens: (batches[caps[0]]["words_embs"], batches[caps[0]]["sent_embs"], batches[caps[0]]["mask"])
    optG = torch.optim.Adam(netG.parameters(), lr=cfg.TRAIN.OPT.G_LR, betas=(cfg.TRAIN.OPT.G_BETA1, cfg.TRAIN.OPT.G_BETA2))
    optD = torch.optim.Adam(netD.parameters(), lr=cfg.TRAIN.OPT.D_LR, betas=(cfg.TRAIN.OPT.D_BETA1, cfg.TRAIN.OPT.D_BETA2))
    tmp = tempfile.mkdtemp()
    tg.img_dir, tg.args, tg.netD = tmp, types.SimpleNamespace(log_type="tb"), netD
    tg.writer = types.SimpleNamespace(add_scalar=lambda *a_, **k: None)
    tg.train(train_loader=Loader(), test_loader=None, state_epoch=0, text_encoder=enc, netG=netG, netD=netD,
             optimizerG=optG, optimizerD=optD, logger=types.SimpleNamespace(info=lambda *a_, **k: None), model_dir=tmp)
    times.append(time.perf_counter())
    ref = [t1 - t0 for t0, t1 in zip(times[2:-1], times[3:])]          # skip the first iteration(s): warm-up
    PG, PD = X.synth_params(X.gen_shapes(h), 7), X.synth_params(X.netd_shapes(h), 8)
    oG, oD = X.AdamState(h.g_lr, h.g_betas), X.AdamState(h.d_lr, h.d_betas)
    port = []
    for i, b in enumerate(batches):
        t0 = time.perf_counter()
        X.train_step(PG, PD, oG, oD, h, b)
        if i >= 2:
            port.append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"reference train(): {med(ref):.3f} s/iteration ({batch / med(ref):.2f} img/s); port train_step: {med(port):.3f} s/iteration "
          f"({batch / med(port):.2f} img/s); {a.threads} threads, 64x64, batch {batch}, {yml}")


if __name__ == "__main__":
    main()
