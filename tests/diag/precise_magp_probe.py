"""diagnostic: the MA-GP iteration of the half mode with and without the precise trunk, loss by loss against the f32 oracle, over seeds"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import xmc_ref as X
from parity_util import LOSS_KEYS, run_oracle_steps, run_product_steps, setup_cfg
from xmc_gan_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "ref"
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ops.set_precision("f16")
cfg, h = setup_cfg("df_gan_damsm.yml")
for seed in range(nseeds):
    if kind == "synth":
        PG, PD = X.synth_params(X.gen_shapes(h), 5 + seed), X.synth_params(X.netd_shapes(h), 6 + seed)
    else:
        PG, PD = X.ref_init_params(X.gen_shapes(h), 5 + seed, 0.1), X.ref_init_params(X.netd_shapes(h), 6 + seed, 0.1)
    batches = [X.synth_batch(h, 8, seed=200 + seed, words_len=cfg.TEXT.MAX_LENGTH)]
    PGo, PDo, o = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
    for on in (True, False):
        ops.precise_trunk(on)
        ops.reset_loss_scalers()
        netG, netD, p, tapG, tapD = run_product_steps(h, PG, PD, batches, eps=1e-3)
        sd = {k: v.detach().float().cpu() for k, v in netD.state_dict().items()}
        dw = max(((sd[k] - PDo[k]).abs().max() / 4e-4).item() for k in sd)       # in units of the learning rate
        print(f"seed {seed} precise={on}: " + " ".join(f"{k} {abs(float(p[0][k]) - float(o[0][k])) / abs(float(o[0][k])):.1e}" for k in LOSS_KEYS if k in o[0])
              + f" | final D weights: worst element {dw:.2f} lr from the oracle's", flush=True)
ops.precise_trunk(None)
