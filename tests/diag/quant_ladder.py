"""diagnostic (CPU only, no GPU needed): WHICH stored tensors of the bf16 mode carry the loss / logit error against the f32 oracle?

The oracle's quantisation-aware mode reproduces the engine's bf16 mode bit for bit in >= 99.6 % of the elements
(tests/test_models_gpu.py::test_bf16_blocks_reproduce_quantisation_aware_oracle), so the ladder can be climbed on the CPU:
the first iteration of the G+D step is evaluated in f32 and with bf16 rounding at every storage site except a chosen set
(`X.quant(True, skip=...)`), on two parameterisations:
  * "synth"  -- oracle.synth_params: Kaiming weights with non-zero biases and block gammas 0.25..0.75 (the parity tests' one,
                chosen so that every branch contributes: a worst case for error growth);
  * "ref"    -- the reference's own start of training (weight_init: Kaiming weights, zero biases) with the block gammas at the
                value given (0 in the reference, 0.1 in bench.py).

    python tests/diag/quant_ladder.py [--size 64] [--nch 32] [--batch 8] [--seeds 3] [--gamma 0.1]

Prints, per rung, the worst relative error over the loss scalars and the relative L2 error of the real / fake logit vectors.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import xmc_ref as X
from parity_util import LOSS_KEYS, setup_cfg

RUNGS = [
    ("all sites bf16 (the round-2 default mode)", ()),
    ("weights f32, activations bf16", ("d.w", "g.w", "h.w")),
    ("activations f32, weights bf16", tuple(t for t in X.QUANT_SITES if not t.endswith(".w"))),
    ("D + head f32, G bf16", ("d.", "h.")),
    ("G f32, D + head bf16", ("g.",)),
    ("D trunk f32 (pool, shortcut, block sum)", ("d.pool", "d.sc", "d.sum")),
    ("D trunk + head f32", ("d.pool", "d.sc", "d.sum", "h.")),
    ("D trunk + head + conv_img/image f32", ("d.pool", "d.sc", "d.sum", "h.", "d.conv_img", "d.img")),
    ("D trunk + head f32, G trunk f32 (stem, shortcut, block sum, image)", ("d.pool", "d.sc", "d.sum", "h.", "g.stem", "g.sc", "g.sum", "g.img", "g.act")),
    ("everything but the residual-branch convolutions (d.r0 d.r2 g.c1 g.c2 g.aff) f32",
     tuple(t for t in X.QUANT_SITES if t not in ("d.r0", "d.r2", "g.c1", "g.c2", "g.aff", "d.w", "g.w"))),
    ("WHAT-IF: all sites IEEE half (f16: 11 significant bits, same MFMA rate) instead of bf16 (8 bits)", "f16"),
]
# round 4: the rungs above the all-half mode (`--fmt f16`): which f32 islands would bring the LOGIT vectors inside 1e-3?
RUNGS_F16 = [
    ("all sites IEEE half (the f16 mode)", ()),
    ("f32 head: last feature map + COND_DNET (condition, joint_conv.0 output, head weights)", ("d.last", "h.")),
    ("f32 head + D weights f32", ("d.last", "h.", "d.w")),
    ("f32 head + D trunk f32 (pool, shortcut, block sum)", ("d.last", "h.", "d.pool", "d.sc", "d.sum")),
    ("D + head f32, G half", ("d.", "h.")),
    ("G f32, D + head half", ("g.",)),
    ("weights f32, activations half", ("d.w", "g.w", "h.w")),
    ("activations f32, weights half", tuple(t for t in X.QUANT_SITES if not t.endswith(".w"))),
]


FMT = torch.bfloat16


def one_step(h, PG, PD, batch, skip=None):
    PG = {k: v.clone() for k, v in PG.items()}
    PD = {k: v.clone() for k, v in PD.items()}
    oG, oD = X.AdamState(h.g_lr, h.g_betas, 1e-3), X.AdamState(h.d_lr, h.d_betas, 1e-3)
    if skip is None:
        return X.train_step(PG, PD, oG, oD, h, batch)
    if skip == "f16":
        with X.quant(True, fmt=torch.float16, precise=False):
            return X.train_step(PG, PD, oG, oD, h, batch)
    with X.quant(True, skip=skip, fmt=FMT, precise=False):      # (the rungs describe the modes WITHOUT the precise trunk of round 5)
        return X.train_step(PG, PD, oG, oD, h, batch)


def errors(o, ref):
    worst, which = 0.0, ""
    for k in LOSS_KEYS:
        if k in ref:
            e = abs(o[k] - ref[k]) / max(abs(ref[k]), 1e-12)
            if e > worst:
                worst, which = e, k
    lg = tuple(((o[k] - ref[k]).norm() / ref[k].norm()).item() for k in ("logit_real", "logit_fake"))
    return worst, which, max(lg), lg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--nch", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--cfg", type=str, default="df_gan_damsm_nomagp.yml")
    ap.add_argument("--fmt", type=str, default="bf16", choices=["bf16", "f16"], help="f16: the rungs above the all-half mode")
    ap.add_argument("--params", type=str, default="synth,ref")
    a = ap.parse_args()
    global FMT, RUNGS
    if a.fmt == "f16":
        FMT, RUNGS = torch.float16, RUNGS_F16
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg, h = setup_cfg(a.cfg, **{"TRAIN.NCH": a.nch, "IMG.SIZE": a.size})
    print(f"# {a.cfg}, {a.size}x{a.size}, NCH={a.nch}, batch {a.batch}, {a.seeds} seeds; worst relative loss error (which loss) | "
          f"logit rel. L2 error; bar 1e-3")
    for pname in a.params.split(","):
        print(f"## parameters: {pname}" + (f" (block gammas {a.gamma})" if pname == "ref" else ""))
        rows = {r[0]: [] for r in RUNGS}
        for s in range(a.seeds):
            if pname == "synth":
                PG, PD = X.synth_params(X.gen_shapes(h), 5 + s), X.synth_params(X.netd_shapes(h), 6 + s)
            else:
                PG, PD = X.ref_init_params(X.gen_shapes(h), 5 + s, a.gamma), X.ref_init_params(X.netd_shapes(h), 6 + s, a.gamma)
            batch = X.synth_batch(h, a.batch, seed=200 + s, words_len=cfg.TEXT.MAX_LENGTH)
            ref = one_step(h, PG, PD, batch)
            for name, skip in RUNGS:
                rows[name].append(errors(one_step(h, PG, PD, batch, skip), ref))
        for name, _ in RUNGS:
            r = rows[name]
            w = max(r, key=lambda t: t[0])
            print(f"{name:90s} loss {w[0]:.2e} ({w[1]:13s}) median {sorted(t[0] for t in r)[len(r) // 2]:.2e} | "
                  f"logit {max(t[2] for t in r):.2e} (real {max(t[3][0] for t in r):.2e}, fake {max(t[3][1] for t in r):.2e})", flush=True)


if __name__ == "__main__":
    main()
