"""diagnostic (CPU only): WHICH LAYER of the generator owns the half mode's error of the generated-image losses?

`errD_fake` / `errG_fake` are means over the batch of a function of D(G(z)): with D exact (f32) the rungs below round one generator
site at a time ("only this tensor rounded") and report the relative error of mean(relu(1 + logit)) (errD_fake), of -mean(logit)
(errG_fake) and the relative L2 error of the logit vector.

    python tests/diag/layer_ladder_g.py [--size 256] [--batch 8] [--seeds 3] [--fmt f16] [--gamma 0.1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F

import xmc_ref as X
from parity_util import setup_cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--fmt", type=str, default="f16", choices=["bf16", "f16"])
    ap.add_argument("--cfg", type=str, default="df_gan_damsm_nomagp.yml")
    a = ap.parse_args()
    fmt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.fmt]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg, h = setup_cfg(a.cfg, **{"IMG.SIZE": a.size})
    ga = X.gen_arch(h.img_size, h.nch)
    depth = ga["depth"]
    learned = [i for i in range(depth) if ga["cin"][i] != ga["cout"][i]]
    trunk_w = [f"g.w@b{i}.sc" for i in learned] + ["g.w@out"]
    rungs = [("ALL generator sites rounded", dict(only=("g.",))),
             ("all generator WEIGHTS rounded", dict(only=("g.w",))),
             ("all generator ACTIVATIONS rounded", dict(only=("g.stem", "g.aff", "g.c1", "g.c2", "g.sc", "g.sum", "g.act", "g.img"))),
             ("trunk only (stem, c_sc + conv_out weights, shortcut, block sums, image)", dict(only=tuple(trunk_w + ["g.stem", "g.sc", "g.sum", "g.act", "g.img"]))),
             ("residual branches only (c1 / c2 weights and outputs, affine outputs)", dict(only=tuple([f"g.w@b{i}.c1" for i in range(depth)] + [f"g.w@b{i}.c2" for i in range(depth)] + ["g.aff", "g.c1", "g.c2"]))),
             ("only g.stem", dict(only=("g.stem",))), ("only g.img (the generated image)", dict(only=("g.img",))),
             ("only g.w@out (conv_out weights)", dict(only=("g.w@out",)))]
    for i in range(depth):
        if i in learned:
            rungs += [(f"only g.w@b{i}.sc", dict(only=(f"g.w@b{i}.sc",))), (f"only g.sc@b{i}", dict(only=(f"g.sc@b{i}",)))]
        rungs.append((f"only g.sum@b{i}", dict(only=(f"g.sum@b{i}",))))
    res = {n: [] for n, _ in rungs}
    for s in range(a.seeds):
        PG, PD = X.ref_init_params(X.gen_shapes(h), 5 + s, a.gamma), X.ref_init_params(X.netd_shapes(h), 6 + s, a.gamma)
        b = X.synth_batch(h, a.batch, seed=300 + s, words_len=cfg.TEXT.MAX_LENGTH)
        with torch.no_grad():
            ps = X.proj_sent(PG, b["sent_embs"])
            lg = lambda img: X.cond_dnet(PD, h, X.netd_forward(PD, h, img), ps)[0].flatten()
            gen = lambda: X.gen_forward(PG, h, b["noise"], b["sent_embs"], words_embs=b.get("words_embs"), mask=b.get("mask"))
            ref = lg(gen())
            for name, kw in rungs:
                with X.quant(True, fmt=fmt, precise=False, **kw):
                    img = gen()
                o = lg(img)
                e1 = abs(F.relu(1 + o).mean() - F.relu(1 + ref).mean()) / F.relu(1 + ref).mean().abs()
                e2 = abs(o.mean() - ref.mean()) / ref.mean().abs()
                res[name].append((float(e1), float(e2), float((o - ref).norm() / ref.norm())))
        print(f"# seed {s} done", file=sys.stderr, flush=True)
    print(f"# {a.cfg}, {a.size}x{a.size}, batch {a.batch}, {a.seeds} seeds, {a.fmt}, reference initialisation (block gammas {a.gamma}), D in f32; "
          "relative error of errD_fake | errG_fake | logit vector (rms over seeds, max)")
    for name, _ in rungs:
        r = res[name]
        f = lambda j: f"{(sum(t[j] ** 2 for t in r) / len(r)) ** 0.5:9.2e} {max(t[j] for t in r):9.2e}"
        print(f"{name:78s} {f(0)} | {f(1)} | {f(2)}", flush=True)


if __name__ == "__main__":
    main()
