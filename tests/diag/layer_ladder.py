"""diagnostic (CPU only): WHICH LAYER of the discriminator owns the 16-bit modes' logit error against the f32 oracle?

Round 5 (review item 1).  The figures `bench.py`'s parity leg reports are D-only quantities: the logit vector of D + COND_DNET on the
real images and on the ORACLE's generated images, relative L2 against the f32 oracle.  Here each storage site of the rounding oracle
(`X.quant`, which reproduces the engine's 16-bit modes bit for bit in > 99 % of the elements) is switched on ALONE, one layer at a
time ("only this tensor is rounded, everything else f32"): independent rounding errors add in quadrature, so the squares of the
column below are the shares of the all-sites figure.  Rungs that bracket the table: all sites, all weights, all activations, and the
candidate fixes ("trunk" = the shortcut path image -> [pool -> conv_s -> block sum] x depth -> head).

    python tests/diag/layer_ladder.py [--size 256] [--batch 8] [--seeds 5] [--fmt f16] [--gamma 0.1] [--params ref]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import xmc_ref as X
from parity_util import setup_cfg


def logits(h, PG, PD, imgs, sent):
    ps = X.proj_sent(PG, h, sent) if hasattr(X, "proj_sent") else None
    return X.cond_dnet(PD, h, X.netd_forward(PD, h, imgs), ps)[0].flatten()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--nch", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--cfg", type=str, default="df_gan_damsm_nomagp.yml")
    ap.add_argument("--fmt", type=str, default="f16", choices=["bf16", "f16"])
    ap.add_argument("--params", type=str, default="ref", choices=["ref", "synth"])
    a = ap.parse_args()
    fmt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.fmt]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg, h = setup_cfg(a.cfg, **{"TRAIN.NCH": a.nch, "IMG.SIZE": a.size})
    depth = X.disc_arch(h.img_size, h.nch)["depth"] - 1
    blocks = [f"b{i}" for i in range(depth)]
    learned = [f"b{i}" for i in range(depth) if X.disc_arch(h.img_size, h.nch)["cin"][i + 1] != X.disc_arch(h.img_size, h.nch)["cout"][i + 1]]
    trunk_w = [f"d.w@{b}.s" for b in learned] + ["h.w"]
    trunk_a = ["d.img", "d.pool", "d.sc", "d.sum", "h.c"]
    rungs = [("ALL sites rounded (the f16 mode of rounds 3-4)", dict(skip=())),
             ("all WEIGHTS rounded, activations f32", dict(only=("d.w", "h.w"))),
             ("all ACTIVATIONS rounded, weights f32", dict(skip=("d.w", "h.w", "g.w"))),
             ("trunk only rounded (image, pool, conv_s, block sums, conv_s weights, head)", dict(only=tuple(trunk_w + trunk_a + ["h.m"]))),
             ("residual branches only rounded (conv_r weights, r0, r2)", dict(only=tuple([f"d.w@{b}.r0" for b in blocks] + [f"d.w@{b}.r2" for b in blocks] + ["d.r0", "d.r2"]))),
             ("FIX A: conv_s + head weights exact (hi+lo), rest rounded", dict(skip=tuple(trunk_w))),
             ("FIX B: A + trunk activations exact (pool, conv_s out, block sum, image)", dict(skip=tuple(trunk_w + trunk_a))),
             ("FIX C: A + conv_s output and pooled input exact, block sums rounded", dict(skip=tuple(trunk_w + ["d.pool", "d.sc"]))),
             ("FIX D: all D weights exact", dict(skip=("d.w", "h.w")))]
    # the plan built in round 5 (`ops.precise_trunk`, the f16 mode): head in f32; conv_s weights as hi + lo pairs (b0's composed
    # shortcut excepted); the blocks on maps <= 8x8 ("small") keep their shortcut, block sum and pooled by-product in f32
    small = [i for i in range(depth) if a.size // (2 ** (i + 1)) <= 8]
    plan = ["h."] + [f"d.w@b{i}.s" for i in range(1, depth) if f"b{i}" in learned]
    for i in small:
        plan += [f"d.sc@b{i}", f"d.sum@b{i}"] + ([f"d.pool@b{i + 1}"] if i + 1 < depth else [])
    rungs.append(("PLAN: f32 head, hi+lo conv_s weights (b1..), f32 trunk on maps <= 8x8", dict(skip=tuple(plan))))
    rungs.append(("PLAN + b0's composed shortcut weights exact", dict(skip=tuple(plan + ["d.w@b0.s"]))))
    rungs.append(("BUILT: the rounding oracle's precise-trunk mode (= the engine's f16 mode since round 5)", dict(precise=True)))
    rungs += [(f"only image rounded", dict(only=("d.img",)))]
    for b in blocks:
        for site, tag in (("d.w", f"{b}.r0"), ("d.w", f"{b}.r2"), ("d.w", f"{b}.s"), ("d.r0", b), ("d.r2", b), ("d.pool", b), ("d.sc", b), ("d.sum", b)):
            if site == "d.w" and tag.endswith(".s") and b not in learned:
                continue
            if site in ("d.sc",) and b not in learned:
                continue
            if site == "d.pool" and b == "b0":
                continue
            rungs.append((f"only {site}@{tag}", dict(only=(f"{site}@{tag}",))))
    rungs += [("only h.w@j0 (joint_conv.0 weights)", dict(only=("h.w@j0",))), ("only h.w@j2", dict(only=("h.w@j2",))),
              ("only h.c (condition)", dict(only=("h.c",))), ("only h.m (joint_conv.0 output)", dict(only=("h.m",)))]
    res = {n: [] for n, _ in rungs}
    for s in range(a.seeds):
        if a.params == "synth":
            PG, PD = X.synth_params(X.gen_shapes(h), 5 + s), X.synth_params(X.netd_shapes(h), 6 + s)
        else:
            PG, PD = X.ref_init_params(X.gen_shapes(h), 5 + s, a.gamma), X.ref_init_params(X.netd_shapes(h), 6 + s, a.gamma)
        b = X.synth_batch(h, a.batch, seed=300 + s, words_len=cfg.TEXT.MAX_LENGTH)
        with torch.no_grad():
            fake = X.gen_forward(PG, h, b["noise"], b["sent_embs"], words_embs=b.get("words_embs"), mask=b.get("mask"))
            ps = F_proj(PG, h, b["sent_embs"])
            ref = [X.cond_dnet(PD, h, X.netd_forward(PD, h, im), ps)[0].flatten() for im in (b["imgs"], fake)]
            for name, kw in rungs:
                with X.quant(True, fmt=fmt, **{"precise": False, **kw}):
                    o = [X.cond_dnet(PD, h, X.netd_forward(PD, h, im), ps)[0].flatten() for im in (b["imgs"], fake)]
                res[name].append(tuple(float((x - y).norm() / y.norm()) for x, y in zip(o, ref)))
        print(f"# seed {s} done", file=sys.stderr, flush=True)
    print(f"# {a.cfg}, {a.size}x{a.size}, NCH={a.nch}, batch {a.batch}, {a.seeds} seeds, {a.fmt}, parameters: {a.params}"
          + (f" (block gammas {a.gamma})" if a.params == "ref" else "") + "; relative L2 error of the logit vector, bar 1e-3")
    print(f"{'rung':82s} {'real: rms':>10s} {'max':>9s} | {'gen.: rms':>10s} {'max':>9s}")
    for name, _ in rungs:
        r = res[name]
        rms = lambda j: (sum(t[j] ** 2 for t in r) / len(r)) ** 0.5
        print(f"{name:82s} {rms(0):10.2e} {max(t[0] for t in r):9.2e} | {rms(1):10.2e} {max(t[1] for t in r):9.2e}", flush=True)


def F_proj(PG, h, sent):
    import torch.nn.functional as F
    return F.linear(sent, PG["proj_sent.weight"], PG["proj_sent.bias"]) if "proj_sent.weight" in PG else sent


if __name__ == "__main__":
    main()
