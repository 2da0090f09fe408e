"""bf16 product vs the plain f32 oracle and vs the quantisation-aware oracle on the attention-modulation generators (first iteration):
how much of the bf16 distance the rounding sites g.c.* explain.  python tests/diag/concept_quant_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
import xmc_ref as X
from xmc_gan_amd import ops
from parity_util import run_oracle_steps, run_product_steps, setup_cfg, mean_abs_err, LOSS_KEYS

CASES = [("concept_in_df_gan_damsm_nomagp.yml", {"TRAIN.NCH": 8}, 3),
         ("concept_in_df_gan_damsm_nomagp.yml", {"TRAIN.NCH": 8, "IMG.SIZE": 128}, 2),
         ("concept_out_df_gan_sbert_damsm_nomagp.yml", {"TRAIN.NCH": 8}, 3),
         ("concept_in_df_gan_sbert_n2_damsm.yml", {"TRAIN.NCH": 8}, 3)]


def agg(rec, ref):
    num = den = 0.0
    worst, wn = 0.0, ""
    big = max(g.norm().item() for g in ref.values() if g is not None)
    for n, go in ref.items():
        if go is None:
            continue
        gp = rec[n]
        num += ((gp - go) ** 2).sum().item(); den += (go ** 2).sum().item()
        e = (gp - go).norm().item() / max(go.norm().item(), 2e-2 * big)
        if e > worst:
            worst, wn = e, n
    return (num / den) ** 0.5, worst, wn


ops.set_precision("bf16")
for yml, over, B in CASES:
    cfg, h = setup_cfg(yml, **over)
    PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
    bs = [X.synth_batch(h, B, seed=200, words_len=cfg.TEXT.MAX_LENGTH)]
    _, _, o = run_oracle_steps(h, PG, PD, bs, eps=1e-3)
    _, _, q = run_oracle_steps(h, PG, PD, bs, eps=1e-3, quant=True)
    with ops.fixed_order():
        _, _, p, tapG, tapD = run_product_steps(h, PG, PD, bs, eps=1e-3)
    print(f"== {yml} {over}")
    for name, ref in (("f32 oracle  ", o[0]), ("quant oracle", q[0])):
        le = max(abs(float(p[0][k]) - float(ref[k])) / (abs(float(ref[k])) + 0.2) for k in LOSS_KEYS if k in ref)
        line = f"  vs {name}: loss {le:.2e} image {mean_abs_err(p[0]['fake'], ref['fake']):.2e}"
        a, w, wn = agg(tapD.records[0], ref["grads_D"]); line += f" | D agg {a:.2e} worst {w:.2e}"
        if "grads_G" in ref:
            a, w, wn = agg(tapG.records[0], ref["grads_G"]); line += f" | G agg {a:.2e} worst {w:.2e} ({wn})"
        print(line, flush=True)
