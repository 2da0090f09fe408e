"""Run-to-run spread of the attention-modulation generator's gradients (BASELINE config 3's generator at 128 px, the size whose
per-tensor bars were widened) with and without ops.fixed_order(): three product runs on identical inputs, worst per-tensor relative
difference to run 0, and the error against the f32 CPU oracle.  Prints, asserts nothing.
usage: python tests/diag/fixed_order_probe.py [fp32|bf16]"""
import os, sys, contextlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
import xmc_ref as X
from parity_util import setup_cfg, run_oracle_steps, run_product_steps
from xmc_gan_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ops.set_precision(mode)
cfg, h = setup_cfg("concept_in_df_gan_damsm_nomagp.yml", **{"TRAIN.NCH": 8, "IMG.SIZE": 128})
PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
batches = [X.synth_batch(h, 2, seed=200, words_len=cfg.TEXT.MAX_LENGTH)]
_, _, o = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
og = o[0]["grads_G"]
for fixed in (False, True):
    recs = []
    for rep in range(3):
        with (ops.fixed_order() if fixed else contextlib.nullcontext()):
            _, _, p, tapG, tapD = run_product_steps(h, PG, PD, batches, eps=1e-3)
        recs.append(tapG.records[0])
    big = max(g.norm().item() for g in og.values() if g is not None)
    spread, err = [], []
    for n, g0 in recs[0].items():
        if g0 is None or og.get(n) is None or og[n].norm().item() < 1e-5 * big:
            continue
        spread.append((max((recs[k][n] - g0).norm().item() / max(g0.norm().item(), 1e-30) for k in (1, 2)), n))
        err.append(((g0 - og[n]).norm().item() / og[n].norm().item(), n))
    spread.sort(reverse=True); err.sort(reverse=True)
    allv = lambda r: torch.cat([v.flatten() for v in r.values() if v is not None])
    agg = ((allv(recs[0]) - allv({k: v for k, v in og.items() if recs[0].get(k) is not None})).norm() / allv({k: v for k, v in og.items() if recs[0].get(k) is not None}).norm()).item()
    print(f"== {mode}, fixed order {fixed}: worst run-to-run spread / worst error vs the f32 oracle (all G tensors as one vector: {agg:.2e})")
    for (s_, n1), (e_, n2) in zip(spread[:6], err[:6]):
        print(f"   spread {s_:9.3e} {n1:58s} error {e_:9.3e} {n2}")
