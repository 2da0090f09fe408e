"""D-step gradients of smoke()'s iteration: sign-bit blocks against stored-branch blocks (same process, same inputs)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import xmc_ref as X
from xmc_gan_amd import ops
from parity_util import run_oracle_steps, run_product_steps, setup_cfg, rel_err


def _patch_ops(name, val):       # (ops is a package since round 5: a name lives in the module that defines it and in those that import it)
    import sys
    for m in list(sys.modules.values()):
        if getattr(m, "__name__", "").startswith("xmc_gan_amd.ops") and hasattr(m, name):
            setattr(m, name, val)


mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
ops.set_precision(mode)
cfg, h = setup_cfg("df_gan_damsm.yml", **{"TRAIN.NCH": 8})
PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
batches = [X.synth_batch(h, 4, seed=200, words_len=cfg.TEXT.MAX_LENGTH)]
_, _, o32 = run_oracle_steps(h, PG, PD, batches, eps=1e-3)
orig = ops._second_order
res = {}
for name, force in (("bits", False), ("values", True)):
    _patch_ops("_second_order", (lambda: True) if force else orig)
    _, _, p, tapG, tapD = run_product_steps(h, PG, PD, batches, eps=1e-3)
    res[name] = (tapD.records[0], tapG.records[0])
_patch_ops("_second_order", orig)
for which, idx, ref in (("D", 0, o32[0]["grads_D"]), ("G", 1, o32[0]["grads_G"])):
    a, b = res["bits"][idx], res["values"][idx]
    rows = sorted(((rel_err(a[n], b[n]), rel_err(a[n], ref[n]), rel_err(b[n], ref[n]), float(ref[n].abs().max()), n) for n in a if n in ref), reverse=True)
    print(f"{which}: bits vs values / bits vs f32 oracle / values vs f32 oracle / max|g|")
    for r in rows[:8]:
        print("   %.3e %.3e %.3e %.2e %s" % r)
