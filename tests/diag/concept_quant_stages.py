"""Teacher-forced, stage by stage: each storage point of an attention-modulation generator in the bf16 engine against the
quantisation-aware oracle's value at the same point (tests/parity_util.py concept_quant_walk).  Prints the bit-equal fraction and the
relative L2 distance per site.  python tests/diag/concept_quant_stages.py [in|out] [f16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from xmc_gan_amd import ops
from parity_util import concept_quant_walk

kind = sys.argv[1] if len(sys.argv) > 1 else "in"
ops.set_precision("f16" if "f16" in sys.argv[2:] else "bf16")
for blk, what, same, rel in concept_quant_walk(kind):
    print(f"block {blk} {what:24s} bit-equal {same:.5f}  rel {rel:.2e}", flush=True)
