"""Shapes of the framework fill / zeros operators of one steady-state eager iteration (torch profiler on the fourth iteration).
usage: python tests/diag/fill_shapes.py [bench flags]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
import xmc_gan.train_gan as tg

orig_it = tg.gan_iteration
calls = [0]
PROF = [None]


def counted(*a, **k):
    calls[0] += 1
    if calls[0] == 4:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            out = orig_it(*a, **k)
            torch.cuda.synchronize()
        PROF[0] = prof
        return out
    return orig_it(*a, **k)


tg.gan_iteration = counted
import runpy
sys.argv = ["bench.py", "--steps", "2", "--warmup", "3", "--graph", "0", "--no_cpu_baseline", "--no_roofline", "--no_parity", "--no_alt_precision",
            "--no_entrypoint"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
rows = []
for e in PROF[0].key_averages(group_by_input_shape=True):
    if e.key in ("aten::zeros", "aten::fill_", "aten::zero_", "aten::zeros_like", "aten::full", "aten::ones_like", "aten::new_zeros", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::cat", "aten::clone", "aten::contiguous"):
        rows.append((e.count, e.key, str(e.input_shapes)[:100], e.device_time_total))
rows.sort(reverse=True)
for c, k, sh, t in rows[:50]:
    print(f"{c:5d}  {k:18s} {t:9.1f} us  {sh}")
