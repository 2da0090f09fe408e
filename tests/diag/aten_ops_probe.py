"""Which framework (ATen) kernels are left in one eager G+D iteration of the bench configuration, with input shapes
(diagnostic: prints the table, asserts nothing)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch
import bench

if __name__ == "__main__":
    import runpy
    from torch.profiler import profile, ProfilerActivity
    # reuse bench.py's set-up by running it with 1 step, no graph, inside the profiler
    sys.argv = ["bench.py", "--steps", "1", "--warmup", "2", "--graph", "0", "--no_cpu_baseline", "--no_roofline"] + sys.argv[1:]
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        try:
            runpy.run_path(os.path.join(os.path.dirname(__file__), "..", "..", "bench.py"), run_name="__main__")
        except SystemExit:
            pass
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        if e.key.startswith("aten::") and e.device_time_total > 0:
            rows.append((e.device_time_total / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    for t, c, k, sh in rows[:40]:
        print(f"{t:9.1f} us/it  n={c:6.1f}  {k:28s} {sh}")
