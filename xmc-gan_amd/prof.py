"""Per-launch timing of the convolution kernels with HIP events recorded on the launch stream
(bench.py's `roofline` object).  Disabled by default: zero overhead on the normal path."""
import torch

_on = False
_recs = []


def enable():
    global _on
    _on = True
    _recs.clear()


def disable():
    global _on
    _on = False


def active():
    return _on


class launch:
    """with prof.launch(family, flops): <enqueue one kernel>"""

    def __init__(self, family, flops, tag=None):
        self.family, self.flops, self.tag = family, flops, tag

    def __enter__(self):
        if _on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()          # current stream == the stream the kernel is enqueued on
        return self

    def __exit__(self, *a):
        if _on:
            self.e1.record()
            _recs.append((self.family, self.flops, self.e0, self.e1, self.tag))


def summary(peak_tflops):
    torch.cuda.synchronize()
    fam = {}
    for family, flops, e0, e1, _tag in _recs:
        f = fam.setdefault(family, dict(launches=0, gflop=0.0, ms=0.0))
        f["launches"] += 1
        f["gflop"] += flops / 1e9
        f["ms"] += e0.elapsed_time(e1)
    for f in fam.values():
        f["tflops"] = round(f["gflop"] / max(f["ms"], 1e-9), 2)
        f["avg_launch_us"] = round(1e3 * f["ms"] / f["launches"], 2)
        f["gflop"], f["ms"] = round(f["gflop"], 1), round(f["ms"], 3)
    dom = max(fam, key=lambda k: fam[k]["ms"]) if fam else None
    if dom is None:
        return None
    d = fam[dom]
    return dict(bound="mfma", kernel=dom, achieved=d["tflops"], peak=peak_tflops, unit="TFLOP/s",
                frac=round(d["tflops"] / peak_tflops, 4), traffic=None, launches=d["launches"],
                avg_launch_us=d["avg_launch_us"], algorithmic_gflop_per_step=d["gflop"], families=fam)


def by_shape():
    """[(family, tag, launches, total_ms, tflops)] sorted by time (debug aid)."""
    torch.cuda.synchronize()
    agg = {}
    for family, flops, e0, e1, tag in _recs:
        a = agg.setdefault((family.split(" ")[0], tag), [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
    rows = [(k[0], k[1], v[0], v[1], v[2] / 1e9 / max(v[1], 1e-9)) for k, v in agg.items()]
    return sorted(rows, key=lambda r: -r[3])
