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
            from . import lib
            kern = lib.load().xmc_last_kernel().decode()       # the instantiation the C dispatcher picked
            _recs.append((self.family, self.flops, self.e0, self.e1, self.tag, kern))


def _aggregate(key):
    agg = {}
    for rec in _recs:
        f = agg.setdefault(key(rec), dict(launches=0, gflop=0.0, ms=0.0))
        f["launches"] += 1
        f["gflop"] += rec[1] / 1e9
        f["ms"] += rec[2].elapsed_time(rec[3])
    for f in agg.values():
        f["tflops"] = round(f["gflop"] / max(f["ms"], 1e-9), 2)
        f["avg_launch_us"] = round(1e3 * f["ms"] / f["launches"], 2)
        f["gflop_per_launch"] = round(f["gflop"] / f["launches"], 3)
        f["gflop"], f["ms"] = round(f["gflop"], 1), round(f["ms"], 3)
    return agg


def summary(peak_tflops, traffic_lookup=None):
    """Roofline object for the dominant kernel = the instantiation (name as rocprof prints it) with the largest summed
    duration in the profiled iteration.  achieved = algorithmic FLOPs of its launches / their HIP-event durations."""
    torch.cuda.synchronize()
    if not _recs:
        return None
    kern = _aggregate(lambda r: r[5])
    fam = _aggregate(lambda r: r[0])
    dom = max(kern, key=lambda k: kern[k]["ms"])
    d = kern[dom]
    traffic = traffic_lookup(dom) if traffic_lookup else None
    return dict(bound="mfma", kernel=dom, achieved=d["tflops"], peak=peak_tflops, unit="TFLOP/s",
                frac=round(d["tflops"] / peak_tflops, 4), traffic=traffic, launches=d["launches"],
                avg_launch_us=d["avg_launch_us"], algorithmic_gflop_per_launch=d["gflop_per_launch"],
                kernels=kern, families=fam)


def by_shape():
    """[(family, tag, launches, total_ms, tflops)] sorted by time (debug aid)."""
    torch.cuda.synchronize()
    agg = {}
    for family, flops, e0, e1, tag, kern in _recs:
        a = agg.setdefault((kern or family.split(" ")[0], tag), [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
    rows = [(k[0], k[1], v[0], v[1], v[2] / 1e9 / max(v[1], 1e-9)) for k, v in agg.items()]
    return sorted(rows, key=lambda r: -r[3])
