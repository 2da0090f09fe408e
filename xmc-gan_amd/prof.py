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

    def __init__(self, family, flops, tag=None, nbytes=0):
        """``nbytes``: algorithmic HBM bytes of the launch = every operand and result tensor once (the `roofline_hbm` view)"""
        self.family, self.flops, self.tag, self.nbytes = family, flops, tag, nbytes

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
            _recs.append((self.family, self.flops, self.e0, self.e1, self.tag, kern, self.nbytes))


def _aggregate_where(pred):
    return _aggregate(lambda r: r[5], [r for r in _recs if pred(r)])


def _aggregate(key, recs=None):
    agg = {}
    for rec in (_recs if recs is None else recs):
        f = agg.setdefault(key(rec), dict(launches=0, gflop=0.0, ms=0.0, gbyte=0.0))
        f["launches"] += 1
        f["gflop"] += rec[1] / 1e9
        f["gbyte"] += rec[6] / 1e9
        f["ms"] += rec[2].elapsed_time(rec[3])
    for f in agg.values():
        f["tflops"] = round(f["gflop"] / max(f["ms"], 1e-9), 2)
        f["gbytes_per_s"] = round(f["gbyte"] / max(f["ms"], 1e-9) * 1e3, 1)
        f["flop_per_byte"] = round(f["gflop"] / max(f["gbyte"], 1e-12), 1)
        f["gbyte"] = round(f["gbyte"], 3)
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
    # the dominant instantiation of each operator family as well (forward + data gradient / weight gradient): the overall dominant
    # kernel changes family from round to round, these two keep a comparable series
    per_family = {}
    for fname in {r[0] for r in _recs}:
        ks = _aggregate_where(lambda r, f=fname: r[0] == f)
        if ks:
            k = max(ks, key=lambda n: ks[n]["ms"])
            per_family[fname.split(" ")[0]] = dict(kernel=k, achieved=ks[k]["tflops"], frac=round(ks[k]["tflops"] / peak_tflops, 4),
                                                   launches=ks[k]["launches"], ms=ks[k]["ms"], avg_launch_us=ks[k]["avg_launch_us"],
                                                   traffic=traffic_lookup(k) if traffic_lookup else None)
    return dict(bound="mfma", kernel=dom, achieved=d["tflops"], peak=peak_tflops, unit="TFLOP/s",
                frac=round(d["tflops"] / peak_tflops, 4), traffic=traffic, launches=d["launches"],
                avg_launch_us=d["avg_launch_us"], algorithmic_gflop_per_launch=d["gflop_per_launch"],
                dominant_per_family=per_family, kernels=kern, families=fam)


def summary_hbm(peak_gbs, ridge_flop_per_byte, top=8):
    """The HBM view ("layers with Cout <= 64 ... report their GB/s, not MFMA %"): the convolution instantiations
    whose algorithmic intensity (FLOPs / bytes of their operand and result tensors, each once) lies below the ridge of the chip
    are bandwidth-bound; for the `top` of them by time: algorithmic bytes / HIP-event duration against the HBM peak."""
    torch.cuda.synchronize()
    kern = _aggregate(lambda r: r[5])
    rows = [(k, v) for k, v in kern.items() if v["gbyte"] > 0 and v["flop_per_byte"] < ridge_flop_per_byte]
    rows.sort(key=lambda kv: -kv[1]["ms"])
    out = [dict(kernel=k, bound="hbm", achieved=v["gbytes_per_s"], peak=peak_gbs, unit="GB/s", frac=round(v["gbytes_per_s"] / peak_gbs, 4),
                launches=v["launches"], ms=v["ms"], algorithmic_mbyte_per_launch=round(1e3 * v["gbyte"] / v["launches"], 1),
                flop_per_byte=v["flop_per_byte"]) for k, v in rows[:top]]
    return dict(ridge_flop_per_byte=ridge_flop_per_byte, total_ms=round(sum(v["ms"] for _, v in rows), 3), kernels=out)


def by_shape():
    """[(family, tag, launches, total_ms, tflops, algorithmic TB/s)] sorted by time (debug aid)."""
    torch.cuda.synchronize()
    agg = {}
    for family, flops, e0, e1, tag, kern, nb in _recs:
        a = agg.setdefault((kern or family.split(" ")[0], tag), [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops; a[3] += nb or 0
    rows = [(k[0], k[1], v[0], v[1], v[2] / 1e9 / max(v[1], 1e-9), v[3] / 1e9 / max(v[1], 1e-9)) for k, v in agg.items()]
    return sorted(rows, key=lambda r: -r[3])
