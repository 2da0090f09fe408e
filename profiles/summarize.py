#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSV summaries committed next to this script.

    python profiles/summarize.py stats  <dir of `rocprofv3 --kernel-trace --stats`>            out.csv
    python profiles/summarize.py pmc    <dir of `--pmc FETCH_SIZE`> <dir of `--pmc WRITE_SIZE`>  out.csv
    python profiles/summarize.py mfma   <dir of `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE`>  out.csv

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  On gfx950 FETCH_SIZE counts the 128-byte requests of wide
coalesced reads at 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is exact for 16-byte-per-lane
stores and float atomics.  The two counters do not fit one pass (TCC slots), hence two runs of the same command.

mfma: MfmaUtil as rocprofv3's own derived metric defines it, 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM),
from the raw counters (the per-dispatch rows report GRBM_GUI_ACTIVE summed over the 8 XCDs, hence the /8; SIMD_NUM = 1024).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def _one(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        sys.exit(f"no *{suffix} under {d}")
    return max(hits, key=os.path.getsize)


def stats(d, out):
    rows = list(csv.DictReader(open(_one(d, "_kernel_stats.csv"))))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], round(int(r["TotalDurationNs"]) / 1e6, 3),
                        round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"]])


def _counter(d, name):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(_one(d, "_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def pmc(dfetch, dwrite, out):
    fe, wr = _counter(dfetch, "FETCH_SIZE"), _counter(dwrite, "WRITE_SIZE")
    names = sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_sum_raw", "fetch_MB_per_dispatch_corrected_x2", "WRITE_SIZE_KB_sum",
                    "write_MB_per_dispatch"])
        for k in names:
            nf, sf = fe.get(k, [0, 0.0])
            nw, sw = wr.get(k, [0, 0.0])
            n = max(nf, nw, 1)
            w.writerow([k, n, round(sf, 1), round(2 * sf / 1024 / max(nf, 1), 3), round(sw, 1), round(sw / 1024 / max(nw, 1), 3)])


def mfma(d, out, xcds=8, simds=1024):
    agg, nd, seen = defaultdict(lambda: defaultdict(float)), defaultdict(int), set()
    for r in csv.DictReader(open(_one(d, "_counter_collection.csv"))):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            nd[k] += 1
    write_mfma([dict(kernel=k, dispatches=nd[k], gui_active=c.get("GRBM_GUI_ACTIVE", 0.0),
                     mfma_busy=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), lds_conf=c.get("SQ_LDS_BANK_CONFLICT", 0.0),
                     lds_act=c.get("SQ_LDS_IDX_ACTIVE", 0.0)) for k, c in agg.items()], out, xcds, simds)


def write_mfma(rows, out, xcds=8, simds=1024):
    rows = sorted((r for r in rows if r["gui_active"] > 0), key=lambda r: -r["gui_active"])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "GRBM_GUI_ACTIVE_sum_over_xcds", "SQ_VALU_MFMA_BUSY_CYCLES_sum", "MfmaUtil_pct",
                    "lds_bank_conflict_cycles_per_lds_active_cycle"])
        for r in rows:
            util = 100.0 * r["mfma_busy"] / (r["gui_active"] / xcds * simds)
            w.writerow([r["kernel"], r["dispatches"], round(r["gui_active"]), round(r["mfma_busy"]), round(util, 2),
                        round(r["lds_conf"] / max(r["lds_act"], 1.0), 4)])


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) == 4 and sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    elif len(sys.argv) == 5 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sys.exit(__doc__)
