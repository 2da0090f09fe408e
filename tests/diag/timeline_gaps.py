"""diagnostic: from a `rocprofv3 --kernel-trace -f csv` directory, how much of an iteration's wall time is covered by kernels, how
much by dependency gaps, and how much two kernels overlap (weight gradients on the side stream / graph branches)

    python tests/diag/timeline_gaps.py <dir> [n_last_dispatches]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = rows[-n:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0          # union of intervals
cur_s, cur_e = rows[0][0], rows[0][1]
overlap = 0
gaps = []
for s, e, _ in rows[1:]:
    if s <= cur_e:
        overlap += min(e, cur_e) - s
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
gaps.sort()
small = [g for g in gaps if g < 50_000]
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.2f} ms: sum of durations {tot / 1e6:.2f} ms, union {busy / 1e6:.2f} ms, overlapped {overlap / 1e6:.2f} ms")
print(f"gaps: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms; of these < 50 us: {len(small)} totalling {sum(small) / 1e6:.2f} ms, median {small[len(small) // 2] / 1e3:.2f} us, "
      f"p90 {small[int(len(small) * 0.9)] / 1e3:.2f} us")
