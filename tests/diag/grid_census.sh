#!/bin/bash
# one eager iteration under the kernel trace: grid / workgroup / LDS / register figures per dispatch, to look for launches whose
# workgroup count is a little over a multiple of what the chip holds at once (a near-empty last round)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/census; mkdir -p $O
rocprofv3 --kernel-trace -f csv -d $O/t -- python bench.py --graph 0 --steps 1 --warmup 1 --no_alt_precision --no_entrypoint --no_parity --no_cpu_baseline --no_roofline "$@" > $O/bench.json 2> $O/err
f=$(find $O/t -name '*kernel_trace.csv' | head -1)
python - "$f" $O/census.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = ["Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Start_Timestamp", "End_Timestamp"]
keep = [k for k in keep if k in rows[0]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows[len(rows) // 2:]:
        w.writerow([r[k][:120] if k == "Kernel_Name" else r[k] for k in keep])
print(len(rows), "dispatches;", list(rows[0].keys()))
PY
rm -rf $O/t
