#!/bin/bash
# what the data-parallel path adds on one rank: kernel stats of `bench.py --force_dp` (RCCL at world size 1) beside the plain step
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/fdp; mkdir -p $O
Q="--steps 10 --warmup 3 --no_parity --no_alt_precision --no_entrypoint --no_cpu_baseline --no_roofline"
rocprofv3 --kernel-trace --stats -f csv -d $O/a -- python bench.py $Q > $O/plain.json 2> $O/plain.err
python profiles/summarize.py stats $O/a $O/plain.csv; rm -rf $O/a
rocprofv3 --kernel-trace --stats -f csv -d $O/b -- python bench.py --force_dp $Q > $O/forced.json 2> $O/forced.err
python profiles/summarize.py stats $O/b $O/forced.csv; rm -rf $O/b
python - <<'PY'
import csv
def load(f): return {r['kernel']: (int(r['calls']), float(r['total_ms'])) for r in csv.DictReader(open(f))}
a, b = load('gpurun_out/fdp/plain.csv'), load('gpurun_out/fdp/forced.csv')
print('total ms', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
rows = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k, a.get(k, (0, 0)), b.get(k, (0, 0))) for k in set(a) | set(b)), reverse=True)
for d, k, va, vb in rows[:14] + rows[-6:]:
    print(f"{d:8.3f} ms  {k[:90]:90s} {va} -> {vb}")
PY
