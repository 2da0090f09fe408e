#!/bin/bash
# what the concept head's backward kernel waits for: instruction-cache and wait counters of one eager config-3 iteration
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/hb; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\b\|SQ_INST_CYCLES[A-Z_]*" $O/counters.txt | sort -u > $O/names.txt || true
cat $O/names.txt
P="--workload config3 --graph 0 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/p1 -- python bench.py $P > /dev/null 2> $O/p1.err
f=$(find $O/p1 -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "concept_head" in k or "gn_sums" in k or "attn_pool_bwd" in k or "affine2_bwd" in k:
        n = max(cnt[(k, c)] for c in agg[k])
        print(k, n, {c: round(v / n, 1) for c, v in agg[k].items()})
PY
rm -rf $O/p1
