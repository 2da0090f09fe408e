#!/bin/bash
# config 3 under the kernel trace: per-kernel averages -> gpurun_out/c3/stats.csv, and the bench line
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/c3; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/t -- python bench.py --workload config3 --no_alt_precision --no_entrypoint --no_parity --no_cpu_baseline --no_roofline "$@" > $O/bench.json 2> $O/err
python profiles/summarize.py stats $O/t $O/stats.csv; rm -rf $O/t
tail -1 $O/bench.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
