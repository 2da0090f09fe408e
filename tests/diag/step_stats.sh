#!/bin/bash
# kernel stats of the headline step on THIS box (condensed csv -> gpurun_out/stats/<tag>.csv): tests/diag/step_stats.sh <tag> [bench flags]
tag=${1:-now}; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/stats
mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/raw_$tag -- python bench.py --steps 12 --warmup 4 --no_alt_precision --no_entrypoint --no_parity --no_cpu_baseline --no_roofline "$@" > $O/$tag.json 2> $O/$tag.err
python profiles/summarize.py stats $O/raw_$tag $O/$tag.csv
rm -rf $O/raw_$tag
cut -c1-160 $O/$tag.json
