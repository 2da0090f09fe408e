#!/bin/bash
# same-box A/B of the row-kernel split rounding (floor: new, XMC_DEBUG_DISPATCH=wrow_ceil_split: old): per-kernel averages from two
# rocprofv3 stats passes of the headline step, then three interleaved whole-step timings
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/wrow; mkdir -p $O
Q="--no_alt_precision --no_entrypoint --no_parity --no_cpu_baseline --no_roofline --steps 10 --warmup 3"
rocprofv3 --kernel-trace --stats -f csv -d $O/new -- python bench.py $Q > $O/new.json 2> $O/new.err
python profiles/summarize.py stats $O/new $O/new.csv; rm -rf $O/new
export XMC_DEBUG_DISPATCH=wrow_ceil_split
rocprofv3 --kernel-trace --stats -f csv -d $O/old -- python bench.py $Q > $O/old.json 2> $O/old.err
python profiles/summarize.py stats $O/old $O/old.csv; rm -rf $O/old
unset XMC_DEBUG_DISPATCH
grep "wgrad_row" $O/new.csv $O/old.csv
bash tests/diag/ab.sh wrow_ceil_split 3
