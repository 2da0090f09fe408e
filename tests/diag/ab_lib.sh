#!/bin/bash
# same-box A/B of two builds of the library on the benched step: tests/diag/ab_lib.sh <other .so> [rounds] [bench flags]
lib=$1; n=${2:-2}; shift; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no_parity --no_alt_precision --no_entrypoint --no_cpu_baseline --no_roofline "$@" > gpurun_out/ab/new_$i.log 2>&1 || exit 1
  XMC_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no_parity --no_alt_precision --no_entrypoint --no_cpu_baseline --no_roofline "$@" > gpurun_out/ab/old_$i.log 2>&1 || exit 1
  echo "round $i new $(tail -1 gpurun_out/ab/new_$i.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])') | other lib $(tail -1 gpurun_out/ab/old_$i.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
