#!/bin/bash
# samples the card's clock / power while the headline step runs: tests/diag/clock_watch.sh  (writes gpurun_out/clock/)
mkdir -p gpurun_out/clock
python bench.py --steps 500 --warmup 5 --no_parity --no_alt_precision --no_entrypoint --no_cpu_baseline --no_roofline > gpurun_out/clock/bench.log 2>&1 &
bp=$!
sleep 6
for i in $(seq 1 80); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|junction" | tr '\n' ' ' >> gpurun_out/clock/smi.log
  echo >> gpurun_out/clock/smi.log
  kill -0 $bp 2>/dev/null || break
  sleep 0.3
done
wait $bp
tail -1 gpurun_out/clock/bench.log | cut -c1-200
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '
