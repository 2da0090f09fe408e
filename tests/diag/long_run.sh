#!/bin/bash
# stability: the entry point for many iterations at the benched size, both 16-bit modes (synthetic batches, 4 distinct, cycled)
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/long
for P in bf16 f16; do
  python xmc_gan/train_gan.py --cfg xmc_gan/cfg/df_gan_damsm_nomagp.yml --synthetic ${1:-1000} --bs 256 --imsize 256 --max_epoch 1 --precision $P \
      --output_dir gpurun_out/long/run_$P > gpurun_out/long/$P.log 2>&1
  echo "$P rc=$?"; grep -i "images/s\|throughput\|skipped\|loss scale\|nan\|inf " gpurun_out/long/$P.log | tail -6
  tail -3 gpurun_out/long/$P.log | cut -c1-300
done
