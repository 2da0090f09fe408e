set -e
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05prof
mkdir -p $O
Q="--no_alt_precision --no_entrypoint --no_parity"
run() { python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline $Q "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', '|', j['value'], 'img/s', j['ms_per_step'], 'ms', j.get('step_algorithmic_tflops'), 'TF/s', j.get('step_frac_of_bf16_peak'))"; }
{ run; run --precision f16; run --workload magp; run --workload magp --precision f16; run --workload config2; run --workload config3;
  run --workload config3 --gen CONCEPT_INATTN_GEN; run --workload config3 --gen CONCEPT_OUTATTN_GEN;
  run --workload config3 --gen CONCEPT_OUT_DF_GEN --cfg concept_out_df_gan_sbert_damsm_nomagp.yml; run --spec_norm; } > $O/r05_bench_workloads.txt
cat $O/r05_bench_workloads.txt
