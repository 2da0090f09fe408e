# round 5: the rocprofv3 passes and bench runs whose summaries are committed next to this script (run on a GPU box from the repo root;
# the program stands directly after `--` in every rocprofv3 command)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05prof
mkdir -p $O
Q="--no_alt_precision --no_entrypoint --no_parity"
python bench.py > $O/r05_bench_default_256px_b256.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py $Q > $O/r05_bench_under_rocprof_256px_b256.json 2> $O/stats.err
python profiles/summarize.py stats $O/stats $O/r05_kernel_stats_256px_b256.csv
rm -rf $O/stats
rocprofv3 --kernel-trace --stats -f csv -d $O/stats16 -- python bench.py $Q --precision f16 --no_cpu_baseline --no_roofline > $O/r05_bench_under_rocprof_f16.json 2> $O/stats16.err
python profiles/summarize.py stats $O/stats16 $O/r05_kernel_stats_f16_256px_b256.csv
rm -rf $O/stats16
echo stats done
P="--graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline $Q"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/fetch -- python bench.py $P > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/write -- python bench.py $P > /dev/null 2> $O/write.err
python profiles/summarize.py pmc $O/fetch $O/write $O/r05_pmc_hbm_traffic_256px_b256.csv
rm -rf $O/fetch $O/write
echo pmc done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $O/mfma -- python bench.py $P > /dev/null 2> $O/mfma.err
python profiles/summarize.py mfma $O/mfma $O/r05_pmc_mfma_util_256px_b256.csv
rm -rf $O/mfma
echo mfma done
rocprofv3 --kernel-trace --stats -f csv -d $O/c3 -- python bench.py --workload config3 $Q --no_cpu_baseline --no_roofline > $O/r05_bench_under_rocprof_config3.json 2> $O/c3.err
python profiles/summarize.py stats $O/c3 $O/r05_kernel_stats_concept_in_128px_b64.csv
rm -rf $O/c3
rocprofv3 --kernel-trace --stats -f csv -d $O/gp -- python bench.py --workload magp $Q --no_cpu_baseline --no_roofline > $O/r05_bench_under_rocprof_magp.json 2> $O/gp.err
python profiles/summarize.py stats $O/gp $O/r05_kernel_stats_magp_256px_b256.csv
rm -rf $O/gp
echo workloads stats done
run() { python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline $Q "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', '|', j['value'], 'img/s', j['ms_per_step'], 'ms', j.get('step_algorithmic_tflops'), 'TF/s', j.get('step_frac_of_bf16_peak'))"; }
{ run; run --precision f16; run --workload magp; run --workload magp --precision f16; run --workload config2; run --workload config3;
  run --workload config3 --gen CONCEPT_INATTN_GEN; run --workload config3 --gen CONCEPT_OUTATTN_GEN;
  run --workload config3 --gen CONCEPT_OUT_DF_GEN --cfg concept_out_df_gan_sbert_damsm_nomagp.yml; run --spec_norm; } > $O/r05_bench_workloads.txt
cat $O/r05_bench_workloads.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_three_modes.txt 2>&1
python tests/diag/contrastive_time.py 2>/dev/null | grep "^n=" > $O/r05_contrastive_now.txt
head -c 300 $O/r05_bench_default_256px_b256.json
