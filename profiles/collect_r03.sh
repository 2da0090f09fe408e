# round 3: the rocprofv3 passes whose summaries are committed next to this script (run on a GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r03prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py --no_alt_precision --no_entrypoint --no_parity > $O/r03_bench_under_rocprof_256px_b256.json 2> $O/stats.err
python profiles/summarize.py stats $O/stats $O/r03_kernel_stats_256px_b256.csv
rm -rf $O/stats
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/fetch -- python bench.py --graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity > /dev/null 2> $O/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/write -- python bench.py --graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity > /dev/null 2> $O/write.err
python profiles/summarize.py pmc $O/fetch $O/write $O/r03_pmc_hbm_traffic_256px_b256.csv
rm -rf $O/fetch $O/write
echo pmc done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $O/mfma -- python bench.py --graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity > /dev/null 2> $O/mfma.err
python profiles/summarize.py mfma $O/mfma $O/r03_pmc_mfma_util_256px_b256.csv
rm -rf $O/mfma
echo mfma done
head -c 400 $O/r03_bench_under_rocprof_256px_b256.json
