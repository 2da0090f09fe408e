set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/c3pmc
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/fetch -- python bench.py --workload config3 --graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/write -- python bench.py --workload config3 --graph 0 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_alt_precision --no_entrypoint --no_parity > /dev/null 2> $O/write.err
python profiles/summarize.py pmc $O/fetch $O/write $O/pmc_c3.csv
rm -rf $O/fetch $O/write
echo done
