set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r04prof/pmc_dstem
mkdir -p "$O"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -f csv -d $O/a -- python tests/diag/dstem_time.py > /dev/null 2> $O/err.txt
python - <<'PY'
import csv,glob,collections
f=max(glob.glob('gpurun_out/r04prof/pmc_dstem/a/**/*counter_collection.csv',recursive=True),key=lambda p:__import__('os').path.getsize(p))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    if 'dstem' in k or 'wgrad_tile_kernel<4, 2, 512, 2' in k:
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    mu=100*v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']/8*1024) if v['GRBM_GUI_ACTIVE'] else 0
    print(f"{k:60s} MfmaUtil {mu:5.1f}%  LDS conflict/active {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f}  lds_idx_active/gui {v['SQ_LDS_IDX_ACTIVE']/max(v['GRBM_GUI_ACTIVE'],1):.3f}")
PY
rm -rf $O/a
