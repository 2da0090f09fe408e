set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2lds
mkdir -p $O
for v in cw8 cw4; do
  if [ $v = cw4 ]; then export XMC_DEBUG_DISPATCH=wtile_cw4; else unset XMC_DEBUG_DISPATCH; fi
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -f csv -d $O/$v -- python tests/kernel_probe.py "256,32,256,256,3,1,1,fwd,10" "256,64,128,128,3,1,1,fwd,10" > $O/$v.txt 2> $O/$v.err
done
python - <<'PY'
import csv, glob, collections
for v in ("cw8","cw4"):
    f = glob.glob(f"gpurun_out/r2lds/{v}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"][:60]
            if "wtile2" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        gui = c["GRBM_GUI_ACTIVE"]
        print(v, k, {a: f"{b:.3e}" for a, b in c.items()}, "LDS_IDX_ACTIVE/(GUI*32)=%.3f MFMA/(GUI*128)=%.3f conflict/active=%.3f" % (c["SQ_LDS_IDX_ACTIVE"]/(gui*32), c["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*128), c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1)))
PY
cat $O/cw8.txt $O/cw4.txt
rm -rf $O/cw8 $O/cw4
