# PMC passes over one conv_bench configuration; one rocprofv3 run per counter group, counters only (no traces)
cd /tmp && export TMPDIR=/tmp
BIN=${BIN:-$GRAFT_REPO_ROOT/tools/conv_bench.out}
ARGS=${ARGS:-"64 64 64 384 384 9 1 128 1 2 1"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
rm -rf $OUT; mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- $BIN $ARGS > $OUT/$tag.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_conv'
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_glds' not in r['Kernel_Name']: continue
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
with open(out+'/summary.txt','w') as o:
    for k,(n,v) in sorted(agg.items()): o.write(f"{k} dispatches={n} avg_per_dispatch={v/n:.1f}\n")
print(open(out+'/summary.txt').read())
PY
