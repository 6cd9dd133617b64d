#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for shape in "64 64 64 384 384 9 0 128 1 2 1" "64 64 64 192 192 9 0 96 1 2 2" "64 32 32 576 576 9 0 96 1 2 1" "64 32 32 960 384 9 0 128 1 3 1" "64 16 16 768 768 9 0 96 1 3 1"; do
  for b in cb_noxcd conv_bench cb_noxcd conv_bench; do echo -n "$b: "; timeout 60 ./$b.out $shape; done
done
cd /tmp && export TMPDIR=/tmp
for b in cb_noxcd conv_bench; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${b}_$c -- $GRAFT_REPO_ROOT/tools/$b.out 64 64 64 384 384 9 0 128 1 2 1 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
tot=0;n=0
for f in glob.glob('/tmp/pmc_${b}_$c/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_glds' in r['Kernel_Name']: tot+=float(r['Counter_Value']); n+=1
print('$b $c per dispatch (KB):', tot/max(n,1), 'dispatches', n)
PY
  done
done
b=conv_bench
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_pp_$c -- $GRAFT_REPO_ROOT/tools/$b.out 64 64 64 384 384 9 0 128 1 5 1 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
tot=0;n=0
for f in glob.glob('/tmp/pmc_pp_$c/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_pp' in r['Kernel_Name']: tot+=float(r['Counter_Value']); n+=1
print('pp $c per dispatch (KB):', tot/max(n,1), 'dispatches', n)
PY
done
