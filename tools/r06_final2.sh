#!/bin/bash
# round 6, last call: the wide tile's -DTD_TRACE build again (guarded DMA / load forms), flavour 9 and the persistent flavour 10; then the end-of-round collection
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_trace_after_guard.txt; : > $O
for a in "4 512 512 64 64 9 0 64 1 9 1" "4 512 512 64 64 9 0 64 1 10 1" "64 64 64 192 192 9 0 96 1 9 1"; do echo "## $a" >> $O; TD_NO_CMP=1 timeout 120 tools/conv_bench_trace.out $a 2>&1 | grep -v "check" | cut -c1-330 >> $O; done
for a in "4 512 512 64 64 9 0 64 1 10 1" "4 512 512 64 64 9 0 64 1 10 2 0 0 1" "4 512 512 128 64 9 0 64 1 10 1"; do echo "## $a" >> $O; timeout 120 tools/conv_bench.out $a 2>&1 | grep -E "us  |check persistent" | cut -c1-200 >> $O; done
bash tools/r06_final.sh
echo ==== ; cat $O
