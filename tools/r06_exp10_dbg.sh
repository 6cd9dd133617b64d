#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp10_dbg.txt; : > $O
for a in "1 512 512 64 64 9 0 64 1 10 0" "4 512 512 64 64 9 0 64 1 10 1" "4 512 512 128 64 9 0 64 1 10 0" "4 512 512 64 64 9 0 64 1 10 2 0 0 1"; do echo "## $a" >> $O; TD_DBG=1 TD_NO_CMP=1 timeout 120 tools/conv_bench.out $a 2>&1 | grep -v "check wide" | cut -c1-700 >> $O; done
cat $O
