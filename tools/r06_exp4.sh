#!/bin/bash
# round 6, experiment 4: the whole GPU suite under the new defaults; wide-tile phase traces with the tap-entry wait split into "own DMA pieces" and
# "barrier"; the counters rocprofv3 offers for the L2 / fabric / memory-side cache on this box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp4.txt; : > $O
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r06_exp4_tests.txt 2>&1
tail -25 gpurun_out/r06_exp4_tests.txt >> $O
echo "# phase traces (flavour 9: 'restage' column = the part of 'tap-entry wait' spent on vmcnt, i.e. this wave's own DMA pieces; body = loop - wait)" >> $O
for L in "64 64 64 192 192 9 0 96 1 9 1" "64 64 64 192 192 9 0 96 1 9 2 0 0 1" "64 64 64 576 192 9 0 96 1 9 1" "4 512 512 64 64 9 0 64 1 9 1"; do
  echo "## $L" >> $O; TD_NO_CMP=1 timeout 120 tools/conv_bench_trace.out $L 2>&1 | grep -E "us  |trace \(|taps per WG|of the epilogue|timeline|shader clock" >> $O
done
echo "# rocprofv3 -L: cache / fabric counters" >> $O
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(TCC|TCP|MALL|GL2C|EA)[A-Za-z0-9_]*\b" | sort -u | tr '\n' ' ') >> $O 2>&1
echo >> $O
cat $O
