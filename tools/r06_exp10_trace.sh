cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_exp10_trace.txt; : > $O
for a in "4 512 512 64 64 9 0 64 1 9 1" "4 512 512 64 64 9 0 64 1 10 1" "4 512 512 128 64 9 0 64 1 9 1" "4 512 512 128 64 9 0 64 1 10 1" "4 512 512 64 64 9 0 64 1 9 0" "4 512 512 64 64 9 0 64 1 10 0"; do echo "## $a" >> $O; TD_NO_CMP=1 timeout 120 tools/conv_bench_trace.out $a 2>&1 | grep -v "check" >> $O; done
cat $O
