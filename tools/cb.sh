cd tools
for b in conv_bench.out conv_bench_nosched.out; do echo $b; for a in "64 64 64 384 384 9 0 128 1 2 0" "64 64 64 384 384 9 1 128 1 2 1" "64 64 64 192 192 9 1 96 1 2 1" "64 32 32 576 576 9 1 96 1 2 1"; do timeout 60 ./$b $a; done; done
timeout 60 ./conv_bench_trace.out 64 64 64 384 384 9 0 128 1 2 0
