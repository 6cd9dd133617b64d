cd tools
for a in "64 64 64 384 384 9 0 128 1 2 0" "64 64 64 384 384 9 1 128 1 2 1" "64 64 64 384 384 9 0 128 1 2 2" "64 64 64 192 192 9 1 96 1 2 1" "64 64 64 192 192 9 0 96 1 2 2" "64 32 32 576 576 9 1 96 1 2 1"; do timeout 60 ./conv_bench.out $a; done
timeout 60 ./conv_bench_trace.out 64 64 64 384 384 9 0 128 1 2 0
timeout 60 ./conv_bench_trace.out 64 64 64 384 384 9 0 128 1 2 2
cd ..; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
