cd tools
for a in "64 8 8 1536 768 9 1 96 1 3 1" "64 8 8 1536 768 9 1 96 2 3 1" "64 8 8 1536 768 9 1 96 3 3 1" "64 8 8 1536 768 9 1 128 2 3 1" "64 8 8 768 768 9 0 96 2 3 2" "1 64 64 384 384 9 1 96 1 3 1" "1 64 64 384 384 9 1 96 2 3 1" "1 64 64 384 384 9 1 96 3 3 1" "1 64 64 384 384 9 1 64 6 0 1" "1 32 32 576 576 9 1 96 4 3 1" "1 32 32 576 576 9 1 64 8 0 1"; do timeout 60 ./conv_bench.out $a; done
cd ..; python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_new.json; python -c "
import json; d=json.load(open('gpurun_out/bench_new.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['end_to_end_achieved'], d['latency_single_tile_ms'])"
