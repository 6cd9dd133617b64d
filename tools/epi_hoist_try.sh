#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
while read -r args; do
  [ -z "$args" ] && continue
  echo "## $args"
  for b in conv_bench conv_bench_hoist; do
    for r in 1 2; do timeout 120 tools/$b.out $args | head -1 | sed "s/^/  $b: /"; done
    TD_DUMP=gpurun_out/$b.bin timeout 120 tools/$b.out $args > /dev/null
  done
  cmp gpurun_out/conv_bench.bin gpurun_out/conv_bench_hoist.bin > /dev/null && echo "  bits: identical" || echo "  bits: DIFFER"
done <<'LAYERS'
64 64 64 192 192 9 0 96 1 3 1
64 64 64 192 192 9 0 96 1 3 2 0 0 1
64 64 64 384 384 9 0 128 1 3 2 0 0 1
64 64 64 384 192 9 0 96 1 3 1
64 32 32 384 384 9 0 128 1 3 1
64 16 16 576 576 9 0 96 1 3 2 0 0 1
LAYERS
rm -f gpurun_out/conv_bench*.bin
