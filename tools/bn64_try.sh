#!/bin/bash
# 64-cout-granular layers (decoder 512x512 / 256x256 levels): per-tap flavour (fl 0, bn 64) vs the LDS-DMA flavour with bn 64 (fl 2 = 8 waves, fl 3 = 4 waves)
cd $GRAFT_REPO_ROOT/tools
for shape in "4 512 512 64 64 9 2 64" "4 512 512 128 64 9 0 64" "4 512 512 64 64 9 0 64" "16 64 64 192 64 9 0 64"; do
  for epi in 1 2; do
    for fl in 0 2 3; do timeout 60 ./conv_bench.out $shape 1 $fl $epi 2>&1 | grep -v check; done
  done
done
cd ..
echo "== decoder forward, batch 4, 512x512 (profile_model)"; 
for o in "glds_bn64=0" "glds_bn64=1" "glds_bn64=1,glds_variant=1" "glds_bn64=1,glds_variant=0"; do echo "[$o]"; TD_OPTS="$o" timeout 200 python tools/profile_model.py decoder 4 512 bf16 2>/dev/null | head -12; done
for o in "glds_bn64=0" "glds_bn64=1"; do echo "[$o]"; TD_OPTS="$o" timeout 200 python tools/profile_ops.py 64 bf16 2>/dev/null | head -1; done
