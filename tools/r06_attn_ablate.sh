#!/bin/bash
# round 6: ablation of the pipelined attention loop.  tools/attn_abl<bits>.out = tools/attn_bench.hip built with -DTD_ATTN_ABL=<bits> (built here when missing: hipcc cross-compiles
# without a GPU); results are WRONG by design, only the time counts.  bit 0 (1) no barrier in the loop, bit 1 (2) no staging (global loads + LDS writes), bit 3 (8) no S MFMAs, bit 4 (16) no PV MFMAs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for a in 0 1 2 3 8 16 24 27; do
  [ -x tools/attn_abl$a.out ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTD_ATTN_ABL=$a -Iterrain_diffusion_amd/csrc tools/attn_bench.hip -o tools/attn_abl$a.out 2>/dev/null &
done; wait
O=gpurun_out/r06_attn_ablate.txt; : > $O
for shape in "2 8 4096 4096 64" "2 8 4096 4096 40"; do
for rep in 1 2; do
for a in 0 1 2 3 8 16 24 27; do
  echo "## ABL=$a $shape : $(timeout 60 tools/attn_abl$a.out $shape 50 2>&1 | head -1)" >> $O
done; done; done
cat $O
