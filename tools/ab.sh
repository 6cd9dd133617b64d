#!/bin/bash
# One A/B driver for engine / plan options and for single conv layers.  It replaces the ~45 one-off scripts of rounds 1-3 (ab_conv, ablate,
# b1_*, bn64_try, dma1x1_ab*, dual_ab, epi_abl, occ*_try, opt_ab, pp_*, ps_*, splitk_weighted_ab, stagger_try*, walk_ab, xcd_try, xseg_ab ...):
# every one of them was "run X under option set A and under option set B, interleaved, and print one line each".
#
#   tools/ab.sh bench  [bench.py args ...] -- "<optsA>" "<optsB>" ...     value / ms_per_step of bench.py under each option set, two rounds interleaved
#   tools/ab.sh perop  N dtype              -- "<optsA>" "<optsB>" ...     kernel time per forward + the five slowest ops (tools/profile_ops.py)
#   tools/ab.sh hash                        -- "<optsA>" "<optsB>" ...     sha256 of the network output at batch 64 / 5 / 1 (tools/out_hash.py)
#   tools/ab.sh layer  BIN                  -- "<args A>" "<args B>" ...   one conv layer through tools/conv_bench.out or tools/sb_bench.out
# Option sets are engine options "k=v,k=v" ("" = defaults).  Ablation builds of the conv kernels are separate binaries:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTD_ABLATE_x -Iterrain_diffusion_amd/csrc tools/conv_bench.hip -o tools/cb_x.out   (x: BLOAD BARRIER BSTORE EPI)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mode=$1; shift
pre=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do pre+=("$1"); shift; done
shift
rounds=${AB_ROUNDS:-2}
for r in $(seq $rounds); do
  for o in "$@"; do
    case $mode in
      bench) echo -n "[$o] "; timeout 900 python bench.py --no-cpu-baseline --no-kernel-profile --no-latency "${pre[@]}" ${o:+--engine-opts "$o"} 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" ;;
      perop) echo "[$o]"; TD_OPTS="$o" TD_TOP=${TD_TOP:-5} timeout 300 python tools/profile_ops.py "${pre[@]}" 2>/dev/null | grep -v amdgpu.ids ;;
      hash)  echo -n "[$o] "; TD_OPTS="$o" timeout 300 python tools/out_hash.py 2>/dev/null | tail -1 ;;
      layer) echo "[$o]"; timeout 300 "${pre[0]}" $o ;;
      *) echo "usage: see the header of tools/ab.sh"; exit 2 ;;
    esac
  done
done
