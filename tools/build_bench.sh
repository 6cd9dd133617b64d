#!/bin/bash
# Builds tools/conv_bench.out from separately compiled objects, so that a change to one kernel family does not recompile the others:
#   tools/build_bench.sh [suffix] [extra hipcc flags ...]     e.g.  tools/build_bench.sh _trace -DTD_TRACE
# Objects live in tools/obj/ (git-ignored); the base object (conv_igemm + conv_glds: ~4 minutes) is rebuilt only when its sources are newer.
cd "$(dirname "$0")/.."
sfx=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -Iterrain_diffusion_amd/csrc $*"
mkdir -p tools/obj
C=terrain_diffusion_amd/csrc
newer() { [ ! -e "$1" ] && return 0; for f in "${@:2}"; do [ "$f" -nt "$1" ] && return 0; done; return 1; }
if newer tools/obj/base$sfx.o $C/conv_igemm.hip $C/conv_glds.hip $C/conv_common.h $C/td_device.h tools/bench_base.hip; then hipcc $F -c tools/bench_base.hip -o tools/obj/base$sfx.o || exit 1; fi
if newer tools/obj/wide$sfx.o $C/conv_glds_wide.hip $C/conv_common.h $C/td_device.h tools/bench_wide.hip; then hipcc $F -c tools/bench_wide.hip -o tools/obj/wide$sfx.o || exit 1; fi
hipcc $F -DTD_BENCH_EXTERN -c tools/conv_bench.hip -o tools/obj/main$sfx.o || exit 1
hipcc --offload-arch=gfx950 tools/obj/main$sfx.o tools/obj/base$sfx.o tools/obj/wide$sfx.o -o tools/conv_bench$sfx.out
