// The slow-to-compile kernel families of tools/conv_bench.hip as one object (see tools/build_bench.sh): the per-tap and the LDS-DMA flavours.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "conv_igemm.hip"
#include "conv_glds.hip"
