// The wide tile of the LDS-DMA flavour as its own object for tools/conv_bench.hip (see tools/build_bench.sh).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "conv_glds_wide.hip"
