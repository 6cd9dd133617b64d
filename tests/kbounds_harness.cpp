// Host-only harness for tests/test_kbounds_cpu.py: prints the split-K slice bounds conv_set_kbounds (csrc/td_device.h) computes.
// args: weighted ksplit chunk (C taps)...
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "td_device.h"
using namespace td;
int main(int argc, char** argv) {
    // args: weighted ksplit chunk (C taps)...
    ConvParams p; memset(&p, 0, sizeof p);
    int weighted = atoi(argv[1]); p.ksplit = atoi(argv[2]); int chunk = atoi(argv[3]);
    int n = 0;
    for (int i = 4; i + 1 < argc; i += 2) { p.seg[p.nseg].C = atoi(argv[i]); p.seg[p.nseg].taps = atoi(argv[i + 1]); n += p.seg[p.nseg].C / chunk; ++p.nseg; }
    p.kgroups = n;
    bool ok = conv_set_kbounds(p, weighted != 0, chunk);
    printf("%d", ok ? 1 : 0);
    if (ok) for (int s = 0; s <= p.ksplit; ++s) printf(" %d", p.kb[s]);
    printf("\n");
    return 0;
}
