/* A plain-C host of include/td_engine.h (tests/test_abi_host.py compiles it with gcc -std=c99 and links libtd_engine.so): the host-side entry
 * points answer without a GPU (version, build id, _tile_seed), and td_engine_create either returns an engine (GPU box: one weight window is
 * computed and printed) or fails LOUDLY with a message — there is no CPU fallback behind the C-ABI either. */
#include <inttypes.h>
#include <stdio.h>

#include "td_engine.h"

int main(void) {
    printf("version %d\n", td_version());
    printf("build %s\n", td_build_id());
    printf("seed %" PRIu64 "\n", td_tile_seed(5861u, -3, 7));
    td_engine* e = NULL;
    const int rc = td_engine_create(0, &e);
    if (rc != TD_OK) {
        printf("engine_create %d: %s\n", rc, td_last_error());
        return 0;
    }
    float w[8];
    const int rc2 = td_linear_weight_window(e, 8, w);
    printf("engine_create 0\nwindow %d: %.6f %.6f %.6f %.6f\n", rc2, (double)w[0], (double)w[1], (double)w[2], (double)w[3]);
    td_engine_destroy(e);
    return 0;
}
