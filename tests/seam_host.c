/* A plain-C host of include/td_seam.h (tests/test_seam_cpu.py compiles it with gcc -std=c99 and links libtd_seam.so): prints the shard plan of one
 * rank — mesh, region, the four window lists and the message cuts — in a fixed text form the test compares with parallel.ShardPlan.  Host
 * arithmetic only: no GPU, no RCCL call.  usage: seam_host H W tile stride world extended rank window_bytes */
#include <stdio.h>
#include <stdlib.h>

#include "td_seam.h"

static int die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, td_seam_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 9) return 2;
    const int H = atoi(argv[1]), W = atoi(argv[2]), tile = atoi(argv[3]), stride = atoi(argv[4]), world = atoi(argv[5]), extended = atoi(argv[6]),
              rank = atoi(argv[7]);
    const long long wb = atoll(argv[8]);
    td_seam_plan* plan = NULL;
    if (td_seam_plan_create(H, W, tile, stride, world, extended, &plan) != TD_SEAM_OK) return die("td_seam_plan_create");
    int32_t mesh[4], region[4];
    if (td_seam_plan_mesh(plan, mesh) != TD_SEAM_OK || td_seam_plan_region(plan, rank, region) != TD_SEAM_OK) return die("mesh/region");
    printf("mesh %d %d %d %d\n", mesh[0], mesh[1], mesh[2], mesh[3]);
    printf("region %d %d %d %d\n", region[0], region[1], region[2], region[3]);
    static const char* names[4] = {"own", "needed", "sends", "recvs"};
    const int cap = mesh[2] * mesh[3] * (world > 1 ? world : 1);
    int32_t* ij = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    int32_t* peer = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    td_seam_msg* s = (td_seam_msg*)malloc(sizeof(td_seam_msg) * (size_t)cap);
    td_seam_msg* r = (td_seam_msg*)malloc(sizeof(td_seam_msg) * (size_t)cap);
    for (int kind = TD_SEAM_OWN; kind <= TD_SEAM_RECVS; ++kind) {
        const int n = td_seam_plan_windows(plan, rank, kind, ij, peer, cap);
        if (n < 0) return die("td_seam_plan_windows");
        printf("%s %d:", names[kind], n);
        for (int k = 0; k < n; ++k) printf(" %d,%d@%d", ij[2 * k], ij[2 * k + 1], peer[k]);
        printf("\n");
    }
    int ns = 0, nr = 0;
    if (td_seam_plan_messages(plan, rank, wb, s, &ns, r, &nr, cap) != TD_SEAM_OK) return die("td_seam_plan_messages");
    printf("send_msgs %d:", ns);
    for (int k = 0; k < ns; ++k) printf(" %d:%lld+%lld", s[k].peer, (long long)s[k].offset, (long long)s[k].bytes);
    printf("\nrecv_msgs %d:", nr);
    for (int k = 0; k < nr; ++k) printf(" %d:%lld+%lld", r[k].peer, (long long)r[k].offset, (long long)r[k].bytes);
    printf("\n");
    free(ij); free(peer); free(s); free(r);
    td_seam_plan_destroy(plan);
    return 0;
}
