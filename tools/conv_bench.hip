// Standalone micro-benchmark of td::conv_igemm_kernel on one synthetic layer (random data), for ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I terrain_diffusion_amd/csrc tools/conv_bench.hip -o tools/conv_bench.out
//   ./conv_bench.out N H W Cin Cout taps xform bn [ksplit]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "conv_igemm.hip"
#include "conv_glds.hip"
using namespace td;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, W = argc > 3 ? atoi(argv[3]) : 64;
    int Cin = argc > 4 ? atoi(argv[4]) : 192, Cout = argc > 5 ? atoi(argv[5]) : 192, taps = argc > 6 ? atoi(argv[6]) : 9;
    int xform = argc > 7 ? atoi(argv[7]) : 0, bn = argc > 8 ? atoi(argv[8]) : 64, ksplit = argc > 9 ? atoi(argv[9]) : 1, flavor = argc > 10 ? atoi(argv[10]) : 0;
    const int chunk = 64;
    size_t M = (size_t)N * H * W;
    int kgroups = Cin / chunk, ksteps = kgroups * taps;
    void *x, *w, *out; float* partial = nullptr;
    CK(hipMalloc(&x, M * Cin * 2)); CK(hipMalloc(&w, (size_t)ksteps * Cout * 128 + 8192)); CK(hipMalloc(&out, M * Cout * 2));
    std::vector<uint16_t> hx(M * Cin), hw((size_t)ksteps * Cout * 64);
    srand(1);
    for (auto& v : hx) v = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);          // ~ +-0.5..1
    for (auto& v : hw) v = 0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15);          // small
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    if (ksplit > 1) CK(hipMalloc(&partial, (size_t)ksplit * M * Cout * 4));
    ConvParams p; memset(&p, 0, sizeof p);
    p.nseg = 1; p.seg[0].src = x; p.seg[0].C = Cin; p.seg[0].cstride = Cin; p.seg[0].Hs = H; p.seg[0].Ws = W; p.seg[0].taps = taps; p.seg[0].xform = xform; p.seg[0].scale = 1.f;
    p.wpack = w; p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.CoutPad = Cout; p.kgroups = kgroups; p.ksplit = ksplit; p.partial = partial;
    bool narrow = W < 16; int TW = narrow ? 8 : 16, NIMG = narrow ? (flavor == 2 ? 4 : 2) : 1; int TH = (flavor == 2 && !narrow) ? 16 : 8;
    p.tiles_x = (W + TW - 1) / TW; p.tiles_y = (H + TH - 1) / TH; p.img_groups = (N + NIMG - 1) / NIMG; p.n_ntiles = Cout / bn;
    p.epi = EPI_PLAIN; p.out = out; p.out_cstride = Cout;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK((flavor >= 2 ? launch_conv_glds(p, narrow, bn, flavor - 2, st) : launch_conv(p, true, narrow, bn, flavor, st)));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK((flavor >= 2 ? launch_conv_glds(p, narrow, bn, flavor - 2, st) : launch_conv(p, true, narrow, bn, flavor, st)));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double flop = 2.0 * M * Cout * Cin * taps;
    printf("N%d %dx%d Cin%d Cout%d taps%d xform%d bn%d ks%d fl%d : %.1f us  %.1f TFLOP/s (%.1f%% of 2500)  wgs=%d\n", N, H, W, Cin, Cout, taps, xform, bn, ksplit, flavor, ms * 1e3,
           flop / ms / 1e9, flop / ms / 1e9 / 25.0, p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups * ksplit);
    return 0;
}
