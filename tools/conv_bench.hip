// Standalone micro-benchmark of td::conv_igemm_kernel on one synthetic layer (random data), for ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I terrain_diffusion_amd/csrc tools/conv_bench.hip -o tools/conv_bench.out
//   ./conv_bench.out N H W Cin Cout taps xform bn [ksplit]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#include <array>
#ifdef TD_BENCH_EXTERN   // kernels linked from separately compiled objects (tools/build_bench.sh: the conv_glds family takes minutes to compile, the flavour under work seconds)
#include <algorithm>
#include "conv_common.h"
namespace td {
hipError_t launch_conv(const ConvParams& p, int dtype, bool narrow, int bn, int ksplit_variant, hipStream_t st);
hipError_t launch_conv_glds(const ConvParams& p, int dtype, bool narrow, int bn, int variant, hipStream_t st);
hipError_t launch_conv_glds_wide(const ConvParams& p, int dtype, int bn, hipStream_t st);
extern int g_bench_extra_lds;
}
#else
#include "conv_igemm.hip"
#include "conv_glds.hip"
#include "conv_glds_wide.hip"
#endif
using namespace td;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void __launch_bounds__(256) bench_prefetch_kernel(const uint4* __restrict__ w, size_t n16, uint4* sink) {
    uint4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = w[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x & acc.y & acc.z & acc.w) == 0xfffffffeu && (acc.x ^ acc.y) == 0x12345678u) sink[threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, W = argc > 3 ? atoi(argv[3]) : 64;
    int Cin = argc > 4 ? atoi(argv[4]) : 192, Cout = argc > 5 ? atoi(argv[5]) : 192, taps = argc > 6 ? atoi(argv[6]) : 9;
    int xform = argc > 7 ? atoi(argv[7]) : 0, bn = argc > 8 ? atoi(argv[8]) : 64, ksplit = argc > 9 ? atoi(argv[9]) : 1, flavor = argc > 10 ? atoi(argv[10]) : 0, epi = argc > 11 ? atoi(argv[11]) : 0;
    const int stagger = argc > 12 ? atoi(argv[12]) : 0, chain = argc > 13 ? atoi(argv[13]) : 0, want_out2 = argc > 14 ? atoi(argv[14]) : 0;  // chain: 1 = x->y->x same walk order, 2 = second layer walks backwards
    const int chunk = 64;
    size_t M = (size_t)N * H * W;
    int kgroups = Cin / chunk, ksteps = kgroups * taps;
    // TD_SEG2="Cin2,taps2": a second K-segment with its own source tensor (e.g. the decoder's fused 1x1 skip conv: TD_SEG2=576,1); TD_DMA1X1=0|1 picks
    // the register-staged or the LDS-DMA path for 1x1 segments (conv_glds flavours), and both are compared bit for bit after the timing
    int Cin2 = 0, taps2 = 1;
    if (getenv("TD_SEG2")) { if (sscanf(getenv("TD_SEG2"), "%d,%d", &Cin2, &taps2) != 2 || Cin2 % chunk || (taps2 != 1 && taps2 != 9)) { printf("bad TD_SEG2\n"); return 1; } }
    kgroups += Cin2 / chunk; ksteps += Cin2 / chunk * taps2;
    void *x, *w, *out; float* partial = nullptr;
    CK(hipMalloc(&x, M * Cin * 2)); CK(hipMalloc(&w, (size_t)(ksteps + 2) * Cout * 128 + 16384)); CK(hipMalloc(&out, M * Cout * 2));
    std::vector<uint16_t> hx(M * Cin), hw((size_t)ksteps * Cout * 64);
    unsigned long long rs_ = 0x9E3779B97F4A7C15ull;   // xorshift64* (the libc generator cost 2-4 s of box time per invocation at 10^8 elements)
    auto rnd = [&]() -> unsigned { rs_ ^= rs_ >> 12; rs_ ^= rs_ << 25; rs_ ^= rs_ >> 27; return (unsigned)((rs_ * 0x2545F4914F6CDD1Dull) >> 40); };
    for (auto& v : hx) { const unsigned r_ = rnd(); v = 0x3f00 + (r_ & 0xff) + (((r_ >> 8) & 1) << 15); }          // ~ +-0.5..1
    for (auto& v : hw) { const unsigned r_ = rnd(); v = 0x3c00 + (r_ & 0xff) + (((r_ >> 8) & 1) << 15); }          // small
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    if (ksplit > 1) CK(hipMalloc(&partial, (size_t)ksplit * M * Cout * 4));
#ifdef TD_TRACE
    const size_t trace_n = (size_t)65536 * 12 * 16;
    CK(hipMalloc(&partial, trace_n * 8)); CK(hipMemset(partial, 0, trace_n * 8));
#endif
    ConvParams p; memset(&p, 0, sizeof p);
    p.nseg = 1; p.seg[0].src = x; p.seg[0].C = Cin; p.seg[0].cstride = Cin; p.seg[0].Hs = H; p.seg[0].Ws = W; p.seg[0].taps = taps; p.seg[0].xform = xform; p.seg[0].scale = 1.f;
    p.wpack = w; p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.CoutPad = Cout; p.kgroups = kgroups; p.ksplit = ksplit; p.partial = partial;
    if (Cin2) {
        void* x2; CK(hipMalloc(&x2, M * Cin2 * 2));
        std::vector<uint16_t> hx2(M * Cin2); for (auto& v : hx2) { const unsigned r_ = rnd(); v = 0x3f00 + (r_ & 0xff) + (((r_ >> 8) & 1) << 15); }
        CK(hipMemcpy(x2, hx2.data(), hx2.size() * 2, hipMemcpyHostToDevice));
        p.nseg = 2; p.seg[1].src = x2; p.seg[1].C = Cin2; p.seg[1].cstride = Cin2; p.seg[1].Hs = H; p.seg[1].Ws = W; p.seg[1].taps = taps2; p.seg[1].xform = 0; p.seg[1].scale = 1.f;
    }
    p.dma1x1 = getenv("TD_DMA1X1") ? atoi(getenv("TD_DMA1X1")) : 1;
    bool narrow = W < 16; int TW = narrow ? 8 : 16, NIMG = narrow ? ((flavor == 2 || flavor == 4) ? 4 : 2) : 1; const bool pers = flavor == 10; if (pers) flavor = 9;   /* 10 = the wide tile's persistent tile loop (p.persist workgroups; TD_PERSIST=G overrides 2 x 256 slots' rule) */ int TH = flavor == 8 ? 4 : ((flavor == 2 || flavor == 4 || flavor == 9) && !narrow) ? 16 : 8;  // flavor 8 = conv_glds variant 2 (tiny tile); 9 = conv_glds_wide.hip (256 px x bn, 4 waves, two workgroups per CU)
    p.tiles_x = (W + TW - 1) / TW; p.tiles_y = (H + TH - 1) / TH; p.img_groups = (N + NIMG - 1) / NIMG; p.n_ntiles = Cout / bn;
    p.epi = epi; p.out = out; p.out_cstride = Cout;
    if (pers) { const long long tiles = (long long)p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups, slots = 512, rounds = (tiles + slots - 1) / slots; long long g = (tiles + rounds - 1) / rounds; if ((tiles & 7) == 0) g = (g + 7) & ~7LL;
        p.persist = getenv("TD_PERSIST") ? atoi(getenv("TD_PERSIST")) : (int)std::min(g, tiles); }
    { void* z; CK(hipMalloc(&z, 4096)); CK(hipMemset(z, 0, 4096)); p.zeros = z; }
    if (epi == EPI_EMB_SILU) { float* cv; CK(hipMalloc(&cv, (size_t)N * Cout * 4)); std::vector<float> hc((size_t)N * Cout); for (size_t i_ = 0; i_ < hc.size(); ++i_) hc[i_] = 0.75f + 0.001f * (float)(i_ % 509);   /* varied: an indexing slip in the kernel's modulation-row staging must change bits */ CK(hipMemcpy(cv, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); p.cvec = cv; p.cvec_stride = Cout; }
    if (epi == EPI_RESIDUAL) { void* r; float* ssq; CK(hipMalloc(&r, M * Cout * 2)); CK(hipMemcpy(r, hx.data(), std::min(hx.size(), M * Cout) * 2, hipMemcpyHostToDevice)); CK(hipMalloc(&ssq, M * (Cout / 32 + 8) * 4)); CK(hipMemset(ssq, 0, M * (Cout / 32 + 8) * 4));
        p.res = r; p.res_cstride = Cout; p.res_Hs = H; p.res_Ws = W; p.res_scale = 0.9f; p.clip = 256.f; p.out_sumsq = ssq; }
    (void)stagger;  // round-3 stagger experiments are recorded in profiles/r03_conv_walk_order_and_stagger.txt; the hook is gone from the kernel
    if (want_out2) { void* o2; CK(hipMalloc(&o2, M * Cout * 2)); p.out2 = o2; p.out2_scale = 1.f; }
    if (!conv_set_kbounds(p, true)) { printf("bad split-K\n"); return 1; }
    ConvParams p2 = p;
    if (chain) { if (Cin != Cout) { printf("chain needs Cin == Cout\n"); return 1; } p2.seg[0].src = out; p2.out = x; p2.reverse = chain == 2 ? 1 : 0; }
    if (getenv("TD_EXTRA_LDS")) g_bench_extra_lds = atoi(getenv("TD_EXTRA_LDS"));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto L = [&](const ConvParams& q) { return flavor == 9 ? launch_conv_glds_wide(q, 1, bn, st) : flavor == 8 ? launch_conv_glds(q, 1, narrow, bn, 2, st) : flavor >= 2 ? launch_conv_glds(q, 1, narrow, bn, flavor - 2, st) : launch_conv(q, 1, narrow, bn, flavor, st); };
    for (int i = 0; i < 4; ++i) CK(L((chain && (i & 1)) ? p2 : p));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(L((chain && (i & 1)) ? p2 : p));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    if (getenv("TD_COLD")) {  // cold-cache timing: evict L2 + MALL (1 GiB memset) before every launch, time each launch by itself
        void* fl; const size_t fb = (size_t)1 << 30; CK(hipMalloc(&fl, fb)); float acc = 0.f;
        const int mode = atoi(getenv("TD_COLD"));   // 2: weights pulled back through the memory-side cache by a prefetch kernel before the timed launch; 3: activations too
        size_t wbytes = 0; for (int s_ = 0; s_ < p.nseg; ++s_) wbytes += (size_t)(p.seg[s_].C / 64) * p.seg[s_].taps * p.CoutPad * 128;
        for (int i = 0; i < reps; ++i) { CK(hipMemsetAsync(fl, i, fb, st));
            if (mode >= 2) hipLaunchKernelGGL(bench_prefetch_kernel, dim3(128), dim3(256), 0, st, (const uint4*)p.wpack, wbytes / 16, (uint4*)fl);
            if (mode >= 3) hipLaunchKernelGGL(bench_prefetch_kernel, dim3(128), dim3(256), 0, st, (const uint4*)p.seg[0].src, (size_t)M * Cin * 2 / 16, (uint4*)fl);
            CK(hipEventRecord(e0, st)); CK(L(p)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); acc += t; }
        printf("   cold caches (mode %d): %.1f us (hot loop above: %.1f us)\n", mode, acc / reps * 1e3, ms * 1e3); CK(hipFree(fl));
    }
    if (const char* df = getenv("TD_DUMP")) {   // the output of one launch, for comparing two builds of the kernels bit by bit (cmp)
        CK(hipMemset(out, 0, M * Cout * 2)); CK(L(p)); CK(hipStreamSynchronize(st));
        std::vector<uint16_t> ho(M * Cout); CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        FILE* fp = fopen(df, "wb"); if (fp) { fwrite(ho.data(), 2, ho.size(), fp);
            if (p.out2) { CK(hipMemcpy(ho.data(), p.out2, ho.size() * 2, hipMemcpyDeviceToHost)); fwrite(ho.data(), 2, ho.size(), fp); }   // the pre-activated second output
            if (p.out_sumsq) { std::vector<float> hs(M * (Cout / 32)); CK(hipMemcpy(hs.data(), p.out_sumsq, hs.size() * 4, hipMemcpyDeviceToHost)); fwrite(hs.data(), 4, hs.size(), fp); }   // and the pixel-norm partials
            fclose(fp); }
    }
    double flop = 2.0 * M * Cout * ((double)Cin * taps + (double)Cin2 * taps2);
    printf("N%d %dx%d Cin%d Cout%d taps%d xform%d bn%d ks%d fl%d epi%d stg%d ch%d : %.1f us  %.1f TFLOP/s (%.1f%% of 2500)  wgs=%d%s\n", N, H, W, Cin, Cout, taps, xform, bn, ksplit, pers ? 10 : flavor, epi, stagger, chain, ms * 1e3,
           flop / ms / 1e9, flop / ms / 1e9 / 25.0, p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups * ksplit, pers ? (" persistent on " + std::to_string(p.persist)).c_str() : "");
    if (pers && !getenv("TD_NO_CMP")) {   // the persistent tile loop against one tile per workgroup: same K order -> out, second output and sum-of-squares planes bit for bit
        std::vector<uint16_t> o[2], o2[2]; std::vector<float> ss[2];
        for (int d = 0; d < 2; ++d) {
            ConvParams q = p; if (d == 0) q.persist = 0;
            CK(hipMemset(out, 0, M * Cout * 2)); if (p.out2) CK(hipMemset(p.out2, 0, M * Cout * 2)); if (p.out_sumsq) CK(hipMemset(p.out_sumsq, 0, M * (Cout / 32) * 4));
            CK(L(q)); CK(hipStreamSynchronize(st));
            o[d].resize(M * Cout); CK(hipMemcpy(o[d].data(), out, M * Cout * 2, hipMemcpyDeviceToHost));
            if (p.out2) { o2[d].resize(M * Cout); CK(hipMemcpy(o2[d].data(), p.out2, M * Cout * 2, hipMemcpyDeviceToHost)); }
            if (p.out_sumsq) { ss[d].resize(M * (Cout / 32)); CK(hipMemcpy(ss[d].data(), p.out_sumsq, ss[d].size() * 4, hipMemcpyDeviceToHost)); }
        }
        size_t bad = 0, bad2 = 0, bads = 0, nz = 0;
        for (size_t i = 0; i < o[0].size(); ++i) { bad += o[0][i] != o[1][i]; nz += (o[1][i] & 0x7fff) != 0; }
        for (size_t i = 0; i < o2[0].size(); ++i) bad2 += o2[0][i] != o2[1][i];
        for (size_t i = 0; i < ss[0].size(); ++i) bads += memcmp(&ss[0][i], &ss[1][i], 4) != 0;
        if (bad && getenv("TD_DBG")) { size_t shown = 0; std::vector<size_t> hy(16, 0), hx(16, 0), hc(4, 0), ht(64, 0);
            for (size_t i = 0; i < o[0].size(); ++i) if (o[0][i] != o[1][i]) { const size_t pix = i / Cout; const int c = (int)(i % Cout), x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
                hy[y & 15]++; hx[x & 15]++; hc[(c / 16) & 3]++; ht[((y >> 4) & 7) * 8 + ((x >> 4) & 7)]++;
                if (shown++ < 12) printf("    n %d y %d x %d c %d: %04x vs %04x\n", n, y, x, c, o[0][i], o[1][i]); }
            printf("    by y&15:"); for (auto v : hy) printf(" %zu", v); printf("\n    by x&15:"); for (auto v : hx) printf(" %zu", v); printf("\n    by (c/16)&3:"); for (auto v : hc) printf(" %zu", v);
            printf("\n    by tile (ty&7, tx&7):"); for (auto v : ht) printf(" %zu", v); printf("\n"); }
        printf("  check persistent vs one tile per workgroup: out %zu / %zu differ (nonzero %zu), out2 %zu / %zu, sumsq %zu / %zu\n", bad, o[0].size(), nz, bad2, o2[0].size(), bads, ss[0].size());
    }
    if (Cin2 && (flavor == 2 || flavor == 3 || flavor == 8)) {  // register-staged vs LDS-DMA 1x1 path of conv_glds: same K order, same MFMA -> same bits
        std::vector<uint16_t> o0(M * Cout), o1(M * Cout);
        for (int d = 0; d < 2; ++d) {
            ConvParams q = p; q.dma1x1 = d;
            CK(hipMemset(out, 0, M * Cout * 2));
            CK(L(q)); CK(hipStreamSynchronize(st));
            CK(hipMemcpy((d ? o1 : o0).data(), out, o0.size() * 2, hipMemcpyDeviceToHost));
        }
        size_t bad = 0, nz = 0; for (size_t i = 0; i < o0.size(); ++i) { bad += o0[i] != o1[i]; nz += (o1[i] & 0x7fff) != 0; }
        printf("  check dma1x1 1 vs 0: %zu / %zu outputs differ, nonzero outputs %zu%s\n", bad, o0.size(), nz, ksplit > 1 ? "  (split-K: `out` is written by the reduce launch)" : "");
    }
    if (flavor == 9 && !getenv("TD_NO_CMP")) {   // the wide tile against the 128-pixel tile of conv_glds: same K order, same MFMA -> out, second output and sum-of-squares planes bit for bit
        const int bn0 = (Cout % bn == 0 && (bn == 96 || bn == 128 || bn == 64)) ? bn : 96;
        std::vector<uint16_t> o[2], o2[2]; std::vector<float> ss[2];
        for (int d = 0; d < 2; ++d) {
            ConvParams q = p;
            if (d == 0) { q.n_ntiles = Cout / bn0; q.tiles_y = (H + 7) / 8; }
            CK(hipMemset(out, 0, M * Cout * 2)); if (p.out2) CK(hipMemset(p.out2, 0, M * Cout * 2)); if (p.out_sumsq) CK(hipMemset(p.out_sumsq, 0, M * (Cout / 32) * 4));
            CK(d == 0 ? launch_conv_glds(q, 1, narrow, bn0, 1, st) : L(q)); CK(hipStreamSynchronize(st));
            o[d].resize(M * Cout); CK(hipMemcpy(o[d].data(), out, M * Cout * 2, hipMemcpyDeviceToHost));
            if (p.out2) { o2[d].resize(M * Cout); CK(hipMemcpy(o2[d].data(), p.out2, M * Cout * 2, hipMemcpyDeviceToHost)); }
            if (p.out_sumsq) { ss[d].resize(M * (Cout / 32)); CK(hipMemcpy(ss[d].data(), p.out_sumsq, ss[d].size() * 4, hipMemcpyDeviceToHost)); }
        }
        // another K order (channel half outside the taps): fp32 sums rounded differently -> the bf16 outputs may differ in the last place; report how many do,
        // by how many bf16 ulps at most, and the relative RMS difference (a wrong tap / channel / cout mapping shows as O(1))
        auto bf = [](uint16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
        auto cmp16 = [&](const std::vector<uint16_t>& a, const std::vector<uint16_t>& b, const char* name) {
            size_t bad = 0, nz = 0; int maxulp = 0; double num = 0, den = 0;
            for (size_t i = 0; i < a.size(); ++i) { nz += (b[i] & 0x7fff) != 0; if (a[i] != b[i]) { ++bad; const int d = abs((int)(a[i] & 0x7fff) - (int)(b[i] & 0x7fff)); maxulp = std::max(maxulp, ((a[i] ^ b[i]) & 0x8000) ? 9999 : d); }
                const double x = bf(a[i]), y = bf(b[i]); num += (x - y) * (x - y); den += x * x; }
            printf("  check wide vs conv_glds bn%d small, %s: %zu / %zu differ (nonzero %zu), max %d bf16 ulp, rel-RMS %.3e\n", bn0, name, bad, a.size(), nz, maxulp, den > 0 ? sqrt(num / den) : 0.0);
        };
        cmp16(o[0], o[1], "out");
        if (!o2[0].empty()) cmp16(o2[0], o2[1], "out2");
        if (!ss[0].empty()) { double num = 0, den = 0; for (size_t i = 0; i < ss[0].size(); ++i) { num += ((double)ss[0][i] - ss[1][i]) * ((double)ss[0][i] - ss[1][i]); den += (double)ss[0][i] * ss[0][i]; }
            printf("  check wide vs conv_glds, sumsq planes: rel-RMS %.3e\n", den > 0 ? sqrt(num / den) : 0.0); }
    }
    if (getenv("TD_CMP_BN") && (flavor == 2 || flavor == 3)) {   // same K order, same MFMA: another cout tile width must give the same bits
        const int bn0 = atoi(getenv("TD_CMP_BN"));
        std::vector<uint16_t> o0(M * Cout), o1(M * Cout);
        CK(hipMemset(out, 0, M * Cout * 2)); CK(L(p)); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(o1.data(), out, o1.size() * 2, hipMemcpyDeviceToHost));
        ConvParams q = p; q.n_ntiles = Cout / bn0; q.tiles_y = (H + 7) / 8;
        CK(hipMemset(out, 0, M * Cout * 2)); CK(launch_conv_glds(q, 1, narrow, bn0, 1, st)); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(o0.data(), out, o0.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, nz = 0; for (size_t i = 0; i < o0.size(); ++i) { bad += o0[i] != o1[i]; nz += (o1[i] & 0x7fff) != 0; }
        printf("  check vs bn%d small: %zu / %zu outputs differ, nonzero outputs %zu\n", bn0, bad, o0.size(), nz);
    }
#ifdef TD_TRACE
    {
        int wgs = p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups; const int nw = flavor == 4 ? 12 : ((flavor == 3 || flavor == 9) ? 4 : 8), TS = 16;
        std::vector<unsigned long long> tb((size_t)wgs * nw * TS);
        CK(hipMemcpy(tb.data(), partial, tb.size() * 8, hipMemcpyDeviceToHost));
        double s[5] = {0, 0, 0, 0, 0}; unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < wgs * nw; ++i) { for (int j = 0; j < 5; ++j) s[j] += (double)tb[(size_t)i * TS + j]; if (tb[(size_t)i*TS+5] && tb[(size_t)i*TS+5] < t0) t0 = tb[(size_t)i*TS+5]; if (tb[(size_t)i*TS+6] > t1) t1 = tb[(size_t)i*TS+6]; }
        {   // per-CU timeline from the 100 MHz realtime stamps of wave 0 of every workgroup: idle gap between consecutive workgroups
            std::vector<std::array<unsigned long long, 3>> ev;
            for (int w = 0; w < wgs; ++w) { const unsigned long long* t = &tb[(size_t)w * nw * TS]; const unsigned long long id = t[9]; const unsigned long long cu = ((id >> 8) & 0xff00) >> 8 | (((id >> 8) >> 13) & 7) << 4 | (id & 15) << 8; ev.push_back({cu, t[8], t[10]}); }
            std::sort(ev.begin(), ev.end());
            double gap = 0, busy = 0; int ngap = 0, ncu = 0; unsigned long long first = ~0ull, last = 0, maxper = 0, cnt = 0;
            for (size_t i = 0; i < ev.size(); ++i) { busy += (double)(ev[i][2] - ev[i][1]); first = std::min(first, ev[i][1]); last = std::max(last, ev[i][2]);
                if (i == 0 || ev[i][0] != ev[i-1][0]) { ++ncu; cnt = 1; } else { gap += (double)ev[i][1] - (double)ev[i-1][2]; ++ngap; ++cnt; } maxper = std::max(maxper, cnt); }
            printf("  timeline: %d distinct CUs, max %llu WGs on one CU, span %.1f us, mean WG %.2f us, mean idle gap between consecutive WGs on a CU %.2f us\n", ncu, maxper, (last - first) / 100.0, busy / wgs / 100.0, ngap ? gap / ngap / 100.0 : 0.0);
        }
        double clk = 0; for (int i = 0; i < wgs * nw; ++i) clk += 100.0 * (double)(tb[(size_t)i*TS+6] - tb[(size_t)i*TS+5]) / (double)tb[(size_t)i*TS+7];
        printf("  shader clock during the kernel (s_memtime / s_memrealtime): %.0f MHz\n", clk / (wgs * nw));
        for (int j = 0; j < 5; ++j) s[j] /= (double)wgs * nw;
        printf("  trace (s_memtime ticks, mean per wave): prologue %.0f  loop %.0f (of which tap-entry wait %.0f, restage %.0f, body %.0f)  epilogue %.0f  | WG total %.0f | kernel span %.0f ticks = %.2f ticks/us\n",
               s[0], s[1], s[3], s[4], s[1] - s[3] - s[4], s[2], s[0] + s[1] + s[2], (double)(t1 - t0), (double)(t1 - t0) / (ms * 1e3));
        printf("  taps per WG: %d  -> body %.0f ticks/tap, wait %.0f ticks/tap\n", ksteps, (s[1] - s[3] - s[4]) / ksteps, s[3] / ksteps);
        { double dr = 0; for (int i = 0; i < wgs * nw; ++i) dr += (double)tb[(size_t)i * TS + 11]; printf("  of the epilogue: %.0f ticks waiting for the stores to drain after the last one was issued\n", dr / ((double)wgs * nw)); }
    }
#endif
    return 0;
}
