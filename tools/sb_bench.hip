// Standalone correctness + timing harness of the small-batch conv flavour (conv_sb.hip) against the per-tap flavour (conv_igemm.hip, bf16, no split-K)
// and the split-K LDS-DMA flavour (conv_glds.hip) on one synthetic layer with random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I terrain_diffusion_amd/csrc tools/sb_bench.hip -o tools/sb_bench.out
//   ./sb_bench.out N H W Cin Cout mt nt [Cin1x1 epi xform order resample glds_ks]      (mt = 0: the 64 px x 16 cout flavour of conv_s16.hip)
//     Cin: channels of the 3x3 segment (0 = none), Cin1x1: channels of a second, 1x1 segment with its own source; epi 0 plain 1 emb-silu 2 residual;
//     xform 0 none 1 mp_silu 2 pixel-norm + mp_silu (3x3 segment); order = sb_order; resample 0 keep 2 up (3x3 source at half resolution);
//     glds_ks: split-K factor of the conv_glds comparison run (0 = skip it)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "conv_igemm.hip"
#ifdef SB_WITH_GLDS
#include "conv_glds.hip"
#endif
#include "conv_sb.hip"
#include "conv_s16.hip"
using namespace td;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float bf2f(uint16_t v) { uint32_t b = (uint32_t)v << 16; float f; memcpy(&f, &b, 4); return f; }
int main(int argc, char** argv) {
    auto A = [&](int i, int d) { return argc > i ? atoi(argv[i]) : d; };
    const int N = A(1, 1), H = A(2, 64), W = A(3, 64), Cin = A(4, 192), Cout = A(5, 192), mt = A(6, 2), nt = A(7, 2), Cin1 = A(8, 0), epi = A(9, 0), xform = A(10, 0),
              order = A(11, 0), resample = A(12, 0), glds_ks = A(13, 0);
    const int chunk = 64, CoutPad = (Cout + 63) / 64 * 64;
    const size_t M = (size_t)N * H * W;
    const int Hs = resample == 2 ? H / 2 : H, Ws = resample == 2 ? W / 2 : W;
    const size_t Ms = (size_t)N * Hs * Ws;
    const int n3 = Cin / chunk, g1 = Cin1 / chunk, ksteps = n3 * 9 + g1;
    void *x, *x1 = nullptr, *w, *wsb, *out, *out_ref;
    CK(hipMalloc(&x, std::max<size_t>(Ms * Cin * 2, 256))); CK(hipMalloc(&w, (size_t)(ksteps + 2) * CoutPad * 128 + 16384)); CK(hipMalloc(&wsb, (size_t)(ksteps + 40) * CoutPad * 128 + 16384));
    CK(hipMemset(wsb, 0, (size_t)(ksteps + 40) * CoutPad * 128 + 16384));
    CK(hipMalloc(&out, M * CoutPad * 4)); CK(hipMalloc(&out_ref, M * CoutPad * 4));
    srand(1);
    std::vector<uint16_t> hx(Ms * Cin), hw((size_t)ksteps * CoutPad * 64);
    for (auto& v : hx) v = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
    for (auto& v : hw) v = 0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15);
    if (Cin) CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    ConvParams p; memset(&p, 0, sizeof p);
    int ns = 0;
    float* ssq_in = nullptr;
    if (Cin) {
        p.seg[ns].src = x; p.seg[ns].C = Cin; p.seg[ns].cstride = Cin; p.seg[ns].Hs = Hs; p.seg[ns].Ws = Ws; p.seg[ns].taps = 9; p.seg[ns].xform = xform; p.seg[ns].scale = 1.f; p.seg[ns].resample = resample;
        if (xform == 2) {
            const int parts = Cin / 32;
            std::vector<float> hs((size_t)parts * Ms);
            for (auto& v : hs) v = 16.f + (rand() & 0xff) / 32.f;   // ~ sum of 32 squares of +-0.5..1 values
            CK(hipMalloc(&ssq_in, hs.size() * 4)); CK(hipMemcpy(ssq_in, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
            p.seg[ns].sumsq = ssq_in; p.seg[ns].nparts = parts; p.seg[ns].inv_c = 1.f / Cin;
        }
        ++ns;
    }
    if (Cin1) {
        std::vector<uint16_t> hx1(M * Cin1); for (auto& v : hx1) v = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
        CK(hipMalloc(&x1, M * Cin1 * 2)); CK(hipMemcpy(x1, hx1.data(), hx1.size() * 2, hipMemcpyHostToDevice));
        p.seg[ns].src = x1; p.seg[ns].C = Cin1; p.seg[ns].cstride = Cin1; p.seg[ns].Hs = H; p.seg[ns].Ws = W; p.seg[ns].taps = 1; p.seg[ns].xform = 0; p.seg[ns].scale = 1.f;
        ++ns;
    }
    p.nseg = ns; p.wpack = w; p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.CoutPad = CoutPad; p.kgroups = n3 + g1; p.ksplit = 1;
    p.epi = epi; p.out_cstride = Cout; p.dma1x1 = 1;
    { void* z; CK(hipMalloc(&z, 4096)); CK(hipMemset(z, 0, 4096)); p.zeros = z; }
    if (epi == EPI_EMB_SILU) { float* cv; std::vector<float> hc((size_t)N * CoutPad); for (auto& v : hc) v = 0.9f + (rand() & 0xff) / 1024.f; CK(hipMalloc(&cv, hc.size() * 4)); CK(hipMemcpy(cv, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); p.cvec = cv; p.cvec_stride = CoutPad; }
    float *ssq = nullptr, *ssq_ref = nullptr;
    if (epi == EPI_RESIDUAL) {
        void* r; std::vector<uint16_t> hr(M * Cout); for (auto& v : hr) v = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
        CK(hipMalloc(&r, hr.size() * 2)); CK(hipMemcpy(r, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&ssq, M * (CoutPad / 16 + 8) * 4)); CK(hipMalloc(&ssq_ref, M * (CoutPad / 16 + 8) * 4));
        p.res = r; p.res_cstride = Cout; p.res_Hs = H; p.res_Ws = W; p.res_scale = 0.9f; p.clip = 256.f;
    }
    void *o2 = nullptr, *o2_ref = nullptr;
    if (A(14, 0)) { CK(hipMalloc(&o2, M * Cout * 2)); CK(hipMalloc(&o2_ref, M * Cout * 2)); p.out2_scale = 1.3f; }
    if (!conv_set_kbounds(p, true)) { printf("bad split-K\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    const bool s16 = mt == 0;
    if (s16) CK(launch_s16_repack(w, wsb, CoutPad, n3, g1, st)); else CK(launch_sb_repack(w, wsb, CoutPad, n3, g1, st));
    const bool narrow = W < 16;
    // ---- reference: per-tap flavour, 8x16 (8x8x2) tile, bn 64, no split-K
    ConvParams pr = p; pr.out = out_ref; pr.out_sumsq = ssq_ref; pr.out2 = o2_ref;
    { const int TW = narrow ? 8 : 16, NIMG = narrow ? 2 : 1; pr.tiles_x = (W + TW - 1) / TW; pr.tiles_y = (H + 7) / 8; pr.img_groups = (N + NIMG - 1) / NIMG; pr.n_ntiles = CoutPad / 64; }
    CK(hipMemset(out_ref, 0, M * CoutPad * 4));
    CK(launch_conv(pr, 1, narrow, 64, 0, st));
    // ---- sb
    ConvParams ps = p; ps.out = out; ps.out_sumsq = ssq; ps.out2 = o2; ps.wpack_sb = wsb; ps.sb_n3 = n3; ps.sb_order = order;
    { const int TW = narrow ? 8 : 16, TH = narrow ? ((mt == 2 || s16) ? 8 : 4) : (mt == 4 ? 8 : (mt == 2 || s16) ? 4 : 2); ps.tiles_x = (W + TW - 1) / TW; ps.tiles_y = (H + TH - 1) / TH; ps.img_groups = N; ps.n_ntiles = s16 ? CoutPad / 16 : CoutPad / (32 * nt); }
    auto LSB = [&](const ConvParams& q_) { return s16 ? launch_conv_s16(q_, 1, narrow, st) : launch_conv_sb(q_, 1, narrow, mt, nt, st); };
    const int sb_ks = A(15, 1);   // split-K over workgroups on top of the in-workgroup split (+ reduce launch)
    if (sb_ks > 1) {
        float* part; CK(hipMalloc(&part, (size_t)sb_ks * M * CoutPad * 4)); ps.partial = part; ps.ksplit = sb_ks;
        if (!conv_set_kbounds(ps, true)) { printf("bad sb split-K\n"); return 1; }
    }
    CK(hipMemset(out, 0, M * CoutPad * 4));
    CK(LSB(ps));
    CK(hipStreamSynchronize(st));
    {
        std::vector<uint16_t> a(M * Cout), b(M * Cout);
        CK(hipMemcpy(a.data(), out, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out_ref, b.size() * 2, hipMemcpyDeviceToHost));
        double se = 0, sr = 0, mx = 0; size_t nz = 0, nbad = 0;
        for (size_t i = 0; i < a.size(); ++i) { const double fa = bf2f(a[i]), fb = bf2f(b[i]); se += (fa - fb) * (fa - fb); sr += fb * fb; mx = std::max(mx, fabs(fa - fb)); nz += (a[i] & 0x7fff) != 0; nbad += !(fabs(fa - fb) <= 0.02 * fabs(fb) + 0.02 * sqrt(sr / (i + 1))); }
        printf("  sb vs per-tap: rel-RMS %.3e  max|d| %.3e (rms %.3e)  outside 2%%: %zu / %zu  nonzero %zu\n", sqrt(se / std::max(sr, 1e-30)), mx, sqrt(sr / a.size()), nbad, a.size(), nz);
        if (ssq) {
            std::vector<float> sa(M * (CoutPad / 32)), sb_(M * (CoutPad / 32));
            if (s16) {   // one plane per 16 couts: planes 2k and 2k + 1 together are the 32-cout partial of the other flavours
                std::vector<float> s2(M * (CoutPad / 16)); CK(hipMemcpy(s2.data(), ssq, s2.size() * 4, hipMemcpyDeviceToHost));
                for (int k = 0; k < CoutPad / 32; ++k) for (size_t i = 0; i < M; ++i) sa[k * M + i] = s2[(2 * k) * M + i] + s2[(2 * k + 1) * M + i];
            } else
            CK(hipMemcpy(sa.data(), ssq, sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb_.data(), ssq_ref, sb_.size() * 4, hipMemcpyDeviceToHost));
            // the per-tap flavour keeps one partial per (cout tile, wave column) = per 32 couts at bn 64 with 2 wave columns: same decomposition
            double e2 = 0, r2 = 0; for (size_t i = 0; i < sa.size(); ++i) { e2 += (sa[i] - sb_[i]) * (double)(sa[i] - sb_[i]); r2 += (double)sb_[i] * sb_[i]; }
            printf("  sumsq partials: rel-RMS %.3e\n", sqrt(e2 / std::max(r2, 1e-30)));
        }
        if (o2) {
            CK(hipMemcpy(a.data(), o2, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2_ref, b.size() * 2, hipMemcpyDeviceToHost));
            double e2 = 0, r2 = 0; for (size_t i = 0; i < a.size(); ++i) { const double fa = bf2f(a[i]), fb = bf2f(b[i]); e2 += (fa - fb) * (fa - fb); r2 += fb * fb; }
            printf("  out2: rel-RMS %.3e\n", sqrt(e2 / std::max(r2, 1e-30)));
        }
    }
    // ---- timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    auto timeit = [&](auto&& L) { for (int i = 0; i < 3; ++i) CK(L()); CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) CK(L()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3; };
    void* fl = nullptr; const size_t fb = (size_t)768 << 20;
    auto cold = [&](auto&& L) {   // L2 + MALL evicted before every launch, each launch timed by itself
        if (!fl) CK(hipMalloc(&fl, fb));
        float acc = 0.f; const int r = 8;
        for (int i = 0; i < r; ++i) { CK(hipMemsetAsync(fl, i, fb, st)); CK(hipEventRecord(e0, st)); CK(L()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); acc += t; }
        return acc / r * 1e3;
    };
    const double flop = 2.0 * M * Cout * ((double)Cin * 9 + Cin1), wmb = (double)ksteps * CoutPad * 128 * 1e-6;
    const float t_sb = timeit([&] { return LSB(ps); });
    const float c_sb = cold([&] { return LSB(ps); });
    printf("N%d %dx%d C%d+%d(1x1) -> %d mt%d nt%d epi%d xf%d ord%d rs%d ks%d | sb: hot %.1f us (%.0f TF/s)  cold %.1f us (%.0f GB/s of %.2f MB weights)  wgs=%d\n", N, H, W, Cin, Cin1, Cout, mt, nt, epi, xform, order, resample, sb_ks,
           t_sb, flop / t_sb * 1e-6, c_sb, wmb / c_sb * 1e3, wmb, ps.n_ntiles * ps.tiles_x * ps.tiles_y * ps.img_groups * sb_ks);
#ifdef SB_WITH_GLDS
    if (glds_ks > 0) {
        ConvParams pg = p; pg.out = out_ref; pg.out_sumsq = ssq_ref; pg.out2 = o2_ref; pg.ksplit = glds_ks;
        float* partial; CK(hipMalloc(&partial, (size_t)glds_ks * M * CoutPad * 4)); pg.partial = partial;
        const int bn = CoutPad % 96 == 0 ? 96 : 64;
        { const int TW = narrow ? 8 : 16, NIMG = narrow ? 2 : 1; pg.tiles_x = (W + TW - 1) / TW; pg.tiles_y = (H + 7) / 8; pg.img_groups = (N + NIMG - 1) / NIMG; pg.n_ntiles = CoutPad / bn; }
        if (!conv_set_kbounds(pg, true)) { printf("  glds: bad split-K\n"); return 0; }
        if (bn == 64 && narrow) { printf("  glds: no narrow bn64\n"); return 0; }
        const float t_g = timeit([&] { return launch_conv_glds(pg, 1, narrow, bn, 1, st); });
        const float c_g = cold([&] { return launch_conv_glds(pg, 1, narrow, bn, 1, st); });
        printf("      conv_glds small tile bn%d ks%d (+ reduce launch): hot %.1f us  cold %.1f us  wgs=%d\n", bn, glds_ks, t_g, c_g, pg.n_ntiles * pg.tiles_x * pg.tiles_y * pg.img_groups * glds_ks);
    }
#endif
#ifdef TD_TRACE
    {
        const int wgs = ps.n_ntiles * ps.tiles_x * ps.tiles_y * ps.img_groups;
        unsigned long long* tbuf; CK(hipMalloc(&tbuf, (size_t)wgs * 4 * 16 * 8)); CK(hipMemset(tbuf, 0, (size_t)wgs * 4 * 16 * 8));
        ConvParams pt = ps; pt.partial = (float*)tbuf;
        CK(launch_conv_sb(pt, 1, narrow, mt, nt, st)); CK(hipStreamSynchronize(st));
        for (int rep = 0; rep < 2; ++rep) {   // rep 0: hot (the launch above warmed the caches), rep 1: cold
            if (rep) { if (!fl) CK(hipMalloc(&fl, fb)); CK(hipMemsetAsync(fl, 3, fb, st)); }
            CK(launch_conv_sb(pt, 1, narrow, mt, nt, st)); CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> tb((size_t)wgs * 4 * 16);
            CK(hipMemcpy(tb.data(), tbuf, tb.size() * 8, hipMemcpyDeviceToHost));
            double s[12] = {0}; unsigned long long r0 = ~0ull, r1 = 0; double mxw = 0;
            for (int i = 0; i < wgs * 4; ++i) { for (int j = 0; j < 12; ++j) s[j] += (double)tb[(size_t)i * 16 + j]; r0 = std::min(r0, tb[(size_t)i * 16 + 8]); r1 = std::max(r1, tb[(size_t)i * 16 + 9]); mxw = std::max(mxw, (double)(tb[(size_t)i * 16 + 9] - tb[(size_t)i * 16 + 8])); }
            for (int j = 0; j < 12; ++j) s[j] /= wgs * 4.0;
            printf("  trace %s (s_memtime ticks, mean per wave): [kernarg + id decode %.0f, coords %.0f] prologue %.0f  first-barrier %.0f  3x3 loop %.0f  1x1 loop %.0f  reduce %.0f  epilogue %.0f  store drain %.0f | total %.0f | kernel span %.2f us, longest wave %.2f us\n",
                   rep ? "cold" : "hot ", s[10], s[11], s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], (r1 - r0) / 100.0, mxw / 100.0);
        }
    }
#endif
    return 0;
}
