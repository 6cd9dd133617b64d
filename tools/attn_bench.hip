// Standalone harness of the attention kernels (csrc/attn_mfma.hip): pack + flash kernel on one (B, H, Lq, Lk, D) problem.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iterrain_diffusion_amd/csrc tools/attn_bench.hip -o tools/attn_bench.out
//   tools/attn_bench.out B H Lq Lk D [reps]        TD_ATTN_DUMP=file writes the output (compare two builds with cmp);
//   a CPU fp64 reference (bf16-rounded operands) is evaluated on 48 sampled queries.
// Under rocprofv3 (--kernel-trace / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) the flash kernel's rows are the ones named attn_mfma*.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "attn_mfma.hip"
using namespace td;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static float bf16r(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, H = argc > 2 ? atoi(argv[2]) : 8, Lq = argc > 3 ? atoi(argv[3]) : 4096, Lk = argc > 4 ? atoi(argv[4]) : 4096, D = argc > 5 ? atoi(argv[5]) : 40;
    const int reps = argc > 6 ? atoi(argv[6]) : 20;
    const float scale = 1.f / sqrtf((float)D);
    const size_t nq = (size_t)B * H * Lq * D, nk = (size_t)B * H * Lk * D;
    std::vector<float> hq(nq), hk(nk), hv(nk);
    srand(7);
    auto rnd = [] { float a = 0.f; for (int i = 0; i < 4; ++i) a += (float)rand() / RAND_MAX - 0.5f; return a * 1.7f; };
    for (auto& x : hq) x = rnd();
    for (auto& x : hk) x = rnd();
    for (auto& x : hv) x = rnd();
    float *dq, *dk, *dv, *dout, *dout2;
    CK(hipMalloc(&dq, nq * 4)); CK(hipMalloc(&dk, nk * 4)); CK(hipMalloc(&dv, nk * 4)); CK(hipMalloc(&dout, nq * 4)); CK(hipMalloc(&dout2, nq * 4));
    CK(hipMemcpy(dq, hq.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), nk * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), nk * 4, hipMemcpyHostToDevice));
    size_t qn, kn, vn;
    const size_t ws = attn_workspace_elems(B, H, Lq, Lk, D, &qn, &kn, &vn);
    __bf16* w; CK(hipMalloc(&w, ws * 2)); CK(hipMemset(w, 0, ws * 2));
    __bf16 *Qp = w, *Kp = w + qn, *Vt = w + qn + kn;
    AttnStrides sq{(long)H * Lq * D, (long)Lq * D, D, 1}, sk{(long)H * Lk * D, (long)Lk * D, D, 1};
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(attn_pack<float>(dq, dk, dv, sq, sk, sk, B, H, Lq, Lk, D, 0, scale, Qp, Kp, Vt, st));
    CK(attn_mfma(Qp, Kp, Vt, dout, nullptr, sq, B, H, Lq, Lk, D, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(attn_mfma(Qp, Kp, Vt, dout, nullptr, sq, B, H, Lq, Lk, D, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double useful = 4.0 * B * H * Lq * Lk * D;
    const int Dp = attn_dp(D), Dm = attn_dm(D), Lkp = (Lk + 63) / 64 * 64, Lqp = (Lq + 255) / 256 * 256;   // (round 6: where D % 16 != 0 a padding channel / row carries the softmax's reference point / row sum, attn_fold)
    const double issued = 2.0 * B * H * Lqp * Lkp * (Dp + Dm);
    printf("B%d H%d %dx%d d%d : %.1f us  useful %.1f TFLOP/s  issued-MFMA %.1f TFLOP/s (%.1f %% of 2500)\n", B, H, Lq, Lk, D, ms * 1e3, useful / ms / 1e9, issued / ms / 1e9, issued / ms / 1e9 / 25.0);
    std::vector<float> ho(nq);
    CK(hipMemcpy(ho.data(), dout, nq * 4, hipMemcpyDeviceToHost));
    // CPU reference on sampled queries: bf16-rounded K, V and (scale * log2 e * q), fp64 softmax
    double se = 0, sr = 0;
    const float qs = scale * 1.4426950408889634f;
    for (int smp = 0; smp < 48; ++smp) {
        const int b = rand() % B, h = rand() % H, qi = (smp < 4) ? (smp & 1 ? Lq - 1 : 0) : rand() % Lq;
        const float* qr = &hq[(((size_t)b * H + h) * Lq + qi) * D];
        std::vector<double> sc(Lk); double mx = -1e300;
        for (int k = 0; k < Lk; ++k) { const float* kr = &hk[(((size_t)b * H + h) * Lk + k) * D]; double a = 0; for (int c = 0; c < D; ++c) a += (double)bf16r(qr[c] * qs) * bf16r(kr[c]); sc[k] = a; mx = a > mx ? a : mx; }
        double den = 0; std::vector<double> acc(D, 0.0);
        for (int k = 0; k < Lk; ++k) { const double p = exp2(sc[k] - mx); den += p; const float* vr = &hv[(((size_t)b * H + h) * Lk + k) * D]; for (int c = 0; c < D; ++c) acc[c] += p * bf16r(vr[c]); }
        for (int c = 0; c < D; ++c) { const double r = acc[c] / den, g = ho[(((size_t)b * H + h) * Lq + qi) * D + c]; se += (g - r) * (g - r); sr += r * r; }
    }
    printf("  rel-RMS vs fp64 reference on 48 sampled queries: %.3e\n", sqrt(se / sr));
#ifdef TD_ATTN_TRACE
    {   // phase trace of the unpipelined kernel (TD_ATTN_PIPE=0): shader cycles per tile and wave
        const int nwg = (Lq + 255) / 256; const size_t nwv = (size_t)B * H * nwg * 8;
        unsigned long long* dtr; CK(hipMalloc(&dtr, nwv * 64)); CK(hipMemset(dtr, 0, nwv * 64));
        CK(attn_mfma(Qp, Kp, Vt, dout, (__bf16*)dtr, sq, B, H, Lq, Lk, D, st)); CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> tr(nwv * 8); CK(hipMemcpy(tr.data(), dtr, nwv * 64, hipMemcpyDeviceToHost));
        double sum[6] = {0, 0, 0, 0, 0, 0}; const double nt = (Lk + 63) / 64;
        for (size_t w = 0; w < nwv; ++w) for (int i = 0; i < 6; ++i) sum[i] += (double)tr[w * 8 + i];
        const char* nm[6] = {"requests for tile t+2 issued", "S = K Q^T (MFMA)", "softmax (VALU)", "O += V P (MFMA, rescale)", "LDS stores of tile t+1 (waits for its loads)", "barrier wait"};
        double tot = 0; for (int i = 0; i < 6; ++i) tot += sum[i];
        printf("  phase trace, cycles per tile and wave (mean over %zu waves): total %.0f\n", nwv, tot / nwv / nt);
        for (int i = 0; i < 6; ++i) printf("    %-42s %7.0f  (%.0f %%)\n", nm[i], sum[i] / nwv / nt, 100.0 * sum[i] / tot);
    }
#endif
    if (const char* f = getenv("TD_ATTN_DUMP")) { FILE* fp = fopen(f, "wb"); if (fp) { fwrite(ho.data(), 4, nq, fp); fclose(fp); } }
    return 0;
}
