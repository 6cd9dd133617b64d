// Implicit-GEMM convolution for the EDMUnet2D conv stack, hand-written for gfx950 (CDNA4).
//
// Replaces the reference's `MPConv.forward` (terrain_diffusion/models/mp_layers.py:201-221) together with
// the element-wise work `UNetBlock.forward` wraps around it (terrain_diffusion/models/unet_block.py:116-156):
// pixel-norm + mp_silu on the way in (fused into LDS staging), emb-scale + mp_silu / residual mp_sum + clip
// on the way out (fused into the epilogue), mp_concat + the 1x1 skip conv as extra K-segments.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (segment, 128-byte channel chunk, tap).
//   * MFMA A operand = weights (rows = couts), B operand = activations (cols = pixels), so each lane ends up
//     with 4 consecutive couts of one pixel -> packed NHWC stores.
//   * Activations: a (TH+2)x(TW+2) halo patch of one channel chunk is staged ONCE in LDS and reused by all
//     9 taps (tap = shifted LDS address); weights stream per tap from a pre-packed, pre-swizzled slab.
//   * LDS rows are 128 B (64 bf16 / 32 fp32 channels); the eight 16-B slots of a row are XOR-swizzled with
//     (row & 7) so the 16-lane ds_read_b128 groups hit 16 distinct slots of the 256-B bank row.
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  fp32: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "conv_common.h"

namespace td {

// One K-step (one tap of one 128-byte channel chunk) of MFMAs: weights from sb, activations from the patch sa shifted by doff.
template <typename T, int MT, int NTL>
__device__ __forceinline__ void mfma_tap(const unsigned char* __restrict__ sa, const unsigned char* __restrict__ sb, const int (&base_pp)[MT],
                                         const int (&nloc)[NTL], int doff, int lg, f32x4 (&acc)[MT][NTL]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kg = ks * 4 + lg;
        u32x4 wf[NTL], xf[MT];
#ifdef TD_ABLATE_DSREAD
#pragma unroll
        for (int j = 0; j < NTL; ++j) { wf[j] = u32x4{(unsigned)kg, 1u, 2u, 3u}; asm volatile("" : "+v"(wf[j])); }
#pragma unroll
        for (int i = 0; i < MT; ++i) { xf[i] = u32x4{(unsigned)doff, 1u, 2u, 3u}; asm volatile("" : "+v"(xf[i])); }
#else
#pragma unroll
        for (int j = 0; j < NTL; ++j) wf[j] = *(const u32x4*)(sb + nloc[j] * 128 + ((kg ^ TD_SWZ(nloc[j])) << 4));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int pp = base_pp[i] + doff;
            xf[i] = *(const u32x4*)(sa + pp * 128 + ((kg ^ TD_SWZ(pp)) << 4));
        }
#endif
#ifdef TD_ABLATE_MFMA
#pragma unroll
        for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(wf[j]));
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(xf[i]));
#else
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = Half<T>::mfma16(__builtin_bit_cast(typename Half<T>::x8, wf[j]), __builtin_bit_cast(typename Half<T>::x8, xf[i]), acc[i][j]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, wf[j])[e], __builtin_bit_cast(f32x4, xf[i])[e], acc[i][j], 0, 0, 0);
        }
#endif
    }
}

// One weight tile per K-step, register-staged and double-buffered in LDS, one barrier per tap; split-K over K-groups.
template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_kernel(const ConvParams p) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int TPIX = TH * TW;
    constexpr int BM = NIMG * TPIX;
    constexpr int PH = TH + 2, PW = TW + 2, PPI = PH * PW, NPATCH = NIMG * PPI;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 16, NTL = WN / 16;
    constexpr int CHUNK = Elem<T>::CHUNK, PER16 = Elem<T>::PER16;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int B_ITERS = (BN * 8) / NTHR;
    constexpr int A_BYTES = NPATCH * 128, B_BYTES = BN * 128;
    static_assert(BM % (16 * WAVES_M) == 0 && BN % (16 * WAVES_N) == 0 && (BN * 8) % NTHR == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NA = 2, NB = 2, BR = 1;
    unsigned char* s_a = smem;                        // activation-patch buffer(s)
    unsigned char* s_b = smem + NA * A_BYTES;         // weight-tile buffers
    float* s_rn = (float*)(s_b + NB * B_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;

    int bid = blockIdx.x;
    const int ntile = bid % p.n_ntiles; bid /= p.n_ntiles;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups;
    const int mtile = bid % mtiles;
    const int ksp = bid / mtiles;
    const int txi = mtile % p.tiles_x, tyi = (mtile / p.tiles_x) % p.tiles_y, ig = mtile / (p.tiles_x * p.tiles_y);
    const int n0 = ig * NIMG, y0 = tyi * TH, x0 = txi * TW, co0 = ntile * BN;
    const int g0 = p.kb[ksp], g1 = p.kb[ksp + 1];  // conv_set_kbounds (host)

    // ---- locate the first K group: (segment, chunk) and its first kstep
    int seg = 0, chunk = 0, kstep = 0;
    {
        int g = 0;
        while (seg < p.nseg) {
            const int nch = p.seg[seg].C / CHUNK;
            if (g0 < g + nch) { chunk = g0 - g; kstep += chunk * p.seg[seg].taps; break; }
            g += nch; kstep += nch * p.seg[seg].taps; ++seg;
        }
    }

    // ---- weights of the first kstep are independent of everything else: get them in flight first
    u32x4 breg[BR * B_ITERS];
    const int kstep_last = p.kgroups == 0 ? 0 : [&] { int t_ = 0; for (int s_ = 0; s_ < p.nseg; ++s_) t_ += (p.seg[s_].C / CHUNK) * p.seg[s_].taps; return t_ - 1; }();
#define TD_LOAD_B(DST, KS)                                                                             \
    {                                                                                                  \
        const int ks_ = (KS) < kstep_last ? (KS) : kstep_last;                                         \
        const u32x4* wsrc_ = (const u32x4*)p.wpack + ((size_t)ks_ * p.CoutPad + co0) * 8 + tid;        \
        _Pragma("unroll") for (int i_ = 0; i_ < B_ITERS; ++i_) DST[i_] = wsrc_[i_ * NTHR];             \
    }
    TD_LOAD_B(breg, kstep);

    // ---- per-thread staging coordinates (constant over the K loop): packed (n, y, x, interior) or -1
    int a_coord[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int e = tid + it * NTHR, pp = e >> 3;
        const int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
        const int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
        const bool ok = (pp < NPATCH) && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
        const bool interior = py >= 1 && py <= TH && px >= 1 && px <= TW;
        a_coord[it] = ok ? ((n << 21) | (y << 11) | (x << 1) | (interior ? 1 : 0)) : -1;
    }
    u32x4 av[A_ITERS];
#define TD_LOAD_A(SEG, CH)                                                                                            \
    {                                                                                                                 \
        const ConvSeg& sg_ = p.seg[SEG];                                                                              \
        const T* src_ = (const T*)sg_.src + (CH) * CHUNK + (tid & 7) * PER16;                                         \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            const int c_ = a_coord[it_];                                                                              \
            av[it_] = u32x4{0u, 0u, 0u, 0u};                                                                          \
            if (c_ >= 0 && (sg_.taps == 9 || (c_ & 1))) {                                                             \
                const int sp_ = src_pixel(c_ >> 21, (c_ >> 11) & 1023, (c_ >> 1) & 1023, sg_.Hs, sg_.Ws, sg_.resample); \
                av[it_] = *(const u32x4*)(src_ + (size_t)sp_ * sg_.cstride);                                          \
            }                                                                                                         \
        }                                                                                                             \
    }
#define TD_STORE_A(SEG, BUF)                                                                           \
    {                                                                                                  \
        const ConvSeg& sg_ = p.seg[SEG];                                                               \
        unsigned char* dst_ = s_a + (BUF) * A_BYTES;                                                   \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = av[it_];                                                                    \
                if (sg_.xform != 0 && a_coord[it_] >= 0) {                                             \
                    float s_ = sg_.scale;                                                              \
                    if (sg_.xform == 2) s_ *= s_rn[pp_];                                               \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(dst_ + pp_ * 128 + ((slot_ ^ TD_SWZ(pp_)) << 4)) = v_;                        \
            }                                                                                          \
        }                                                                                              \
    }
    if (g0 < g1) TD_LOAD_A(seg, chunk);

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source, for the patch pixels
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (p.seg[0].xform == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = p.seg[0].Hs; rn_Ws = p.seg[0].Ws; rn_res = p.seg[0].resample; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    if (rn_sumsq) {
        const size_t npix = (size_t)p.N * rn_Hs * rn_Ws;
        for (int pp = tid; pp < NPATCH; pp += NTHR) {
            const int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
            const int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
            float rn = 0.f;
            if (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W)
                rn = pixel_rn(rn_sumsq, rn_parts, npix, src_pixel(n, y, x, rn_Hs, rn_Ws, rn_res), rn_invc);
            s_rn[pp] = rn;
        }
    }

    // ---- MFMA operand addressing
    int base_pp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = wm * WM + i * 16 + lr;
        const int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        base_pp[i] = img * PPI + (ty + 1) * PW + (tx + 1);
    }
    int nloc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) nloc[j] = wn * WN + j * 16 + lr;

    // fp32 mode accumulates each K-group (one chunk x taps, <= 288 products) in a fresh accumulator and adds it to the running
    // total afterwards, which keeps the fp32 summation error at the level of a blocked CPU conv.
    constexpr bool TWO_LEVEL = sizeof(T) == 4;
    f32x4 acc[MT][NTL];
    f32x4 tot[TWO_LEVEL ? MT : 1][TWO_LEVEL ? NTL : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (TWO_LEVEL) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    __syncthreads();  // s_rn visible
    if (g0 < g1) TD_STORE_A(seg, 0);
    int abuf = 0, bbuf = 0;

    for (int g = g0; g < g1; ++g) {
        const int taps = p.seg[seg].taps;
        // next group's (segment, chunk); its activation patch is fetched now and lands in the other buffer after this group's taps
        int nseg_ = seg, nchunk_ = chunk + 1;
        if (nchunk_ == p.seg[seg].C / CHUNK) { nchunk_ = 0; ++nseg_; }
        const bool has_next = g + 1 < g1;
        if (has_next) TD_LOAD_A(nseg_, nchunk_);
        const unsigned char* sa = s_a + abuf * A_BYTES;
#define TD_TAP(DOFF, MORE)                                                                              \
    {                                                                                                   \
        unsigned char* sb_ = s_b + bbuf * B_BYTES;                                                      \
        TD_ABL_BSTORE(_Pragma("unroll") for (int i_ = 0; i_ < B_ITERS; ++i_) *(u32x4*)(sb_ + (tid + i_ * NTHR) * 16) = breg[i_]); \
        ++kstep;                                                                                        \
        TD_ABL_BLOAD(TD_LOAD_B(breg, kstep));                                                           \
        TD_ABL_BARRIER(__syncthreads());                                                                                \
        mfma_tap<T, MT, NTL>(sa, sb_, base_pp, nloc, (DOFF), lg, acc);                                  \
        bbuf ^= 1;                                                                                      \
    }
        if (taps == 9) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) TD_TAP((tap / 3 - 1) * PW + (tap % 3 - 1), (tap < 8) || has_next);
        } else {
            TD_TAP(0, has_next);
        }
        if (has_next) TD_STORE_A(nseg_, abuf ^ 1);
        abuf ^= 1;
        if constexpr (TWO_LEVEL) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        seg = nseg_; chunk = nchunk_;
    }
#undef TD_TAP
#undef TD_LOAD_A
#undef TD_STORE_A
#undef TD_LOAD_B
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = tot[i][j];
    }

    // ---------------- epilogue
    const size_t M = (size_t)p.N * p.H * p.W;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = wm * WM + i * 16 + lr;
        const int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        const int n = n0 + img, y = y0 + ty, x = x0 + tx;
        const bool ok = n < p.N && y < p.H && x < p.W;
        float ss = 0.f;
        if (ok) {
            if (p.ksplit > 1) {
                const size_t pix = ((size_t)n * p.H + y) * p.W + x;
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    const int co = co0 + wn * WN + j * 16 + lg * 4;
                    *(f32x4*)(p.partial + ((size_t)ksp * M + pix) * p.CoutPad + co) = acc[i][j];
                }
            } else {
                const float rn = (p.res_sumsq != nullptr) ? s_rn[base_pp[i]] : 1.f;
                const int cobase = co0 + wn * WN + lg * 4;
                f32x4 aux[NTL];  // residual / modulation operands fetched together, then consumed
                if (p.epi == EPI_EMB_SILU) {
#pragma unroll
                    for (int j = 0; j < NTL; ++j) aux[j] = *(const f32x4*)(p.cvec + (size_t)n * p.cvec_stride + cobase + j * 16);
                } else if (p.epi == EPI_RESIDUAL && p.res) {
                    const int sp = src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample);
#pragma unroll
                    for (int j = 0; j < NTL; ++j) aux[j] = load4<T>(p.res, (size_t)sp * p.res_cstride + cobase + j * 16);
                }
#pragma unroll
                for (int j = 0; j < NTL; ++j) ss += epilogue4<T>(p, n, y, x, cobase + j * 16, acc[i][j], rn, aux[j]);
            }
        }
        if (p.out_sumsq && p.ksplit == 1) {
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (ok && lg == 0) {
                const size_t pix = ((size_t)n * p.H + y) * p.W + x;
                p.out_sumsq[(size_t)(ntile * WAVES_N + wn) * M + pix] = ss;
            }
        }
    }
}

// Split-K tail: sums the fp32 partial slabs in fixed order (deterministic) and applies the same epilogue.
// One wave per (pixel, 256-cout chunk); the K-split loop is unrolled 8-deep so the loads of a slab row are in flight together.
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
    const int lane = threadIdx.x & 63;
    const size_t M = (size_t)p.N * p.H * p.W;
    const int nchunks = (p.CoutPad + 255) / 256;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= M * nchunks) return;
    const size_t pix = wid / nchunks;
    const int chunk = (int)(wid % nchunks);
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), n = (int)(pix / ((size_t)p.W * p.H));
    const int co = chunk * 256 + lane * 4;
    float ss = 0.f;
    if (co < p.CoutPad) {
        float rn = 1.f;
        if (p.res_sumsq)
            rn = pixel_rn(p.res_sumsq, p.res_nparts, (size_t)p.N * p.res_Hs * p.res_Ws, src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample), p.res_inv_c);
        f32x4 aux = {0.f, 0.f, 0.f, 0.f};
        if (p.epi == EPI_EMB_SILU) aux = *(const f32x4*)(p.cvec + (size_t)n * p.cvec_stride + co);
        else if (p.epi == EPI_RESIDUAL && p.res) aux = load4<T>(p.res, (size_t)src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample) * p.res_cstride + co);
        const float* base = p.partial + pix * p.CoutPad + co;
        const size_t kstride = M * p.CoutPad;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= p.ksplit; k += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(base + (size_t)(k + u) * kstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        if (k + 4 <= p.ksplit) {   // tails of 4 and 2 in flight together as well (ksplit = 9 used to be 8 + 1 round trips, 3 was 3)
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const f32x4*)(base + (size_t)(k + u) * kstride);
#pragma unroll
            for (int u = 0; u < 4; ++u) v += t[u];
            k += 4;
        }
        if (k + 2 <= p.ksplit) {
            const f32x4 t0 = *(const f32x4*)(base + (size_t)k * kstride), t1 = *(const f32x4*)(base + (size_t)(k + 1) * kstride);
            v += t0; v += t1;
            k += 2;
        }
        if (k < p.ksplit) v += *(const f32x4*)(base + (size_t)k * kstride);
        ss = epilogue4<T>(p, n, y, x, co, v, rn, aux);
    }
    if (p.out_sumsq) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) p.out_sumsq[(size_t)chunk * M + pix] = ss;
    }
}

// ------------------------------------------------------------------------------------------ host launcher
template <typename T, int TH, int TW, int NIMG, int BN>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int WAVES_M = 2, WAVES_N = 2;
    constexpr int NPATCH = NIMG * (TH + 2) * (TW + 2);
    size_t lds = (size_t)2 * NPATCH * 128 + 2 * BN * 128 + NPATCH * 4;
    int grid = p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups * p.ksplit;
    auto kern = conv_igemm_kernel<T, TH, TW, NIMG, BN, WAVES_M, WAVES_N>;
    if (lds > 65536) {
        // per (instantiation, device): hipFuncSetAttribute applies to the CURRENT device's copy of the kernel only
        static bool attr_set[64] = {};
        int dev_ = 0; (void)hipGetDevice(&dev_);
        if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
            hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
            if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (p.ksplit > 1) {
        size_t W_ = (size_t)p.N * p.H * p.W * ((p.CoutPad + 255) / 256);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<T>, dim3((unsigned)((W_ + 3) / 4)), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    return e;
}

// Tile geometry is chosen by the plan builder and recorded in the params (tiles_x/tiles_y/img_groups/n_ntiles);
// `narrow` selects the 8x8x2 tile for feature maps narrower than 16, `bn` the cout tile (64, 96 or 128; 192 spills registers).
template <typename T>
static hipError_t launch_t(const ConvParams& p, bool narrow, int bn, int flavor, hipStream_t st) {
    (void)flavor;
    if (!narrow) {
        switch (bn) {
            case 128: return launch_cfg<T, 8, 16, 1, 128>(p, st);
            case 96: return launch_cfg<T, 8, 16, 1, 96>(p, st);
            default: return launch_cfg<T, 8, 16, 1, 64>(p, st);
        }
    }
    switch (bn) {
        case 128: return launch_cfg<T, 8, 8, 2, 128>(p, st);
        case 96: return launch_cfg<T, 8, 8, 2, 96>(p, st);
        default: return launch_cfg<T, 8, 8, 2, 64>(p, st);
    }
}
// dtype: 0 fp32 (exact-fp32 MFMA), 1 bf16, 2 fp16
hipError_t launch_conv(const ConvParams& p, int dtype, bool narrow, int bn, int flavor, hipStream_t st) {
    return dtype == 1 ? launch_t<__bf16>(p, narrow, bn, flavor, st) : dtype == 2 ? launch_t<_Float16>(p, narrow, bn, flavor, st) : launch_t<float>(p, narrow, bn, flavor, st);
}

}  // namespace td
