// Small-batch ("latency") flavour of the implicit-GEMM convolution (bf16 / fp16, gfx950): the kernel behind the convs of a forward whose grid
// would not fill the chip with the throughput tiles of conv_glds.hip (one 64x64 latent tile = BASELINE configs[1], the 1-16-window batches of
// the cascade's latent stage).  Same maths / parameter block / fused prologues and epilogues as conv_glds.hip
// (mp_layers.py:201-221, unet_block.py:116-156); what differs is how K is split:
//   * conv_glds in this regime splits K over WORKGROUPS: fp32 partial planes written to HBM, read back and reduced by a second launch behind
//     78 of the 79 convs of a forward (profiles/r03_batch1_hbm_traffic.json: 1.74 GB read + 0.95 GB written per forward against 0.507 GB of
//     weights).  Here K is split over the FOUR WAVES of a workgroup and reduced through LDS: wave s contracts channels [16 s, 16 s + 16) of every
//     64-channel K-group, all taps -- perfectly balanced for any number of K-groups, 3x3 and 1x1 alike -- so there is no partial plane in HBM,
//     no reduce launch, and the workgroup tile can be as small as 32 px x 32 couts (144-384 workgroups per layer at batch 1);
//   * every weight byte is used by exactly one wave of one workgroup, so weights do not go through LDS at all: they are stored a second time in
//     MFMA-fragment order ([K-group][16-channel slice][tap][32-cout tile][lane][16 B], sb_repack_kernel) and stream HBM/L2 -> VGPR with plain
//     1-KiB-per-instruction loads, one K-group (9 taps) ahead of their use; the first nine are requested before anything else in the kernel;
//   * the activation halo patch of a K-group is shared by the four waves (each reads its own 32-byte column of the 144-byte rows: same
//     conflict-free lane -> pixel map and compile-time tap offsets as conv_glds), double-buffered in LDS: ONE barrier per 9 taps;
//   * 1x1 K-segments (the decoder's fused skip conv, attention projections) touch neither LDS nor a barrier: their MFMA B fragment is 16 bytes
//     of one pixel, fetched straight from global memory three K-groups ahead;
//   * the four partial accumulator sets are exchanged through LDS (lane-linear 1 KiB pieces, conflict-free) and summed in the fixed order
//     ((w0 + w1) + w2) + w3; wave q then runs the usual wide epilogue for 32x32 tile q.
// Results depend on the tile configuration only (never on the batch a window rides in), but differ in rounding from conv_glds (other K order):
// engine option "batch_invariant" keeps conv_glds everywhere.
#include "conv_common.h"

namespace td {

#ifdef TD_TRACE  // in-kernel phase timing with s_memtime (tools/sb_bench.hip only)
#define TD_ST(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define TD_ST(v)
#endif

// dst (fragment order) <- src (conv_glds slab [kstep][CoutPad][128 B], slots swizzled).  One thread per 16-byte piece.
// n3 = number of leading 3x3 K-groups (every 3x3 segment precedes every 1x1 segment), g1 = number of 1x1 K-groups behind them.
__global__ __launch_bounds__(256) void sb_repack_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int CoutPad, int n3, int g1) {
    const int NCT = CoutPad / 32;
    const size_t total = ((size_t)n3 * 36 + (size_t)g1 * 4) * NCT * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const size_t r0 = idx >> 6;
    const int ct = (int)(r0 % NCT);
    const size_t kunit = r0 / NCT;
    int kstep, s;
    if (kunit < (size_t)n3 * 36) { const int g = (int)(kunit / 36), r = (int)(kunit % 36); s = r / 9; kstep = g * 9 + r % 9; }
    else { const size_t r = kunit - (size_t)n3 * 36; s = (int)(r & 3); kstep = n3 * 9 + (int)(r >> 2); }
    const int cout = ct * 32 + (lane & 31), q = s * 2 + (lane >> 5);
    dst[idx] = src[((size_t)kstep * CoutPad + cout) * 8 + (q ^ TD_SWZ(cout))];
}

// Loads / stores with an explicit GLOBAL address space.  The kernel pins its kernel arguments in scalar registers through empty asm statements
// (see below); a pointer that went through one loses hipcc's address-space inference and would be accessed with FLAT instructions, which count
// on lgkmcnt as well as vmcnt and so collide with every LDS wait of the K loop.
template <typename V> __device__ __forceinline__ V sb_gld(const void* q) { return *(const __attribute__((address_space(1))) V*)q; }
template <typename V> __device__ __forceinline__ void sb_gst(void* q, V v) { *(__attribute__((address_space(1))) V*)q = v; }

// src_pixel of conv_common.h without branches: resample 0 keep, 1 down (src[2y, 2x]), 2 up (src[y/2, x/2])
__device__ __forceinline__ int sb_src_pixel(int n, int y, int x, int Hs, int Ws, int dn, int up) { return (n * Hs + ((y << dn) >> up)) * Ws + ((x << dn) >> up); }

// What dominates a batch-1 conv is not arithmetic but the chain "kernarg miss (~1 us) -> first loads (~1 us) -> ... -> epilogue operands (~1 us)":
// (tools/sb_trace.sh).  The kernel is therefore written around three rules: (1) kernel arguments are read in a few bursts into scalar registers
// (hipcc otherwise re-reads them one at a time behind every uniform branch: a dozen serialised scalar-cache round trips in the prologue, five
// more before the epilogue); (2) every load of the K loop is useful and unconditional (exact counted vmcnt, nothing left to drain at the end);
// (3) whatever the tail needs from memory is requested before the last K-group is contracted.
template <typename T, int TH, int TW, int NT>
__global__ __launch_bounds__(256, 2) void conv_sb_kernel(const ConvParams p) {
    typedef typename Half<T>::x8 hx8;
    typedef typename Half<T>::x4 hx4;
    constexpr int NTHR = 256, NW = 4;
    constexpr int TPIX = TH * TW, MT = TPIX / 32;
    constexpr int PH = TH + 2, PW = TW == 8 ? 12 : TW + 2, NPATCH = PH * PW;   // 8-wide: 2 pad columns (frag_pixel)
    constexpr int CHUNK = 64, PER16 = 8, PITCH = 144;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int A_BYTES = NPATCH * PITCH;
    constexpr int RED_BYTES = NW * MT * NT * 4096;                 // four partial accumulator sets, 4 KiB per 32x32 tile
    constexpr int RN_BASE = (2 * A_BYTES > RED_BYTES ? 2 * A_BYTES : RED_BYTES);
    static_assert(TPIX % 32 == 0 && A_BYTES % 16 == 0 && RN_BASE % 16 == 0 && MT * NT <= NW, "tile shape");
    static_assert(2 * A_BYTES + 2 * PW * PITCH + 2 * PITCH + 128 < 65536, "ds_read offset field");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0
    float* s_rn = (float*)(smem + RN_BASE);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    TD_ST(tr0);
#ifdef TD_TRACE
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- kernel arguments of the prologue, one burst (the empty asm pins them: loaded together, waited for once)
    unsigned k_d1 = p.sb_d1, k_m1 = p.sb_m1, k_m2 = p.sb_m2, k_m3 = p.sb_m3, k_g8 = p.sb_grid8, k_d0 = p.sb_d0, k_m0 = p.sb_m0;
    int k_ks = p.ksplit;
    int k_order = p.sb_order, k_tx = p.tiles_x, k_ty = p.tiles_y, k_N = p.N, k_H = p.H, k_W = p.W, k_cpad = p.CoutPad, k_n3 = p.sb_n3, k_ng = p.kgroups;
    const unsigned char* k_wsb = (const unsigned char*)p.wpack_sb;
    const T* k_s0src = (const T*)p.seg[0].src;
    int k_s0C = p.seg[0].C, k_s0cs = p.seg[0].cstride, k_s0Hs = p.seg[0].Hs, k_s0Ws = p.seg[0].Ws, k_s0rs = p.seg[0].resample, k_s0xf = p.seg[0].xform;
    float k_s0sc = p.seg[0].scale;
    asm volatile("" : "+s"(k_d0), "+s"(k_m0), "+s"(k_ks));
    asm volatile("" : "+s"(k_d1), "+s"(k_m1), "+s"(k_m2), "+s"(k_m3), "+s"(k_g8), "+s"(k_order), "+s"(k_tx), "+s"(k_ty), "+s"(k_N), "+s"(k_H), "+s"(k_W),
                 "+s"(k_cpad), "+s"(k_n3), "+s"(k_ng), "+s"(k_wsb), "+s"(k_s0src), "+s"(k_s0C), "+s"(k_s0cs), "+s"(k_s0Hs), "+s"(k_s0Ws), "+s"(k_s0rs), "+s"(k_s0xf), "+s"(k_s0sc));

    unsigned bid = blockIdx.x;
    if (k_g8) bid = (bid & 7) * k_g8 + (bid >> 3);   // XCD x takes a contiguous range of logical ids (speed only; k_g8 = grid / 8 when 8 divides it)
    // logical id -> (cout tile, pixel tile).  sb_order 0: the cout tiles of a pixel tile are adjacent (they share the halo patch in an XCD's L2),
    // 1: the pixel tiles of a cout tile are adjacent (an XCD's L2 then holds few cout tiles' weights).  Divisions by host-made magic numbers.
    // split-K over workgroups (k_ks > 1, the weight-streaming-bound deep levels at batch 1: see the planner): slice ksp is the outermost index
    const unsigned ksp = k_ks > 1 ? td_udiv(bid, k_d0, k_m0) : 0u;
    bid -= ksp * k_d0;
    const unsigned q1 = td_udiv(bid, k_d1, k_m1), r1 = bid - q1 * k_d1;
    const unsigned ntile = k_order ? q1 : r1, mtile = k_order ? r1 : q1;
    const unsigned q2 = td_udiv(mtile, (unsigned)k_tx, k_m2), txi = mtile - q2 * (unsigned)k_tx;
    const int n0 = (int)td_udiv(q2, (unsigned)k_ty, k_m3), tyi = (int)(q2 - (unsigned)n0 * (unsigned)k_ty);
    const int y0 = tyi * TH, x0 = (int)txi * TW, co0 = (int)ntile * (NT * 32);
    const int NCT = k_cpad / 32, n3 = k_n3;
    // K-groups of this workgroup: [g_lo, g_hi) (everything without split-K; conv_set_kbounds slices balanced by K-steps otherwise): the 3x3 groups
    // [a3, b3) and the 1x1 groups [a1, ngroups) behind them
    int g_lo = 0, g_hi = k_ng;
    if (k_ks > 1) { g_lo = p.kb[ksp]; g_hi = p.kb[ksp + 1]; }
    const int a3 = g_lo < n3 ? g_lo : n3, b3 = g_hi < n3 ? g_hi : n3, a1 = g_lo > n3 ? g_lo : n3, ngroups = g_hi;

    // ---- weight stream: this lane's 16 bytes of (K-group g, slice `wave`, tap t, cout tile ct0 + j) live at wl3 + ((g * 36 + t) * NCT + j) * 1024
    // for the 3x3 groups and at wl1 + (((g - n3) * 4) * NCT + j) * 1024 for the 1x1 groups behind them.  Group 0's nine taps are requested first.
    const size_t tstep = (size_t)NCT * 1024;
    const unsigned char* wl3 = k_wsb + ((size_t)(wave * 9) * NCT + co0 / 32) * 1024 + lane * 16;
    const unsigned char* wl1 = k_wsb + ((size_t)n3 * 36 * NCT + (size_t)wave * NCT + co0 / 32) * 1024 + lane * 16;
    u32x4 wr[9][NT];
    if (a3 < b3) {
        const unsigned char* w0 = wl3 + (size_t)a3 * 36 * tstep;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j) wr[t][j] = sb_gld<u32x4>(w0 + t * tstep + j * 1024);
    }
#ifdef TD_TRACE
    TD_ST(tra);
#endif

    // ---- activation-patch staging (3x3 groups): per thread A_ITERS 16-byte pieces (patch pixel e>>3, slot e&7), as in conv_glds.hip
    int a_coord[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int e = tid + it * NTHR, pp = e >> 3;
        const int py = pp / PW, px = pp % PW;
        const int y = y0 + py - 1, x = x0 + px - 1;
        const bool ok = (pp < NPATCH) && px < TW + 2 && n0 < k_N && y >= 0 && y < k_H && x >= 0 && x < k_W;  // px >= TW+2: pad columns
        a_coord[it] = ok ? ((y << 11) | x) : -1;
    }
    u32x4 av[A_ITERS];
    int caoff[A_ITERS];          // element offset of this thread's piece in the cursor segment's source (0 for zero-fill pieces)
    // cursor = the K-group whose patch is requested next (one ahead of the contraction); cur_* = transform of the group held in `av`
    const T* csrc = k_s0src;
    int cseg = 0, cchunk = 0, cnch = k_s0C / CHUNK, c_xf = k_s0xf, cur_xf = 0;
    float c_sc = k_s0sc, cur_sc = 1.f;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int c = a_coord[it];
        caoff[it] = c >= 0 ? sb_src_pixel(n0, c >> 11, c & 2047, k_s0Hs, k_s0Ws, k_s0rs == 1, k_s0rs == 2) * k_s0cs + (tid & 7) * PER16 : 0;
    }
#define TD_SEG_NEXT()                                                                                                 \
    {                                                                                                                 \
        ++cseg; cchunk = 0;                                                                                           \
        const ConvSeg& sg_ = p.seg[cseg];                                                                             \
        csrc = (const T*)sg_.src; c_xf = sg_.xform; c_sc = sg_.scale; cnch = sg_.C / CHUNK;                           \
        const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            const int c_ = a_coord[it_];                                                                              \
            caoff[it_] = c_ >= 0 ? sb_src_pixel(n0, c_ >> 11, c_ & 2047, Hs_, Ws_, rs_ == 1, rs_ == 2) * cs_ + (tid & 7) * PER16 : 0; \
        }                                                                                                             \
    }
    // always issued (zero-fill pieces read offset 0) so that the vmcnt bookkeeping of the main loop is exact
#define TD_LOAD_A()                                                                                    \
    {                                                                                                  \
        cur_xf = c_xf; cur_sc = c_sc;                                                                  \
        const T* src_ = csrc + cchunk * CHUNK;                                                         \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) av[it_] = sb_gld<u32x4>(src_ + caoff[it_]); \
        ++cchunk;                                                                                      \
    }
#define TD_STORE_A(BUF)                                                                                \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = a_coord[it_] >= 0 ? av[it_] : u32x4{0u, 0u, 0u, 0u};                        \
                if (cur_xf != 0 && a_coord[it_] >= 0) {                                                \
                    float s_ = cur_sc;                                                                 \
                    if (cur_xf == 2) s_ *= s_rn[pp_];                                                  \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(smem + (BUF) * A_BYTES + pp_ * PITCH + (slot_ << 4)) = v_;                   \
            }                                                                                          \
        }                                                                                              \
    }
    if (a3 < b3) {
        int gl = a3;   // a later split-K slice starts inside the K range: walk the cursor to its first group
        while (gl >= cnch) { gl -= cnch; TD_SEG_NEXT(); }
        cchunk = gl;
        TD_LOAD_A();
    }
#ifdef TD_TRACE
    TD_ST(trb);
#endif
    TD_ST(tr1);

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source, for the patch pixels (the partial sums of squares are independent loads:
    // requested eight at a time, added in ascending order)
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (k_s0xf == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = k_s0Hs; rn_Ws = k_s0Ws; rn_res = k_s0rs; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    if (rn_sumsq) {
        const size_t npix = (size_t)k_N * rn_Hs * rn_Ws;
        for (int pp = tid; pp < NPATCH; pp += NTHR) {
            const int py = pp / PW, px = pp % PW;
            const int y = y0 + py - 1, x = x0 + px - 1;
            float rn = 0.f;
            if (n0 < k_N && y >= 0 && y < k_H && x >= 0 && x < k_W) {
                const float* sp = rn_sumsq + sb_src_pixel(n0, y, x, rn_Hs, rn_Ws, rn_res == 1, rn_res == 2);
                float s = 0.f;
                int q = 0;
                for (; q + 16 <= rn_parts; q += 16) {   // (sixteen at a time first: pixel_rn of conv_common.h says why)
                    float t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = sp[(size_t)(q + u) * npix];
#pragma unroll
                    for (int u = 0; u < 16; ++u) s += t[u];
                }
                for (; q + 8 <= rn_parts; q += 8) {
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = sp[(size_t)(q + u) * npix];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += t[u];
                }
                if (q < rn_parts) {   // the last 1 ... 7 planes in one round trip (pixel_rn of conv_common.h)
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = sp[(size_t)(q + u < rn_parts ? q + u : rn_parts - 1) * npix];   // index-clamped: no branch around a load
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += q + u < rn_parts ? t[u] : 0.f;
                }
                rn = 1.f / (1e-4f + sqrtf(s * rn_invc));   // mp_layers.py:9-12 with dim=1 (pixel_rn of conv_common.h)
            }
            s_rn[pp] = rn;
        }
    }

    // ---- MFMA operand addressing: weights = A operand (rows = couts, straight from the ring registers), activations = B operand
    // (cols = pixels): xbase = LDS address of the TOP-LEFT tap of this lane's pixel, this wave's 16-channel slice, this lane's k-half
    unsigned xbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(i * 32, l31, img, ty, tx);
        xbase[i] = (unsigned)(ty * PW + tx) * PITCH + (unsigned)wave * 32u + (unsigned)lh * 16u;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TD_ST(tr2);
    // the pixel-norm table is needed by the first restage only when the 3x3 source is normed (the residual's table is read in the epilogue,
    // behind the K loop's barriers)
    if (k_s0xf == 2) __syncthreads();

#define TD_TOFF(TAP) ((((TAP) / 3) * PW + ((TAP) % 3)) * PITCH)
    // One 3x3 K-group: restage, barrier, nine taps.  REFILL: the next group exists -- its patch pieces are requested before the barrier and ring
    // slot t is refilled right behind tap t.  The loop body (REFILL = 1) is straight-line as far as loads go, so hipcc's counted vmcnt(N) are
    // exact: nothing is waited for early.  (Loads inside wave-uniform branches make its wait-count pass merge the branch states and drain the
    // whole weight queue in front of every restage; refills sunk to the end of the group -- hipcc's choice when left alone -- shrink the
    // prefetch distance; LDS reads behind the MFMAs expose LDS latency nine times per group: hence the two sched_barriers per tap.)
#define TD_GROUP(G, REFILL)                                                                                           \
    {                                                                                                                 \
        const int buf_ = (G) & 1;                                                                                     \
        TD_STORE_A(buf_);                                                                                             \
        if (REFILL) { if (cchunk == cnch) TD_SEG_NEXT(); TD_LOAD_A(); }                                               \
        /* passing this barrier: patch G is visible; every wave has finished reading the buffer patch G+1 will be written to */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        __builtin_amdgcn_s_barrier();                                                                                 \
        asm volatile("" ::: "memory");                                                                                \
        const unsigned char* wnext_ = wl3 + (size_t)((G) + 1) * 36 * tstep;                                           \
        unsigned xb_[MT];                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xb_[i_] = xbase[i_] + (unsigned)buf_ * (unsigned)A_BYTES;   \
        u32x4 xf_[2][MT];   /* fragments of tap t+1 are requested before the MFMAs of tap t */                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xf_[0][i_] = *(const u32x4*)(smem + xb_[i_] + TD_TOFF(0));  \
        _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                                            \
            if (t_ < 8) {                                                                                             \
                _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xf_[(t_ + 1) & 1][i_] = *(const u32x4*)(smem + xb_[i_] + TD_TOFF(t_ + 1)); \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                         \
                _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                     \
                    acc[i_][j_] = Half<T>::mfma32(__builtin_bit_cast(hx8, wr[t_][j_]), __builtin_bit_cast(hx8, xf_[t_ & 1][i_]), acc[i_][j_]); \
            if (REFILL) {                                                                                             \
                _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) wr[t_][j_] = sb_gld<u32x4>(wnext_ + t_ * tstep + j_ * 1024); \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    }
    // the first group is peeled: the prologue requests weights BEFORE the patch (they depend on nothing but the workgroup id), the loop the other
    // way round, and hipcc merges the loop-entry and back-edge wait-count states -- entering the loop from the prologue would cost a drain of
    // the refill queue (vmcnt(3)) in front of every restage
    if (b3 - a3 > 1) TD_GROUP(a3, true);
    for (int g = a3 + 1; g + 1 < b3; ++g) TD_GROUP(g, true);

    TD_ST(tr3);
    // ---------------- before the last 3x3 group: everything the tail needs from memory.  Epilogue kernel arguments in one burst ...
    int e_epi = p.epi, e_of32 = p.out_f32, e_Cout = p.Cout, e_ocs = p.out_cstride, e_cvs = p.cvec_stride, e_rcs = p.res_cstride, e_rHs = p.res_Hs, e_rWs = p.res_Ws,
        e_rrs = p.res_resample;
    const void* e_res = p.res; const float* e_rss = p.res_sumsq; const float* e_cvec = p.cvec;
    void* e_out = p.out; void* e_out2 = p.out2; float* e_oss = p.out_sumsq;
    float e_rsc = p.res_scale, e_clip = p.clip, e_o2s = p.out2_scale;
    asm volatile("" : "+s"(e_epi), "+s"(e_of32), "+s"(e_Cout), "+s"(e_ocs), "+s"(e_cvs), "+s"(e_rcs), "+s"(e_rHs), "+s"(e_rWs), "+s"(e_rrs), "+s"(e_res), "+s"(e_rss),
                 "+s"(e_cvec), "+s"(e_out), "+s"(e_out2), "+s"(e_oss), "+s"(e_rsc), "+s"(e_clip), "+s"(e_o2s));
    // ... and the operands of the 32x32 tile this wave will finish (tile q = wave)
    const size_t M = (size_t)k_N * k_H * k_W;
    const bool wide = !e_of32 && (e_Cout & 7) == 0;
    const int q = wave, ei = q / NT, ej = q % NT;
    int ety = 0, etx = 0;
    { int img; frag_pixel<TW, TPIX>(ei * 32, l31, img, ety, etx); }
    const int en = n0, ey = y0 + ety, ex = x0 + etx;
    const int cot = co0 + ej * 32;   // first cout of this tile
    const bool eok = q < MT * NT && en < k_N && ey < k_H && ex < k_W && cot < k_cpad;
    const bool has_res = e_epi == EPI_RESIDUAL && e_res != nullptr;
    const int esp = (eok && has_res) ? sb_src_pixel(en, ey, ex, e_rHs, e_rWs, e_rrs == 1, e_rrs == 2) : 0;
    // UNCONDITIONAL loads (a conditional one in front of the last group makes hipcc drain the queue at its restage and at its first tap): the
    // modulation row or the residual run where this wave has one, else 32 harmless bytes of the weight slab
    f32x4 ca[2], cb[2];
    u32x4 rw[2];
    {
        const bool use = eok && wide && cot < e_Cout;
        const bool use_c = use && e_epi == EPI_EMB_SILU, use_r = use && has_res;
        const float* crow = use_c ? e_cvec + (size_t)en * e_cvs + cot + 4 * lh : (const float*)k_wsb;
        const T* rrow = use_r ? (const T*)e_res + (size_t)esp * e_rcs + cot + 8 * lh : (const T*)k_wsb;
#pragma unroll
        for (int m = 0; m < 2; ++m) { ca[m] = sb_gld<f32x4>(crow + m * 16); cb[m] = sb_gld<f32x4>(crow + m * 16 + 8); rw[m] = sb_gld<u32x4>(rrow + m * 16); }
    }

    if (a3 < b3) TD_GROUP(b3 - 1, false);
#undef TD_GROUP
#undef TD_TOFF
#undef TD_SEG_NEXT
#undef TD_LOAD_A
#undef TD_STORE_A

    // ---------------- 1x1 K-groups: both operands straight from global memory, no LDS and no barrier; batches of D1 groups (a batch's loads
    // are requested while the previous batch is contracted; whole 1x1 phases of <= D1 groups -- most of them -- are one request burst)
    if (a1 < ngroups) {
        constexpr int D1 = MT >= 4 ? 4 : 8;   // (the 128-pixel tile holds four B fragments per group: half the depth for the same registers)
        int sg1 = 0;
        while (sg1 < p.nseg && p.seg[sg1].taps == 9) ++sg1;   // first 1x1 segment
        u32x4 wa[D1][NT], xa[D1][MT];
        int poff[MT];           // element offset of this lane's pixel (+ its 8 channels of the wave's slice) in the segment's source
        const T* psrc = nullptr; int pn = 0, pseg = sg1 - 1, pchunk = 0;   // prefetch cursor
        const unsigned char* pw = wl1 + (size_t)(a1 - n3) * 4 * tstep;
        auto seg_open = [&]() {
            ++pseg; pchunk = 0;
            const ConvSeg& sg_ = p.seg[pseg];
            psrc = (const T*)sg_.src; pn = sg_.C / CHUNK;
            const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int img, ty, tx;
                frag_pixel<TW, TPIX>(i * 32, l31, img, ty, tx);
                const int y = y0 + ty, x = x0 + tx;
                // a pixel outside the image is an MFMA column nobody stores: any readable address will do
                const int pix = (n0 < k_N && y < k_H && x < k_W) ? sb_src_pixel(n0, y, x, Hs_, Ws_, rs_ == 1, rs_ == 2) : 0;
                poff[i] = pix * cs_ + wave * 16 + lh * 8;
            }
        };
        int pleft = ngroups - a1;   // groups the prefetch cursor has not issued yet
        auto issue = [&](int slot_) {   // slot_ is a compile-time constant at every call site
            if (pleft > 0) {
                if (pchunk == pn) seg_open();
#pragma unroll
                for (int j = 0; j < NT; ++j) wa[slot_][j] = sb_gld<u32x4>(pw + j * 1024);
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[slot_][i] = sb_gld<u32x4>(psrc + poff[i] + pchunk * CHUNK);
                ++pchunk; --pleft; pw += 4 * tstep;
            }
        };
        seg_open();
        { int gl = a1 - n3; while (gl >= pn) { gl -= pn; seg_open(); } pchunk = gl; }   // a later split-K slice starts inside the 1x1 range
#pragma unroll
        for (int d = 0; d < D1; ++d) issue(d);
        for (int g = a1; g < ngroups; g += D1) {
#pragma unroll
            for (int d = 0; d < D1; ++d) {
                if (g + d < ngroups) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = Half<T>::mfma32(__builtin_bit_cast(hx8, wa[d][j]), __builtin_bit_cast(hx8, xa[d][i]), acc[i][j]);
                    issue(d);
                }
            }
        }
    }

    TD_ST(tr4);
    // ---------------- in-workgroup K reduction: four partial accumulator sets through LDS (over the dead patch buffers)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done reading the patch
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *(f32x4*)(smem + ((wave * MT * NT + i * NT + j) * 4 + rg) * 1024 + lane * 16) =
                    f32x4{acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    TD_ST(tr5);
    // ---------------- epilogue: wave q owns 32x32 tile q = i * NT + j (lane: 4 groups of 4 consecutive couts of pixel column l31)
    if (q < MT * NT) {
        f32x4 a4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 s = *(const f32x4*)(smem + ((0 * MT * NT + q) * 4 + rg) * 1024 + lane * 16);
#pragma unroll
            for (int w = 1; w < NW; ++w) s += *(const f32x4*)(smem + ((w * MT * NT + q) * 4 + rg) * 1024 + lane * 16);
            a4[rg] = s;
        }
        if (k_ks > 1) {   // raw fp32 partial sums [ksplit][pixel][CoutPad]; conv_splitk_reduce_kernel adds the slices in fixed order + epilogue
            if (eok) {
                float* prow = p.partial + ((size_t)ksp * M + ((size_t)en * k_H + ey) * k_W + ex) * k_cpad + cot + 4 * lh;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) *(f32x4*)(prow + rg * 8) = a4[rg];
            }
            return;
        }
        float ssj = 0.f;
        if (eok) {
            const float rn = (e_rss != nullptr) ? s_rn[(ety + 1) * PW + (etx + 1)] : 1.f;
            if (wide) {
                if (cot < e_Cout) {
                    const size_t pix = ((size_t)en * k_H + ey) * k_W + ex;
                    T* orow = (T*)e_out + pix * e_ocs + cot + 8 * lh;
                    const float rs = e_rsc * rn;
                    const bool want_ss = e_oss != nullptr, want_o2 = e_out2 != nullptr;
                    const SiluK k_o2 = silu_k(e_o2s);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        u32x4 o, o2;
                        epi_unit8<T>(e_epi, has_res, e_clip, want_ss, want_o2, a4[2 * m], a4[2 * m + 1], ca[m], cb[m], rw[m], rs, k_o2, o, o2, ssj);
                        sb_gst<u32x4>(orow + m * 16, o);
                        if (want_o2) sb_gst<u32x4>((T*)e_out2 + (orow - (T*)e_out) + m * 16, o2);
                    }
                }
            } else {   // few-channel / fp32 outputs (the U-Net's output conv, incl. its fused solver step): the shared scalar epilogue
                const int cobase = cot + 4 * lh;
                f32x4 aux[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    aux[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (e_epi == EPI_EMB_SILU) aux[rg] = sb_gld<f32x4>(e_cvec + (size_t)en * e_cvs + cobase + rg * 8);
                    else if (has_res) aux[rg] = load4<T>(p.res, (size_t)esp * e_rcs + cobase + rg * 8);
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) ssj += epilogue4<T>(p, en, ey, ex, cobase + rg * 8, a4[rg], rn, aux[rg]);
            }
        }
        if (e_oss) {   // one partial per 32-cout block, as conv_glds.hip writes them (the consumer adds CoutPad / 32 planes in ascending order)
            const float ss = ssj + __shfl_xor(ssj, 32);
            if (eok && lh == 0) sb_gst<float>(e_oss + (size_t)(cot / 32) * M + ((size_t)en * k_H + ey) * k_W + ex, ss);
        }
    }
#ifdef TD_TRACE
    TD_ST(tr6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TD_ST(tr7);
    if (lane == 0 && p.partial) {
        unsigned long long* tb = (unsigned long long*)p.partial + ((size_t)blockIdx.x * NW + wave) * 16;
        tb[0] = tr1 - tr0; tb[1] = tr2 - tr1; tb[2] = tr3 - tr2; tb[3] = tr4 - tr3; tb[4] = tr5 - tr4; tb[5] = tr6 - tr5; tb[6] = tr7 - tr6; tb[7] = tr7 - tr0;
        tb[8] = tr_rt0; tb[9] = __builtin_amdgcn_s_memrealtime(); tb[10] = tra - tr0; tb[11] = trb - tra;
    }
#endif
}

template <typename T, int TH, int TW, int NT>
static hipError_t launch_sb_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int MT = TH * TW / 32, NPATCH = (TH + 2) * (TW == 8 ? 12 : TW + 2);
    constexpr int A2 = 2 * NPATCH * 144, RED = 4 * MT * NT * 4096;
    constexpr size_t lds = (size_t)(A2 > RED ? A2 : RED) + NPATCH * 4;
    if (!p.wpack_sb || p.ksplit < 1 || p.ksplit > 64 || (p.ksplit > 1 && !p.partial) || p.CoutPad % (NT * 32) != 0) return hipErrorInvalidValue;
    bool seen1 = false;   // every 3x3 segment before every 1x1 segment (the weight stream's addressing relies on it), 1x1 sources untransformed
    int n3 = 0;
    for (int s = 0; s < p.nseg; ++s) {
        if (p.seg[s].taps == 9) { if (seen1) return hipErrorInvalidValue; n3 += p.seg[s].C / 64; }
        else { seen1 = true; if (p.seg[s].xform != 0) return hipErrorInvalidValue; }
    }
    if (n3 != p.sb_n3) return hipErrorInvalidValue;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups, grid1 = p.n_ntiles * mtiles, grid = grid1 * p.ksplit;
    if (grid1 <= 0 || (long long)grid * std::max(grid1, std::max(mtiles, p.n_ntiles)) >= ((long long)1 << 32)) return hipErrorInvalidValue;   // sb_udiv's range
    ConvParams q = p;
    q.sb_d0 = grid1; q.sb_m0 = td_magic(grid1);
    q.sb_d1 = p.sb_order ? mtiles : p.n_ntiles; q.sb_m1 = td_magic(q.sb_d1); q.sb_m2 = td_magic(p.tiles_x); q.sb_m3 = td_magic(p.tiles_y);
    q.sb_grid8 = (grid & 7) == 0 ? (unsigned)grid >> 3 : 0u;
    auto kern = conv_sb_kernel<T, TH, TW, NT>;
    if (lds > 65536) {
        static bool attr_set[64] = {};
        int dev_ = 0; (void)hipGetDevice(&dev_);
        if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, q);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
        const size_t W_ = (size_t)p.N * p.H * p.W * ((p.CoutPad + 255) / 256);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<T>, dim3((unsigned)((W_ + 3) / 4)), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    return e;
}

// Tile configurations: mt = 32-pixel MFMA blocks per workgroup (4: 8x16 pixels, 16-wide maps and nt = 1 only; 2: 4x16 or 8x8 pixels, 1: 2x16 or 4x8),
// nt = 32-cout blocks (1 or 2).  Round 5, the 128 px x 32 cout tile: the K loop is bound by what one CU can ingest, and per K-group this tile ingests
// 36.9 KB of weights + 23 KB of patch = 60 KB against the 87.5 KB of the 64 x 64 tile with the SAME number of workgroups and MFMAs per wave.  Measured
// level (slower in a hot loop, equal cold, -1 ... +2.5 % of a forward depending on the batch: profiles/r05_conv_sb_128px_tile.txt): opt-in, engine option sb_m4.
// The caller sets tiles_x / tiles_y for that tile, img_groups = N, n_ntiles = CoutPad / (32 nt), ksplit (+ kb[], partial when > 1).
template <typename T>
static hipError_t launch_conv_sb_t(const ConvParams& p, bool narrow, int mt, int nt, hipStream_t st) {
    if (!narrow) {
        if (mt == 4) return nt == 1 ? launch_sb_cfg<T, 8, 16, 1>(p, st) : hipErrorInvalidValue;   // round 5: 128 px x 32 couts
        if (mt == 2) return nt == 2 ? launch_sb_cfg<T, 4, 16, 2>(p, st) : launch_sb_cfg<T, 4, 16, 1>(p, st);
        return nt == 2 ? launch_sb_cfg<T, 2, 16, 2>(p, st) : launch_sb_cfg<T, 2, 16, 1>(p, st);
    }
    if (mt == 2) return nt == 2 ? launch_sb_cfg<T, 8, 8, 2>(p, st) : launch_sb_cfg<T, 8, 8, 1>(p, st);
    return nt == 2 ? launch_sb_cfg<T, 4, 8, 2>(p, st) : launch_sb_cfg<T, 4, 8, 1>(p, st);
}

// dtype: 1 bf16, 2 fp16 (this flavour has no fp32 form)
hipError_t launch_conv_sb(const ConvParams& p, int dtype, bool narrow, int mt, int nt, hipStream_t st) {
    if ((mt != 1 && mt != 2 && !(mt == 4 && nt == 1 && !narrow)) || (nt != 1 && nt != 2)) return hipErrorInvalidValue;
    return dtype == 2 ? launch_conv_sb_t<_Float16>(p, narrow, mt, nt, st) : launch_conv_sb_t<__bf16>(p, narrow, mt, nt, st);
}

// Fragment-order copy of a conv's packed weights (device to device).  `dst` holds ksteps * CoutPad * 128 bytes.
hipError_t launch_sb_repack(const void* src, void* dst, int CoutPad, int n3, int g1, hipStream_t st) {
    const size_t total = ((size_t)n3 * 36 + (size_t)g1 * 4) * (CoutPad / 32) * 64;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(sb_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, CoutPad, n3, g1);
    return hipGetLastError();
}

}  // namespace td
