// Deep-level latency flavour of the implicit-GEMM convolution (bf16 / fp16, gfx950), round 5: 64 pixels x 16 couts per workgroup on
// v_mfma_f32_16x16x32, for the launches of a small batch whose maps are so small (16x16 level of one 64x64 latent tile: BASELINE configs[1]) that even
// the 64 x 32 tile of conv_sb.hip leaves most of the chip idle.  conv_sb fills the chip there by splitting K over WORKGROUPS as well -- fp32 partial
// planes and a reduce launch behind 44 of the 79 convs of a single-tile forward (4.7 us each, plus 0.3 GB of partial-plane traffic per forward).
// Here the cout tile is 16 wide instead: twice the workgroups per weight byte, every weight byte still read by exactly one wave of one workgroup,
// K split only over the four waves of the workgroup -- no partial plane in HBM and NO reduce launch.
// MEASURED (profiles/r05_conv_s16_deep_levels.txt): it wins where its grid reaches about half the chip -- the 16x16 level of one tile, 144
// workgroups: 576 -> 576 8.9 / 11.8 us hot / cold against 10.1 / 14.5 for conv_sb + 3 K-slices + reduce -- and LOSES at the 8x8 level (48 workgroups:
// 768 -> 768 12.1 us against 9.0): a workgroup streams 16 x K x 2 bytes (221 KB there) with 9 KiB in flight per wave = 36 KB per CU, ~20 GB/s per CU at
// ~2 us of loaded latency, and 48 CUs cannot match the 216 workgroups of the split-K plan.  The planner (engine.hip, option "s16") therefore takes this
// flavour only where its grid has >= "s16_min_wgs" (128) workgroups; 24 of the 44 reduce launches stay.
// Same maths / parameter block / fused prologues and epilogues as the other flavours (mp_layers.py:201-221, unet_block.py:116-156):
//   * wave w = (half, gsel): it contracts channels [32 half, 32 half + 32) of every K-group whose index is gsel modulo 2 -- the K loop walks the
//     3x3 K-groups in PAIRS, one halo patch per group of the pair, double-buffered (four patch buffers, ONE barrier per 18 taps);
//   * weights: a third copy in THIS kernel's fragment order ([K-group][half][tap][16-cout tile][lane][16 B], s16_repack_kernel; made on first use),
//     one contiguous 1-KiB load per wave and tap straight into the MFMA A registers, one group of the wave (9 taps) ahead;
//   * 1x1 K-groups: both operands straight from global memory, four groups ahead, no LDS and no barrier (as in conv_sb.hip);
//   * the four partial accumulator sets (16 KB) meet in LDS and are summed in the fixed order ((w0 + w1) + w2) + w3; wave q finishes pixels
//     16 q ... 16 q + 15: a lane holds 4 consecutive couts of one pixel (the 16x16 C layout), i.e. one 8-byte store;
//   * sums of squares: ONE partial plane per 16 couts (this flavour's ops are planned with CoutPad / 16 planes; the consumer adds them ascending).
// Results depend on the tile configuration only, never on the batch; they differ in rounding from the other flavours (another K order): engine
// option "batch_invariant" keeps conv_glds everywhere.
#include "conv_common.h"

namespace td {

// dst (this kernel's fragment order) <- src (conv_glds slab [kstep][CoutPad][128 B], slots swizzled).  One thread per 16-byte piece.
// n3 = leading 3x3 K-groups (every 3x3 segment precedes every 1x1 segment), g1 = 1x1 K-groups behind them.
__global__ __launch_bounds__(256) void s16_repack_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int CoutPad, int n3, int g1) {
    const int NCT = CoutPad / 16;
    const size_t total = ((size_t)n3 * 18 + (size_t)g1 * 2) * NCT * 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const size_t r0 = idx >> 6;
    const int ct = (int)(r0 % NCT);
    const size_t kunit = r0 / NCT;
    int kstep, half;
    if (kunit < (size_t)n3 * 18) { const int g = (int)(kunit / 18), r = (int)(kunit % 18); half = r / 9; kstep = g * 9 + r % 9; }
    else { const size_t r = kunit - (size_t)n3 * 18; half = (int)(r & 1); kstep = n3 * 9 + (int)(r >> 1); }
    const int cout = ct * 16 + (lane & 15), q = half * 4 + (lane >> 4);   // A operand of v_mfma_f32_16x16x32: row lane & 15, channels 8 (lane >> 4) ... + 7
    dst[idx] = src[((size_t)kstep * CoutPad + cout) * 8 + (q ^ TD_SWZ(cout))];
}

template <typename V> __device__ __forceinline__ V s16_gld(const void* q) { return *(const __attribute__((address_space(1))) V*)q; }
template <typename V> __device__ __forceinline__ void s16_gst(void* q, V v) { *(__attribute__((address_space(1))) V*)q = v; }
__device__ __forceinline__ int s16_src_pixel(int n, int y, int x, int Hs, int Ws, int dn, int up) { return (n * Hs + ((y << dn) >> up)) * Ws + ((x << dn) >> up); }

template <typename T, int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv_s16_kernel(const ConvParams p) {
    typedef typename Half<T>::x8 hx8;
    typedef typename Half<T>::x4 hx4;
    constexpr int NTHR = 256, NW = 4;
    constexpr int TPIX = TH * TW, MT = TPIX / 16;
    constexpr int PH = TH + 2, PW = TW == 8 ? 12 : TW + 2, NPATCH = PH * PW;
    constexpr int CHUNK = 64, PER16 = 8, PITCH = 144;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int A_BYTES = NPATCH * PITCH;
    constexpr int RN_BASE = 4 * A_BYTES;   // [pair buffer 0: group 0, group 1][pair buffer 1: group 0, group 1][1/rms per patch pixel]
    static_assert(TPIX == 64 && A_BYTES % 16 == 0 && NW * MT * 1024 <= 4 * A_BYTES, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object: its offset is 0
    float* s_rn = (float*)(smem + RN_BASE);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int half = wave & 1, gsel = wave >> 1;

    // ---- kernel arguments of the prologue, one burst (conv_sb.hip says why)
    unsigned k_d1 = p.sb_d1, k_m1 = p.sb_m1, k_m2 = p.sb_m2, k_m3 = p.sb_m3, k_g8 = p.sb_grid8;
    int k_order = p.sb_order, k_tx = p.tiles_x, k_ty = p.tiles_y, k_N = p.N, k_H = p.H, k_W = p.W, k_cpad = p.CoutPad, k_n3 = p.sb_n3, k_ng = p.kgroups;
    const unsigned char* k_ws = (const unsigned char*)p.wpack_sb;
    const T* k_s0src = (const T*)p.seg[0].src;
    int k_s0C = p.seg[0].C, k_s0cs = p.seg[0].cstride, k_s0Hs = p.seg[0].Hs, k_s0Ws = p.seg[0].Ws, k_s0rs = p.seg[0].resample, k_s0xf = p.seg[0].xform;
    float k_s0sc = p.seg[0].scale;
    asm volatile("" : "+s"(k_d1), "+s"(k_m1), "+s"(k_m2), "+s"(k_m3), "+s"(k_g8), "+s"(k_order), "+s"(k_tx), "+s"(k_ty), "+s"(k_N), "+s"(k_H), "+s"(k_W),
                 "+s"(k_cpad), "+s"(k_n3), "+s"(k_ng), "+s"(k_ws), "+s"(k_s0src), "+s"(k_s0C), "+s"(k_s0cs), "+s"(k_s0Hs), "+s"(k_s0Ws), "+s"(k_s0rs), "+s"(k_s0xf), "+s"(k_s0sc));

    unsigned bid = blockIdx.x;
    if (k_g8) bid = (bid & 7) * k_g8 + (bid >> 3);   // XCD x takes a contiguous range of logical ids (speed only)
    const unsigned q1 = td_udiv(bid, k_d1, k_m1), r1 = bid - q1 * k_d1;
    const unsigned ntile = k_order ? q1 : r1, mtile = k_order ? r1 : q1;   // sb_order as in conv_sb.hip
    const unsigned q2 = td_udiv(mtile, (unsigned)k_tx, k_m2), txi = mtile - q2 * (unsigned)k_tx;
    const int n0 = (int)td_udiv(q2, (unsigned)k_ty, k_m3), tyi = (int)(q2 - (unsigned)n0 * (unsigned)k_ty);
    const int y0 = tyi * TH, x0 = (int)txi * TW, co0 = (int)ntile * 16;
    const int NCT = k_cpad / 16, n3 = k_n3, n1 = k_ng - k_n3;

    // ---- weight stream.  This lane's 16 bytes of (3x3 group g, half, tap t) live at wl3 + (g * 18 + t) * tstep, of (1x1 group j, half) at
    // wl1 + j * 2 * tstep.
    const size_t tstep = (size_t)NCT * 1024;
    const unsigned char* wl3 = k_ws + ((size_t)(half * 9) * NCT + co0 / 16) * 1024 + lane * 16;
    const unsigned char* wl1 = k_ws + ((size_t)n3 * 18 * NCT + (size_t)half * NCT + co0 / 16) * 1024 + lane * 16;
    u32x4 wr[9];

    // ---- activation-patch staging (3x3 groups): per thread A_ITERS 16-byte pieces (patch pixel e>>3, slot e&7), as in conv_sb.hip; the two
    // groups of a pair are fetched into av[0] / av[1]
    int a_coord[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int e = tid + it * NTHR, pp = e >> 3;
        const int py = pp / PW, px = pp % PW;
        const int y = y0 + py - 1, x = x0 + px - 1;
        const bool ok = (pp < NPATCH) && px < TW + 2 && n0 < k_N && y >= 0 && y < k_H && x >= 0 && x < k_W;
        a_coord[it] = ok ? ((y << 11) | x) : -1;
    }
    u32x4 av[2][A_ITERS];
    int caoff[A_ITERS];
    const T* csrc = k_s0src;
    int cseg = 0, cchunk = 0, cnch = k_s0C / CHUNK, c_xf = k_s0xf, cur_xf[2] = {0, 0};
    float c_sc = k_s0sc, cur_sc[2] = {1.f, 1.f};
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int c = a_coord[it];
        caoff[it] = c >= 0 ? s16_src_pixel(n0, c >> 11, c & 2047, k_s0Hs, k_s0Ws, k_s0rs == 1, k_s0rs == 2) * k_s0cs + (tid & 7) * PER16 : 0;
    }
#define TD_SEG_NEXT()                                                                                                 \
    {                                                                                                                 \
        ++cseg; cchunk = 0;                                                                                           \
        const ConvSeg& sg_ = p.seg[cseg];                                                                             \
        csrc = (const T*)sg_.src; c_xf = sg_.xform; c_sc = sg_.scale; cnch = sg_.C / CHUNK;                           \
        const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            const int c_ = a_coord[it_];                                                                              \
            caoff[it_] = c_ >= 0 ? s16_src_pixel(n0, c_ >> 11, c_ & 2047, Hs_, Ws_, rs_ == 1, rs_ == 2) * cs_ + (tid & 7) * PER16 : 0; \
        }                                                                                                             \
    }
    // the cursor moves to the next 3x3 K-group: its source pointer and transform go to slot K of the pair being fetched
    const T* asrc[2] = {k_s0src, k_s0src};
#define TD_CURSOR(K)                                                                                   \
    {                                                                                                  \
        if (cchunk == cnch) TD_SEG_NEXT();                                                             \
        cur_xf[K] = c_xf; cur_sc[K] = c_sc;                                                            \
        asrc[K] = csrc + cchunk * CHUNK;                                                               \
        ++cchunk;                                                                                      \
    }
    // ALWAYS issued (a group that does not exist re-reads the start of a live tensor): every load of the K loop is unconditional, so that hipcc's
    // counted vmcnt(N) are exact -- a load inside a wave-uniform branch makes its wait-count pass merge the branch states, and each tap would
    // wait for patch loads requested a few hundred cycles earlier (conv_sb.hip has the measurements)
#define TD_FETCH(K, EXISTS)                                                                            \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) av[K][it_] = s16_gld<u32x4>(asrc[K] + ((EXISTS) ? caoff[it_] : 0)); \
    }
#define TD_STORE_A(K, BUF)                                                                             \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = a_coord[it_] >= 0 ? av[K][it_] : u32x4{0u, 0u, 0u, 0u};                     \
                if (cur_xf[K] != 0 && a_coord[it_] >= 0) {                                             \
                    float s_ = cur_sc[K];                                                              \
                    if (cur_xf[K] == 2) s_ *= s_rn[pp_];                                               \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(smem + ((BUF) * 2 + (K)) * A_BYTES + pp_ * PITCH + (slot_ << 4)) = v_;       \
            }                                                                                          \
        }                                                                                              \
    }
    if (n3 > 0) TD_CURSOR(0);
    TD_FETCH(0, n3 > 0);
    if (n3 > 1) TD_CURSOR(1);
    TD_FETCH(1, n3 > 1);
    // the nine taps of the wave's first group: requested BEHIND the first pair's patches -- the order of the K loop (patches of the next pair, then
    // the refills).  hipcc merges the loop-entry and back-edge wait-count states: weights first here made every restage of the loop wait for all but
    // seven of the loads in flight, i.e. for the refills issued a moment earlier.  Unconditional and index-clamped like every load of the loop.
    {
        const unsigned char* w0 = gsel < n3 ? wl3 + (size_t)gsel * 18 * tstep : k_ws + lane * 16;
        const size_t ws0 = gsel < n3 ? tstep : 0;   // (a wave without a 3x3 group re-reads the first KiB of the slab: it may be shorter than nine taps)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t] = s16_gld<u32x4>(w0 + t * ws0);
    }

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source, for the patch pixels
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (k_s0xf == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = k_s0Hs; rn_Ws = k_s0Ws; rn_res = k_s0rs; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    if (rn_sumsq) {
        const size_t npix = (size_t)k_N * rn_Hs * rn_Ws;
        for (int pp = tid; pp < NPATCH; pp += NTHR) {
            const int py = pp / PW, px = pp % PW;
            const int y = y0 + py - 1, x = x0 + px - 1;
            float rn = 0.f;
            if (n0 < k_N && y >= 0 && y < k_H && x >= 0 && x < k_W)
                rn = pixel_rn(rn_sumsq, rn_parts, npix, s16_src_pixel(n0, y, x, rn_Hs, rn_Ws, rn_res == 1, rn_res == 2), rn_invc);
            s_rn[pp] = rn;
        }
    }

    // ---- MFMA operand addressing.  B operand of v_mfma_f32_16x16x32: column = pixel lr of the 16-pixel block, channels 8 lg ... + 7 of the wave's
    // 32-channel half.  xbase = LDS offset of the TOP-LEFT tap of this lane's pixel inside a patch buffer (+ its channel slot)
    unsigned xbase[MT];
    int ety[MT], etx[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = i * 16 + lr;
        ety[i] = q / TW; etx[i] = q % TW;
        xbase[i] = (unsigned)(ety[i] * PW + etx[i]) * PITCH + (unsigned)half * 64u + (unsigned)lg * 16u;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (k_s0xf == 2) __syncthreads();   // the pixel-norm table is needed by the first restage only when the 3x3 source is normed

#define TD_TOFF(TAP) ((((TAP) / 3) * PW + ((TAP) % 3)) * PITCH)
    // nine taps of this wave's group in pair buffer BUF; REFILL: ring slot t is refilled right behind tap t from WNEXT (two sched_barriers per tap,
    // as in conv_sb.hip: left alone, hipcc sinks the nine refills to the end of the group -- a prefetch distance of one restage instead of one
    // whole pair -- and lets the LDS reads drift behind the MFMAs)
#define TD_TAPS(BUF, REFILL, WNEXT)                                                                                   \
    {                                                                                                                 \
        unsigned xb_[MT];                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xb_[i_] = xbase[i_] + (unsigned)((BUF) * 2 + gsel) * (unsigned)A_BYTES; \
        u32x4 xf_[2][MT];                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xf_[0][i_] = *(const u32x4*)(smem + xb_[i_] + TD_TOFF(0));  \
        _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) {                                                            \
            if (t_ < 8) {                                                                                             \
                _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xf_[(t_ + 1) & 1][i_] = *(const u32x4*)(smem + xb_[i_] + TD_TOFF(t_ + 1)); \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                         \
                acc[i_] = Half<T>::mfma16(__builtin_bit_cast(hx8, wr[t_]), __builtin_bit_cast(hx8, xf_[t_ & 1][i_]), acc[i_]); \
            if (REFILL) wr[t_] = s16_gld<u32x4>((WNEXT) + t_ * tstep);                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    }
    // whole pairs: straight-line as far as loads go (both groups exist; what is fetched ahead may not: index-clamped)
    const int nfull = n3 >> 1;
    for (int pi = 0; pi < nfull; ++pi) {
        const int buf = pi & 1;
        TD_STORE_A(0, buf);
        TD_STORE_A(1, buf);
        const bool more0 = 2 * pi + 2 < n3, more1 = 2 * pi + 3 < n3;
        if (more0) TD_CURSOR(0);
        TD_FETCH(0, more0);
        if (more1) TD_CURSOR(1);
        TD_FETCH(1, more1);
        // passing this barrier: the patches of pair pi are visible; every wave has finished reading the buffers pair pi + 1 will be written to
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int g = 2 * pi + gsel;
        // the wave's next group (two further on); past the end: its own group again, never used
        const unsigned char* wnext = wl3 + (size_t)(g + 2 < n3 ? g + 2 : g) * 18 * tstep;
        TD_TAPS(buf, true, wnext);
    }
    if (n3 & 1) {   // the odd last group: one patch, contracted by the waves gsel = 0 (their ring holds it)
        const int buf = nfull & 1;
        TD_STORE_A(0, buf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (gsel == 0) TD_TAPS(buf, false, wl3);
    }
#undef TD_TAPS
#undef TD_TOFF
#undef TD_SEG_NEXT
#undef TD_CURSOR
#undef TD_FETCH
#undef TD_STORE_A

    // ---- epilogue kernel arguments in one burst, and the operands of the 16 pixels x 4 couts this lane will finish (block q = wave), requested
    // BEFORE the 1x1 phase and the reduction (unconditional, index-clamped: a lane without a valid output reads 16 harmless bytes of the weights)
    int e_epi = p.epi, e_of32 = p.out_f32, e_Cout = p.Cout, e_ocs = p.out_cstride, e_cvs = p.cvec_stride, e_rcs = p.res_cstride, e_rHs = p.res_Hs, e_rWs = p.res_Ws,
        e_rrs = p.res_resample;
    const void* e_res = p.res; const float* e_rss = p.res_sumsq; const float* e_cvec = p.cvec;
    void* e_out = p.out; void* e_out2 = p.out2; float* e_oss = p.out_sumsq;
    float e_rsc = p.res_scale, e_clip = p.clip, e_o2s = p.out2_scale;
    asm volatile("" : "+s"(e_epi), "+s"(e_of32), "+s"(e_Cout), "+s"(e_ocs), "+s"(e_cvs), "+s"(e_rcs), "+s"(e_rHs), "+s"(e_rWs), "+s"(e_rrs), "+s"(e_res), "+s"(e_rss),
                 "+s"(e_cvec), "+s"(e_out), "+s"(e_out2), "+s"(e_oss), "+s"(e_rsc), "+s"(e_clip), "+s"(e_o2s));
    const size_t M = (size_t)k_N * k_H * k_W;
    const int q = wave;
    const int qp = q * 16 + lr, qty = qp / TW, qtx = qp % TW;
    const int ey = y0 + qty, ex = x0 + qtx, cot = co0 + 4 * lg;   // this lane's pixel and its first cout
    const bool eok = n0 < k_N && ey < k_H && ex < k_W && cot < k_cpad;
    const bool lean = !e_of32 && sizeof(T) == 2 && (e_Cout & 3) == 0 && e_epi != EPI_DPM_STEP;   // 16-bit NHWC output, 4 couts = one 8-byte store
    const bool has_res = e_epi == EPI_RESIDUAL && e_res != nullptr;
    const int esp = (eok && has_res) ? s16_src_pixel(n0, ey, ex, e_rHs, e_rWs, e_rrs == 1, e_rrs == 2) : 0;
    f32x4 aux_c; u32x2 aux_r;
    {
        const bool use = eok && lean && cot < e_Cout;
        const float* crow = (use && e_epi == EPI_EMB_SILU) ? e_cvec + (size_t)n0 * e_cvs + cot : (const float*)k_ws;
        const T* rrow = (use && has_res) ? (const T*)e_res + (size_t)esp * e_rcs + cot : (const T*)k_ws;
        aux_c = s16_gld<f32x4>(crow); aux_r = s16_gld<u32x2>(rrow);
    }

    // ---------------- 1x1 K-groups of this wave (gsel, gsel + 2, ...): both operands straight from global memory, D1 groups ahead
    if (n1 > 0) {
        constexpr int D1 = 4;
        int sg1 = 0;
        while (sg1 < p.nseg && p.seg[sg1].taps == 9) ++sg1;   // first 1x1 segment
        u32x4 wa[D1], xa[D1][MT];
        int poff[MT];
        const T* psrc = nullptr; int pn = 0, pseg = sg1 - 1, pchunk = 0;
        auto seg_open = [&]() {
            ++pseg; pchunk = 0;
            const ConvSeg& sg_ = p.seg[pseg];
            psrc = (const T*)sg_.src; pn = sg_.C / CHUNK;
            const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int y = y0 + ety[i], x = x0 + etx[i];
                // a pixel outside the image is an MFMA column nobody stores: any readable address will do
                const int pix = (n0 < k_N && y < k_H && x < k_W) ? s16_src_pixel(n0, y, x, Hs_, Ws_, rs_ == 1, rs_ == 2) : 0;
                poff[i] = pix * cs_ + half * 32 + lg * 8;
            }
        };
        auto advance = [&]() { ++pchunk; if (pchunk == pn && pseg + 1 < p.nseg) seg_open(); };   // the cursor moves ONE group
        int pj = gsel;                       // 1x1 group index the prefetch cursor stands on (this wave takes every second one)
        auto issue = [&](int slot_) {        // slot_ is a compile-time constant at every call site
            if (pj < n1) {
                wa[slot_] = s16_gld<u32x4>(wl1 + (size_t)pj * 2 * tstep);
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[slot_][i] = s16_gld<u32x4>(psrc + poff[i] + pchunk * CHUNK);
                pj += 2; advance(); advance();   // (past the last group the cursor is never used again)
            }
        };
        seg_open();
        if (gsel) advance();
#pragma unroll
        for (int d = 0; d < D1; ++d) issue(d);
        for (int j = gsel; j < n1; j += 2 * D1) {
#pragma unroll
            for (int d = 0; d < D1; ++d) {
                if (j + 2 * d < n1) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = Half<T>::mfma16(__builtin_bit_cast(hx8, wa[d]), __builtin_bit_cast(hx8, xa[d][i]), acc[i]);
                    issue(d);
                }
            }
        }
    }

    // ---------------- in-workgroup K reduction: four partial accumulator sets through LDS (over the dead patch buffers)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done reading the patches
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < MT; ++i) *(f32x4*)(smem + (wave * MT + i) * 1024 + lane * 16) = acc[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x4 v = *(const f32x4*)(smem + (0 * MT + q) * 1024 + lane * 16);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *(const f32x4*)(smem + (w * MT + q) * 1024 + lane * 16);

    // ---------------- epilogue: 4 consecutive couts of one pixel per lane
    float ss = 0.f;
    if (eok) {
        const float rn = (e_rss != nullptr) ? s_rn[(qty + 1) * PW + (qtx + 1)] : 1.f;
        if (lean) {
            if (cot < e_Cout) {   // the arithmetic of epilogue4 (conv_common.h) on pinned scalars
                if (e_epi == EPI_EMB_SILU) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = Elem<T>::silu(v[k] * aux_c[k]);
                } else if (e_epi == EPI_RESIDUAL) {
                    if (has_res) {
                        const hx4 rh = __builtin_bit_cast(hx4, aux_r);
                        const float s = e_rsc * rn;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += s * (float)rh[k];
                    }
                    if (e_clip > 0.f) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(v[k], -e_clip), e_clip);
                    }
                }
                const size_t pix = ((size_t)n0 * k_H + ey) * k_W + ex;
                const hx4 h = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                s16_gst<u32x2>((T*)e_out + pix * e_ocs + cot, __builtin_bit_cast(u32x2, h));
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float f = (float)h[k]; ss += f * f; }
                if (e_out2) {   // from the ROUNDED value: what the consumer's patch staging would compute
                    hx4 a;
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = (T)Elem<T>::silu_scaled((float)h[k], e_o2s);
                    s16_gst<u32x2>((T*)e_out2 + pix * e_ocs + cot, __builtin_bit_cast(u32x2, a));
                }
            }
        } else {   // fp32 / few-channel outputs, the fused solver step: the shared scalar epilogue
            f32x4 aux = {0.f, 0.f, 0.f, 0.f};
            if (e_epi == EPI_EMB_SILU) aux = s16_gld<f32x4>(e_cvec + (size_t)n0 * e_cvs + cot);
            else if (has_res) aux = load4<T>(p.res, (size_t)esp * e_rcs + cot);
            ss = epilogue4<T>(p, n0, ey, ex, cot, v, rn, aux);
        }
    }
    if (e_oss) {   // one partial per 16-cout tile (ops on this flavour are planned with CoutPad / 16 planes)
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (eok && lg == 0) s16_gst<float>(e_oss + (size_t)(co0 / 16) * M + ((size_t)n0 * k_H + ey) * k_W + ex, ss);
    }
}

template <typename T, int TH, int TW>
static hipError_t launch_s16_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int NPATCH = (TH + 2) * (TW == 8 ? 12 : TW + 2);
    constexpr size_t lds = (size_t)4 * NPATCH * 144 + NPATCH * 4;
    if (!p.wpack_sb || p.ksplit != 1 || p.CoutPad % 16 != 0) return hipErrorInvalidValue;
    bool seen1 = false;   // every 3x3 segment before every 1x1 segment, 1x1 sources untransformed
    int n3 = 0;
    for (int s = 0; s < p.nseg; ++s) {
        if (p.seg[s].taps == 9) { if (seen1) return hipErrorInvalidValue; n3 += p.seg[s].C / 64; }
        else { seen1 = true; if (p.seg[s].xform != 0) return hipErrorInvalidValue; }
    }
    if (n3 != p.sb_n3) return hipErrorInvalidValue;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups, grid = p.n_ntiles * mtiles;
    if (grid <= 0 || p.n_ntiles != p.CoutPad / 16 || (long long)grid * std::max(mtiles, p.n_ntiles) >= ((long long)1 << 32)) return hipErrorInvalidValue;   // td_udiv's range
    ConvParams q = p;
    q.sb_d1 = p.sb_order ? mtiles : p.n_ntiles; q.sb_m1 = td_magic(q.sb_d1); q.sb_m2 = td_magic(p.tiles_x); q.sb_m3 = td_magic(p.tiles_y);
    q.sb_grid8 = (grid & 7) == 0 ? (unsigned)grid >> 3 : 0u;
    auto kern = conv_s16_kernel<T, TH, TW>;
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
    return hipGetLastError();
}

// dtype: 1 bf16, 2 fp16.  Tile: 64 pixels (4 x 16, narrow maps 8 x 8) x 16 couts; the caller sets tiles_x / tiles_y for it, img_groups = N,
// n_ntiles = CoutPad / 16, ksplit = 1, wpack_sb = THIS flavour's weight copy (launch_s16_repack), sb_n3, sb_order.
hipError_t launch_conv_s16(const ConvParams& p, int dtype, bool narrow, hipStream_t st) {
    if (dtype == 2) return narrow ? launch_s16_cfg<_Float16, 8, 8>(p, st) : launch_s16_cfg<_Float16, 4, 16>(p, st);
    return narrow ? launch_s16_cfg<__bf16, 8, 8>(p, st) : launch_s16_cfg<__bf16, 4, 16>(p, st);
}

// Fragment-order copy of a conv's packed weights for this flavour (device to device).  `dst` holds ksteps * CoutPad * 128 bytes.
hipError_t launch_s16_repack(const void* src, void* dst, int CoutPad, int n3, int g1, hipStream_t st) {
    const size_t total = ((size_t)n3 * 18 + (size_t)g1 * 2) * (CoutPad / 16) * 64;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(s16_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, CoutPad, n3, g1);
    return hipGetLastError();
}

}  // namespace td
