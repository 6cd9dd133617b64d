// Persistent-stream flavour of the implicit-GEMM convolution (bf16 / fp16, gfx950), round 3.
// Same maths, parameter block, weight slab, LDS fragment layout, tile shapes and K order as conv_glds.hip -- every output is accumulated by the
// same v_mfma_f32_32x32x16 sequence and leaves through the same epi_unit8 arithmetic, so the results are BIT-IDENTICAL to conv_glds.hip
// (tests/test_gpu_parity.py::test_conv_tile_variants_bit_identical, tools/conv_bench.hip flavour 6).  What differs is how a workgroup's time is
// organised.  Measured on MI355X (profiles/r03_conv_phase_trace.txt): in conv_glds a workgroup of the 192-channel 64x64 layers lives 52 k
// cycles of which 21 k are its tap loop; prologue (8.5 k: kernel-argument fetch, first weight tiles, an HBM round trip for the halo patch) and
// epilogue (15 k: one dependent load -> arithmetic -> store round per 8 couts) are latency-bound, so the second workgroup of the CU cannot
// compress them, and the pair settles into a stable lockstep (both in their memory phases, then both competing for the matrix pipe): neither
// staggering the pair nor ramping the start times over the chip changes the total.  Here:
//   * PERSISTENT: a workgroup walks over many (pixel tile, cout tile) work items.  The weight ring streams straight on into the next item's
//     slab, and the patch staging cursor runs one K-group ahead of the compute cursor ACROSS segment and item boundaries, so after the first
//     item nobody waits for a prologue (and the exposed HBM round trip conv_glds pays at every K-segment boundary is gone too).
//   * DRIPPED EPILOGUE: when an item's last tap is done its accumulators move to a second register set and the next item starts at once;
//     the epilogue units (8 couts x 1 pixel per lane each: <= 8 per wave) are executed one per tap during the next item's first nine taps,
//     right after the tap's barrier where the wave's memory queue is empty anyway.  The residual operands of all units are requested in one
//     go at the item boundary and have two and a half taps to arrive; modulation vectors are fetched one unit ahead.  A wave's epilogue
//     arithmetic (two transcendental pairs per output for the activated second output) then runs under the MFMAs of the other wave of its
//     SIMD instead of in a phase where the matrix pipe has nothing to do.
//   * the pixel-norm factors (1/rms) of the next item's patch are fetched during the current item's last nine taps into the other half of a
//     double-buffered LDS table.
//   * work order: XCD x (workgroups b with b % 8 == x) owns the pixel tiles m % 8 == x and walks them with the cout tile fastest, so the
//     siblings that share a halo patch run side by side on one L2.  `reverse` walks the pixel tiles backwards: the host alternates it from
//     layer to layer so that a layer first reads what its producer wrote last (still in the 256 MB Infinity Cache; +7 % on the 192-channel
//     64x64 layers in a chained A/B).
// Restrictions (the host falls back to conv_glds.hip otherwise): no split-K, 16-bit NHWC output with Cout % 8 == 0, the first K segment is
// 3x3, 3x3 segments precede 1x1 segments and the 1x1 segments hold a multiple of 3 K-groups (the ring slot of an item's first K-step is then
// always 0, which the compile-time slot numbers of the nine-tap groups rely on).
#include "conv_common.h"
#include <type_traits>

namespace td {

#ifdef TD_TRACE  // in-kernel phase timing with s_memtime (tools/conv_bench.hip only)
#define PS_T(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PS_TACC(acc_, a, b) acc_ += (b) - (a)
#else
#define PS_T(v)
#define PS_TACC(acc_, a, b)
#endif

template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N + 3) / 4 < 2 ? 2 : (WAVES_M * WAVES_N + 3) / 4) void conv_ps_kernel(const ConvParams p) {
    typedef typename Half<T>::x8 hx8;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int TPIX = TH * TW, BM = NIMG * TPIX;
    constexpr int PH = TH + 2, PW = TW == 8 ? 12 : TW + 2, PPI = PH * PW, NPATCH = NIMG * PPI;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int CHUNK = 64, PER16 = 8;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16);
    constexpr int B_BYTES = NBI * NTHR * 16, RING = 3;
    constexpr int PITCH = 144;
    constexpr int A_BASE = RING * B_BYTES, A_BYTES = NPATCH * PITCH, RN_BASE = A_BASE + A_BYTES;
    constexpr int NU = MT * NT * 2;  // epilogue units per wave: 8 couts x 1 pixel per lane each
    static_assert(WM % 32 == 0 && WN % 32 == 0, "tile shape");
    static_assert(NU <= 8 && NPATCH <= NTHR, "one epilogue unit per tap of a nine-tap group; one patch pixel per thread for the 1/rms table");
    static_assert((RING - 1) * B_BYTES + (NT - 1) * 4096 + 128 < 65536 && 2 * PW * PITCH + 2 * PITCH + 128 < 65536, "ds_read offset field");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0
    unsigned char* s_a = smem + A_BASE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef TD_TRACE
    unsigned long long tr_group = 0, tr_stage = 0, tr_epi = 0, tr_adv = 0, tr_items = 0;
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    PS_T(tr_start);

    // ---- work order (speed only).  Workgroup b sits on XCD b % 8; XCD x owns a CONTIGUOUS range of pixel tiles (spatial neighbours share halo
    // rows through its L2) and its workgroups take (pixel tile, cout tile) pairs round-robin, cout tile fastest (the siblings of a pixel tile
    // run side by side).  For a fixed workgroup the pixel tile grows with j, so the first invalid item ends its walk.
    const int n_nt = p.n_ntiles, tiles_xy = p.tiles_x * p.tiles_y, mtiles = tiles_xy * p.img_groups;
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int mper = (mtiles + 7) >> 3, m_lo = xcd * mper, m_cnt = max(0, min(mper, mtiles - m_lo));
    auto decode = [&](int j, int& n0, int& y0, int& x0, int& co0) -> bool {
        const int q = j * per_x + wslot, ml = q / n_nt;
        const int nt = q - ml * n_nt;
        const bool ok = ml < m_cnt;
        int mt = ok ? m_lo + ml : 0;
        if (p.reverse & 1) mt = mtiles - 1 - mt;
        const int ig = mt / tiles_xy, r = mt - ig * tiles_xy, tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
        // the divisions run on the vector ALU: tell the compiler the results are wave-uniform (they feed scalar address registers)
        n0 = __builtin_amdgcn_readfirstlane(ig * NIMG); y0 = __builtin_amdgcn_readfirstlane(tyi * TH); x0 = __builtin_amdgcn_readfirstlane(txi * TW);
        co0 = __builtin_amdgcn_readfirstlane(nt * BN);
        return __builtin_amdgcn_readfirstlane((int)ok) != 0;
    };
    int cn0, cy0, cx0, cco0;  // the item under the compute cursor
    if (!decode(0, cn0, cy0, cx0, cco0)) return;  // uniform over the workgroup
    int ksteps_item = 0;
    for (int s = 0; s < p.nseg; ++s) ksteps_item += (p.seg[s].C / CHUNK) * p.seg[s].taps;

    // ---- weight ring: every tap fetches the tile two K-steps ahead; the stream runs on into the next item's slab (or, past the last item,
    // harmlessly re-fetches tile 0, which nobody reads).
    const size_t wstep = (size_t)p.CoutPad * 128;
    const unsigned char* wnext = (const unsigned char*)p.wpack + (size_t)cco0 * 128;
    int wleft = ksteps_item, wj = 0;
    unsigned wvoff[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) wvoff[i] = (unsigned)tid * 16u + (unsigned)i * NTHR * 16u;
    const unsigned ldsw = (unsigned)wave * 1024u;
#define PS_GLDS_B(SLOT)                                                                                      \
    {                                                                                                        \
        const unsigned long long wa_ = (unsigned long long)wnext;                                            \
        const unsigned char* wu_ = (const unsigned char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wa_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)wa_)); \
        _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) TD_GLDS16(wvoff[i_], wu_, ldsw, (SLOT) * B_BYTES + i_ * NTHR * 16); \
        wnext += wstep;                                                                                      \
        if (--wleft == 0) {                                                                                  \
            int a_, b_, c_, co_;                                                                             \
            ++wj;                                                                                            \
            if (decode(wj, a_, b_, c_, co_)) { wnext = (const unsigned char*)p.wpack + (size_t)co_ * 128; wleft = ksteps_item; } \
            else { wnext = (const unsigned char*)p.wpack; wleft = 1 << 30; }                                 \
        }                                                                                                    \
    }
    PS_GLDS_B(0);
    PS_GLDS_B(1);

    // ---- patch staging.  The STAGING cursor (item sj, segment s_seg, chunk s_chunk) runs one K-group ahead of the compute cursor; its
    // descriptor lives in registers: a_coord (patch coordinates of this thread's pieces in the staged item), aoff (element offsets in the
    // staged segment's source, -1 = zero fill), s_src / s_xform / s_scale.
    int a_coord[A_ITERS], aoff[A_ITERS];
    u32x4 av[A_ITERS];
    auto set_coords = [&](int n0, int y0, int x0) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int e = tid + it * NTHR, pp = e >> 3;
            const int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
            const int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
            const bool ok = (pp < NPATCH) && px < TW + 2 && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;  // px >= TW+2: pad columns
            const bool interior = py >= 1 && py <= TH && px >= 1 && px <= TW;
            a_coord[it] = ok ? ((n << 21) | (y << 11) | (x << 1) | (interior ? 1 : 0)) : -1;
        }
    };
    const T* s_src = nullptr;
    int s_taps = 9, s_xform = 0, s_nchunks = 0;
    float s_scale = 1.f;
#define PS_SEG_BEGIN(SEG)                                                                                             \
    {                                                                                                                 \
        const ConvSeg& sg_ = p.seg[SEG];                                                                              \
        s_src = (const T*)sg_.src; s_taps = sg_.taps; s_xform = sg_.xform; s_scale = sg_.scale; s_nchunks = sg_.C / CHUNK; \
        const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            const int c_ = a_coord[it_];                                                                              \
            aoff[it_] = -1;                                                                                           \
            if (c_ >= 0 && (s_taps == 9 || (c_ & 1)))                                                                 \
                aoff[it_] = src_pixel(c_ >> 21, (c_ >> 11) & 1023, (c_ >> 1) & 1023, Hs_, Ws_, rs_) * cs_ + (tid & 7) * PER16; \
        }                                                                                                             \
    }
#define PS_LOAD_A(CH)                                                                                  \
    {                                                                                                  \
        const T* src_ = s_src + (CH) * CHUNK;                                                          \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            /* always issued (offset 0 for zero-fill pieces) so that the vmcnt bookkeeping of the main loop is exact */ \
            av[it_] = *(const u32x4*)(src_ + (aoff[it_] >= 0 ? aoff[it_] : 0));                        \
        }                                                                                              \
    }
#define PS_STORE_A(RNBUF)                                                                              \
    {                                                                                                  \
        const float* s_rn_ = (const float*)(smem + RN_BASE + (RNBUF) * (NPATCH * 4));                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = aoff[it_] >= 0 ? av[it_] : u32x4{0u, 0u, 0u, 0u};                           \
                if (s_xform != 0 && aoff[it_] >= 0) {                                                  \
                    float sc_ = s_scale;                                                               \
                    if (s_xform == 2) sc_ *= s_rn_[pp_];                                               \
                    v_ = xform_piece<T>(v_, sc_);                                                      \
                }                                                                                      \
                *(u32x4*)(s_a + pp_ * PITCH + (slot_ << 4)) = v_;                                      \
            }                                                                                          \
        }                                                                                              \
    }
    set_coords(cn0, cy0, cx0);
    PS_SEG_BEGIN(0);
    PS_LOAD_A(0);

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source (first segment with xform 2, else the normed residual), one patch pixel per thread
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (p.seg[0].xform == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = p.seg[0].Hs; rn_Ws = p.seg[0].Ws; rn_res = p.seg[0].resample; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    const size_t rn_npix = (size_t)p.N * rn_Hs * rn_Ws;
    auto tile_rn = [&](int n0, int y0, int x0) -> float {  // value for patch pixel tid (< NPATCH)
        float rn = 0.f;
        if (tid < NPATCH) {
            const int img = tid / PPI, r = tid % PPI, py = r / PW, px = r % PW;
            const int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
            if (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W) rn = pixel_rn(rn_sumsq, rn_parts, rn_npix, src_pixel(n, y, x, rn_Hs, rn_Ws, rn_res), rn_invc);
        }
        return rn;
    };
    int rb = 0;  // half of the 1/rms table that belongs to the compute item
    if (rn_sumsq) {
        const float r0 = tile_rn(cn0, cy0, cx0);
        if (tid < NPATCH) ((float*)(smem + RN_BASE))[tid] = r0;
    }

    // ---- MFMA operand addressing (identical to conv_glds.hip)
    int base_pp[MT];
    unsigned xbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
        base_pp[i] = img * PPI + (ty + 1) * PW + (tx + 1);
        xbase[i] = (unsigned)A_BASE + (unsigned)(base_pp[i] - PW - 1) * PITCH + (unsigned)lh * 16u;
    }
    unsigned wbase[4];
    {
        const int nl = wn * WN + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wbase[ks] = (unsigned)(nl * 128 + (((ks * 2 + lh) ^ TD_SWZ(nl)) << 4));
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __syncthreads();  // 1/rms table visible (first item only: this one may drain the two weight tiles, they are needed next anyway)
    PS_STORE_A(0);

    // staging cursor -> the group after (item 0, segment 0, chunk 0)
    int sj = 0, s_seg = 0, s_chunk = 0, sn0 = cn0, sy0 = cy0, sx0 = cx0, sco0 = cco0;
    bool s_valid = true;
#define PS_STAGE_ADVANCE()                                                                                   \
    {                                                                                                        \
        ++s_chunk;                                                                                           \
        if (s_chunk >= s_nchunks) {                                                                          \
            s_chunk = 0; ++s_seg;                                                                            \
            if (s_seg >= p.nseg) {                                                                           \
                s_seg = 0; ++sj;                                                                             \
                s_valid = decode(sj, sn0, sy0, sx0, sco0);                                                   \
                if (s_valid) set_coords(sn0, sy0, sx0);                                                      \
            }                                                                                                \
            if (s_valid) { PS_SEG_BEGIN(s_seg); }                                                            \
        }                                                                                                    \
    }
    int c_seg = 0, c_chunk = 0, c_nchunks = s_nchunks, c_taps = 9;
    PS_STAGE_ADVANCE();

    // ---- epilogue of the item under the compute cursor.  It runs in one go when the item's last tap is done, but nothing in it waits for
    // memory: the residual runs of all units (16 bytes per lane each) are requested one per tap during the item's LAST nine taps (after the
    // tap's weight tile, so the counted wait of the next tap simply leaves one more request in flight), modulation vectors come from L2 in one
    // batch, and the stores are fire-and-forget.  conv_glds.hip's epilogue did one dependent load -> arithmetic -> store round per unit.
    const size_t M = (size_t)p.N * p.H * p.W;
    const bool has_res = p.epi == EPI_RESIDUAL && p.res != nullptr, is_emb = p.epi == EPI_EMB_SILU, want_ss = p.out_sumsq != nullptr, want_o2 = p.out2 != nullptr;
    const SiluK k_o2 = silu_k(p.out2_scale);
    int pixv[MT], e_sp[MT]; bool okv[MT];
    u32x4 rwv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) rwv[u] = u32x4{0u, 0u, 0u, 0u};
    auto epi_coords = [&]() {  // output pixel / residual source pixel of this lane's pixels in the compute item
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int img, ty, tx;
            frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
            const int n = cn0 + img, y = cy0 + ty, x = cx0 + tx;
            okv[i] = n < p.N && y < p.H && x < p.W;
            pixv[i] = okv[i] ? (n * p.H + y) * p.W + x : 0;
            e_sp[i] = (has_res && okv[i]) ? src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample) : 0;
        }
    };
    auto epi_fetch_r = [&](auto UC) {  // residual run of unit U (EPI_RESIDUAL): 16 bytes per lane in the stored layout
        constexpr int U = decltype(UC)::value < 0 ? 0 : (decltype(UC)::value >= NU ? NU - 1 : decltype(UC)::value);
        constexpr int i = U / (NT * 2), j = (U / 2) % NT, m = U % 2;
        const bool in = cco0 + wn * WN + j * 32 < p.Cout;
        rwv[U] = *(const u32x4*)((const T*)p.res + (size_t)e_sp[i] * p.res_cstride + cco0 + wn * WN + 8 * lh + (in ? j * 32 + m * 16 : 0));
    };
    auto item_epilogue = [&](bool prefetched) {
        if (!prefetched) {
            epi_coords();
            if (has_res) {
#define PS_FR(K) if (NU > (K)) epi_fetch_r(std::integral_constant<int, (K)>{});
                PS_FR(0) PS_FR(1) PS_FR(2) PS_FR(3) PS_FR(4) PS_FR(5) PS_FR(6) PS_FR(7)
#undef PS_FR
            }
        }
        f32x4 cav[NU], cbv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            cav[u] = f32x4{0.f, 0.f, 0.f, 0.f}; cbv[u] = cav[u];
            if (is_emb) {  // modulation values of all units in one batch (L2-resident)
                const int i = u / (NT * 2), j = (u / 2) % NT, m = u % 2;
                int img, ty, tx;
                frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
                const bool in = cco0 + wn * WN + j * 32 < p.Cout;
                const float* crow = p.cvec + (size_t)min(cn0 + img, p.N - 1) * p.cvec_stride + cco0 + wn * WN + 4 * lh + (in ? j * 32 + m * 16 : 0);
                cav[u] = *(const f32x4*)crow; cbv[u] = *(const f32x4*)(crow + 8);
            }
        }
        float rnv[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) rnv[i] = (p.res_sumsq != nullptr) ? ((const float*)(smem + RN_BASE + rb * (NPATCH * 4)))[base_pp[i]] : 1.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float ss = 0.f;
                if (cco0 + wn * WN + j * 32 < p.Cout) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int u = (i * NT + j) * 2 + m;
                        const f32x4 va = {acc[i][j][8 * m + 0], acc[i][j][8 * m + 1], acc[i][j][8 * m + 2], acc[i][j][8 * m + 3]};
                        const f32x4 vb = {acc[i][j][8 * m + 4], acc[i][j][8 * m + 5], acc[i][j][8 * m + 6], acc[i][j][8 * m + 7]};
                        u32x4 o, o2;
                        epi_unit8<T>(p.epi, has_res, p.clip, want_ss, want_o2, va, vb, cav[u], cbv[u], rwv[u], p.res_scale * rnv[i], k_o2, o, o2, ss);
                        const size_t oo = (size_t)pixv[i] * p.out_cstride + cco0 + wn * WN + 8 * lh + j * 32 + m * 16;
                        if (okv[i]) *(u32x4*)((T*)p.out + oo) = o;
                        if (want_o2 && okv[i]) *(u32x4*)((T*)p.out2 + oo) = o2;
                    }
                }
                if (want_ss) {  // one partial per 32-cout MFMA block, own half + partner half (the order conv_glds.hip uses)
                    unsigned a_ = __builtin_bit_cast(unsigned, ss), b_ = a_;
                    swap_halves(a_, b_);  // lanes 0-31: b_ = the value of lane + 32
                    const float st = ss + __builtin_bit_cast(float, b_);
                    if (okv[i] && lh == 0 && cco0 + wn * WN + j * 32 < p.CoutPad) p.out_sumsq[(size_t)((cco0 + wn * WN) / 32 + j) * M + pixv[i]] = st;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        }
    };

#define PS_TOFF(T_) ((((T_) / 3) * PW + ((T_) % 3)) * PITCH)
    int slot = 0;  // ring slot of the current K-step; compile-time inside a nine-tap group, tracked for 1x1 segments
    u32x4 wfA_[NT], xfA_[MT], wfB_[NT], xfB_[MT];
#define PS_FRAG_READ(WF, XF, SLOT, KS, TOFF)                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(smem + wbase[KS] + ((SLOT) * B_BYTES + j_ * 4096)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(smem + xbase[i_] + ((TOFF) + (KS) * 32)); \
    }
#define PS_FRAG_MFMA(WF, XF)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                acc[i_][j_] = Half<T>::mfma32(__builtin_bit_cast(hx8, WF[j_]), __builtin_bit_cast(hx8, XF[i_]), acc[i_][j_]); \
    }
    // One tap (conv_glds.hip's cross-tap pipelined schedule).  After the tap's barrier the wave's memory queue holds nothing older than the
    // patch loads of tap 0 / the residual operands of the pending epilogue, so this is where the dripped work goes, BEFORE the weight tile of
    // two K-steps ahead is requested: the counted waits of the following taps then need no knowledge of it.
#define PS_TAPP(TAPIDX, SLOT, TOFF, TOFF_NEXT)                                                               \
    {                                                                                                        \
        PS_FRAG_MFMA(wfA_, xfA_);                                                                            \
        PS_FRAG_READ(wfA_, xfA_, SLOT, 2, TOFF);                                                             \
        PS_FRAG_MFMA(wfB_, xfB_);                                                                            \
        PS_FRAG_READ(wfB_, xfB_, SLOT, 3, TOFF);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        /* tile k+1 (issued one tap ago) has landed; younger than it: tap 6's patch loads (at tap 7) and ONE residual run (item's last group) */ \
        if ((TAPIDX) >= 1 && (TAPIDX) <= NU && pf_res) {                                                     \
            if ((TAPIDX) == 7 && stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS + 1) : "memory");   \
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                            \
        }                                                                                                    \
        else if ((TAPIDX) == 7 && stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS) : "memory");      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT + MT)) : "memory");                               \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if ((TAPIDX) == 8 && stage) {  /* the patch pieces requested at tap 6 are older than the tile just waited for */ \
            _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) asm volatile("" : "+v"(av[it_]));      \
        }                                                                                                    \
        if ((TAPIDX) == 8 && pf_res) {  /* so are the residual runs (all but, with 8 units, the last one): tell the compiler here, where its wait is free */ \
            _Pragma("unroll") for (int u_ = 0; u_ < NU; ++u_) asm volatile("" : "+v"(rwv[u_]));              \
        }                                                                                                    \
        if ((TAPIDX) == 3 && stage_rn) rn_next = tile_rn(sn0, sy0, sx0);                                     \
        if ((TAPIDX) == 5 && stage_rn && tid < NPATCH) ((float*)(smem + RN_BASE + (rb ^ 1) * (NPATCH * 4)))[tid] = rn_next; \
        PS_GLDS_B(((SLOT) + 2) % RING);                                                                      \
        if ((TAPIDX) == 6 && stage) PS_LOAD_A(s_chunk);  /* late: the registers are free while the epilogue units run (taps 1..NU) */ \
        if ((TAPIDX) < NU && pf_res) epi_fetch_r(std::integral_constant<int, (TAPIDX)>{});                   \
        PS_FRAG_MFMA(wfA_, xfA_);                                                                            \
        if ((TAPIDX) < 8) PS_FRAG_READ(wfA_, xfA_, ((SLOT) + 1) % RING, 0, TOFF_NEXT);                       \
        PS_FRAG_MFMA(wfB_, xfB_);                                                                            \
        if ((TAPIDX) < 8) PS_FRAG_READ(wfB_, xfB_, ((SLOT) + 1) % RING, 1, TOFF_NEXT);                       \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        if ((TAPIDX) < 8) __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        if ((TAPIDX) < 8) __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                           \
    }
    // entry of a nine-tap group: this wave's patch ds_writes are out; tile k (slot 0) was issued >= 1 tap ago, tile k+1 and (right after an
    // item boundary) the residual operands may still be in flight
#define PS_GROUP_ENTRY()                                                                                     \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        PS_FRAG_READ(wfA_, xfA_, 0, 0, PS_TOFF(0));                                                          \
        PS_FRAG_READ(wfB_, xfB_, 0, 1, PS_TOFF(0));                                                          \
    }
    // 1x1 K-step (centre tap of the patch): conv_glds.hip's unpipelined step, with the staging cursor instead of "next chunk of this segment"
#define PS_TAP1(SLOT, TOFF)                                                                                  \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if (stage) PS_LOAD_A(s_chunk);                                                                       \
        PS_FRAG_READ(wfA_, xfA_, SLOT, 0, TOFF);                                                             \
        PS_FRAG_READ(wfB_, xfB_, SLOT, 1, TOFF);                                                             \
        PS_FRAG_MFMA(wfA_, xfA_);                                                                            \
        PS_FRAG_READ(wfA_, xfA_, SLOT, 2, TOFF);                                                             \
        PS_FRAG_MFMA(wfB_, xfB_);                                                                            \
        PS_FRAG_READ(wfB_, xfB_, SLOT, 3, TOFF);                                                             \
        PS_FRAG_MFMA(wfA_, xfA_);                                                                            \
        PS_FRAG_MFMA(wfB_, xfB_);                                                                            \
    }

    PS_T(tr_pro);
    for (;;) {
        PS_T(tq0_);
        const bool last_of_item = c_seg == p.nseg - 1 && c_chunk == c_nchunks - 1;
        const bool stage = s_valid;                                   // a next K-group exists: fetch its patch during this group
        const bool stage_rn = stage && last_of_item && rn_sumsq != nullptr;  // ... and it opens a new item: its 1/rms factors too
        const bool pf_res = last_of_item && has_res && c_taps == 9;   // the item's last nine taps: request its residual runs, one per tap
        float rn_next = 0.f;
        if (c_taps == 9) {  // slot == 0 here (see the restrictions at the top)
            if (pf_res) epi_coords();
            PS_GROUP_ENTRY();
            PS_TAPP(0, 0, PS_TOFF(0), PS_TOFF(1)); PS_TAPP(1, 1, PS_TOFF(1), PS_TOFF(2)); PS_TAPP(2, 2, PS_TOFF(2), PS_TOFF(3));
            PS_TAPP(3, 0, PS_TOFF(3), PS_TOFF(4)); PS_TAPP(4, 1, PS_TOFF(4), PS_TOFF(5)); PS_TAPP(5, 2, PS_TOFF(5), PS_TOFF(6));
            PS_TAPP(6, 0, PS_TOFF(6), PS_TOFF(7)); PS_TAPP(7, 1, PS_TOFF(7), PS_TOFF(8)); PS_TAPP(8, 2, PS_TOFF(8), PS_TOFF(8));
        } else {
            if (slot == 0) PS_TAP1(0, PS_TOFF(4)) else if (slot == 1) PS_TAP1(1, PS_TOFF(4)) else PS_TAP1(2, PS_TOFF(4));
            if (stage_rn) {  // an item that ends on a 1x1 K-step: the next item's factors are fetched here, exposed (no layer of the models does this)
                rn_next = tile_rn(sn0, sy0, sx0);
                if (tid < NPATCH) ((float*)(smem + RN_BASE + (rb ^ 1) * (NPATCH * 4)))[tid] = rn_next;
            }
        }
        PS_T(tq1_); PS_TACC(tr_group, tq0_, tq1_);
        if (stage) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave is done reading the current patch (and the other half of the 1/rms table is visible)
            asm volatile("" ::: "memory");
            PS_STORE_A((last_of_item ? (rb ^ 1) : rb));  // visible to the others after the next tap's lgkmcnt(0) + barrier
        }
        if (c_taps != 9) {
            if (slot == 0) PS_GLDS_B(2) else if (slot == 1) PS_GLDS_B(0) else PS_GLDS_B(1);
            slot = slot == 2 ? 0 : slot + 1;
        }
        PS_T(tq2_); PS_TACC(tr_stage, tq1_, tq2_);
        if (last_of_item) {
            item_epilogue(pf_res);
#ifdef TD_TRACE
            ++tr_items;
#endif
            PS_T(tq3_); PS_TACC(tr_epi, tq2_, tq3_);
            if (!stage) break;
            cn0 = sn0; cy0 = sy0; cx0 = sx0; cco0 = sco0;
            if (rn_sumsq) rb ^= 1;
        }
        PS_T(tq4_);
        c_seg = s_seg; c_chunk = s_chunk; c_nchunks = s_nchunks; c_taps = s_taps;
        PS_STAGE_ADVANCE();
        PS_T(tq5_); PS_TACC(tr_adv, tq4_, tq5_);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two over-fetched tail tiles must not land in a successor's LDS
#ifdef TD_TRACE
    {
        PS_T(tr_end);
        if (lane == 0) {
            unsigned long long* tb = (unsigned long long*)p.partial + ((size_t)blockIdx.x * (WAVES_M * WAVES_N) + wave) * 16;  // 16 u64 per wave
            tb[0] = (tr_pro - tr_start) + tr_adv; tb[1] = tr_group + tr_stage; tb[2] = tr_epi; tb[3] = 0; tb[4] = tr_stage; tb[5] = tr_start; tb[6] = tr_end;
            tb[8] = tr_rt0; tb[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | __builtin_amdgcn_s_getreg((3 << 11) | 20);
            tb[10] = __builtin_amdgcn_s_memrealtime(); tb[7] = tb[10] - tr_rt0; tb[11] = tr_items; tb[12] = tr_pro - tr_start; tb[13] = tr_adv;
        }
    }
#endif
#undef PS_TOFF
#undef PS_TAP1
#undef PS_TAPP
#undef PS_GROUP_ENTRY
#undef PS_FRAG_READ
#undef PS_FRAG_MFMA
#undef PS_STAGE_ADVANCE
#undef PS_LOAD_A
#undef PS_STORE_A
#undef PS_SEG_BEGIN
#undef PS_GLDS_B
}

// true when the persistent-stream flavour can run this conv (see the restrictions in the header comment)
static bool conv_ps_eligible(const ConvParams& p) {
    if (p.ksplit != 1 || p.out_f32 || (p.Cout & 7) || p.nseg < 1 || p.seg[0].taps != 9) return false;
    bool seen1 = false;
    int c1 = 0;
    for (int s = 0; s < p.nseg; ++s) {
        if (p.seg[s].taps == 9 && seen1) return false;
        if (p.seg[s].taps != 9) { seen1 = true; c1 += p.seg[s].C / 64; }
    }
    return c1 % 3 == 0;
}

template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
static hipError_t launch_ps_cfg(const ConvParams& p, int n_cus, hipStream_t st) {
    constexpr int NPATCH = NIMG * (TH + 2) * (TW == 8 ? 12 : TW + 2);
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    const size_t lds = (size_t)NPATCH * 144 + 3 * (size_t)(((BN * 128 + NTHR * 16 - 1) / (NTHR * 16)) * NTHR * 16) + 2 * NPATCH * 4;
    auto kern = conv_ps_kernel<T, TH, TW, NIMG, BN, WAVES_M, WAVES_N>;
    static bool attr_set[64] = {};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    // 4-wave tiles: two workgroups per CU (<= 80 KB of LDS each), 8-wave tiles: one; a multiple of 8 so that every XCD gets the same number;
    // never more than there is work
    const long items = (long)p.tiles_x * p.tiles_y * p.img_groups * p.n_ntiles;
    int grid = std::max(8, (n_cus * (WAVES_M * WAVES_N <= 4 ? 2 : 1) / 8) * 8);
    while (grid > 8 && (long)(grid - 8) >= items) grid -= 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_conv_ps_t(const ConvParams& p, bool narrow, int bn, int variant, int n_cus, hipStream_t st) {
    if (bn == 64) {
        if (narrow) return hipErrorInvalidValue;
        return variant == 1 ? launch_ps_cfg<T, 8, 16, 1, 64, 4, 1>(p, n_cus, st) : launch_ps_cfg<T, 16, 16, 1, 64, 8, 1>(p, n_cus, st);
    }
    if (variant == 1) {
        if (!narrow) return bn == 128 ? launch_ps_cfg<T, 8, 16, 1, 128, 2, 2>(p, n_cus, st) : launch_ps_cfg<T, 8, 16, 1, 96, 4, 1>(p, n_cus, st);
        return bn == 128 ? launch_ps_cfg<T, 8, 8, 2, 128, 2, 2>(p, n_cus, st) : launch_ps_cfg<T, 8, 8, 2, 96, 4, 1>(p, n_cus, st);
    }
    if (!narrow) return bn == 128 ? launch_ps_cfg<T, 16, 16, 1, 128, 4, 2>(p, n_cus, st) : launch_ps_cfg<T, 16, 16, 1, 96, 8, 1>(p, n_cus, st);
    return bn == 128 ? launch_ps_cfg<T, 8, 8, 4, 128, 4, 2>(p, n_cus, st) : launch_ps_cfg<T, 8, 8, 4, 96, 8, 1>(p, n_cus, st);
}

// dtype: 1 bf16, 2 fp16.  Same tile configurations as launch_conv_glds (variant 0 "big" = 8 waves, 1 "small" = 4 waves).
hipError_t launch_conv_ps(const ConvParams& p, int dtype, bool narrow, int bn, int variant, int n_cus, hipStream_t st) {
    if ((bn != 64 && bn != 96 && bn != 128) || !conv_ps_eligible(p)) return hipErrorInvalidValue;
    return dtype == 2 ? launch_conv_ps_t<_Float16>(p, narrow, bn, variant, n_cus, st) : launch_conv_ps_t<__bf16>(p, narrow, bn, variant, n_cus, st);
}

}  // namespace td
