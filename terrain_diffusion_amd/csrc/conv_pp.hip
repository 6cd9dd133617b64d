// Persistent ping-pong flavour of the implicit-GEMM 3x3 convolution (bf16, gfx950): the throughput kernel for layers whose grid fills
// the chip several times over.  Same maths, parameter block, LDS fragment layout and K order as conv_glds.hip (results are bit-identical:
// every output is accumulated by the same v_mfma_f32_32x32x16_bf16 sequence) -- what differs is how the time is organised:
//   * PERSISTENT: one 8-wave workgroup per CU walks over many (pixel tile, cout tile) work items.  The weight stream (LDS-DMA ring), the
//     halo patch of the next K-group (double-buffered in LDS) and the next tile's pixel-norm factors are fetched while the current tile's
//     taps run, so a tile boundary costs one epilogue instead of a workgroup launch gap + an exposed HBM round trip + a drain;
//     the residual operand of the epilogue is prefetched during the last taps.
//   * PING-PONG: the two waves that share a SIMD alternate roles every half tap.  One issues its 16 MFMAs of the tap back to back at raised
//     priority while its partner reads its next fragments from LDS, issues its share of the weight DMA and does the tile epilogue / patch
//     staging work; a workgroup barrier swaps the roles (both waves run the same code, one a phase behind the other).  The matrix pipe of each SIMD sees one uninterrupted MFMA stream (profiles/r02_tap_schedule_microbench.txt: 1028 cycles per 1024-cycle tap,
//     against 1107 for the interleaved schedule with all operands resident in LDS).
//   * XCD-aware work order: the cout-tile siblings of a pixel tile are taken in the same step by neighbouring workgroups of ONE XCD, so the
//     halo patch is read from HBM once and from that XCD's L2 by the siblings.
// Restrictions (the host falls back to conv_glds.hip otherwise): all K segments are 3x3, W >= 16, bf16 NHWC output with Cout % 8 == 0,
// no split-K.  DESIGN.md has the measurements.
#include "conv_common.h"

namespace td {

#ifdef TD_PP_TRACE  // in-kernel phase timing (tools/conv_bench.hip only): s_memtime at points where the LDS queue is empty anyway
#define PP_T(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_TACC(acc, a, b) acc += (b) - (a)
#else
#define PP_T(v)
#define PP_TACC(acc, a, b)
#endif
#ifdef TD_PP_ABL_FETCH
#define PP_ABL_FETCH(X)
#else
#define PP_ABL_FETCH(X) X
#endif
#ifdef TD_PP_ABL_STAGE
#define PP_ABL_STAGE(X)
#else
#define PP_ABL_STAGE(X) X
#endif
#ifdef TD_PP_ABL_EPI
#define PP_ABL_EPI(X) { float t_ = 0.f; _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) t_ += acc[i_][j_][r_]; if (t_ == 12345.678f) ((float*)p.out)[tid] = t_; PP_ZERO_ACC(); }
#else
#define PP_ABL_EPI(X) X
#endif

// DMAP = true: every K segment is staged untransformed (xform 0), so the halo patch goes HBM -> LDS by LDS-DMA, issued by four "patch waves"
// (waves 2,3,6,7) while the other four ("weight waves") carry the whole weight stream.  The point is the per-wave, in-order vmcnt queue: a
// wave that must see its weight pieces land every tap cannot leave patch loads in flight for more than two taps, and the patch comes from HBM
// (~2.5 us under load).  With the roles split, patch pieces have eight taps to land and nobody ever waits for HBM in the steady state.
// DMAP = false: register staging with the fused transform (pixel-norm / mp_silu), by all waves.
template <typename T, int BN, int WAVES_M, int WAVES_N, bool DMAP>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const ConvParams p) {
    typedef typename Half<T>::x8 hx8;
    typedef typename Half<T>::x4 hx4;
    constexpr int TH = 16, TW = 16, NTHR = 512;
    constexpr int TPIX = TH * TW, BM = TPIX;
    constexpr int PH = TH + 2, PW = TW + 2, NPATCH = PH * PW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int CHUNK = 64, PER16 = 8;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16);
    constexpr int B_BYTES = NBI * NTHR * 16, RING = 3;
    constexpr int PITCH = 144;
    constexpr int NPIECE = (NPATCH * 9 + 63) / 64;             // 1-KiB DMA pieces of a patch image (9 x 16 B per pixel, the 9th is row padding)
    constexpr int A_BASE = RING * B_BYTES, A_BYTES = NPIECE * 1024, RN_BASE = A_BASE + 2 * A_BYTES;
    constexpr int NPW = B_BYTES / 1024 / 4;                    // DMAP: weight pieces per weight wave per tile
    constexpr int NPP = (NPIECE + 3) / 4;                      // DMAP: patch pieces per patch wave per K-group
    constexpr int NWQ = DMAP ? NPW : NBI;                      // weight DMA instructions a (weight) wave issues per tap
    constexpr int NRES = MT * NT * 2;  // 16-byte residual pieces per lane
    static_assert(WAVES_M * WAVES_N == 8 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    static_assert(A_BYTES >= NPATCH * PITCH && NPATCH <= NTHR && NPP <= 12, "patch layout");
    static_assert((RING - 1) * B_BYTES + (NT - 1) * 4096 + 128 < 65536 && 2 * PW * PITCH + 2 * PITCH + 128 < 65536, "ds_read offset field");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);  // waves w and w+4 share a SIMD: group 0 = waves 0-3, group 1 = waves 4-7
    const bool wrole = !DMAP || (wave & 2) == 0;                 // DMAP: waves 0,1,4,5 stream weights, waves 2,3,6,7 stream patches
    const int rslot = (wave & 1) | ((wave >> 2) << 1);           // 0..3 among the waves of the same role

    // ---- work order.  Workgroup b sits on XCD b % 8 (observed dispatch rule; only speed depends on it).  XCD x owns the pixel tiles
    // m = 8*i + x; its workgroups take (m, cout tile) pairs round-robin with the cout tile fastest, so siblings run side by side.
    const int n_nt = p.n_ntiles;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups;
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int tiles_xy = p.tiles_x * p.tiles_y;
    auto decode = [&](int j, int& n0, int& y0, int& x0, int& nt) -> bool {
        const int q = j * per_x + wslot, m8 = q / n_nt;
        const int mt = m8 * 8 + xcd;
        nt = q - m8 * n_nt;
        const int ig = mt / tiles_xy, r = mt - ig * tiles_xy, tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
        n0 = ig; y0 = tyi * TH; x0 = txi * TW;
        return mt < mtiles;
    };
    int ksteps_tile = 0;
    for (int s = 0; s < p.nseg; ++s) ksteps_tile += (p.seg[s].C / CHUNK) * 9;

    // compute cursor: tile (cn0, cy0, cx0, cnt) ; staging cursor (one K-group ahead): tile (sn0, ...), segment, chunk
    int cj = 0, cn0, cy0, cx0, cnt;
    if (!decode(0, cn0, cy0, cx0, cnt)) return;  // uniform over the workgroup

    // ---- weight stream: every wave copies its own 1-KiB pieces of each [BN x 128 B] tile, two tiles ahead of the tap it is about to
    // compute (see the phase table below).  The stream continues seamlessly into the next work item's slab.
    const size_t wstep = (size_t)p.CoutPad * 128;
    int wj = 0, wleft = ksteps_tile;
    const unsigned char* wnext = (const unsigned char*)p.wpack + (size_t)(cnt * BN) * 128;
    // piece i of this wave: DMAP: pieces rslot*NPW + i (consecutive KiBs); else: pieces wave + 8*i (one per 8-KiB round)
    unsigned wvoff[NWQ];
#pragma unroll
    for (int i = 0; i < NWQ; ++i) wvoff[i] = DMAP ? (unsigned)((rslot * NPW + i) * 1024 + lane * 16) : (unsigned)tid * 16u + (unsigned)i * NTHR * 16u;
    const unsigned ldsw = DMAP ? (unsigned)(rslot * NPW) * 1024u : (unsigned)wave * 1024u;
    constexpr int WPSTEP = DMAP ? 1024 : NTHR * 16;
#define PP_FETCH(SLOT)                                                                                       \
    {                                                                                                        \
        if (wrole) { _Pragma("unroll") for (int i_ = 0; i_ < NWQ; ++i_) TD_GLDS16(wvoff[i_], wnext, ldsw, (SLOT) * B_BYTES + i_ * WPSTEP); } \
        wnext += wstep;                                                                                      \
        if (--wleft == 0) {                                                                                  \
            int a_, b_, c_, nt_;                                                                             \
            ++wj;                                                                                            \
            if (decode(wj, a_, b_, c_, nt_)) { wnext = (const unsigned char*)p.wpack + (size_t)(nt_ * BN) * 128; wleft = ksteps_tile; } \
            else { wnext = (const unsigned char*)p.wpack; wleft = 1 << 30; } /* past the end: harmless refetch of tile 0, never read */ \
        }                                                                                                    \
    }

    // ---- patch staging of the K-group under the staging cursor (tile origin sn0/sy0/sx0, segment, chunk).
    // Register path: per thread A_ITERS 16-byte pieces (patch pixel e>>3, slot e&7), fetched to registers, transformed, written to LDS.
    // DMA path: per patch wave NPP 1-KiB pieces; lane l of piece q carries 16-byte unit 64*q + l of the padded image (unit = pixel*9 + slot,
    // slot 8 = padding), out-of-image pixels read a zero page.
    int sj = 0, sn0 = cn0, sy0 = cy0, sx0 = cx0, snt = cnt, sseg = 0, schunk = 0;
    constexpr int NST = DMAP ? NPP : A_ITERS;
    int a_tc[NST], aoff[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        if constexpr (DMAP) {
            const int q = min(rslot + 4 * it, NPIECE - 1), u = 64 * q + lane, pp = u / 9, sl = min(u % 9, 7);
            a_tc[it] = pp < NPATCH ? (((pp / PW) << 12) | ((pp % PW) << 4) | sl) : -1;
        } else {
            const int e = tid + it * NTHR, pp = e >> 3;
            a_tc[it] = pp < NPATCH ? (((pp / PW) << 12) | ((pp % PW) << 4) | (e & 7)) : -1;
        }
    }
    u32x4 av[DMAP ? 1 : A_ITERS];
    const T* seg_src = nullptr;
    int seg_xform = 0, seg_nchunks = 0;
    float seg_scale = 1.f;
#define PP_SEG_BEGIN(SEG)                                                                                    \
    {                                                                                                        \
        const ConvSeg& sg_ = p.seg[SEG];                                                                     \
        seg_src = (const T*)sg_.src; seg_xform = sg_.xform; seg_scale = sg_.scale; seg_nchunks = sg_.C / CHUNK; \
        const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                         \
        _Pragma("unroll") for (int it_ = 0; it_ < NST; ++it_) {                                              \
            const int y_ = sy0 + (a_tc[it_] >> 12) - 1, x_ = sx0 + ((a_tc[it_] >> 4) & 255) - 1;             \
            const bool ok_ = a_tc[it_] >= 0 && y_ >= 0 && y_ < p.H && x_ >= 0 && x_ < p.W;                   \
            aoff[it_] = ok_ ? src_pixel(sn0, y_, x_, Hs_, Ws_, rs_) * cs_ + (a_tc[it_] & 15) * PER16 : -1;   \
        }                                                                                                    \
    }
#define PP_LOAD_A(CH)                                                                                        \
    {                                                                                                        \
        const T* src_ = seg_src + (CH) * CHUNK;                                                              \
        _Pragma("unroll") for (int it_ = 0; it_ < NST; ++it_) av[DMAP ? 0 : it_] = *(const u32x4*)(src_ + (aoff[it_] >= 0 ? aoff[it_] : 0)); \
    }
#define PP_STORE_A(PBUF, RNBUF)                                                                              \
    {                                                                                                        \
        unsigned char* s_a_ = smem + A_BASE + (PBUF) * A_BYTES;                                              \
        const float* s_rn_ = (const float*)(smem + RN_BASE + (RNBUF) * (NPATCH * 4));                        \
        _Pragma("unroll") for (int it_ = 0; it_ < NST; ++it_) {                                              \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                                  \
            if (pp_ < NPATCH) {                                                                              \
                u32x4 v_ = aoff[it_] >= 0 ? av[DMAP ? 0 : it_] : u32x4{0u, 0u, 0u, 0u};                      \
                if (seg_xform != 0 && aoff[it_] >= 0) {                                                      \
                    float s_ = seg_scale;                                                                    \
                    if (seg_xform == 2) s_ *= s_rn_[pp_];                                                    \
                    v_ = xform_piece<T>(v_, s_);                                                             \
                }                                                                                            \
                *(u32x4*)(s_a_ + pp_ * PITCH + (slot_ << 4)) = v_;                                           \
            }                                                                                                \
        }                                                                                                    \
    }
    // DMA path: pieces [I0, I1) of this patch wave for chunk CH of the staging segment into patch buffer PBUF
#define PP_DMA_PATCH(I0, I1, CH, PBUF)                                                                       \
    {                                                                                                        \
        const T* src_ = seg_src + (CH) * CHUNK;                                                              \
        _Pragma("unroll") for (int it_ = (I0); it_ < (I1) && it_ < NST; ++it_) {                             \
            const void* g_ = aoff[it_] >= 0 ? (const void*)(src_ + aoff[it_]) : p.zeros;                     \
            const unsigned m0_ = (unsigned)(A_BASE + min(rslot + 4 * it_, NPIECE - 1) * 1024) + (unsigned)(PBUF) * A_BYTES; \
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(m0_) : "memory"); \
        }                                                                                                    \
    }
    // per-pixel 1/(eps + rms) of the pixel-normed source (first segment with xform 2, else the normed residual), for a tile's patch pixels
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (p.seg[0].xform == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = p.seg[0].Hs; rn_Ws = p.seg[0].Ws; rn_res = p.seg[0].resample; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    const size_t rn_npix = (size_t)p.N * rn_Hs * rn_Ws;
    auto tile_rn = [&](int n0, int y0, int x0) -> float {  // value for patch pixel tid (< NPATCH)
        float rn = 0.f;
        if (tid < NPATCH) {
            const int py = tid / PW, px = tid % PW, y = y0 + py - 1, x = x0 + px - 1;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) rn = pixel_rn(rn_sumsq, rn_parts, rn_npix, src_pixel(n0, y, x, rn_Hs, rn_Ws, rn_res), rn_invc);
        }
        return rn;
    };

    // ---- MFMA operand addressing (identical to conv_glds.hip): xoff = byte offset of the TOP-LEFT tap of this lane's pixel inside a patch
    // buffer (+ its k-half); wbase[ks] = this lane's cout row inside a ring slot with the slab's XOR swizzle applied
    int base_pp[MT];
    unsigned xoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
        base_pp[i] = (ty + 1) * PW + (tx + 1);
        xoff[i] = (unsigned)(base_pp[i] - PW - 1) * PITCH + (unsigned)lh * 16u;
    }
    unsigned wbase[4];
    {
        const int nl = wn * WN + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wbase[ks] = (unsigned)(nl * 128 + (((ks * 2 + lh) ^ TD_SWZ(nl)) << 4));
    }
    f32x16 acc[MT][NT];
#define PP_ZERO_ACC()                                                                                        \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.f;                     \
    }
    PP_ZERO_ACC();

    // ---- prologue of the first tile (the only exposed one)
    PP_FETCH(0);
    PP_FETCH(1);
    PP_SEG_BEGIN(0);
    if constexpr (DMAP) { if (!wrole) PP_DMA_PATCH(0, NST, 0, 0); } else { PP_LOAD_A(0); }
    int rb = 0, pb = 0;  // pixel-norm buffer of the compute tile, patch buffer of the compute group
    if (rn_sumsq) {
        const float rn0 = tile_rn(cn0, cy0, cx0);
        if (tid < NPATCH) ((float*)(smem + RN_BASE))[tid] = rn0;
    }
    __syncthreads();  // s_rn visible; this one may drain the first weight tiles, they are needed next anyway
    if constexpr (!DMAP) PP_STORE_A(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // advance the staging cursor past group (tile 0, seg 0, chunk 0)
    bool s_valid = true;    // staging cursor points at a real group
#define PP_STAGE_ADVANCE()                                                                                   \
    {                                                                                                        \
        ++schunk;                                                                                            \
        if (schunk >= seg_nchunks) {                                                                         \
            schunk = 0; ++sseg;                                                                              \
            if (sseg >= p.nseg) {                                                                            \
                sseg = 0; ++sj;                                                                              \
                s_valid = decode(sj, sn0, sy0, sx0, snt);                                                    \
            }                                                                                                \
            if (s_valid) { PP_SEG_BEGIN(sseg); }                                                             \
        }                                                                                                    \
    }
    PP_STAGE_ADVANCE();

    u32x4 wf[4][NT], xf[4][MT];
#define PP_TOFF(T) ((((T) / 3) * PW + ((T) % 3)) * PITCH)
#define PP_FRAG_READ(KS, SLOT, XB, TOFF)                                                                     \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) wf[KS][j_] = *(const u32x4*)(smem + wbase[KS] + ((SLOT) * B_BYTES + j_ * 4096)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) xf[KS][i_] = *(const u32x4*)(smem + XB[i_] + ((TOFF) + (KS) * 32)); \
    }
#define PP_LOAD_TAP(SLOT, XB, TOFF) { PP_FRAG_READ(0, SLOT, XB, TOFF); PP_FRAG_READ(1, SLOT, XB, TOFF); PP_FRAG_READ(2, SLOT, XB, TOFF); PP_FRAG_READ(3, SLOT, XB, TOFF); }
#define PP_MFMA_TAP()                                                                                        \
    {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                \
                _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                            \
                    acc[i_][j_] = Half<T>::mfma32(__builtin_bit_cast(hx8, wf[ks_][j_]), __builtin_bit_cast(hx8, xf[ks_][i_]), acc[i_][j_]); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    }
#define PP_BARRIER()                                                                                         \
    {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

    // ---- epilogue of a finished tile (wide bf16 path of conv_glds.hip): 2*MT*NT units of 8 couts x 1 pixel per lane.  It runs at the
    // START of the wave's next fragment-load phase, i.e. while the partner wave of the SIMD issues MFMAs; the per-unit operand (modulation
    // vector or residual) is fetched one unit ahead and the units are fenced from each other so that the compiler cannot hoist all the
    // loads to the top (which costs more registers than the loop around it can spare).
    const size_t M = (size_t)p.N * p.H * p.W;
    int en0 = 0, ey0 = 0, ex0 = 0, ent = 0, erb = 0;  // the tile whose accumulators are waiting for their epilogue
    bool epi_pending = false;
    auto epilogue = [&]() {
        constexpr int NU = MT * NT * 2;
        const int co0 = ent * BN;
        f32x4 ca[2] = {}, cb[2] = {};
        u32x4 rw[2] = {};
        int pixv[MT]; bool okv[MT]; float rnv[MT], ssv[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int img, ty, tx;
            frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
            const int y = ey0 + ty, x = ex0 + tx;
            okv[i] = y < p.H && x < p.W;
            pixv[i] = okv[i] ? (en0 * p.H + y) * p.W + x : 0;
            rnv[i] = (p.res_sumsq != nullptr) ? ((const float*)(smem + RN_BASE + erb * (NPATCH * 4)))[base_pp[i]] : 1.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) ssv[i][j] = 0.f;
        }
        const int cobase = co0 + wn * WN + 4 * lh;
        const bool has_res = p.epi == EPI_RESIDUAL && p.res != nullptr, want_ss = p.out_sumsq != nullptr, want_o2 = p.out2 != nullptr;
        const SiluK k_o2 = silu_k(p.out2_scale);
        auto fetch = [&](int u, int s) {
            const int i = u / (NT * 2), j = (u / 2) % NT, m = u % 2;
            const bool in = co0 + wn * WN + j * 32 < p.Cout;
            if (p.epi == EPI_EMB_SILU) {
                const float* crow = p.cvec + (size_t)en0 * p.cvec_stride + (in ? cobase + j * 32 + m * 16 : 0);
                ca[s] = *(const f32x4*)crow; cb[s] = *(const f32x4*)(crow + 8);
            } else if (has_res) {
                int img, ty, tx;
                frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
                const int sp = okv[i] ? src_pixel(en0, ey0 + ty, ex0 + tx, p.res_Hs, p.res_Ws, p.res_resample) : 0;
                rw[s] = *(const u32x4*)((const T*)p.res + (size_t)sp * p.res_cstride + (in ? co0 + wn * WN + 8 * lh + j * 32 + m * 16 : 0));
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = u / (NT * 2), j = (u / 2) % NT, m = u % 2, s = u & 1;
            if (u + 1 < NU) fetch(u + 1, s ^ 1);
            if (co0 + wn * WN + j * 32 < p.Cout) {
                const f32x4 va = {acc[i][j][8 * m + 0], acc[i][j][8 * m + 1], acc[i][j][8 * m + 2], acc[i][j][8 * m + 3]};
                const f32x4 vb = {acc[i][j][8 * m + 4], acc[i][j][8 * m + 5], acc[i][j][8 * m + 6], acc[i][j][8 * m + 7]};
                u32x4 o, o2;
                epi_unit8<T>(p.epi, has_res, p.clip, want_ss, want_o2, va, vb, ca[s], cb[s], rw[s], p.res_scale * rnv[i], k_o2, o, o2, ssv[i][j]);
                const size_t oo = (size_t)pixv[i] * p.out_cstride + co0 + wn * WN + 8 * lh + j * 32 + m * 16;
                if (okv[i]) *(u32x4*)((T*)p.out + oo) = o;
                if (want_o2 && okv[i]) *(u32x4*)((T*)p.out2 + oo) = o2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.out_sumsq) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {  // one partial per 32-cout MFMA block, as in conv_glds.hip (tile-shape independent order)
                    const float ss = ssv[i][j] + __shfl_xor(ssv[i][j], 32);
                    if (okv[i] && lh == 0 && co0 + wn * WN + j * 32 < p.CoutPad) p.out_sumsq[(size_t)((co0 + wn * WN) / 32 + j) * M + pixv[i]] = ss;
                }
        }
        PP_ZERO_ACC();
    };

    // ---- main loop.  Both groups run the SAME code, group 0 one phase behind group 1 (it passes one extra barrier first), so whenever
    // one wave of a SIMD is in its MFMA phase its partner is in its fragment-load phase.  One tap k (ring slot k % 3) of a wave:
    //   load phase : [epilogue of the previous tile, if one is pending] DMA tile k+2 -> slot (k+2)%3, LOAD fragments(k), counted wait | barrier
    //   MFMA phase : 16 MFMAs at raised priority                                                                                   | barrier
    // Slot (k+2)%3 held tile k-1, last read in the load phase of tap k-1 of the lagging group, one barrier before the leading group's
    // load phase of tap k.  A DMA piece issued in a load phase is waited for at the end of the same wave's NEXT load phase (one full tap
    // later) and is first read two barriers after that.  The patch of the next K-group is loaded to registers in the load phase of tap 0,
    // written to the OTHER patch buffer at the START of the load phase of tap 6 (six taps of HBM latency cover; the compiler's own wait for
    // those registers then drains nothing younger than the previous tap's DMA), and first read in the next group's tap 0.
    unsigned xcur[MT], xnxt[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { xcur[i] = (unsigned)A_BASE + xoff[i]; xnxt[i] = (unsigned)(A_BASE + A_BYTES) + xoff[i]; }
    int cg = 0;                      // K-group index inside the compute tile
    const int groups_tile = p.kgroups;

#ifdef TD_PP_TRACE
    unsigned long long tr_epi = 0, tr_load = 0, tr_bar1 = 0, tr_mfma = 0, tr_bar2 = 0;
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    PP_T(tr_start);
    if (grp == 0) PP_BARRIER();
    for (;;) {
        const bool last_group = cg + 1 == groups_tile;
        const bool stage = s_valid;                       // a next K-group exists: fetch its patch during this group
        const bool stage_rn = stage && rn_sumsq && sj != cj && sseg == 0 && schunk == 0;  // ... and it opens a new tile: its pixel-norm factors too
        float rn_next = 0.f;
        const bool after_epi = epi_pending;               // this group's first load phase starts with the previous tile's epilogue
#define PP_ITER(K, SLOT, TOFF_CUR)                                                                           \
    {                                                                                                        \
        PP_T(t0_);                                                                                           \
        if ((K) == 0 && epi_pending) {                                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* DMA queue empty: the counted waits below need not look past the stores */ \
            PP_ABL_EPI(epilogue());                                                                          \
            epi_pending = false;                                                                             \
        }                                                                                                    \
        PP_T(t1_); PP_TACC(tr_epi, t0_, t1_);                                                                \
        if ((K) == 1 && stage_rn) { if (tid < NPATCH) ((float*)(smem + RN_BASE + (rb ^ 1) * (NPATCH * 4)))[tid] = rn_next; } \
        if (!DMAP && (K) == 2 && stage) { PP_ABL_STAGE(PP_STORE_A(pb ^ 1, (sj != cj) ? (rb ^ 1) : rb)); }    \
        PP_ABL_FETCH(PP_FETCH(((SLOT) + 2) % RING));                                                         \
        PP_LOAD_TAP(SLOT, xcur, TOFF_CUR);                                                                   \
        if (!DMAP && (K) == 0 && stage) { PP_ABL_STAGE(PP_LOAD_A(schunk)); }                                 \
        if (DMAP && (K) < 4 && stage && !wrole) { PP_ABL_STAGE(PP_DMA_PATCH(3 * (K), 3 * (K) + 3, schunk, pb ^ 1)); } \
        if ((K) == 0 && after_epi) {}  /* everything older than this phase was drained before the epilogue; its stores are not waited for */ \
        else if (DMAP) {                                                                                     \
            if (wrole) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWQ) : "memory"); }                        \
            else if ((K) == 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } /* the next group's patch pieces: eight taps to land */ \
        }                                                                                                    \
        else if (((K) == 0 || (K) == 1) && stage) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI + A_ITERS) : "memory"); } \
        else { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory"); }                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        if ((K) == 0 && stage_rn) rn_next = tile_rn(sn0, sy0, sx0);  /* loads issued AFTER the counted wait of this phase */ \
        PP_T(t2_); PP_TACC(tr_load, t1_, t2_);                                                               \
        PP_BARRIER();                                                                                        \
        PP_T(t3_); PP_TACC(tr_bar1, t2_, t3_);                                                               \
        PP_MFMA_TAP();                                                                                       \
        PP_T(t4_); PP_TACC(tr_mfma, t3_, t4_);                                                               \
        PP_BARRIER();                                                                                        \
        PP_T(t5_); PP_TACC(tr_bar2, t4_, t5_);                                                               \
    }
        PP_ITER(0, 0, PP_TOFF(0)); PP_ITER(1, 1, PP_TOFF(1)); PP_ITER(2, 2, PP_TOFF(2));
        PP_ITER(3, 0, PP_TOFF(3)); PP_ITER(4, 1, PP_TOFF(4)); PP_ITER(5, 2, PP_TOFF(5));
        PP_ITER(6, 0, PP_TOFF(6)); PP_ITER(7, 1, PP_TOFF(7)); PP_ITER(8, 2, PP_TOFF(8));
        // ---- group done: swap patch buffers, move both cursors
        if (stage) { PP_STAGE_ADVANCE(); }
        pb ^= 1;
#pragma unroll
        for (int i = 0; i < MT; ++i) { const unsigned t_ = xcur[i]; xcur[i] = xnxt[i]; xnxt[i] = t_; }
        if (last_group) {
            en0 = cn0; ey0 = cy0; ex0 = cx0; ent = cnt; erb = rb; epi_pending = true;
            ++cj;
            if (!decode(cj, cn0, cy0, cx0, cnt)) break;
            cg = 0;
            if (rn_sumsq) rb ^= 1;
        } else {
            ++cg;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead DMA pieces must not land in a successor workgroup's LDS
    PP_ABL_EPI(epilogue());                           // the last tile's
    if (grp == 1) PP_BARRIER();
#ifdef TD_PP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_T(tr_end);
    if (lane == 0) {
        unsigned long long* tb = (unsigned long long*)p.partial + ((size_t)blockIdx.x * 8 + wave) * 16;
        tb[0] = tr_epi; tb[1] = tr_load; tb[2] = tr_bar1; tb[3] = tr_mfma; tb[4] = tr_bar2; tb[5] = tr_end - tr_start;
        tb[6] = __builtin_amdgcn_s_memrealtime() - tr_rt0;  // 100 MHz ticks
    }
#endif
#undef PP_ITER
#undef PP_BARRIER
#undef PP_MFMA_TAP
#undef PP_LOAD_TAP
#undef PP_FRAG_READ
#undef PP_TOFF
#undef PP_STAGE_ADVANCE
#undef PP_ZERO_ACC
#undef PP_STORE_A
#undef PP_LOAD_A
#undef PP_SEG_BEGIN
#undef PP_DMA_PATCH
#undef PP_FETCH
}

template <typename T, int BN, int WAVES_M, int WAVES_N, bool DMAP>
static hipError_t launch_pp_cfg(const ConvParams& p, int n_cus, hipStream_t st) {
    constexpr int NPATCH = 18 * 18, NTHR = 512;
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16);
    const size_t lds = 3 * (size_t)NBI * NTHR * 16 + 2 * (size_t)((NPATCH * 9 + 63) / 64) * 1024 + 2 * NPATCH * 4;
    auto kern = conv_pp_kernel<T, BN, WAVES_M, WAVES_N, DMAP>;
    static bool attr_set[64] = {};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups;
    // one workgroup per CU (145 KB of LDS each), a multiple of 8 so that every XCD gets the same number; never more than there is work
    int grid = std::max(8, (n_cus / 8) * 8);
    const long items = (long)mtiles * p.n_ntiles;
    while (grid > 8 && (long)(grid - 8) >= items) grid -= 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, st, p);
    return hipGetLastError();
}

// bn 128 -> waves 4x2 (64 px x 64 co per wave), bn 96 -> 8x1 (32 px x 96 co).  Preconditions are the caller's (see conv_pp_eligible).
template <typename T>
static hipError_t launch_conv_pp_t(const ConvParams& p, int bn, int n_cus, bool dmap, hipStream_t st) {
    if (bn == 64) return dmap ? launch_pp_cfg<T, 64, 8, 1, true>(p, n_cus, st) : launch_pp_cfg<T, 64, 8, 1, false>(p, n_cus, st);
    if (dmap) return bn == 128 ? launch_pp_cfg<T, 128, 4, 2, true>(p, n_cus, st) : launch_pp_cfg<T, 96, 8, 1, true>(p, n_cus, st);
    return bn == 128 ? launch_pp_cfg<T, 128, 4, 2, false>(p, n_cus, st) : launch_pp_cfg<T, 96, 8, 1, false>(p, n_cus, st);
}

hipError_t launch_conv_pp(const ConvParams& p, int dtype, int bn, int n_cus, hipStream_t st) {
    for (int s = 0; s < p.nseg; ++s) if (p.seg[s].taps != 9) return hipErrorInvalidValue;
    if (p.ksplit != 1 || p.out_f32 || (p.Cout & 7) || p.W < 16 || p.img_groups != p.N) return hipErrorInvalidValue;
    bool dmap = p.zeros != nullptr;  // LDS-DMA patch staging needs untransformed sources (and the zero page for the halo outside the image)
    for (int s = 0; s < p.nseg; ++s) dmap = dmap && p.seg[s].xform == 0;
#ifdef TD_PP_NO_DMAP
    dmap = false;
#endif
    return dtype == 2 ? launch_conv_pp_t<_Float16>(p, bn, n_cus, dmap, st) : launch_conv_pp_t<__bf16>(p, bn, n_cus, dmap, st);
}

}  // namespace td
