// LDS-DMA flavour of the implicit-GEMM convolution (bf16, gfx950): the kernel behind almost every conv of a bf16 forward.
// Same maths / parameter block / fused prologues and epilogues as conv_igemm.hip; what differs is the pipeline:
//   * weights stream HBM/L2 -> LDS with `global_load_lds_dwordx4` (LDS-DMA: no VGPR staging, no ds_write) into a 3-slot ring, one
//     tile ahead of the tap that uses it, behind COUNTED `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` and raw `s_barrier`s; the slab is
//     stored pre-swizzled in HBM and copied linearly (the LDS destination of an LDS-DMA is wave-uniform base + lane*16);
//   * 8 waves x 256 pixels or 4 waves x 128 pixels per workgroup, BN = 96/128 couts; each wave owns a 64x64 (or 32x96) block of
//     v_mfma_f32_32x32x16_bf16 tiles, so one weight tile is shared by all pixels and one activation patch by all couts;
//   * the activation halo patch ((TH+2)x(TW+2) pixels x 64 channels) is fetched one K-group ahead into registers, transformed
//     where the producer could not (pixel-norm + mp_silu) and written to LDS once per 9 taps, rows PADDED to 144 bytes;
//   * the tap loop has no VALU: every fragment address is "lane base + compile-time (slot, tap, k-step) offset" (ds_read offset
//     field), the lane -> pixel map makes every ds_read_b128 lane group conflict-free, and the loop is software-pipelined ACROSS
//     taps (mid-tap barrier, next tap's first fragments requested behind the current MFMAs);
//   * bf16 results leave as dwordx4 stores after v_permlane32_swap pairs; optional pre-activated second output; split-K.
// DESIGN.md §4 has the measurements behind each of these choices and the list of variants that were tried and dropped.
#include "conv_common.h"

namespace td {

#ifdef TD_TRACE  // in-kernel phase timing with s_memtime (tools/conv_bench.hip only): per wave, cycles spent per phase
#define TD_T(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define TD_TACC(acc, a, b) acc += (b) - (a)
#else
#define TD_T(v)
#define TD_TACC(acc, a, b)
#endif

// DMA1: the instantiation that streams 1x1 segments by LDS-DMA (launch_glds_cfg picks it per launch; the plain instantiation is untouched by it)
// (two workgroups of a 4-wave tile per CU.  Three of the 64-cout tile fit the LDS and, at 168 VGPRs, cost 40 bytes of scratch per lane: measured 17 % SLOWER on the
// decoder's 64 -> 64 layers, level on 128 / 192 -> 64 -- profiles/r05_decoder_512_level_experiments.txt)
#define TD_GLDS_MIN_WAVES(BN, W, DMA) (((W) + 3) / 4 < 2 ? 2 : ((W) + 3) / 4)
template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N, bool DMA1 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, TD_GLDS_MIN_WAVES(BN, WAVES_M * WAVES_N, DMA1)) void conv_glds_kernel(const ConvParams p) {
    typedef typename Half<T>::x8 hx8;
    typedef typename Half<T>::x4 hx4;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int TPIX = TH * TW, BM = NIMG * TPIX;
    constexpr int PH = TH + 2, PW = TW == 8 ? 12 : TW + 2, PPI = PH * PW, NPATCH = NIMG * PPI;  // 8-wide: 2 pad columns (see frag_pixel)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int CHUNK = 64, PER16 = 8;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    // a ring slot holds a whole number of LDS-DMA rounds (NTHR x 16 B); for BN = 96 that is 128 rows: the 32 extra rows belong to
    // the next cout tile (or the slab's tail padding) and are never read
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16);  // LDS-DMA instructions per thread per weight tile
    constexpr int B_BYTES = NBI * NTHR * 16, RING = 3;
    // 1x1 segments by LDS-DMA (p.dma1x1): a K-group's 64 channels of the tile's BM pixels, 128-byte rows in MFMA column order, 16-byte slots
    // XOR-swizzled like the weight rows; NST buffers laid over the (then dead) halo patch, DMA_N pieces per thread and buffer
    constexpr int STAGE_BYTES = BM * 128, NST = WAVES_M * WAVES_N >= 8 ? 3 : 2, DMA_N = BM * 8 / NTHR;
    static_assert(BM * 8 % NTHR == 0, "1x1 stage: whole DMA rounds");
    // LDS map: [0, RING*B_BYTES) weight ring | activation patch, PITCH bytes per pixel | s_rn.  The ring comes first so that
    // "slot*B_BYTES + 32-row step" fits the 16-bit ds_read offset field; the patch rows are PADDED to 144 B instead of swizzled:
    // 16 consecutive rows then start at 16 different 16-byte positions of the 256-byte bank window (9 is odd), and a fragment
    // address is "lane base + compile-time (tap, k-step) offset" -- no VALU in the tap loop (the matrix pipe hides only ~5 issue
    // slots per MFMA per SIMD, MI355X guide: every address instruction in the loop is paid in full).
    constexpr int PITCH = 144;
    constexpr int A_BASE = RING * B_BYTES, A_BYTES = NPATCH * PITCH;
    // round 5: the modulation rows c[n][co0 .. co0 + BN) of the tile's images (EPI_EMB_SILU) are staged in LDS by the prologue -- the epilogue then
    // reads them with ds_read_b128 instead of 2 x NT x 2 global loads per pixel row group between the last MFMA and the first store.  Behind
    // the patch and the 1/rms table.
    constexpr int CV_BASE = A_BASE + (A_BYTES + NPATCH * 4 + 15) / 16 * 16;   // (plain instantiation only)
    constexpr int NU = NT * 2;   // epilogue units (8 couts of one pixel) per 32-pixel row group
    static_assert(NIMG * BN <= NTHR, "modulation rows: one element per thread");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "tile shape");
    static_assert((RING - 1) * B_BYTES + (NT - 1) * 4096 + 128 < 65536 && 2 * PW * PITCH + 2 * PITCH + 128 < 65536, "ds_read offset field");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0
    unsigned char* s_a = smem + A_BASE;
    float* s_rn = (float*)(smem + A_BASE + A_BYTES);
    float* s_cv = (float*)(smem + CV_BASE);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef TD_TRACE
    unsigned long long tr_wait = 0, tr_stage = 0;
#endif
    TD_T(tr_start);
#ifdef TD_TRACE
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // XCD-aware order (workgroup b runs on XCD b % 8; only speed depends on it): XCD x takes the contiguous range [x*G/8, (x+1)*G/8) of
    // logical ids, so the n_ntiles cout-tile siblings of a pixel tile (consecutive logical ids) run on ONE XCD at about the same time and
    // share its L2 copy of the halo patch instead of fetching it from HBM once per sibling.
    // Workgroup-id decomposition.  Round 4: the decode constants come in ONE burst of kernel-argument loads pinned by an empty asm statement, the
    // divisions are multiplications by launcher-made magic numbers (td_udiv), and the grid size is a kernel argument instead of a read of the
    // dispatch packet: hipcc used to emit a dozen serialised scalar loads (each one a scalar-cache round trip, the first a ~1 us miss) plus six
    // ~40-instruction integer divisions here, i.e. several thousand cycles before the first weight request of every workgroup.
    unsigned k_d0 = p.sb_d0, k_m0 = p.sb_m0, k_d1 = p.sb_d1, k_m1 = p.sb_m1, k_m2 = p.sb_m2, k_m3 = p.sb_m3, k_g8 = p.sb_grid8, k_grid = p.sb_grid;
    int k_tx = p.tiles_x, k_ty = p.tiles_y, k_rev = p.reverse, k_ks = p.ksplit, k_kg = p.kgroups, k_cpad = p.CoutPad, k_N = p.N, k_H = p.H, k_W = p.W;
    const unsigned char* k_wpack = (const unsigned char*)p.wpack;   // only ever the scalar base of an LDS-DMA statement: no address space to lose
    int k_epi = p.epi, k_Cout = p.Cout, k_of32 = p.out_f32, k_cvs = p.cvec_stride;   // round 5: the epilogue kind is known before the K loop
    const bool k_hres = p.res != nullptr;
    asm volatile("" : "+s"(k_cpad), "+s"(k_N), "+s"(k_H), "+s"(k_W), "+s"(k_wpack), "+s"(k_epi), "+s"(k_Cout), "+s"(k_of32), "+s"(k_cvs));
    asm volatile("" : "+s"(k_d0), "+s"(k_m0), "+s"(k_d1), "+s"(k_m1), "+s"(k_m2), "+s"(k_m3), "+s"(k_g8), "+s"(k_grid), "+s"(k_tx), "+s"(k_ty), "+s"(k_rev), "+s"(k_ks), "+s"(k_kg));
    unsigned ubid = blockIdx.x;
#ifndef TD_NO_XCD_REMAP
    if (k_g8) ubid = (ubid & 7) * k_g8 + (ubid >> 3);
#endif
    // the host alternates `reverse` from layer to layer: a layer then reads first what its producer wrote last, which is still in the 256 MB
    // Infinity Cache (chained A/B on the 192-channel 64x64 layers: -7...-11 % time; profiles/r03_conv_walk_order_and_stagger.txt)
    if (k_rev) ubid = k_grid - 1 - ubid;
    const unsigned q1 = td_udiv(ubid, k_d1, k_m1);
    const int ntile = (int)(ubid - q1 * k_d1);
    const int ksp = (int)td_udiv(q1, k_d0, k_m0);  // split-K index: this workgroup reduces K-groups [g0, g1) and leaves fp32 partials to the reduce kernel
    const unsigned mtile = q1 - (unsigned)ksp * k_d0;
    const unsigned q2 = td_udiv(mtile, (unsigned)k_tx, k_m2);
    const int txi = (int)(mtile - q2 * (unsigned)k_tx), ig = (int)td_udiv(q2, (unsigned)k_ty, k_m3), tyi = (int)(q2 - (unsigned)ig * (unsigned)k_ty);
    const int n0 = ig * NIMG, y0 = tyi * TH, x0 = txi * TW, co0 = ntile * BN;
    int g0 = 0, g1 = k_kg;   // conv_set_kbounds (host); the byte table is only read when K is split
    if (k_ks > 1) { g0 = p.kb[ksp]; g1 = p.kb[ksp + 1]; }
    int seg_first = 0, chunk_first = 0, kstep_first = 0;  // (segment, chunk) of K-group g0 and the K-step it starts at
    if (g0 > 0) {   // (without split-K there is nothing to look for: the walk costs two or three serialised kernel-argument loads per segment)
        int g = 0;
        while (seg_first < p.nseg) {
            const int nch = p.seg[seg_first].C / CHUNK;
            if (g0 < g + nch) { chunk_first = g0 - g; kstep_first += chunk_first * p.seg[seg_first].taps; break; }
            g += nch; kstep_first += nch * p.seg[seg_first].taps; ++seg_first;
        }
    }

    // ---- weight ring.  wnext is the (wave-uniform, SGPR) address of the next tile to fetch; every tap fetches the tile two K-steps
    // ahead UNCONDITIONALLY (the packed slab carries two K-steps of tail padding), so the loop has no tail tests and a fixed vmcnt.
    const size_t wstep = (size_t)k_cpad * 128;
    const unsigned char* wnext = k_wpack + (size_t)co0 * 128 + (size_t)kstep_first * wstep;
    unsigned wvoff[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) wvoff[i] = (unsigned)tid * 16u + (unsigned)i * NTHR * 16u;
    const unsigned ldsw = (unsigned)wave * 1024u;
#define TD_GLDS_B(SLOT)                                                                                      \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) TD_GLDS16(wvoff[i_], wnext, ldsw, (SLOT) * B_BYTES + i_ * NTHR * 16); \
        wnext += wstep;                                                                                      \
    }
    TD_GLDS_B(0);
    TD_GLDS_B(1);
    // modulation rows of this tile (EPI_EMB_SILU; zero where there is no image / no cout): requested here, written to LDS in front of the prologue's
    // barrier.  Never in the DMA instantiation: launch_glds_cfg keeps EPI_EMB_SILU launches on the plain one, and 512 more bytes of LDS would cost
    // the bn 128 DMA tile -- 48 KB ring + 32 KB of stage buffers = exactly half a CU's LDS -- its second workgroup per CU.
    const bool cv_stage = !DMA1 && k_epi == EPI_EMB_SILU && tid < NIMG * BN;
    float cv_val = 0.f;
    if (cv_stage) {
        const int im = tid / BN, c = tid - im * BN;
        if (n0 + im < k_N && co0 + c < k_Cout) cv_val = p.cvec[(size_t)(n0 + im) * k_cvs + co0 + c];
    }

    // ---- activation-patch staging.  Per thread A_ITERS 16-byte pieces (patch pixel e>>3, slot e&7).  The packed coordinate is
    // segment independent; the element offset of the piece inside a segment's source tensor (aoff) is recomputed once per SEGMENT,
    // together with the segment descriptor (kept in registers: re-reading p.seg[] from kernarg memory inside the K loop costs a
    // serialised s_load + s_waitcnt per field), so staging a K-group is just "base + chunk*64".
    int a_coord[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int e = tid + it * NTHR, pp = e >> 3;
        const int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
        const int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
        const bool ok = (pp < NPATCH) && px < TW + 2 && n < k_N && y >= 0 && y < k_H && x >= 0 && x < k_W;  // px >= TW+2: pad columns
        const bool interior = py >= 1 && py <= TH && px >= 1 && px <= TW;
        a_coord[it] = ok ? ((n << 21) | (y << 11) | (x << 1) | (interior ? 1 : 0)) : -1;
    }
    u32x4 av[A_ITERS];
    int aoff[A_ITERS];          // element offset of this thread's piece in the current segment's source, or -1 (zero fill)
    const T* seg_src = nullptr;  // current segment descriptor, in registers
    int seg_taps = 9, seg_xform = 0, seg_nchunks = 0;
    float seg_scale = 1.f;
#define TD_SEG_BEGIN(SEG)                                                                                             \
    {                                                                                                                 \
        const ConvSeg& sg_ = p.seg[SEG];                                                                              \
        seg_src = (const T*)sg_.src; seg_taps = sg_.taps; seg_xform = sg_.xform; seg_scale = sg_.scale; seg_nchunks = sg_.C / CHUNK; \
        const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            int c_ = a_coord[it_];                                                                                    \
            /* opaque: hipcc otherwise decodes (n, y, x) of every piece ONCE, in front of the segment loop, and keeps the 18 values alive across \
               the K loop -- in scratch (bn 128: 72 bytes per lane written and read back per workgroup = ~30 MB of traffic per launch) */       \
            asm volatile("" : "+v"(c_));                                                                              \
            aoff[it_] = -1;                                                                                           \
            if (c_ >= 0 && (seg_taps == 9 || (c_ & 1)))                                                               \
                aoff[it_] = src_pixel(c_ >> 21, (c_ >> 11) & 1023, (c_ >> 1) & 1023, Hs_, Ws_, rs_) * cs_ + (tid & 7) * PER16; \
        }                                                                                                             \
    }
#define TD_LOAD_A(CH)                                                                                  \
    {                                                                                                  \
        const T* src_ = seg_src + (CH) * CHUNK;                                                        \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            /* always issued (offset 0 for zero-fill pieces) so that the vmcnt bookkeeping of the main loop is exact */ \
            av[it_] = *(const u32x4*)(src_ + (aoff[it_] >= 0 ? aoff[it_] : 0));                        \
        }                                                                                              \
    }
#define TD_STORE_A()                                                                                   \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = aoff[it_] >= 0 ? av[it_] : u32x4{0u, 0u, 0u, 0u};                           \
                if (seg_xform != 0 && aoff[it_] >= 0) {                                                \
                    float s_ = seg_scale;                                                              \
                    if (seg_xform == 2) s_ *= s_rn[pp_];                                               \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(s_a + pp_ * PITCH + (slot_ << 4)) = v_;                                      \
            }                                                                                          \
        }                                                                                              \
    }
    // Round 5: the residual runs of the wide epilogue (EPI_RESIDUAL: 16 bytes per lane and unit, MT * NU units) are requested at tap 6 of the LAST
    // 3x3 K-group into the patch-prefetch registers `av`, which are dead there -- exactly A_ITERS loads behind that tap's weight tile, with the
    // counted waits of taps 7 and 8 (TD_TAPP); the loop carries only the run addresses (2 MT registers) on top.  Index-clamped and unconditional (a pixel outside the image or a
    // cout tile past Cout reads a valid address nobody uses).  Units beyond A_ITERS (bn 128: 8 > 6) are fetched at the top of the epilogue.
    // (never in the DMA instantiation: its launches end in 1x1 K-groups by construction)
    const bool r_want = !DMA1 && k_ks == 1 && !k_of32 && (k_Cout & 7) == 0 && k_epi == EPI_RESIDUAL && k_hres;
    bool r_pref = false;   // `av` holds the residual runs
    // Per 32-pixel row group, the address of this lane's first residual run -- made HERE, in the prologue, when the launch will prefetch: the first
    // version read the residual's geometry (five kernel-argument fields) and did the pixel arithmetic inside tap 0 of the last K-group, and the scalar
    // loads' lgkmcnt(0) drained the fragment reads in flight there: +2.8 k cycles in the K loop of a 27-tap workgroup, +3.2 k in the decoder's 9-tap
    // ones (s_memtime traces, profiles/r05_conv_glds_phase_traces.txt) -- most of what the prefetch saved in the epilogue.
    const T* r_ptr[MT];
#define TD_R_ADDR()                                                                                                   \
    {                                                                                                                 \
        const int rHs_ = p.res_Hs, rWs_ = p.res_Ws, rrs_ = p.res_resample, rcs_ = p.res_cstride;                      \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                                           \
            int img_, ty_, tx_;                                                                                       \
            frag_pixel<TW, TPIX>(wm * WM + i_ * 32, l31, img_, ty_, tx_);                                             \
            const int n_ = n0 + img_, y_ = y0 + ty_, x_ = x0 + tx_;                                                   \
            const int sp_ = (n_ < k_N && y_ < k_H && x_ < k_W) ? src_pixel(n_, y_, x_, rHs_, rWs_, rrs_) : 0;         \
            r_ptr[i_] = (const T*)p.res + (sp_ * rcs_ + co0 + wn * WN + 8 * lh);                                      \
        }                                                                                                             \
    }
#define TD_R_UNIT(Q) (*(const u32x4*)(r_ptr[(Q) / NU] + ((co0 + wn * WN + (((Q) % NU) >> 1) * 32 < k_Cout) ? (((Q) % NU) >> 1) * 32 + ((Q) & 1) * 16 : 0)))
#define TD_LOAD_R()                                                                                                   \
    {                                                                                                                 \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) av[it_] = TD_R_UNIT(it_ < MT * NU ? it_ : MT * NU - 1); \
        r_pref = true;                                                                                                \
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) r_ptr[i] = nullptr;
    if (r_want) TD_R_ADDR()
    // DMA instantiation, K range starting in a 1x1 segment (a pure 1x1 conv, or a later split-K slice): nothing is staged through registers
    const bool dma_first = DMA1 && p.seg[seg_first].taps != 9;
    if (!dma_first) {
        TD_SEG_BEGIN(seg_first);
        TD_LOAD_A(chunk_first);
    }

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

    if (cv_stage) s_cv[tid] = cv_val;   // (requested at the top of the prologue)

    // ---- MFMA operand addressing: weights = A operand (rows = couts), activations = B operand (cols = pixels).
    // xbase: LDS byte address of the TOP-LEFT tap of this lane's pixel (+ its k-half); a tap adds ((dy*PW + dx) * PITCH), a 16-deep
    // k-step adds 32 -- both compile-time.  wbase[ks]: this lane's cout row inside a ring slot with the slab's XOR swizzle applied
    // (tap- and slot-invariant); the slot and the 32-row step j are compile-time offsets.
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
    __syncthreads();  // s_rn visible (prologue only: this one may drain the two weight tiles, they are needed next anyway)
    if (!dma_first) TD_STORE_A();
    // (zeroed HERE, behind the prologue: zeroed in front of it the 32-64 accumulator registers were live across the patch-address arithmetic and the
    // 1/rms table, and the bn 128 instantiations spilled 18-22 dwords per lane there -- ~35 MB of scratch written and read back per launch)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TD_T(tr_pro);

#define TD_TOFF(T) ((((T) / 3) * PW + ((T) % 3)) * PITCH)
    int slot = 0;  // ring slot of the current K-step; compile-time inside a 9-tap group (RING divides 9), tracked for 1x1 segments
    u32x4 wfA_[NT], xfA_[MT], wfB_[NT], xfB_[MT];
#define TD_FRAG_READ(WF, XF, SLOT, KS, TOFF)                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(smem + wbase[KS] + ((SLOT) * B_BYTES + j_ * 4096)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(smem + xbase[i_] + ((TOFF) + (KS) * 32)); \
    }
#define TD_FRAG_MFMA(WF, XF)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                acc[i_][j_] = Half<T>::mfma32(__builtin_bit_cast(hx8, WF[j_]), __builtin_bit_cast(hx8, XF[i_]), acc[i_][j_]); \
    }
    // The tap's barrier sits between k-steps 1 and 2 and does NOT drain the LDS queue; the next tap's first
    // two k-steps of fragments are requested behind this tap's last MFMAs, so no fragment read is ever waited for right after
    // it was issued.  Passing barrier(k): weight tile k+1 is visible, nobody reads tile k-1 any more (its slot takes tile k+2).
#define TD_TAPP(TAPIDX, SLOT, TOFF, TOFF_NEXT)                                                               \
    {                                                                                                        \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        TD_FRAG_READ(wfA_, xfA_, SLOT, 2, TOFF);                                                             \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
        TD_FRAG_READ(wfB_, xfB_, SLOT, 3, TOFF);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        TD_T(tA_);                                                                                           \
        /* tile k+1 (issued one tap ago) has landed; only tap 0's patch loads may be younger (tap 1).  Last K-group of the launch (r_now): the    \
           residual runs go out at tap 6, BEHIND the weight tile of tap 8 -- tap 7 lets them stay in flight, and tap 8 needs nothing that was    \
           issued after that tile (tile 9 belongs to a K-group that does not exist), so nothing ever waits for them inside the loop (requested \
           at tap 0 they had two taps to land -- patch loads hit the L2, residual rows come from HBM: +1 k cycles of tap-entry wait) */        \
        if ((TAPIDX) == 1 && has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS) : "memory");        \
        else if ((TAPIDX) == 7 && r_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS) : "memory");      \
        else if ((TAPIDX) == 8 && r_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS + NBI) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        /* LDS returns in order: everything older than the two k-steps just requested is back */             \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT + MT)) : "memory");                               \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        TD_T(tB_); TD_TACC(tr_wait, tA_, tB_);                                                               \
        if ((TAPIDX) == 3 && has_next) {                                                                     \
            _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) asm volatile("" : "+v"(av[it_]));      \
        }                                                                                                    \
        TD_ABL_BLOAD(TD_GLDS_B(((SLOT) + 2) % RING));                                                        \
        if ((TAPIDX) == 0 && has_next) TD_LOAD_A(chunk + 1);                                                 \
        if ((TAPIDX) == 6 && r_now) TD_LOAD_R()                                                              \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        if ((TAPIDX) < 8) TD_FRAG_READ(wfA_, xfA_, ((SLOT) + 1) % RING, 0, TOFF_NEXT);                       \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
        if ((TAPIDX) < 8) TD_FRAG_READ(wfB_, xfB_, ((SLOT) + 1) % RING, 1, TOFF_NEXT);                       \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        if ((TAPIDX) < 8) __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        if ((TAPIDX) < 8) __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                           \
    }
    // entry of a pipelined 9-tap group: this wave's patch ds_writes are out; tile k (slot 0) was issued >= 1 tap ago
#define TD_GROUP_ENTRY()                                                                                     \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        TD_FRAG_READ(wfA_, xfA_, 0, 0, TD_TOFF(0));                                                          \
        TD_FRAG_READ(wfB_, xfB_, 0, 1, TD_TOFF(0));                                                          \
    }
    // 1x1 segment: one K-step per group.  The next group's patch loads go out first; the weight tile two steps ahead is fetched
    // AFTER the restage (whose compiler-inserted vmcnt(0) for the patch registers would otherwise drain a just-issued tile).
#define TD_TAP1(SLOT, TOFF)                                                                                  \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if (has_next) TD_LOAD_A(chunk + 1);                                                                  \
        TD_FRAG_READ(wfA_, xfA_, SLOT, 0, TOFF);                                                             \
        TD_FRAG_READ(wfB_, xfB_, SLOT, 1, TOFF);                                                             \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        TD_FRAG_READ(wfA_, xfA_, SLOT, 2, TOFF);                                                             \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
        TD_FRAG_READ(wfB_, xfB_, SLOT, 3, TOFF);                                                             \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
    }
    int gidx = g0;
    for (int seg = seg_first; seg < p.nseg && gidx < g1; ++seg) {
        if constexpr (DMA1) if (p.seg[seg].taps != 9) {
            // ---- every remaining segment is 1x1 with no input transform (host contract): their K-groups are streamed by LDS-DMA, NST - 1
            // groups ahead, one barrier per group and no staging through registers (the register path pays a global-load round trip, a
            // ds_write pass and two barriers per 64 channels: profiles/r03_conv_1x1_dma_and_weighted_splitk.txt)
            const T* psrc = nullptr; int pseg = seg, pchunk = seg == seg_first ? chunk_first : 0, pn = 1;   // a split-K slice may start inside a 1x1 segment
            unsigned poff[DMA_N];
#define TD_P_BEGIN()                                                                                                  \
            {                                                                                                         \
                const ConvSeg& sg_ = p.seg[pseg];                                                                     \
                psrc = (const T*)sg_.src; pn = sg_.C / CHUNK;                                                         \
                const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                          \
                _Pragma("unroll") for (int i_ = 0; i_ < DMA_N; ++i_) {                                                \
                    const int e_ = tid + i_ * NTHR, row_ = e_ >> 3, sl_ = (e_ & 7) ^ TD_SWZ(row_);                    \
                    int img_, ty_, tx_;                                                                               \
                    frag_pixel<TW, TPIX>(row_ & ~31, row_ & 31, img_, ty_, tx_);                                      \
                    const int n_ = n0 + img_, y_ = y0 + ty_, x_ = x0 + tx_;                                           \
                    /* a pixel outside the image is an MFMA column nobody stores: any readable address will do */     \
                    const int pix_ = (n_ < p.N && y_ < p.H && x_ < p.W) ? src_pixel(n_, y_, x_, Hs_, Ws_, rs_) : 0;   \
                    poff[i_] = (unsigned)(pix_ * cs_ + sl_ * PER16) * (unsigned)sizeof(T);                            \
                }                                                                                                     \
            }
#define TD_P_PREP(BUF)                                                                                                \
                const unsigned long long sa_ = (unsigned long long)(psrc + (size_t)pchunk * CHUNK);                   \
                const unsigned char* su_ = td_uniform_ptr((const unsigned char*)sa_); \
                const unsigned lb_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsw + (unsigned)A_BASE + (unsigned)(BUF) * (unsigned)STAGE_BYTES));
#define TD_P_PIECES(I0, I1)                                                                                           \
                _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_) TD_GLDS16(poff[i_], su_, lb_, i_ * NTHR * 16);
            /* past the last group the cursor stays on it: the loop issues unconditionally (fixed vmcnt), the copies are never read */
#define TD_P_ADVANCE()                                                                                                \
                if (pchunk + 1 < pn) ++pchunk;                                                                        \
                else if (pseg + 1 < p.nseg) { ++pseg; pchunk = 0; TD_P_BEGIN(); }
#define TD_P_ISSUE(BUF)                                                                                               \
            {                                                                                                         \
                TD_P_PREP(BUF)                                                                                        \
                TD_P_PIECES(0, DMA_N)                                                                                 \
                TD_P_ADVANCE()                                                                                        \
            }
            unsigned xb1[4], xcur[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xb1[ks] = (unsigned)A_BASE + (unsigned)((wm * WM + l31) * 128 + (((ks * 2 + lh) ^ TD_SWZ(l31)) << 4));
#define TD_FRAG_READ1(WF, XF, SLOT, KS)                                                                      \
            {                                                                                                \
                _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(smem + wbase[KS] + ((SLOT) * B_BYTES + j_ * 4096)); \
                _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(smem + xcur[KS] + i_ * 4096); \
            }
            // One 1x1 K-group.  The DMA pieces (the stage NST-1 groups ahead, then the weight tile two groups ahead: the order the vmcnt
            // bookkeeping assumes) are issued BEHIND the fragment reads and between the MFMA groups, so that their issue cost (60-185 cycles a
            // piece, MI355X guide) runs under the LDS latency and the matrix pipe instead of in front of both
#define TD_TAP1D(SLOT)                                                                                       \
            {                                                                                                \
                TD_FRAG_READ1(wfA_, xfA_, SLOT, 0);                                                          \
                TD_FRAG_READ1(wfB_, xfB_, SLOT, 1);                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_P_PREP(bprev)                                                                             \
                TD_P_PIECES(0, DMA_N / 2)                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_FRAG_MFMA(wfA_, xfA_);                                                                    \
                TD_FRAG_READ1(wfA_, xfA_, SLOT, 2);                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_P_PIECES(DMA_N / 2, DMA_N)                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_FRAG_MFMA(wfB_, xfB_);                                                                    \
                TD_FRAG_READ1(wfB_, xfB_, SLOT, 3);                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_GLDS_B(((SLOT) + 2) % RING);                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_FRAG_MFMA(wfA_, xfA_);                                                                    \
                TD_FRAG_MFMA(wfB_, xfB_);                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                TD_P_ADVANCE()                                                                               \
            }
            TD_P_BEGIN();
            // the builtin (not asm) form is seen by the compiler's wait-count pass: it then knows that no register-path patch load is pending
            // and does not guard the reuse of those registers with vmcnt waits of its own (which would drain the DMA stream below).
            // In flight here: the two weight tiles ahead, needed by the first group anyway.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave is done with the halo patch: the stage buffers take its place
            asm volatile("" ::: "memory");
#pragma unroll
            for (int d = 0; d < NST - 1; ++d) TD_P_ISSUE(d);
            int bcur = 0;
            for (bool first = true; gidx < g1; ++gidx, first = false) {
                // in flight, oldest first: [W(g), W(g+1) at entry |] S(g) .. S(g+NST-2) interleaved with W(g+1): everything but the youngest
                // NST-2 stages and (after the first group) the youngest weight tile has to be back
                if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DMA_N) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DMA_N + NBI) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // stage g and weight tile g are visible; nobody reads stage g-1 / tile g-1 any more
                asm volatile("" ::: "memory");
                const int bprev = bcur == 0 ? NST - 1 : bcur - 1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xcur[ks] = xb1[ks] + (unsigned)bcur * (unsigned)STAGE_BYTES;
                if (slot == 0) TD_TAP1D(0) else if (slot == 1) TD_TAP1D(1) else TD_TAP1D(2);
                slot = slot == 2 ? 0 : slot + 1;
                bcur = bcur + 1 == NST ? 0 : bcur + 1;
            }
#undef TD_TAP1D
#undef TD_FRAG_READ1
#undef TD_P_ISSUE
#undef TD_P_PREP
#undef TD_P_PIECES
#undef TD_P_ADVANCE
#undef TD_P_BEGIN
            break;
        }
        if (seg > seg_first) {  // first K-group of a later segment: its patch could not be prefetched (different source tensor / transform)
            TD_SEG_BEGIN(seg);
            TD_LOAD_A(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave is done reading the previous patch
            asm volatile("" ::: "memory");
            TD_STORE_A();                  // visible after the next tap's lgkmcnt(0) + barrier
        }
        for (int chunk = (seg == seg_first ? chunk_first : 0); chunk < seg_nchunks && gidx < g1; ++chunk, ++gidx) {
            const bool has_next = chunk + 1 < seg_nchunks && gidx + 1 < g1;
            const bool r_now = r_want && gidx + 1 == g1;   // last K-group of the launch (never true in front of 1x1 groups)
            if (DMA1 || seg_taps == 9) {  // slot == 0 here: the host orders 3x3 segments before 1x1 segments, and RING divides 9
                TD_GROUP_ENTRY();
                TD_TAPP(0, 0, TD_TOFF(0), TD_TOFF(1)); TD_TAPP(1, 1, TD_TOFF(1), TD_TOFF(2)); TD_TAPP(2, 2, TD_TOFF(2), TD_TOFF(3));
                TD_TAPP(3, 0, TD_TOFF(3), TD_TOFF(4)); TD_TAPP(4, 1, TD_TOFF(4), TD_TOFF(5)); TD_TAPP(5, 2, TD_TOFF(5), TD_TOFF(6));
                TD_TAPP(6, 0, TD_TOFF(6), TD_TOFF(7)); TD_TAPP(7, 1, TD_TOFF(7), TD_TOFF(8)); TD_TAPP(8, 2, TD_TOFF(8), TD_TOFF(8));
            } else {  // centre tap only
                if (slot == 0) TD_TAP1(0, TD_TOFF(4)) else if (slot == 1) TD_TAP1(1, TD_TOFF(4)) else TD_TAP1(2, TD_TOFF(4));
            }
            if (has_next) {
                TD_T(tS0_);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // every wave is done reading the current patch
                asm volatile("" ::: "memory");
                TD_ABL_BSTORE(TD_STORE_A());   // visible to the others after the next tap's lgkmcnt(0) + barrier
                TD_T(tS1_); TD_TACC(tr_stage, tS0_, tS1_);
            }
            if (!DMA1 && seg_taps != 9) {
                if (slot == 0) TD_GLDS_B(2) else if (slot == 1) TD_GLDS_B(0) else TD_GLDS_B(1);
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two over-fetched tail tiles must not land in a successor's LDS
#undef TD_TOFF
#undef TD_TAP1
#undef TD_TAPP
#undef TD_GROUP_ENTRY
#undef TD_FRAG_READ
#undef TD_FRAG_MFMA
#undef TD_LOAD_A
#undef TD_STORE_A
#undef TD_SEG_BEGIN
#undef TD_GLDS_B

    TD_T(tr_loop);
    // ---------------- epilogue: lane holds, per 32x32 tile, 4 groups of 4 consecutive couts of pixel column (lane & 31)
    // Its scalar kernel arguments in ONE burst (round 4; hipcc otherwise re-reads them one at a time behind the epilogue's uniform branches:
    // five serialised scalar-cache round trips between the last MFMA and the first store of every workgroup).  Pointers are not pinned (a
    // pointer that went through an asm statement would be accessed with FLAT instructions); only their null tests are.
    int e_of32 = p.out_f32, e_Cout = p.Cout, e_epi = p.epi, e_ocs = p.out_cstride, e_cvs = p.cvec_stride, e_rcs = p.res_cstride, e_rHs = p.res_Hs, e_rWs = p.res_Ws,
        e_rrs = p.res_resample;
    float e_rsc = p.res_scale, e_clip = p.clip, e_o2s = p.out2_scale;
    const bool e_hres = p.res != nullptr, e_hrss = p.res_sumsq != nullptr, e_hoss = p.out_sumsq != nullptr, e_ho2 = p.out2 != nullptr;
    asm volatile("" : "+s"(e_of32), "+s"(e_Cout), "+s"(e_epi), "+s"(e_ocs), "+s"(e_cvs), "+s"(e_rcs), "+s"(e_rHs), "+s"(e_rWs), "+s"(e_rrs), "+s"(e_rsc), "+s"(e_clip),
                 "+s"(e_o2s));
    const size_t M = (size_t)k_N * k_H * k_W;
#ifdef TD_ABLATE_EPI
    {
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t_ += acc[i][j][r];
        if (t_ == 12345.678f) ((float*)p.out)[tid] = t_;
    }
    if (p.N >= 0) return;
#endif
    // Wide path (bf16 output, Cout % 8 == 0): the C layout gives a lane 4 consecutive couts (8 B) per row group and its partner
    // lane (l ^ 32) the next 4; one v_permlane32_swap per dword turns two row groups into one 16-byte run per lane, so the tile
    // leaves as dwordx4 stores (the store tail is issue-bound: half the instructions, half the time; guide T21).  The residual is
    // fetched the same way in reverse (16-byte loads, then the same swap restores the MFMA layout).
    const bool wide = !e_of32 && (e_Cout & 7) == 0;
    if (k_ks > 1) {  // raw fp32 partial sums [ksplit][pixel][CoutPad]; conv_splitk_reduce_kernel adds them in fixed order + epilogue
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int img, ty, tx;
            frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
            const int n = n0 + img, y = y0 + ty, x = x0 + tx;
            if (n < k_N && y < k_H && x < k_W) {
                float* prow = p.partial + ((size_t)ksp * M + ((size_t)n * k_H + y) * k_W + x) * k_cpad + co0 + wn * WN + 4 * lh;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        *(f32x4*)(prow + j * 32 + rg * 8) = f32x4{acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
            }
        }
        return;
    }
    const bool has_res = e_epi == EPI_RESIDUAL && e_hres;
    constexpr int NRX = MT * NU > A_ITERS ? MT * NU - A_ITERS : 1;
    u32x4 rx[NRX];
#pragma unroll
    for (int q = 0; q < NRX; ++q) rx[q] = u32x4{0u, 0u, 0u, 0u};
    if (wide && has_res) {
        if (!r_want) TD_R_ADDR()   // (the DMA instantiation, split-K: no prefetch was planned, the addresses are made here)
        if (!r_pref) TD_LOAD_R()   // the launch ended in 1x1 K-groups (attention projection): nothing was requested yet
        if constexpr (MT * NU > A_ITERS) {
#pragma unroll
            for (int q = 0; q < NRX; ++q) rx[q] = TD_R_UNIT(A_ITERS + q);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
        const int n = n0 + img, y = y0 + ty, x = x0 + tx;
        const bool ok = n < k_N && y < k_H && x < k_W;
        // sums of squares (pixel-norm statistic of the consumer) are kept per 32-cout MFMA block: the partial decomposition -- and with it
        // the fp32 summation order the consumer sees -- is then the same for every tile shape (bn 96 / 128, 4 or 8 waves)
        float ssj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ssj[j] = 0.f;
        if (ok) {
            const float rn = e_hrss ? s_rn[base_pp[i]] : 1.f;
            const int cobase = co0 + wn * WN + 4 * lh;
            const int sp = (e_epi == EPI_RESIDUAL && e_hres) ? src_pixel(n, y, x, e_rHs, e_rWs, e_rrs) : 0;
            if (wide) {
                // units of 8 couts (shared arithmetic: epi_unit8).  Round 5: no operand is waited for inside the unit loop -- the modulation rows
                // come from LDS (s_cv, staged by the prologue), the residual runs were requested during the last K-group (`av`, TD_LOAD_R) or, for
                // the units `av` does not hold, at the top of the epilogue (`rx`); the loop is straight-line code per epilogue kind (a load in a
                // wave-uniform branch between the stores of the previous unit made hipcc wait for it right behind its issue).
                const size_t pix = ((size_t)n * k_H + y) * k_W + x;
                T* orow = (T*)p.out + pix * e_ocs + co0 + wn * WN + 8 * lh;
                const float rs = e_rsc * rn;
                const bool want_ss = e_hoss, want_o2 = e_ho2;
                const SiluK k_o2 = silu_k(e_o2s);
                auto body = [&](auto KIND) {
                    constexpr int K = decltype(KIND)::value;
                    f32x4 ca[NU], cb[NU];
                    u32x4 rw[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        ca[u] = f32x4{0.f, 0.f, 0.f, 0.f}; cb[u] = ca[u]; rw[u] = u32x4{0u, 0u, 0u, 0u};
                        if constexpr (K == 1) {
                            const float* c_ = s_cv + img * BN + wn * WN + (u >> 1) * 32 + (u & 1) * 16 + 4 * lh;
                            ca[u] = *(const f32x4*)c_; cb[u] = *(const f32x4*)(c_ + 8);
                        }
                        if constexpr (K == 2) rw[u] = i * NU + u < A_ITERS ? av[i * NU + u < A_ITERS ? i * NU + u : 0] : rx[i * NU + u >= A_ITERS ? i * NU + u - A_ITERS : 0];
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int j = u >> 1, m = u & 1;
                        if (co0 + wn * WN + j * 32 >= e_Cout) continue;
                        const f32x4 va = {acc[i][j][8 * m + 0], acc[i][j][8 * m + 1], acc[i][j][8 * m + 2], acc[i][j][8 * m + 3]};
                        const f32x4 vb = {acc[i][j][8 * m + 4], acc[i][j][8 * m + 5], acc[i][j][8 * m + 6], acc[i][j][8 * m + 7]};
                        u32x4 o, o2;
                        epi_unit8<T>(e_epi, has_res, e_clip, want_ss, want_o2, va, vb, ca[u], cb[u], rw[u], rs, k_o2, o, o2, ssj[j]);
                        *(u32x4*)(orow + j * 32 + m * 16) = o;
                        if (want_o2) *(u32x4*)((T*)p.out2 + (orow - (T*)p.out) + j * 32 + m * 16) = o2;
                    }
                };
                if (e_epi == EPI_EMB_SILU) { if constexpr (!DMA1) body(std::integral_constant<int, 1>{}); }
                else if (has_res) body(std::integral_constant<int, 2>{});
                else body(std::integral_constant<int, 0>{});
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    f32x4 aux[4];
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        aux[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.epi == EPI_EMB_SILU) aux[rg] = *(const f32x4*)(p.cvec + (size_t)n * p.cvec_stride + cobase + j * 32 + rg * 8);
                        else if (p.epi == EPI_RESIDUAL && p.res) aux[rg] = load4<T>(p.res, (size_t)sp * p.res_cstride + cobase + j * 32 + rg * 8);
                    }
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        f32x4 v = {acc[i][j][rg * 4 + 0], acc[i][j][rg * 4 + 1], acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]};
                        ssj[j] += epilogue4<T>(p, n, y, x, cobase + j * 32 + rg * 8, v, rn, aux[rg]);
                    }
                }
            }
        }
        if (e_hoss) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float ss = ssj[j] + __shfl_xor(ssj[j], 32);
                if (ok && lh == 0 && co0 + wn * WN + j * 32 < k_cpad) {
                    const size_t pix = ((size_t)n * k_H + y) * k_W + x;
                    p.out_sumsq[(size_t)((co0 + wn * WN) / 32 + j) * M + pix] = ss;
                }
            }
        }
    }
#ifdef TD_TRACE
    TD_T(tr_eissue);  // epilogue instructions issued; the stores are still in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TD_T(tr_end);
    if (lane == 0) {
        unsigned long long* tb = (unsigned long long*)p.partial + ((size_t)blockIdx.x * (WAVES_M * WAVES_N) + wave) * 16;  // 16 u64 per wave
        tb[0] = tr_pro - tr_start; tb[1] = tr_loop - tr_pro; tb[2] = tr_end - tr_loop; tb[3] = tr_wait; tb[4] = tr_stage; tb[5] = tr_start; tb[6] = tr_end;
        tb[8] = tr_rt0; tb[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | __builtin_amdgcn_s_getreg((3 << 11) | 20);
        tb[10] = __builtin_amdgcn_s_memrealtime();
        tb[7] = tb[10] - tr_rt0; tb[11] = tr_end - tr_eissue;  // 100 MHz constant clock: shader clock = 100 MHz * (tb[6]-tb[5]) / tb[7]
    }
#endif
}

int g_bench_extra_lds = 0;  // tools/conv_bench.hip only: extra dynamic LDS per workgroup, to force one workgroup per CU in occupancy experiments

template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
static hipError_t launch_glds_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int NPATCH = NIMG * (TH + 2) * (TW == 8 ? 12 : TW + 2);
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr size_t RING_BYTES = 3 * (size_t)(((BN * 128 + NTHR * 16 - 1) / (NTHR * 16)) * NTHR * 16);
    constexpr size_t STAGES = (size_t)(WAVES_M * WAVES_N >= 8 ? 3 : 2) * (NIMG * TH * TW * 128);   // 1x1 stage buffers over the patch (+ s_rn)
    constexpr size_t PATCH_RN = (size_t)NPATCH * 144 + NPATCH * 4, CV_BYTES = (size_t)NIMG * BN * 4;   // the kernel's CV_BASE follows the same rule
    bool seen1 = false;  // the kernel's compile-time ring slots need every 3x3 segment to start on a multiple of 3 K-steps
    for (int s = 0; s < p.nseg; ++s) { if (p.seg[s].taps == 9 && seen1) return hipErrorInvalidValue; if (p.seg[s].taps != 9) seen1 = true; }
    // LDS-DMA streaming of the 1x1 segments (pure 1x1 convs and split-K slices that start inside a 1x1 segment included): no input transform on
    // the 1x1 sources, and no pixel-norm table (s_rn lies under the stage buffers)
    ConvParams pd = p;
    bool dma = p.dma1x1 != 0 && seen1 && p.seg[0].xform != 2 && p.res_sumsq == nullptr;
    for (int s = 0; s < p.nseg && dma; ++s) if (p.seg[s].taps != 9 && p.seg[s].xform != 0) dma = false;
    const int grid = p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups * p.ksplit;
    constexpr bool HAS_DMA = true;
    if (p.epi == EPI_EMB_SILU) dma = false;   // the modulation rows live in LDS behind the patch: the plain instantiation only (no such launch has 1x1 segments in the U-Net)
    const size_t lds = RING_BYTES + (dma ? std::max(PATCH_RN, STAGES) : (PATCH_RN + 15) / 16 * 16 + CV_BYTES) + (size_t)g_bench_extra_lds;
    pd.dma1x1 = dma ? 1 : 0;
    {   // workgroup-id decode constants of the kernel prologue
        const int mtiles_ = p.tiles_x * p.tiles_y * p.img_groups;
        if (grid <= 0 || (long long)grid * std::max(mtiles_, p.n_ntiles) >= ((long long)1 << 32)) return hipErrorInvalidValue;   // td_udiv's range
        pd.sb_d0 = mtiles_; pd.sb_m0 = td_magic(mtiles_); pd.sb_d1 = p.n_ntiles; pd.sb_m1 = td_magic(p.n_ntiles); pd.sb_m2 = td_magic(p.tiles_x); pd.sb_m3 = td_magic(p.tiles_y);
        pd.sb_grid = grid; pd.sb_grid8 = (grid & 7) == 0 ? (unsigned)grid >> 3 : 0u;
    }
    auto kern = dma ? conv_glds_kernel<T, TH, TW, NIMG, BN, WAVES_M, WAVES_N, HAS_DMA> : conv_glds_kernel<T, TH, TW, NIMG, BN, WAVES_M, WAVES_N, false>;
    // per (instantiation, device): hipFuncSetAttribute applies to the CURRENT device's copy of the kernel only
    static bool attr_set[2][64] = {};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dma][dev_] || g_bench_extra_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dma][dev_] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, st, pd);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
        const size_t W_ = (size_t)p.N * p.H * p.W * ((p.CoutPad + 255) / 256);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<T>, dim3((unsigned)((W_ + 3) / 4)), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    return e;
}

// Tile configurations (variant):
//   0 "big"   : 8 waves, 256-pixel tile (16x16, narrow maps 8x8 x 4 images), bn 128 -> waves 4x2 (64 px x 64 co), bn 96 -> 8x1 (32 px x 96 co);
//               ~91 KB LDS -> one workgroup per CU
//   1 "small" : 4 waves, 128-pixel tile (8x16, narrow maps 8x8 x 2 images), bn 128 -> waves 2x2 (64 px x 64 co), bn 96 -> 4x1 (32 px x 96 co);
//               ~71 KB LDS -> two independent workgroups per CU whose prologues / epilogues / barrier stalls overlap
//   bn 64 (16-wide maps only): big = 8 waves 8x1 (32 px x 64 co), small = 4 waves 4x1 on 128 pixels (32 px x 64 co).  For the 64-cout-granular
//   layers (the decoder's 64 / 320-channel levels, 5-channel output convs) that would otherwise fall to the per-tap flavour.
template <typename T>
static hipError_t launch_conv_glds_t(const ConvParams& p, bool narrow, int bn, int variant, hipStream_t st) {
    if (bn == 64) {
        if (narrow) return hipErrorInvalidValue;
        // variant 2 "tiny" (round 3, the latency regime): 4 waves on 64 pixels (4x16) x 64 couts, 32x32 per wave.  A small batch then fills the chip
        // with (pixel tile, cout tile) workgroups instead of K slices: fewer or no fp32 partial slabs to write, read back and reduce
        if (variant == 2) return launch_glds_cfg<T, 4, 16, 1, 64, 2, 2>(p, st);
        return variant == 1 ? launch_glds_cfg<T, 8, 16, 1, 64, 4, 1>(p, st) : launch_glds_cfg<T, 16, 16, 1, 64, 8, 1>(p, st);
    }
    if (variant == 2) return hipErrorInvalidValue;
#ifdef TD_BN192   // tools/conv_bench.hip only: all 192 couts of the 64x64 level in one workgroup (profiles/r04_conv_bn192_tile.txt)
    if (bn == 192) {
        if (narrow) return hipErrorInvalidValue;
        return variant == 1 ? launch_glds_cfg<T, 8, 16, 1, 192, 2, 2>(p, st) : launch_glds_cfg<T, 16, 16, 1, 192, 4, 2>(p, st);
    }
#endif
    if (variant == 1) {
        if (!narrow) return bn == 128 ? launch_glds_cfg<T, 8, 16, 1, 128, 2, 2>(p, st) : launch_glds_cfg<T, 8, 16, 1, 96, 4, 1>(p, st);
        return bn == 128 ? launch_glds_cfg<T, 8, 8, 2, 128, 2, 2>(p, st) : launch_glds_cfg<T, 8, 8, 2, 96, 4, 1>(p, st);
    }
    if (!narrow) return bn == 128 ? launch_glds_cfg<T, 16, 16, 1, 128, 4, 2>(p, st) : launch_glds_cfg<T, 16, 16, 1, 96, 8, 1>(p, st);
    return bn == 128 ? launch_glds_cfg<T, 8, 8, 4, 128, 4, 2>(p, st) : launch_glds_cfg<T, 8, 8, 4, 96, 8, 1>(p, st);
}

// dtype: 1 bf16, 2 fp16 (this flavour has no fp32 form)
hipError_t launch_conv_glds(const ConvParams& p, int dtype, bool narrow, int bn, int variant, hipStream_t st) {
#ifdef TD_BN192
    if (bn == 192) return launch_conv_glds_t<__bf16>(p, narrow, bn, variant, st);
#endif
    if (bn != 64 && bn != 96 && bn != 128) return hipErrorInvalidValue;
    return dtype == 2 ? launch_conv_glds_t<_Float16>(p, narrow, bn, variant, st) : launch_conv_glds_t<__bf16>(p, narrow, bn, variant, st);
}

}  // namespace td
