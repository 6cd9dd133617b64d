// Wide tile of the LDS-DMA implicit-GEMM convolution (round 6): 256 pixels x BN couts per 4-wave workgroup, TWO workgroups per CU, 32-channel
// K-groups with a DOUBLE-BUFFERED activation patch -- one seamless MFMA pipeline over the whole 3x3 part of a launch.
//
// conv_glds.hip's tiles are either 128 px x 96 / 128 couts on 4 waves (two workgroups per CU, but a wave owns 32 px x 96 couts or 64 x 64:
// 4 fragment reads per 3 - 4 MFMAs, and every pixel tile re-streams the cout tile's whole weight tensor through the CU's L2 -> LDS path:
// 12 KB per tap and 128 pixels) or 256 px on 8 waves (half the weight bytes per pixel, but 91 KB of LDS: ONE workgroup per CU, and what the
// kernel lives on is two independent workgroups per CU whose prologues / restages / epilogues run under each other's MFMA stream -- DESIGN.md
// section 4).  This flavour has both: a wave owns 64 px x BN couts (MT = 2, NT = BN / 32: 5 fragment reads per 6 MFMAs at BN = 96), the
// workgroup 256 px (16 x 16), and the LDS footprint stays under half a CU because everything is kept in HALF K-steps:
//   * a K-group is 32 channels x 9 taps; a ring slot holds the 32 channels x BN couts of one tap (BN x 64 B), three slots, one barrier per
//     half K-step = per 2 x MT x NT MFMAs of a wave (the MFMA count the 128-pixel tile has per tap);
//   * the halo patch holds 32 channels (18 x 18 pixels x 80-byte rows = 26 KB) and exists TWICE: while the taps of K-group g read one buffer, the
//     patch of g + 1 is fetched (6 x 16 bytes per thread, registers) and written into the other.  There is no restage barrier and no pipeline
//     bubble at a group boundary: the last half-step of a group requests the first fragments of the next one, across 64-channel chunks and
//     across 3x3 segments alike.
// Per MFMA the workgroup ingests half the weight bytes, reads 0.6x the LDS bytes and runs half as many prologues / epilogue drains.
//
// Same maths, parameter block, packed weight slab ([K-step][cout][128 B], conv_glds.hip's layout: no extra copy) and fused prologues / epilogues as
// conv_glds.hip.  The K ORDER differs (64-channel chunk -> channel half -> tap -> two 16-deep k-steps, against chunk -> tap -> four k-steps): fp32
// accumulation in another order, so results agree with the other tiles to fp32 rounding of the sums (<= 1 bf16 ulp on ~1 % of the outputs), not bit
// for bit; a launch's flavour is a function of its shape and batch size only (planner), like conv_sb / conv_s16.
//
// The half K-step in LDS: row r (cout in tile) is 64 B = four 16-byte pieces; piece j of row r lies at r * 64 + ((j ^ ((r >> 2) & 3)) << 4).
// A ds_read_b128 is serviced in the 16-lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X guide, LDS): each group holds four
// runs of four consecutive rows whose (r >> 2) & 3 are pairwise different, so the 16 pieces of a fragment read fall on 16 different 16-byte
// columns of the 256-byte bank window: conflict-free.  The LDS-DMA writes wave-linear (LDS[base + 16 * lane]); each lane's GLOBAL address
// is free, so lane p copies the piece that belongs at position p: row p >> 2, piece (p & 3) ^ ((row >> 2) & 3), which the slab stores at
// slot (4 h + piece) ^ TD_SWZ(row) of the row's 128 bytes -- i.e. the two channel halves of a K-step differ by `offset ^ 64`.
// Patch rows are 80 bytes (64 + 16 of padding): 16 consecutive rows start at 16 different 16-byte columns of the bank window (5 is odd), and a
// fragment address is "lane base + compile-time (buffer, tap, k-step) offset" -- no VALU in the tap loop, as in conv_glds.hip.
#include "conv_common.h"

namespace td {

#ifdef TD_TRACE
#define TDW_T(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define TDW_TACC(acc, a, b) acc += (b) - (a)
#else
#define TDW_T(v)
#define TDW_TACC(acc, a, b)
#endif

// TAIL: the instantiation for launches whose K ends in 1x1 segments.  Its own code object because the tail's registers (stage addresses, cursor) would
// otherwise push the pure-3x3 launches -- the majority -- over 256 VGPRs: hipcc then spills the LDS-DMA offset registers and reloads them from scratch inside
// the half-step loop behind a vmcnt(0), which drains the weight stream.  The TAIL instantiation pays for its tail by not prefetching the residual runs.
//
// PERS (round 6, second step): the PERSISTENT instantiation.  A workgroup walks the tiles blockIdx.x, blockIdx.x + p.persist, ... of the launch: the K pipeline
// treats "first 64-channel unit of my next tile" exactly like "next unit of this tile" -- its halo patch is requested at half-step 9 of the tile's last unit
// and written at 14, the weight ring wraps to the next tile's cout rows at half tile 18, the first fragments are read in half-step 17 -- so that the epilogue
// of tile t is followed at once by the first MFMAs of tile t + 1 with three weight half tiles already in flight.  No prologue per tile (a third of a
// workgroup's life on the decoder's 64-channel 512x512 layers: profiles/r06_wide_tile_phase_traces.txt).  Per-tile LDS state (the pixel-norm factors of the
// patch, the modulation row) is double-buffered.  Same K order as the non-persistent wide tile: bit-identical results.  Pure 3x3 launches over whole
// 16 x 16 tiles and whole cout tiles only (the launcher checks).
template <typename T, int BN, bool TAIL, bool PERS>
__global__ __launch_bounds__(256, 2) void conv_glds_kernel_wide(const ConvParams p) {
    static_assert(!(TAIL && PERS), "the persistent tile loop serves pure 3x3 launches");
    typedef typename Half<T>::x8 hx8;
    constexpr int NTHR = 256, TH = 16, TW = 16, TPIX = 256;
    constexpr int PW = 18, NPATCH = 18 * 18;
    constexpr int WM = 64, MT = 2, NT = BN / 32;
    constexpr int CHUNK = 64, HALF = 32, PER16 = 8;
    // PERS splits the staging by wave: waves 0 / 1 stream the weight half tiles, waves 2 / 3 stage the halo patches.  s_waitcnt vmcnt completes IN ORDER, so a
    // wave that waits for a weight half tile requested two half-steps ago has thereby waited for every patch load it issued before that: with both roles in
    // every wave a patch never gets more than ~3 half-steps (~1 us) of flight, less than an HBM round trip under load, and the wave sits in the wait.  The
    // patch waves wait for nothing but their own loads, 5 - 7 half-steps after they asked.
    constexpr int WNT = PERS ? 128 : NTHR, PNT = PERS ? 128 : NTHR;       // threads of the weight role / of the patch role
    constexpr int A_ITERS = (NPATCH * 4 + PNT - 1) / PNT;                 // 6 / 11 (four 16-byte pieces per patch pixel and channel half)
    constexpr int A_FULL = NPATCH * 4 / PNT, A_REM = NPATCH * 4 - A_FULL * PNT;   // pieces 0 .. A_FULL - 1 of every patch thread lie inside the patch, piece A_FULL for the first A_REM threads
    // Weight ring: FOUR slots of exactly BN x 64 bytes, half tiles requested THREE half-steps before their taps (two before the barrier that publishes
    // them).  First build: three slots, one half-step of lead -- the s_memtime split of the tap-entry wait said 60 % of it (9 - 17 % of a workgroup's
    // life) was the wave's own LDS-DMA pieces still in flight, the rest barrier skew (profiles/r06_wide_tile_phase_traces.txt).
    // A half tile is BN x 4 sixteen-byte pieces: 256 / 384 / 512 for BN 64 / 96 / 128.  At BN 96 the second piece of waves 2 and 3 has no rows left;
    // they issue it all the same -- into a 2 KB dump area, from the address of their first piece (an L1 hit) -- so that every wave counts the same
    // number of pieces per half tile (the s_waitcnt immediates are compile-time).
    constexpr int NBI = (BN * 64 + WNT * 16 - 1) / (WNT * 16), PSTR = WNT * 16;   // pieces per weight thread and half tile (PERS: 2 / 3, no rowless piece); LDS bytes between a thread's pieces
    constexpr int H_BYTES = BN * 64, RING = 4;
    constexpr bool HAS_DUMP = (BN * 64) % (WNT * 16) != 0;
    constexpr int DUMP_BASE = RING * H_BYTES;
    constexpr int PITCH = 80;
    constexpr int A_BASE = DUMP_BASE + (HAS_DUMP ? 2048 : 0), A_BYTES = NPATCH * PITCH;      // two patch buffers: A_BASE, A_BASE + A_BYTES
    constexpr int RN_BASE = A_BASE + 2 * A_BYTES, RN_BYTES = (NPATCH * 4 + 15) / 16 * 16;
    constexpr int CV_BASE = RN_BASE + (PERS ? 2 : 1) * RN_BYTES;       // PERS: two sets of per-tile state, `par` = the current tile's
    constexpr int CV_BYTES = PERS ? 512 : BN * 4;                      // (PERS: the modulation row arrives by a one-dword LDS-DMA of BOTH weight waves, 256 bytes each)
    constexpr int NU = NT * 2;
    static_assert(BN % 32 == 0 && BN <= NTHR, "tile shape");
    static_assert((NT - 1) * 2048 + 64 < 65536 && A_BYTES + 2 * PW * PITCH + 2 * PITCH + 64 < 65536, "ds_read offset field");
    static_assert(!HAS_DUMP || NBI == 2, "dump piece");
    static_assert(!(PERS && HAS_DUMP) && A_FULL * PNT <= NPATCH * 4 && A_REM < PNT && A_ITERS == A_FULL + 1, "role split");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0
    float* s_rn = (float*)(smem + RN_BASE);
    float* s_cv = (float*)(smem + CV_BASE);
    int par = 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const bool w_role = !PERS || wave < 2, p_role = !PERS || wave >= 2;   // (wave-uniform)
    const int wt = PERS ? (tid & 127) : tid, pt = PERS ? tid - 128 : tid;  // index inside the role (pt < 0 in the weight waves: never used there)
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef TD_TRACE
    unsigned long long tr_wait = 0, tr_stage = 0;
#endif
    TDW_T(tr_start);
#ifdef TD_TRACE
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // PERS: kernel arguments used once per tile (segment descriptors, epilogue parameters, the tile decode's divisors) are RE-READ from the kernarg segment where
    // they are used -- TDW_FRESH() hides the segment pointer from the optimiser, so the s_loads behind it cannot be hoisted out of the tile loop.  Left to
    // itself hipcc hoists all ~80 of them, runs out of SGPRs and moves them through VGPR lanes: 1300 v_readlane / v_writelane in the first build's loop.
    typedef const ConvParams __attribute__((address_space(4)))* tdw_kp_t;
    tdw_kp_t kp_ = (tdw_kp_t)__builtin_amdgcn_kernarg_segment_ptr();
#define TDW_FRESH() { if constexpr (PERS) asm volatile("" : "+s"(kp_)); }
#define KA(F) (PERS ? kp_->F : p.F)
    // workgroup-id decode: conv_glds.hip's (XCD-aware sibling order, alternating walk direction, magic-number divisions, one kernel-argument burst)
    unsigned k_d1 = p.sb_d1, k_m1 = p.sb_m1, k_m2 = p.sb_m2, k_m3 = p.sb_m3, k_g8 = p.sb_grid8, k_grid = p.sb_grid;
    int k_tx = p.tiles_x, k_ty = p.tiles_y, k_rev = p.reverse, k_cpad = p.CoutPad, k_N = p.N, k_H = p.H, k_W = p.W, k_nseg = p.nseg;
    const unsigned char* k_wpack = (const unsigned char*)p.wpack;
    int k_epi = p.epi, k_Cout = p.Cout, k_cvs = p.cvec_stride;
    int k_c0 = p.seg[0].C, k_c1 = p.seg[1].C, k_c2 = p.seg[2].C, k_t0 = p.seg[0].taps, k_t1 = p.seg[1].taps, k_t2 = p.seg[2].taps;
    const bool k_hres = p.res != nullptr;
    asm volatile("" : "+s"(k_cpad), "+s"(k_N), "+s"(k_H), "+s"(k_W), "+s"(k_wpack), "+s"(k_epi), "+s"(k_Cout), "+s"(k_cvs), "+s"(k_nseg));
    asm volatile("" : "+s"(k_d1), "+s"(k_m1), "+s"(k_m2), "+s"(k_m3), "+s"(k_g8), "+s"(k_grid), "+s"(k_tx), "+s"(k_ty), "+s"(k_rev), "+s"(k_c0), "+s"(k_c1), "+s"(k_c2), "+s"(k_t0), "+s"(k_t1), "+s"(k_t2));
    // (virtual workgroup id VB -> image, tile origin, first cout.  PERS: VB = blockIdx.x + k * p.persist, p.persist a multiple of 8 when the grid is, so that
    // every tile of a workgroup lies in the range of its own XCD)
#define TDW_DECODE(VB, N0, Y0, X0, CO0)                                                                      \
    {                                                                                                        \
        unsigned ubid_ = (VB);                                                                               \
        if (k_g8) ubid_ = (ubid_ & 7) * k_g8 + (ubid_ >> 3);                                                 \
        if (k_rev) ubid_ = k_grid - 1 - ubid_;                                                               \
        const unsigned mtile_ = td_udiv(ubid_, k_d1, k_m1);                                                  \
        const int ntile_ = (int)(ubid_ - mtile_ * k_d1);                                                     \
        const unsigned q2_ = td_udiv(mtile_, (unsigned)k_tx, k_m2);                                          \
        const int txi_ = (int)(mtile_ - q2_ * (unsigned)k_tx), ig_ = (int)td_udiv(q2_, (unsigned)k_ty, k_m3), tyi_ = (int)(q2_ - (unsigned)ig_ * (unsigned)k_ty); \
        N0 = ig_; Y0 = tyi_ * TH; X0 = txi_ * TW; CO0 = ntile_ * BN;                                         \
    }
    unsigned vb = blockIdx.x;
    const unsigned vstep = PERS ? (unsigned)p.persist : 0u;
    int n0, y0, x0, co0;
    TDW_DECODE(vb, n0, y0, x0, co0)
    int pn0 = n0, py0 = y0, px0 = x0, pco0 = co0;   // the tile whose patch is being staged (PERS: runs ahead of the tile being accumulated from half-step 9 of its last unit)
    // 64-channel units of the leading 3x3 segments (the host orders 3x3 segments before 1x1 segments), per segment and in all
    const int u0 = k_t0 == 9 ? k_c0 / CHUNK : 0, u1 = (k_nseg > 1 && k_t0 == 9 && k_t1 == 9) ? k_c1 / CHUNK : 0, u2 = (k_nseg > 2 && u1 > 0 && k_t2 == 9) ? k_c2 / CHUNK : 0;
    const int n3 = u0 + u1 + u2;

    // ---- weight ring: half K-steps.  wnext = (SGPR) address of the K-step whose half is fetched next, wchunk = the first K-step (tap 0) of the 64-channel
    // unit being fetched: a unit is 18 half tiles -- taps 0 .. 8 of channel half 0, then taps 0 .. 8 of half 1 (the same nine 128-byte rows per cout,
    // `offset ^ 64`).  Every half-step fetches the half tile two half-steps ahead UNCONDITIONALLY (the slab carries two K-steps of tail padding).
    const size_t wstep = (size_t)k_cpad * 128;
    const unsigned char* wchunk = k_wpack + (size_t)co0 * 128;
    const unsigned char* wnext = wchunk;
    unsigned wvoff[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int pos = wt + i * WNT, r = pos >> 2, j = (pos & 3) ^ ((r >> 2) & 3), x = TD_SWZ(r);
        wvoff[i] = (unsigned)(r * 128 + ((((x & 4) | (j ^ (x & 3)))) << 4));
    }
    // LDS destination of piece i of a half tile in the slot at byte offset SLOT (wave-uniform): wave * 1 KB + i * 4 KB inside the slot; the rowless
    // second piece of waves 2 / 3 at BN 96 goes to the dump area.  wslot = slot of the next half tile to REQUEST, rslot = slot of the next to READ.
    const unsigned ldsw = (unsigned)(PERS ? (wave & 1) : wave) * 1024u;
    const bool dump1 = HAS_DUMP && wave >= 2;
    const unsigned voff1 = NBI > 1 ? (dump1 ? wvoff[0] : wvoff[NBI - 1]) : 0u;
    unsigned wslot = 0, rslot = 0;
#define TDW_SLOT_NEXT(X) { X += H_BYTES; if (X == RING * H_BYTES) X = 0; }
    // half tile T (0 .. 20; 18 .. 20 = taps 0 .. 2 of the NEXT unit) of the unit at wchunk
#define TDW_DMA(TILE)                                                                                        \
    if (w_role) {                                                                                            \
        if ((TILE) == 9) wnext = wchunk;                                                                     \
        if ((TILE) == 18) { if (PERS && wrap) wnext = k_wpack + (size_t)pco0 * 128; else wnext += wstep; wchunk = wnext; } \
        const unsigned x_ = ((TILE) >= 9 && (TILE) < 18) ? 64u : 0u;                                         \
        if constexpr (PERS) {                                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) { const unsigned v_ = wvoff[i_] ^ x_; const unsigned m_ = ldsw + wslot + (unsigned)(i_ * PSTR); TD_GLDS16G(v_, wnext, m_, 0); } \
        } else {                                                                                             \
            { const unsigned v_ = wvoff[0] ^ x_; const unsigned m_ = ldsw + wslot; TD_GLDS16(v_, wnext, m_, 0); } \
            if constexpr (NBI > 1) { const unsigned v_ = voff1 ^ x_; const unsigned m_ = dump1 ? (unsigned)DUMP_BASE + ldsw - 2048u : ldsw + wslot + 4096u; TD_GLDS16(v_, wnext, m_, 0); } \
        }                                                                                                    \
        TDW_SLOT_NEXT(wslot)                                                                                 \
        if ((TILE) != 8 && (TILE) != 17) wnext += wstep;                                                     \
    }
    { constexpr bool wrap = false; (void)wrap; TDW_DMA(0); TDW_DMA(1); TDW_DMA(2); }
    const bool cv_stage = k_epi == EPI_EMB_SILU && tid < BN;
    float cv_val = 0.f;
    if (cv_stage) {
        if (n0 < k_N && co0 + tid < k_Cout) cv_val = p.cvec[(size_t)n0 * k_cvs + co0 + tid];
    }

    // ---- activation-patch staging: 16-byte pieces (patch pixel e >> 2, piece e & 3 of the 32-channel half), one K-group ahead through registers,
    // transformed on the way into LDS.  aoff = element offset of the thread's piece in the current segment's source, or -1 (zero fill).
    u32x4 av[A_ITERS];
    int aoff[A_ITERS];
    const T* seg_src = nullptr;
    int seg_taps = 9, seg_xform = 0, seg_nchunks = 0;
    float seg_scale = 1.f;
#define TDW_SEG_BEGIN(SEG)                                                                                            \
    {                                                                                                                 \
        TDW_FRESH()                                                                                                   \
        seg_src = (const T*)KA(seg[SEG].src); seg_taps = KA(seg[SEG].taps); seg_xform = KA(seg[SEG].xform); seg_scale = KA(seg[SEG].scale); seg_nchunks = KA(seg[SEG].C) / CHUNK; \
        const int Hs_ = KA(seg[SEG].Hs), Ws_ = KA(seg[SEG].Ws), rs_ = KA(seg[SEG].resample), cs_ = KA(seg[SEG].cstride); \
        if (p_role) { _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                     \
            int e_ = pt + it_ * PNT;                                                                                  \
            asm volatile("" : "+v"(e_));   /* (the patch coordinates are re-derived per segment: nothing of them lives across the K loop) */ \
            const int pp_ = e_ >> 2, py_ = pp_ / PW, px_ = pp_ - py_ * PW;                                            \
            const int y_ = py0 + py_ - 1, x_ = px0 + px_ - 1;                                                         \
            const bool ok_ = pp_ < NPATCH && pn0 < k_N && y_ >= 0 && y_ < k_H && x_ >= 0 && x_ < k_W;                 \
            const bool in_ = py_ >= 1 && py_ <= TH && px_ >= 1 && px_ <= TW;                                          \
            aoff[it_] = -1;                                                                                           \
            if (ok_ && (seg_taps == 9 || in_)) aoff[it_] = src_pixel(pn0, y_, x_, Hs_, Ws_, rs_) * cs_ + (tid & 3) * PER16; \
        } }                                                                                                           \
    }
    // Always issued (offset 0 for zero-fill pieces) so that the vmcnt bookkeeping of the loop is exact -- and issued by INLINE ASM: hipcc's wait-count pass
    // does not see these loads, so it never guards their registers with waits of its own (with two half tiles always in flight, its `vmcnt(5 ... 0)` in
    // front of the LDS writes would drain the weight stream); the loop's counted waits cover them, and TDW_PIN_A -- placed behind such a wait -- is the
    // point from which the compiler may use the values.
#define TDW_LOAD_A(CH, HF)                                                                             \
    {                                                                                                  \
        const T* src_ = seg_src + (CH) * CHUNK + (HF) * HALF;                                          \
        if constexpr (PERS) src_ = td_uniform_ptr(src_);   /* (hipcc keeps the segment pointer of the tile loop in a VGPR pair and hands THAT to the "s" operand) */ \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const unsigned o_ = (unsigned)(aoff[it_] >= 0 ? aoff[it_] : 0) * (unsigned)sizeof(T);      \
            if constexpr (PERS) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(av[it_]) : "v"(o_), "s"(src_) : "memory");   /* (guarded: conv_common.h, TD_GLDS16G) */ \
            else asm volatile(TD_SGPR_GUARD_LOAD "global_load_dwordx4 %0, %1, %2" : "=&v"(av[it_]) : "v"(o_), "s"(src_) : "memory"); \
        }                                                                                              \
    }
#define TDW_PIN_R(I0, I1) { _Pragma("unroll") for (int it_ = (I0); it_ < (I1); ++it_) asm volatile("" : "+v"(av[it_])); }
#define TDW_PIN_A() TDW_PIN_R(0, A_ITERS)
    // (ONE address register for the six pieces: piece it_ of thread tid is patch pixel (tid >> 2) + 64 it_, 16-byte slot tid & 3 -- written out so, because left
    // to itself hipcc kept six piece addresses live across the K loop, spilled two of them, and reloaded them from scratch inside the loop behind a vmcnt(0))
    const unsigned st_base = (unsigned)A_BASE + (unsigned)(pt >> 2) * PITCH + (unsigned)((pt & 3) << 4);
#define TDW_STORE_A(BUF, RNP) TDW_STORE_R(BUF, RNP, 0, A_ITERS)
#define TDW_STORE_R(BUF, RNP, I0, I1)                                                                  \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = (I0); it_ < (I1); ++it_) {                                    \
            if (it_ < A_FULL || pt < A_REM) {                                                          \
                u32x4 v_ = aoff[it_] >= 0 ? av[it_] : u32x4{0u, 0u, 0u, 0u};                           \
                if (seg_xform != 0 && aoff[it_] >= 0) {                                \
                    float s_ = seg_scale;                                                              \
                    if (seg_xform == 2) s_ *= (RNP)[(pt >> 2) + it_ * (PNT / 4)];                      \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(smem + st_base + ((BUF) * A_BYTES + it_ * (PNT / 4) * PITCH)) = v_;          \
            }                                                                                          \
        }                                                                                              \
    }
    // residual runs of the wide epilogue: the first A_ITERS of the MT * NU units are requested during the LAST 3x3 unit into `av` (dead there), the rest
    // at the top of the epilogue (conv_glds.hip round 5)
    const bool r_want = !TAIL && k_epi == EPI_RESIDUAL && k_hres;
    bool r_pref = false;
    const T* r_ptr[MT];
#define TDW_R_ADDR()                                                                                                  \
    {                                                                                                                 \
        TDW_FRESH()                                                                                                   \
        const int rHs_ = KA(res_Hs), rWs_ = KA(res_Ws), rrs_ = KA(res_resample), rcs_ = KA(res_cstride);              \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                                           \
            int img_, ty_, tx_;                                                                                       \
            frag_pixel<TW, TPIX>(wm * WM + i_ * 32, l31, img_, ty_, tx_);                                             \
            const int y_ = y0 + ty_, x_ = x0 + tx_;                                                                   \
            const int sp_ = (n0 < k_N && y_ < k_H && x_ < k_W) ? src_pixel(n0, y_, x_, rHs_, rWs_, rrs_) : 0;         \
            r_ptr[i_] = (const T*)KA(res) + (sp_ * rcs_ + co0 + 8 * lh);                                              \
        }                                                                                                             \
    }
#define TDW_R_UNIT(Q) (*(const u32x4*)(r_ptr[(Q) / NU] + ((co0 + (((Q) % NU) >> 1) * 32 < k_Cout) ? (((Q) % NU) >> 1) * 32 + ((Q) & 1) * 16 : 0)))
// (PERS: by inline asm like the patch loads -- loads the compiler knows of and cannot prove consumed on every path to the loop's back edge make it guard the next
// tile's first patch loads, which overwrite `av`, with waits of its own: `vmcnt(1)` in the middle of half-step 0 = a whole HBM round trip per tile)
#define TDW_LOAD_R()                                                                                                  \
    {                                                                                                                 \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                                   \
            if constexpr (PERS) { const u32x4* q_ = &TDW_R_UNIT(it_ < MT * NU ? it_ : MT * NU - 1); asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(av[it_]) : "v"(q_) : "memory"); } \
            else av[it_] = TDW_R_UNIT(it_ < MT * NU ? it_ : MT * NU - 1);                                             \
        }                                                                                                             \
        r_pref = true;                                                                                                \
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) r_ptr[i] = nullptr;
    if (r_want && !PERS) TDW_R_ADDR()   // (PERS: made per tile at half-step 17 of its last unit)
    TDW_SEG_BEGIN(0);
    if (p_role) TDW_LOAD_A(0, 0);

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source, for the patch pixels
#define TDW_RN_SETUP()                                                                                       \
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f; \
    if (KA(seg[0].xform) == 2) { rn_sumsq = KA(seg[0].sumsq); rn_parts = KA(seg[0].nparts); rn_Hs = KA(seg[0].Hs); rn_Ws = KA(seg[0].Ws); rn_res = KA(seg[0].resample); rn_invc = KA(seg[0].inv_c); } \
    else if (KA(res_sumsq)) { rn_sumsq = KA(res_sumsq); rn_parts = KA(res_nparts); rn_Hs = KA(res_Hs); rn_Ws = KA(res_Ws); rn_res = KA(res_resample); rn_invc = KA(res_inv_c); }
    const bool rn_any = p.seg[0].xform == 2 || p.res_sumsq != nullptr;
    {
    TDW_RN_SETUP()
#define TDW_RN_FILL(DST, N0_, Y0_, X0_)                                                                      \
    if (rn_sumsq) {                                                                                          \
        const size_t npix_ = (size_t)k_N * rn_Hs * rn_Ws;                                                    \
        for (int pp_ = tid; pp_ < NPATCH; pp_ += NTHR) {                                                     \
            const int py_ = pp_ / PW, px_ = pp_ % PW;                                                        \
            const int y_ = (Y0_) + py_ - 1, x_ = (X0_) + px_ - 1;                                            \
            float rn_ = 0.f;                                                                                 \
            if ((N0_) < k_N && y_ >= 0 && y_ < k_H && x_ >= 0 && x_ < k_W)                                   \
                rn_ = pixel_rn(rn_sumsq, rn_parts, npix_, src_pixel(N0_, y_, x_, rn_Hs, rn_Ws, rn_res), rn_invc); \
            (DST)[pp_] = rn_;                                                                                \
        }                                                                                                    \
    }
    TDW_RN_FILL(s_rn, n0, y0, x0)
    }
    if (cv_stage) s_cv[tid] = cv_val;
    // PERS: the next tile's modulation row goes global -> LDS by a one-dword LDS-DMA of the weight waves at EVERY half-step 9 that has a successor (counted in
    // the loop's waits; lanes past the cout tile re-read a valid word).  Not through a register: hipcc copied the asm-loaded register at the next control-flow
    // merge, before the data had landed (first build of the role split: wrong modulation on every tile but a workgroup's first).
    const int cv_mul = k_epi == EPI_EMB_SILU ? k_cvs : 0;

    // ---- MFMA operand addressing: weights = A operand (rows = couts), activations = B operand (cols = pixels)
    unsigned xbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
        xbase[i] = (unsigned)A_BASE + (unsigned)(ty * PW + tx) * PITCH + (unsigned)lh * 16u;   // top-left tap of the lane's pixel
    }
    unsigned wbase[2];   // k-step (0 / 1) of the half: piece (2 ks + lh) ^ ((row >> 2) & 3) of this lane's row; the 32-row block j and the slot are compile-time
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wbase[ks] = (unsigned)(l31 * 64 + (((ks * 2 + lh) ^ ((l31 >> 2) & 3)) << 4));
    __syncthreads();  // s_rn visible
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first patch (and the three half tiles requested in front of it)
    if (p_role) { TDW_PIN_A() TDW_STORE_A(0, s_rn); }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TDW_T(tr_pro);

#define TDW_TOFF(TP) ((((TP) / 3) * PW + ((TP) % 3)) * PITCH)
    u32x4 wfA_[NT], xfA_[MT], wfB_[NT], xfB_[MT];
    // fragments of k-step KS (0 / 1) of the half K-step in ring slot SLOT, patch buffer BUF, tap offset TOFF
#define TDW_FRAG_READ(WF, XF, SLOT, KS, BUF, TOFF)                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(smem + (wbase[KS] + (SLOT)) + j_ * 2048);   /* SLOT: wave-uniform byte offset of the ring slot */ \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(smem + xbase[i_] + ((BUF) * A_BYTES + (TOFF) + (KS) * 32)); \
    }
#define TDW_FRAG_MFMA(WF, XF)                                                                                \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                acc[i_][j_] = Half<T>::mfma32(__builtin_bit_cast(hx8, WF[j_]), __builtin_bit_cast(hx8, XF[i_]), acc[i_][j_]); \
    }
    // One half K-step S (0 .. 17) of a 64-channel unit: channel half S / 9 (= patch buffer), tap S % 9, ring slot S % 3; its fragments were requested
    // during the previous half-step.  Passing its barrier: half tile S + 1 is visible (every wave waited for its own pieces), nobody reads half tile
    // S - 1 any more (its slot takes S + 2), and the patch written two half-steps ago is visible.
    //   S = 0: the patch of channel half 1 of this unit is requested (always exists); pinned at S = 2 (the loads are complete there: see the waits),
    //          written into buffer 1 at S = 5 (its last readers -- the previous unit's half-steps 9 .. 17 -- passed this unit's barrier 1);
    //   S = 9: the patch of half 0 of the NEXT unit (same segment, or the next 3x3 segment: `nxt` / `newseg`), written into buffer 0 at S = 14;
    //   S = 17 requests the first fragments of the next unit (slot 0, buffer 0).
    // Waits: in flight at the barrier of S are the pieces of S + 1 (issued one half-step ago) and, younger, the patch loads (issued at S - 1 = 0 / 9:
    // counted past; complete one step later) or, in the last unit, the residual runs (issued at S = 15: counted past at 16; S = 17 needs nothing that
    // was issued after tile 17 -- tile 18 belongs to a unit that does not exist).
#define TDW_HS(S)                                                                                            \
    {                                                                                                        \
        TDW_T(tA_);                                                                                          \
        /* in flight, oldest first: half tile S + 1 (needed now), [patch loads of S - 2], half tile S + 2, [patch loads of S - 1] */ \
        if ((S) == 1 || (S) == 2 || (((S) == 10 || (S) == 11) && nxt)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS + NBI) : "memory"); \
        else if ((S) >= 16 && !nxt) { if (r_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                      \
        if ((S) == 6 || (S) == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave's patch writes of the previous half-step are out */ \
        TDW_T(tM_); TDW_TACC(tr_stage, tA_, tM_);   /* (trace builds: the share of the wait that is this wave's own DMA pieces landing; the rest is the barrier) */ \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        TDW_T(tB_); TDW_TACC(tr_wait, tA_, tB_);                                                             \
        if ((S) == 3 || ((S) == 12 && nxt)) TDW_PIN_A()   /* (the wait above completed the patch loads) */ \
        if ((S) + 3 < 18 || nxt) TDW_DMA((S) + 3);   /* (nothing is requested past the end of the launch's 3x3 part) */ \
        if ((S) == 0) TDW_LOAD_A(chunk, 1);                                                                  \
        if ((S) == 9 && nxt) { if (newseg) TDW_SEG_BEGIN(seg + 1); TDW_LOAD_A(newseg ? 0 : chunk + 1, 0); }  \
        if ((S) == 15 && r_now) TDW_LOAD_R()                                                                 \
        if ((S) == 5) TDW_STORE_A(1, s_rn);                                                                        \
        if ((S) == 14 && nxt) TDW_STORE_A(0, s_rn);                                                                \
        TDW_FRAG_MFMA(wfA_, xfA_);                                                                           \
        if ((S) < 17 || nxt) TDW_FRAG_READ(wfA_, xfA_, rslot, 0, (((S) + 1) / 9) & 1, TDW_TOFF(((S) + 1) % 9)); \
        TDW_FRAG_MFMA(wfB_, xfB_);                                                                           \
        if ((S) < 17 || nxt) { TDW_FRAG_READ(wfB_, xfB_, rslot, 1, (((S) + 1) / 9) & 1, TDW_TOFF(((S) + 1) % 9)); TDW_SLOT_NEXT(rslot) } \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
    }
// The half-step of the PERSISTENT instantiation (roles split by wave, see A_ITERS above).  Weight waves: half tile S + 3 requested in half-step S, waited for at
// the top of S + 2 (in flight there: S + 1 -- needed now -- and S + 2; at 10 / 11 also the next tile's modulation word).  Patch waves: channel half 1 of the
// unit requested at S = 0, the first channel half of the successor unit (next unit of this tile, or the first unit of the workgroup's next tile) at S = 9;
// their pieces return in order and are written in three batches -- 5 / 6 / 7 and 14 / 15 / 16 -- so that the transform + ds_write work of the two patch waves
// does not sit in ONE half-step; the last batch is visible from the barrier of 8 / 17, whose half-step reads the first fragments of that buffer.  The residual
// runs of the epilogue are requested at 17 (`av` is free from 16).
#define TDW_B0 4
#define TDW_B1 8
#define TDW_HSP(S)                                                                                           \
    {                                                                                                        \
        if (w_role) {                                                                                        \
            if ((S) >= 16 && !nxt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          \
            else if (((S) == 10 || (S) == 11) && nxt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI + 1) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                  \
        } else {                                                                                             \
            if ((S) == 5 || ((S) == 14 && nxt)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS - TDW_B0) : "memory"); \
            if ((S) == 6 || ((S) == 15 && nxt)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS - TDW_B1) : "memory"); \
            if ((S) == 7 || ((S) == 16 && nxt)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             \
        }                                                                                                    \
        if (((S) >= 6 && (S) <= 8) || ((S) >= 15 && (S) <= 17)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave's patch writes of the previous half-step are out */ \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if (p_role) {                                                                                        \
            if ((S) == 5) { TDW_PIN_R(0, TDW_B0) TDW_STORE_R(1, s_rn, 0, TDW_B0) }                           \
            if ((S) == 6) { TDW_PIN_R(TDW_B0, TDW_B1) TDW_STORE_R(1, s_rn, TDW_B0, TDW_B1) }                 \
            if ((S) == 7) { TDW_PIN_R(TDW_B1, A_ITERS) TDW_STORE_R(1, s_rn, TDW_B1, A_ITERS) }               \
            if (((S) == 14 || (S) == 15 || (S) == 16) && nxt) {                                              \
                constexpr int i0_ = (S) == 14 ? 0 : (S) == 15 ? TDW_B0 : TDW_B1, i1_ = (S) == 14 ? TDW_B0 : (S) == 15 ? TDW_B1 : A_ITERS; \
                TDW_PIN_R(i0_, i1_)                                                                          \
                if (wrap) TDW_STORE_R(0, (const float*)(smem + RN_BASE + (par ^ 1) * RN_BYTES), i0_, i1_) else TDW_STORE_R(0, s_rn, i0_, i1_) \
            }                                                                                                \
        }                                                                                                    \
        if ((S) == 9 && wrap) {   /* the successor tile: its coordinates, its pixel-norm factors (plain loads: the compiler's own wait drains the wave's queue once per tile) */ \
            TDW_FRESH()                                                                                      \
            { const unsigned k_d1 = kp_->sb_d1, k_m1 = kp_->sb_m1, k_m2 = kp_->sb_m2, k_m3 = kp_->sb_m3, k_g8 = kp_->sb_grid8; \
              const int k_tx = kp_->tiles_x, k_ty = kp_->tiles_y, k_rev = kp_->reverse;                      \
              TDW_DECODE(vb + vstep, pn0, py0, px0, pco0) }                                                  \
            if (rn_any) { TDW_RN_SETUP() TDW_RN_FILL((float*)(smem + RN_BASE + (par ^ 1) * RN_BYTES), pn0, py0, px0) } \
        }                                                                                                    \
        if ((S) + 3 < 18 || nxt) TDW_DMA((S) + 3);                                                           \
        if ((S) == 0 && p_role) TDW_LOAD_A(chunk, 1);                                                        \
        if ((S) == 9 && nxt) {                                                                               \
            if (wrap) TDW_SEG_BEGIN(0) else if (newseg) TDW_SEG_BEGIN(seg + 1);                              \
            if (p_role) TDW_LOAD_A((newseg || wrap) ? 0 : chunk + 1, 0)                                      \
            else {                                                                                           \
                TDW_FRESH()                                                                                  \
                const float* cv_src_ = k_epi == EPI_EMB_SILU ? kp_->cvec : (const float*)k_wpack;            \
                const float* cs_ = td_uniform_ptr(cv_src_ + ((size_t)pn0 * cv_mul + pco0));                  \
                const unsigned co_ = (unsigned)(tid < BN ? tid : 0) * 4u;                                    \
                const unsigned m_ = (unsigned)CV_BASE + (unsigned)((par ^ 1) * CV_BYTES) + (unsigned)(wave & 1) * 256u; \
                TD_GLDS4(co_, cs_, m_, 0);                                                                   \
            }                                                                                                \
        }                                                                                                    \
        if ((S) == 17 && r_now) { TDW_R_ADDR() TDW_LOAD_R() }                                                \
        TDW_FRAG_MFMA(wfA_, xfA_);                                                                           \
        if ((S) < 17 || nxt) TDW_FRAG_READ(wfA_, xfA_, rslot, 0, (((S) + 1) / 9) & 1, TDW_TOFF(((S) + 1) % 9)); \
        TDW_FRAG_MFMA(wfB_, xfB_);                                                                           \
        if ((S) < 17 || nxt) { TDW_FRAG_READ(wfB_, xfB_, rslot, 1, (((S) + 1) / 9) & 1, TDW_TOFF(((S) + 1) % 9)); TDW_SLOT_NEXT(rslot) } \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
    }
    int seg = 0, chunk = 0;
    if (n3 > 0) {
        // entry of the pipeline: this wave's patch writes are out, half tiles 0 and 1 were issued a prologue ago
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBI) : "memory");   // half tile 0 (1 and 2 stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TDW_FRAG_READ(wfA_, xfA_, rslot, 0, 0, TDW_TOFF(0));
        TDW_FRAG_READ(wfB_, xfB_, rslot, 1, 0, TDW_TOFF(0));
        TDW_SLOT_NEXT(rslot)
    }
    const bool no_tail = n3 == p.kgroups;
#ifdef TD_TRACE
    unsigned long long tr_loop = 0;
#endif
    for (;;) {   // the tiles of this workgroup: one, unless PERS
    const bool has_next = PERS && vb + vstep < k_grid;
    if (n3 > 0) {
        for (int u = 0; u < n3; ++u) {
            const bool last = u + 1 == n3;
            const bool wrap = PERS && last && has_next;   // the successor of this unit is the first unit of the workgroup's next tile
            const bool nxt = !last || wrap;
            const bool newseg = !last && chunk + 1 == seg_nchunks;
            const bool r_now = r_want && last && no_tail;   // last unit of a tile of a launch without a 1x1 tail
            if constexpr (PERS) {
                TDW_HSP(0); TDW_HSP(1); TDW_HSP(2); TDW_HSP(3); TDW_HSP(4); TDW_HSP(5); TDW_HSP(6); TDW_HSP(7); TDW_HSP(8);
                TDW_HSP(9); TDW_HSP(10); TDW_HSP(11); TDW_HSP(12); TDW_HSP(13); TDW_HSP(14); TDW_HSP(15); TDW_HSP(16); TDW_HSP(17);
            } else {
                TDW_HS(0); TDW_HS(1); TDW_HS(2); TDW_HS(3); TDW_HS(4); TDW_HS(5); TDW_HS(6); TDW_HS(7); TDW_HS(8);
                TDW_HS(9); TDW_HS(10); TDW_HS(11); TDW_HS(12); TDW_HS(13); TDW_HS(14); TDW_HS(15); TDW_HS(16); TDW_HS(17);
            }
            if (wrap) { seg = 0; chunk = 0; } else if (newseg) { ++seg; chunk = 0; } else ++chunk;
        }
    }
    // ---- 1x1 tail (the fused skip conv of the decoder's conv_res1, pure 1x1 convs): centre tap, one half K-step per (64-channel chunk, channel half).
    // Untransformed sources (every 1x1 segment of the U-Net; p.dma1x1 = the launcher's verdict): BOTH operands by LDS-DMA.  The 32 channels of the
    // tile's 256 pixels (16 KB, 64-byte rows in MFMA column order, pieces swizzled like the weight rows) go into one of THREE stage buffers laid over
    // the dead halo patches, two half-steps ahead, with the weight half tile of the same step behind them in the in-order vmcnt queue; one barrier
    // per half-step, fragment reads not carried across it (the stage of step t + 1 may still be landing).  An operand without tap reuse costs six DMA
    // pieces per 12 MFMAs of a wave in any structure: ~1.6x a 3x3 half-step.
    // Transformed 1x1 sources: the unpipelined register path below (correct and slow; no such launch exists in the U-Net).
    if constexpr (TAIL) {
    if (n3 < p.kgroups && p.dma1x1) {
        constexpr int STAGE = TPIX * 64, NSTG = 3;
        static_assert(NSTG * STAGE <= 2 * A_BYTES, "stage buffers over the two halo patches");
        int pseg = n3 > 0 ? seg + 1 : 0, pchunk = 0, phf = 0, pn = 1;
        const T* psrc = nullptr;
        unsigned poff[4];
#define TDW_P_BEGIN()                                                                                                 \
        {                                                                                                             \
            const ConvSeg& sg_ = p.seg[pseg];                                                                         \
            psrc = (const T*)sg_.src; pn = sg_.C / CHUNK;                                                             \
            const int Hs_ = sg_.Hs, Ws_ = sg_.Ws, rs_ = sg_.resample, cs_ = sg_.cstride;                              \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
                const int e_ = tid + i_ * NTHR, row_ = e_ >> 2, j_ = (e_ & 3) ^ ((row_ >> 2) & 3);                    \
                int img_, ty_, tx_;                                                                                   \
                frag_pixel<TW, TPIX>(row_ & ~31, row_ & 31, img_, ty_, tx_);                                          \
                const int y_ = y0 + ty_, x_ = x0 + tx_;                                                               \
                /* a pixel outside the image is an MFMA column nobody stores: any readable address will do */         \
                const int pix_ = (n0 < k_N && y_ < k_H && x_ < k_W) ? src_pixel(n0, y_, x_, Hs_, Ws_, rs_) : 0;       \
                poff[i_] = (unsigned)(pix_ * cs_ + j_ * PER16) * (unsigned)sizeof(T);                                 \
            }                                                                                                         \
        }
        unsigned sbuf_w = 0, sbuf_r = 0;   // stage buffer (byte offset) of the next stage to request / to read
        const unsigned char* w1 = k_wpack + (size_t)co0 * 128 + (size_t)n3 * 9 * wstep;   // K-step of the half tile requested next
        int w1h = 0;
        // stage of the issue cursor + the weight half tile of the same step; the cursor stops on the last step (the loop issues unconditionally)
#define TDW_P_ISSUE()                                                                                                 \
        {                                                                                                             \
            const unsigned long long sa_ = (unsigned long long)(psrc + (size_t)pchunk * CHUNK + phf * HALF);          \
            const unsigned char* su_ = td_uniform_ptr((const unsigned char*)sa_);                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { const unsigned m_ = (unsigned)A_BASE + sbuf_w + ldsw + (unsigned)i_ * 4096u; TD_GLDS16(poff[i_], su_, m_, 0); } \
            sbuf_w += STAGE; if (sbuf_w == NSTG * STAGE) sbuf_w = 0;                                                  \
            { const unsigned x_ = w1h ? 64u : 0u;                                                                     \
              { const unsigned v_ = wvoff[0] ^ x_; const unsigned m_ = ldsw + wslot; TD_GLDS16(v_, w1, m_, 0); }      \
              if constexpr (NBI > 1) { const unsigned v_ = voff1 ^ x_; const unsigned m_ = dump1 ? (unsigned)DUMP_BASE + ldsw - 2048u : ldsw + wslot + 4096u; TD_GLDS16(v_, w1, m_, 0); } } \
            TDW_SLOT_NEXT(wslot)                                                                                      \
            if (w1h) w1 += wstep;                                                                                     \
            w1h ^= 1;                                                                                                 \
            if (phf == 0) phf = 1;                                                                                    \
            else if (pchunk + 1 < pn) { phf = 0; ++pchunk; }                                                          \
            else if (pseg + 1 < k_nseg) { phf = 0; pchunk = 0; ++pseg; TDW_P_BEGIN(); }                               \
        }
        int nt1 = 0;   // half-steps of the tail
        for (int sg = pseg; sg < k_nseg; ++sg) nt1 += 2 * (p.seg[sg].C / CHUNK);
        TDW_P_BEGIN();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave is done with the halo patches and the ring
        asm volatile("" ::: "memory");
        wslot = 0; rslot = 0;
        TDW_P_ISSUE();
        TDW_P_ISSUE();
        unsigned xb1[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xb1[ks] = (unsigned)A_BASE + (unsigned)((wm * WM + l31) * 64 + (((ks * 2 + lh) ^ ((l31 >> 2) & 3)) << 4));
        for (int t = 0; t < nt1; ++t) {
            // in flight, oldest first: stage t, half tile t, stage t + 1, half tile t + 1
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NBI) : "memory");
            __builtin_amdgcn_s_barrier();   // stage t and half tile t are visible; nobody reads stage t - 1 / half tile t - 1 any more
            asm volatile("" ::: "memory");
            TDW_P_ISSUE();
#pragma unroll
            for (int j_ = 0; j_ < NT; ++j_) wfA_[j_] = *(const u32x4*)(smem + (wbase[0] + rslot) + j_ * 2048);
#pragma unroll
            for (int i_ = 0; i_ < MT; ++i_) xfA_[i_] = *(const u32x4*)(smem + (xb1[0] + sbuf_r) + i_ * 2048);
#pragma unroll
            for (int j_ = 0; j_ < NT; ++j_) wfB_[j_] = *(const u32x4*)(smem + (wbase[1] + rslot) + j_ * 2048);
#pragma unroll
            for (int i_ = 0; i_ < MT; ++i_) xfB_[i_] = *(const u32x4*)(smem + (xb1[1] + sbuf_r) + i_ * 2048);
            TDW_SLOT_NEXT(rslot)
            sbuf_r += STAGE; if (sbuf_r == NSTG * STAGE) sbuf_r = 0;
            TDW_FRAG_MFMA(wfA_, xfA_);
            TDW_FRAG_MFMA(wfB_, xfB_);
        }
#undef TDW_P_ISSUE
#undef TDW_P_BEGIN
    } else if (n3 < p.kgroups) {   // (register-staged, unpipelined)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two half tiles fetched past the end of the 3x3 part
        const unsigned char* w1 = k_wpack + (size_t)co0 * 128 + (size_t)n3 * 9 * wstep;
        int s1 = n3 == 0 ? 0 : (chunk == 0 && seg > 0 ? seg : seg + 1);
        if (n3 > 0 && !(chunk == 0 && seg > 0)) s1 = seg + 1;
        if (n3 == 0) s1 = 0;
        for (int sg = s1; sg < k_nseg; ++sg) {
            TDW_SEG_BEGIN(sg);
            for (int ch = 0; ch < seg_nchunks; ++ch, w1 += wstep) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    TDW_LOAD_A(ch, hf);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // every wave is done with patch buffer 0 and ring slot 0
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    TDW_PIN_A()
                    TDW_STORE_A(0, s_rn);
                    { const unsigned x_ = hf ? 64u : 0u;
                      { const unsigned v_ = wvoff[0] ^ x_; TD_GLDS16(v_, w1, ldsw, 0); }
                      if constexpr (NBI > 1) { const unsigned v_ = voff1 ^ x_; const unsigned m_ = dump1 ? (unsigned)DUMP_BASE + ldsw - 2048u : ldsw + 4096u; TD_GLDS16(v_, w1, m_, 0); } }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    TDW_FRAG_READ(wfA_, xfA_, 0u, 0, 0, TDW_TOFF(4));
                    TDW_FRAG_READ(wfB_, xfB_, 0u, 1, 0, TDW_TOFF(4));
                    TDW_FRAG_MFMA(wfA_, xfA_);
                    TDW_FRAG_MFMA(wfB_, xfB_);
                }
            }
        }
    }
    }
    if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-fetched tail tiles must not land in a successor's LDS


#ifdef TD_TRACE
    tr_loop = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // ---------------- epilogue (conv_glds.hip's wide / narrow paths; no split-K here)
    TDW_FRESH()
    int e_Cout = KA(Cout), e_epi = KA(epi), e_ocs = KA(out_cstride);
    float e_rsc = KA(res_scale), e_clip = KA(clip), e_o2s = KA(out2_scale);
    const bool e_hres = KA(res) != nullptr, e_hrss = KA(res_sumsq) != nullptr, e_hoss = KA(out_sumsq) != nullptr, e_ho2 = KA(out2) != nullptr;
    asm volatile("" : "+s"(e_Cout), "+s"(e_epi), "+s"(e_ocs), "+s"(e_rsc), "+s"(e_clip), "+s"(e_o2s));
    const size_t M = (size_t)k_N * k_H * k_W;
    // (bf16 / fp16 output with Cout % 8 == 0 only -- the dwordx4 store path; the launcher refuses everything else: fp32 outputs and the solver-step epilogue
    // of the few-channel output convs stay on conv_glds.hip)
    const bool has_res = e_epi == EPI_RESIDUAL && e_hres;
    constexpr int NRX = MT * NU > A_ITERS ? MT * NU - A_ITERS : 1;
    u32x4 rx[NRX];
#pragma unroll
    for (int q = 0; q < NRX; ++q) rx[q] = u32x4{0u, 0u, 0u, 0u};
    if (has_res) {
        if constexpr (PERS) {   // the residual runs of half-step 17: the youngest requests of the wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TDW_PIN_A()
        }
        if (!r_want) TDW_R_ADDR()   // (the TAIL instantiation: the run addresses are made here, not carried across the K loop)
        if (!r_pref) TDW_LOAD_R()   // the launch ended in 1x1 K-groups: nothing was requested yet
        if constexpr (MT * NU > A_ITERS) {
#pragma unroll
            for (int q = 0; q < NRX; ++q) rx[q] = TDW_R_UNIT(A_ITERS + q);
        }
    }
    // (one instance per 32-pixel row group, by hand: left to `#pragma unroll` the loop stayed rolled -- hipcc gave up on its size -- and the accumulators,
    // the prefetched residual runs and `rx` were then indexed by a run-time `i`, i.e. lived in scratch memory for the WHOLE kernel)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        {
        int img, ty, tx;
        frag_pixel<TW, TPIX>(wm * WM + i * 32, l31, img, ty, tx);
        const int n = n0, y = y0 + ty, x = x0 + tx;
        const bool ok = PERS || (n < k_N && y < k_H && x < k_W);   // (PERS: whole tiles only)
        float ssj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ssj[j] = 0.f;
        if (ok) {
            const float rn = e_hrss ? s_rn[(ty + 1) * PW + (tx + 1)] : 1.f;
            {
                const size_t pix = ((size_t)n * k_H + y) * k_W + x;
                T* orow = (T*)KA(out) + pix * e_ocs + co0 + 8 * lh;
                const float rs = e_rsc * rn;
                const bool want_ss = e_hoss, want_o2 = e_ho2;
                const SiluK k_o2 = silu_k(e_o2s);
                auto body = [&](auto KIND) __attribute__((always_inline)) {
                    constexpr int K = decltype(KIND)::value;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int j = u >> 1, m = u & 1;
                        f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = ca;
                        u32x4 rw = {0u, 0u, 0u, 0u};
                        if constexpr (K == 1) {
                            const float* c_ = s_cv + (u >> 1) * 32 + (u & 1) * 16 + 4 * lh;
                            ca = *(const f32x4*)c_; cb = *(const f32x4*)(c_ + 8);
                        }
                        if constexpr (K == 2) rw = i * NU + u < A_ITERS ? av[i * NU + u < A_ITERS ? i * NU + u : 0] : rx[i * NU + u >= A_ITERS ? i * NU + u - A_ITERS : 0];
                        if (!PERS && co0 + j * 32 >= e_Cout) continue;
                        const f32x4 va = {acc[i][j][8 * m + 0], acc[i][j][8 * m + 1], acc[i][j][8 * m + 2], acc[i][j][8 * m + 3]};
                        const f32x4 vb = {acc[i][j][8 * m + 4], acc[i][j][8 * m + 5], acc[i][j][8 * m + 6], acc[i][j][8 * m + 7]};
                        u32x4 o, o2;
                        epi_unit8<T>(e_epi, has_res, e_clip, want_ss, want_o2, va, vb, ca, cb, rw, rs, k_o2, o, o2, ssj[j]);
#if defined(TDW_EPI_EXP) && TDW_EPI_EXP == 1      /* experiment (tools/conv_bench.hip only): no stores at all */
                        if (o[0] == 0x12345678u && o2[1] == 0x9abcdef0u) *(u32x4*)(orow + j * 32 + m * 16) = o;
#elif defined(TDW_EPI_EXP) && TDW_EPI_EXP == 2    /* experiment: the same bytes to WRONG, wave-linear addresses -- every store instruction covers 1 KB of whole lines */
                        { T* lin_ = (T*)p.out + (((size_t)blockIdx.x * 4 + wave) * (MT * NU) + (i * NU + u)) * 512 + lane * 8;
                          *(u32x4*)lin_ = o;
                          if (want_o2) *(u32x4*)((T*)p.out2 + (lin_ - (T*)p.out)) = o2; }
#else
                        *(u32x4*)(orow + j * 32 + m * 16) = o;
                        if (want_o2) *(u32x4*)((T*)KA(out2) + (orow - (T*)KA(out)) + j * 32 + m * 16) = o2;
#endif
                    }
                };
                if (e_epi == EPI_EMB_SILU) body(std::integral_constant<int, 1>{});
                else if (has_res) body(std::integral_constant<int, 2>{});
                else body(std::integral_constant<int, 0>{});
            }
        }
        if (e_hoss) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float ss = ssj[j] + __shfl_xor(ssj[j], 32);
                if (ok && lh == 0 && co0 + j * 32 < k_cpad) {
                    const size_t pix = ((size_t)n * k_H + y) * k_W + x;
                    KA(out_sumsq)[(size_t)(co0 / 32 + j) * M + pix] = ss;
                }
            }
        }
        }
    }
    if constexpr (!PERS) break;
    else {
        if (!has_next) break;
        // ---- next tile: its patch (buffer 0), its per-tile LDS state (set par ^ 1) and its first three weight half tiles are in place or in flight
        vb += vstep; n0 = pn0; y0 = py0; x0 = px0; co0 = pco0; par ^= 1;
        s_rn = (float*)(smem + RN_BASE + par * RN_BYTES); s_cv = (float*)(smem + CV_BASE + par * CV_BYTES);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    }   // tiles
#undef TDW_TOFF
#undef TDW_HS
#undef TDW_HSP
#undef TDW_B0
#undef TDW_B1
#undef TDW_PIN_R
#undef TDW_STORE_R
#undef TDW_FRAG_READ
#undef TDW_FRAG_MFMA
#undef TDW_LOAD_A
#undef TDW_PIN_A
#undef TDW_STORE_A
#undef TDW_SEG_BEGIN
#undef TDW_DMA
#undef TDW_SLOT_NEXT
#undef TDW_DECODE
#undef TDW_RN_SETUP
#undef TDW_FRESH
#undef KA
#undef TDW_RN_FILL
#ifdef TD_TRACE
    TDW_T(tr_eissue);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDW_T(tr_end);
    if (lane == 0) {
        unsigned long long* tb = (unsigned long long*)p.partial + ((size_t)blockIdx.x * 4 + wave) * 16;
        tb[0] = tr_pro - tr_start; tb[1] = tr_loop - tr_pro; tb[2] = tr_end - tr_loop; tb[3] = tr_wait; tb[4] = tr_stage; tb[5] = tr_start; tb[6] = tr_end;
        tb[8] = tr_rt0; tb[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | __builtin_amdgcn_s_getreg((3 << 11) | 20);
        tb[10] = __builtin_amdgcn_s_memrealtime();
        tb[7] = tb[10] - tr_rt0; tb[11] = tr_end - tr_eissue;
    }
#endif
#undef TDW_R_ADDR
#undef TDW_R_UNIT
#undef TDW_LOAD_R
}

template <typename T, int BN, bool TAIL, bool PERS>
static hipError_t launch_glds_wide_cfg2(const ConvParams& p, hipStream_t st) {
    constexpr int NPATCH = 18 * 18;
    constexpr size_t RING_BYTES = 4 * (size_t)(BN * 64) + ((BN * 64) % 4096 ? 2048 : 0);   // four exact slots (+ the dump area of the rowless pieces, BN 96)
    constexpr size_t LDS = RING_BYTES + 2 * (size_t)NPATCH * 80 + (PERS ? 2 : 1) * ((size_t)(NPATCH * 4 + 15) / 16 * 16 + (PERS ? (size_t)512 : (size_t)BN * 4));
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    if (p.ksplit != 1 || p.W < 16 || p.CoutPad % BN || p.nseg < 1 || p.nseg > 3) return hipErrorInvalidValue;
    if (p.out_f32 || (p.Cout & 7) || p.epi == EPI_DPM_STEP) return hipErrorInvalidValue;   // the 16-byte-run epilogue only
    bool seen1 = false;   // 3x3 segments first
    for (int s = 0; s < p.nseg; ++s) { if (p.seg[s].taps == 9 && seen1) return hipErrorInvalidValue; if (p.seg[s].taps != 9) seen1 = true; }
    ConvParams pd = p;
    for (int s = p.nseg; s < 3; ++s) { pd.seg[s].C = 0; pd.seg[s].taps = 0; }   // (the kernel reads all three descriptors in one burst)
    pd.dma1x1 = 1;   // the 1x1 tail by LDS-DMA unless a 1x1 source wants a transform at staging
    for (int s = 0; s < p.nseg; ++s) if (p.seg[s].taps != 9 && p.seg[s].xform != 0) pd.dma1x1 = 0;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups, grid = p.n_ntiles * mtiles;
    if (grid <= 0 || (long long)grid * std::max(mtiles, p.n_ntiles) >= ((long long)1 << 32)) return hipErrorInvalidValue;
    pd.sb_d0 = mtiles; pd.sb_m0 = td_magic(mtiles); pd.sb_d1 = p.n_ntiles; pd.sb_m1 = td_magic(p.n_ntiles); pd.sb_m2 = td_magic(p.tiles_x); pd.sb_m3 = td_magic(p.tiles_y);
    pd.sb_grid = grid; pd.sb_grid8 = (grid & 7) == 0 ? (unsigned)grid >> 3 : 0u;
    int launch_grid = grid;
    if (PERS) {   // p.persist = the number of persistent workgroups = the stride of a workgroup's tile walk (the planner's / the bench's choice)
        if (p.persist <= 0 || p.persist > grid || (p.H & 15) || (p.W & 15) || p.Cout != p.CoutPad || p.kgroups < 1) return hipErrorInvalidValue;
        if (pd.sb_grid8 && (p.persist & 7)) return hipErrorInvalidValue;   // a workgroup's tiles must stay on its XCD's range
        launch_grid = p.persist;
    }
    auto kern = conv_glds_kernel_wide<T, BN, TAIL, PERS>;
    static bool attr_set[64] = {};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    hipLaunchKernelGGL(kern, dim3(launch_grid), dim3(256), LDS, st, pd);
    return hipGetLastError();
}

// Number of persistent workgroups for a launch of `tiles` workgroup tiles on `slots` resident workgroup slots (2 per CU): every workgroup walks
// ceil(tiles / slots) tiles (the last round may be short); a multiple of 8 when the launch's tile count is (XCD ranges).  0 = not worth it (one round).
inline int conv_wide_persist_grid(long long tiles, int slots) {
    if (tiles < 2LL * slots) return 0;
    const long long rounds = (tiles + slots - 1) / slots;
    long long g = (tiles + rounds - 1) / rounds;
    if ((tiles & 7) == 0) g = (g + 7) & ~7LL;
    return (int)std::min<long long>(g, tiles);
}

template <typename T, int BN>
static hipError_t launch_glds_wide_cfg(const ConvParams& p, hipStream_t st) {
    bool tail = false;
    for (int s = 0; s < p.nseg; ++s) if (p.seg[s].taps != 9) tail = true;
    if (p.persist > 0) {   // the persistent instantiation exists for the 64-cout tile (at 96 couts it does not fit 256 registers)
        if constexpr (BN == 64) return tail ? hipErrorInvalidValue : launch_glds_wide_cfg2<T, BN, false, true>(p, st);
        else return hipErrorInvalidValue;
    }
    return tail ? launch_glds_wide_cfg2<T, BN, true, false>(p, st) : launch_glds_wide_cfg2<T, BN, false, false>(p, st);
}

// dtype: 1 bf16, 2 fp16; bn: 64 / 96 (a 128-cout tile needs 86 KB of LDS and 256+ registers: one workgroup per CU, which is what this flavour exists to avoid).  16-wide maps only, tiles_y = ceil(H / 16), tiles_x = ceil(W / 16), img_groups = N, no split-K.
hipError_t launch_conv_glds_wide(const ConvParams& p, int dtype, int bn, hipStream_t st) {
    if (dtype == 2) {
        if (bn == 96) return launch_glds_wide_cfg<_Float16, 96>(p, st);
        if (bn == 64) return launch_glds_wide_cfg<_Float16, 64>(p, st);
        return hipErrorInvalidValue;
    }
    if (bn == 96) return launch_glds_wide_cfg<__bf16, 96>(p, st);
    if (bn == 64) return launch_glds_wide_cfg<__bf16, 64>(p, st);
    return hipErrorInvalidValue;
}

}  // namespace td
