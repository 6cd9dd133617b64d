// Throughput flavour of the implicit-GEMM convolution (bf16, gfx950): used when a layer has enough output pixels to fill the chip.
// Same maths / parameter block / fused prologues and epilogues as conv_igemm.hip; what differs is the pipeline:
//   * weights stream HBM/L2 -> LDS with `global_load_lds_dwordx4` (LDS-DMA): no VGPR staging, no ds_write, into a 3-slot ring
//     that runs two K-steps ahead of the MFMAs behind COUNTED `s_waitcnt vmcnt(N)` and raw `s_barrier`s (never drained to 0 in
//     the steady state) — MI355X guide §5 "Pipelining across barriers" / T3+T4;
//   * 8 waves per workgroup (2 per SIMD) on a 256-pixel x BN-cout tile; each wave owns 64 pixels x BN/2 couts of
//     v_mfma_f32_32x32x16_bf16 tiles, so one weight tile is shared by 256 pixels and one activation patch by BN couts;
//   * the activation halo patch ((TH+2)x(TW+2) pixels x 64 channels) is fetched one K-group ahead into registers, transformed
//     (pixel-norm / mp_silu) and written to LDS once per 9 taps;
//   * LDS rows are 128 B with the 16-byte slot index XOR-ed by ((row >> 1) & 7): conflict-free for the 32-row ds_read_b128
//     fragments of the 32x32x16 MFMA (and for the 16-row fragments of conv_igemm.hip).
// The LDS destination of an LDS-DMA is wave-uniform base + lane*16, so the packed weight slab is stored pre-swizzled in HBM and
// copied linearly (guide rule 21: swizzle on the source side).
#include "td_device.h"

namespace td {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// one 1-KiB LDS-DMA piece per wave: lane l copies 16 bytes from its own global address to LDS[lds_addr + 16*l]
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_glds_kernel(const ConvParams p) {
    typedef __bf16 T;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int TPIX = TH * TW, BM = NIMG * TPIX;
    constexpr int PH = TH + 2, PW = TW + 2, PPI = PH * PW, NPATCH = NIMG * PPI;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int CHUNK = 64, PER16 = 8;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    // a ring slot holds a whole number of LDS-DMA rounds (NTHR x 16 B); for BN = 96 that is 128 rows: the 32 extra rows belong to
    // the next cout tile (or the slab's tail padding) and are never read
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16);  // LDS-DMA instructions per thread per weight tile
    constexpr int A_BYTES = NPATCH * 128, B_BYTES = NBI * NTHR * 16, RING = 3;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: its offset is 0
    unsigned char* s_a = smem;
    unsigned char* s_b = smem + A_BYTES;
    float* s_rn = (float*)(s_b + RING * B_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    const int ntile = bid % p.n_ntiles; bid /= p.n_ntiles;
    const int mtile = bid;
    const int txi = mtile % p.tiles_x, tyi = (mtile / p.tiles_x) % p.tiles_y, ig = mtile / (p.tiles_x * p.tiles_y);
    const int n0 = ig * NIMG, y0 = tyi * TH, x0 = txi * TW, co0 = ntile * BN;

    // total K-steps of this conv (ksplit == 1 in this flavour)
    int nk = 0;
    for (int s = 0; s < p.nseg; ++s) nk += (p.seg[s].C / CHUNK) * p.seg[s].taps;

    // ---- weight ring: two tiles in flight before anything else
    const unsigned char* wbase = (const unsigned char*)p.wpack + (size_t)co0 * 128 + (size_t)tid * 16;
    const size_t wstep = (size_t)p.CoutPad * 128;
    const unsigned ldsb0 = (unsigned)A_BYTES + (unsigned)wave * 1024u;
#define TD_GLDS_B(K, SLOT)                                                                                   \
    {                                                                                                        \
        const unsigned char* g_ = wbase + (size_t)(K) * wstep;                                               \
        const unsigned l_ = ldsb0 + (unsigned)(SLOT) * (unsigned)B_BYTES;                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) glds16(g_ + (size_t)i_ * NTHR * 16, l_ + (unsigned)i_ * NTHR * 16); \
    }
    TD_GLDS_B(0, 0);
    if (nk > 1) TD_GLDS_B(1, 1);

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
        const bool ok = (pp < NPATCH) && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
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
            const int c_ = a_coord[it_];                                                                              \
            aoff[it_] = -1;                                                                                           \
            if (c_ >= 0 && (seg_taps == 9 || (c_ & 1)))                                                               \
                aoff[it_] = src_pixel(c_ >> 21, (c_ >> 11) & 1023, (c_ >> 1) & 1023, Hs_, Ws_, rs_) * cs_ + (tid & 7) * PER16; \
        }                                                                                                             \
    }
#define TD_LOAD_A(CH)                                                                                  \
    {                                                                                                  \
        const T* src_ = seg_src + (CH) * CHUNK;                                                        \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            av[it_] = u32x4{0u, 0u, 0u, 0u};                                                           \
            if (aoff[it_] >= 0) av[it_] = *(const u32x4*)(src_ + aoff[it_]);                           \
        }                                                                                              \
    }
#define TD_STORE_A()                                                                                   \
    {                                                                                                  \
        _Pragma("unroll") for (int it_ = 0; it_ < A_ITERS; ++it_) {                                    \
            const int e_ = tid + it_ * NTHR, pp_ = e_ >> 3, slot_ = e_ & 7;                            \
            if (pp_ < NPATCH) {                                                                        \
                u32x4 v_ = av[it_];                                                                    \
                if (seg_xform != 0 && aoff[it_] >= 0) {                                                \
                    float s_ = seg_scale;                                                              \
                    if (seg_xform == 2) s_ *= s_rn[pp_];                                               \
                    v_ = xform_piece<T>(v_, s_);                                                       \
                }                                                                                      \
                *(u32x4*)(s_a + pp_ * 128 + ((slot_ ^ TD_SWZ(pp_)) << 4)) = v_;                        \
            }                                                                                          \
        }                                                                                              \
    }
    TD_SEG_BEGIN(0);
    TD_LOAD_A(0);

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

    // ---- MFMA operand addressing: weights = A operand (rows = couts), activations = B operand (cols = pixels)
    int base_pp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = wm * WM + i * 32 + l31;
        const int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        base_pp[i] = img * PPI + (ty + 1) * PW + (tx + 1);
    }
    int woff[NT][4];  // byte offset of this lane's 16-byte weight fragment inside a tile, per 16-deep k-step (tap-invariant)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nl = wn * WN + j * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) woff[j][ks] = nl * 128 + (((ks * 2 + lh) ^ TD_SWZ(nl)) << 4);
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __syncthreads();  // s_rn visible (prologue only: this one may drain the two weight tiles, they are needed next anyway)
    TD_STORE_A();

    int k = 0, slot = 0;
#ifdef TD_ABLATE_DSREAD
#define TD_FRAG_READ(WF, XF, KS)                                                                             \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) { WF[j_] = u32x4{(unsigned)woff[j_][KS], 1u, 2u, 3u}; asm volatile("" : "+v"(WF[j_])); } \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) { XF[i_] = u32x4{(unsigned)xrow_[i_], 1u, 2u, 3u}; asm volatile("" : "+v"(XF[i_])); }   \
    }
#else
#define TD_FRAG_READ(WF, XF, KS)                                                                             \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(sb_ + woff[j_][KS]);     \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(s_a + xrow_[i_] + ((((KS) * 2) ^ xswz_[i_]) << 4)); \
    }
#endif
#ifdef TD_ABLATE_MFMA
#define TD_FRAG_MFMA(WF, XF)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) asm volatile("" ::"v"(WF[j_]));                    \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) asm volatile("" ::"v"(XF[i_]));                    \
    }
#else
#define TD_FRAG_MFMA(WF, XF)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[j_]), __builtin_bit_cast(bf16x8, XF[i_]), acc[i_][j_], 0, 0, 0); \
    }
#endif
#define TD_TAP(TAPIDX, DOFF)                                                                                 \
    {                                                                                                        \
        /* weight tile k must have landed: only the tile issued after it (k+1) may still be in flight */     \
        if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                           \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        TD_ABL_BARRIER(__builtin_amdgcn_s_barrier());                                                        \
        asm volatile("" ::: "memory");                                                                       \
        TD_ABL_BSTORE(if ((TAPIDX) == 0 && has_next) TD_LOAD_A(chunk + 1));                                  \
        TD_ABL_BLOAD(if (k + 2 < nk) TD_GLDS_B(k + 2, slot == 0 ? 2 : slot - 1));                            \
        const unsigned char* sb_ = s_b + slot * B_BYTES;                                                     \
        int xrow_[MT], xswz_[MT];  /* recomputed per tap on purpose: hoisting 9 taps x 4 k-steps of addresses spills */ \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                                  \
            int pp_ = base_pp[i_];                                                                           \
            asm volatile("" : "+v"(pp_));                                                                    \
            pp_ += (DOFF);                                                                                   \
            xrow_[i_] = pp_ * 128; xswz_[i_] = TD_SWZ(pp_) ^ lh;                                             \
        }                                                                                                    \
        u32x4 wfA_[NT], xfA_[MT], wfB_[NT], xfB_[MT];                                                        \
        TD_FRAG_READ(wfA_, xfA_, 0);                                                                         \
        TD_FRAG_READ(wfB_, xfB_, 1);                                                                         \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        TD_FRAG_READ(wfA_, xfA_, 2);                                                                         \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
        TD_FRAG_READ(wfB_, xfB_, 3);                                                                         \
        TD_FRAG_MFMA(wfA_, xfA_);                                                                            \
        TD_FRAG_MFMA(wfB_, xfB_);                                                                            \
        slot = slot == 2 ? 0 : slot + 1;                                                                     \
        ++k;                                                                                                 \
    }
    for (int seg = 0; seg < p.nseg; ++seg) {
        if (seg > 0) {  // first K-group of a later segment: its patch could not be prefetched (different source tensor / transform)
            TD_SEG_BEGIN(seg);
            TD_LOAD_A(0);
            __builtin_amdgcn_s_barrier();  // every wave is done reading the previous patch
            asm volatile("" ::: "memory");
            TD_STORE_A();                  // visible after the next tap's lgkmcnt(0) + barrier
        }
        for (int chunk = 0; chunk < seg_nchunks; ++chunk) {
            const bool has_next = chunk + 1 < seg_nchunks;
            if (seg_taps == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) TD_TAP(tap, (tap / 3 - 1) * PW + (tap % 3 - 1));
            } else {
                TD_TAP(0, 0);
            }
            if (has_next) {
                __builtin_amdgcn_s_barrier();  // every wave is done reading the current patch
                asm volatile("" ::: "memory");
                TD_ABL_BSTORE(TD_STORE_A());   // visible to the others after the next tap's lgkmcnt(0) + barrier
            }
        }
    }
#undef TD_TAP
#undef TD_FRAG_READ
#undef TD_FRAG_MFMA
#undef TD_LOAD_A
#undef TD_STORE_A
#undef TD_SEG_BEGIN
#undef TD_GLDS_B

    // ---------------- epilogue: lane holds, per 32x32 tile, 4 groups of 4 consecutive couts of pixel column (lane & 31)
    const size_t M = (size_t)p.N * p.H * p.W;
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
    if (p.N < 0)
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = wm * WM + i * 32 + l31;
        const int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        const int n = n0 + img, y = y0 + ty, x = x0 + tx;
        const bool ok = n < p.N && y < p.H && x < p.W;
        float ss = 0.f;
        if (ok) {
            const float rn = (p.res_sumsq != nullptr) ? s_rn[base_pp[i]] : 1.f;
            const int cobase = co0 + wn * WN + 4 * lh;
            const int sp = (p.epi == EPI_RESIDUAL && p.res) ? src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample) : 0;
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
                    ss += epilogue4<T>(p, n, y, x, cobase + j * 32 + rg * 8, v, rn, aux[rg]);
                }
            }
        }
        if (p.out_sumsq) {
            ss += __shfl_xor(ss, 32);
            if (ok && lh == 0) {
                const size_t pix = ((size_t)n * p.H + y) * p.W + x;
                p.out_sumsq[(size_t)(ntile * WAVES_N + wn) * M + pix] = ss;
            }
        }
    }
}

template <int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
static hipError_t launch_glds_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int NPATCH = NIMG * (TH + 2) * (TW + 2);
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    const size_t lds = (size_t)NPATCH * 128 + 3 * (size_t)(((BN * 128 + NTHR * 16 - 1) / (NTHR * 16)) * NTHR * 16) + NPATCH * 4;
    const int grid = p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups;
    auto kern = conv_glds_kernel<TH, TW, NIMG, BN, WAVES_M, WAVES_N>;
    static bool attr_set = false;  // one per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
    return hipGetLastError();
}

// Tile configurations (variant):
//   0 "big"   : 8 waves, 256-pixel tile (16x16, narrow maps 8x8 x 4 images), bn 128 -> waves 4x2 (64 px x 64 co), bn 96 -> 8x1 (32 px x 96 co);
//               ~91 KB LDS -> one workgroup per CU
//   1 "small" : 4 waves, 128-pixel tile (8x16, narrow maps 8x8 x 2 images), bn 128 -> waves 2x2 (64 px x 64 co), bn 96 -> 4x1 (32 px x 96 co);
//               ~71 KB LDS -> two independent workgroups per CU whose prologues / epilogues / barrier stalls overlap
hipError_t launch_conv_glds(const ConvParams& p, bool narrow, int bn, int variant, hipStream_t st) {
    if (variant == 1) {
        if (!narrow) return bn == 128 ? launch_glds_cfg<8, 16, 1, 128, 2, 2>(p, st) : launch_glds_cfg<8, 16, 1, 96, 4, 1>(p, st);
        return bn == 128 ? launch_glds_cfg<8, 8, 2, 128, 2, 2>(p, st) : launch_glds_cfg<8, 8, 2, 96, 4, 1>(p, st);
    }
    if (!narrow) return bn == 128 ? launch_glds_cfg<16, 16, 1, 128, 4, 2>(p, st) : launch_glds_cfg<16, 16, 1, 96, 8, 1>(p, st);
    return bn == 128 ? launch_glds_cfg<8, 8, 4, 128, 4, 2>(p, st) : launch_glds_cfg<8, 8, 4, 96, 8, 1>(p, st);
}

}  // namespace td
