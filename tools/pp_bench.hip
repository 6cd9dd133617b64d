// Tap-loop schedule micro-benchmark (gfx950): what MFMA rate does the conv tap loop reach with LDS-resident operands, as a function of the
// wave schedule?  One 512-thread workgroup per CU, the LDS image of conv_glds.hip (3-slot weight ring, 18x18 halo patch with 144-byte rows,
// same fragment addressing), no global traffic except the optional LDS-DMA weight stream.
//   MODE 0: "interleaved" -- every wave mixes ds_reads and MFMAs, one barrier per tap (the conv_glds.hip schedule)
//   MODE 1: MODE 0 + LDS-DMA weight stream
//   MODE 2: "ping-pong"   -- the two waves of a SIMD alternate: one issues its 16 MFMAs back to back while the other reads the next tap's
//                            fragments; two barriers per tap
//   MODE 3: MODE 2 + LDS-DMA weight stream
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pp_bench.hip -o tools/pp_bench.out && ./tools/pp_bench.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define TD_SWZ(r) (((r) >> 1) & 7)
#define TD_GLDS16(VOFF, SBASE, LDS_BASE, IMM)                                                                 \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                          \
                 ::"v"(VOFF), "s"(SBASE), "s"(LDS_BASE), "n"(IMM) : "memory", "scc")

template <int MODE, int WAVES_M, int WAVES_N, int BN>
__global__ __launch_bounds__(512, 2) void pp_kernel(const unsigned char* __restrict__ wglobal, const unsigned char* __restrict__ aglobal, int ngroups, float* out,
                                                    unsigned long long* cyc) {
    constexpr int NTHR = 512, BM = 256, TW = 16, PW = 18, PITCH = 144, NPATCH = 18 * 18;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
    constexpr int NBI = (BN * 128 + NTHR * 16 - 1) / (NTHR * 16), B_BYTES = NBI * NTHR * 16, RING = 3;
    constexpr int A_BASE = RING * B_BYTES;
    constexpr bool DMA = (MODE & 1) != 0, PP = MODE >= 2 && MODE < 4, HP = MODE >= 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, l31 = lane & 31, lh = lane >> 5;
    // fill LDS with the (random) images
    for (int i = tid; i < (A_BASE + NPATCH * PITCH) / 16; i += NTHR)
        *(u32x4*)(smem + i * 16) = *(const u32x4*)((i * 16 < A_BASE ? wglobal : aglobal) + (size_t)i * 16);
    __syncthreads();
    unsigned xbase[MT], wbase[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // conflict-free lane -> pixel map of conv_glds.hip (16-wide tile): a 16-lane ds_read_b128 group takes one tile row
        const bool g2 = (l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28;
        const int u = g2 ? (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16)) : (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12));
        const int q0 = wm * WM + i * 32, ty = q0 / 16 + (g2 ? 1 : 0), tx = u;
        xbase[i] = (unsigned)A_BASE + (unsigned)(ty * PW + tx) * PITCH + (unsigned)lh * 16u;
    }
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
    const unsigned char* wnext = wglobal;
    const size_t wstep = (size_t)BN * 128;
    unsigned wvoff[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) wvoff[i] = (unsigned)tid * 16u + (unsigned)i * NTHR * 16u;
    const unsigned ldsw = (unsigned)wave * 1024u;
#define GLDS_B(SLOT)                                                                                         \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NBI; ++i_) TD_GLDS16(wvoff[i_], wnext, ldsw, (SLOT) * B_BYTES + i_ * NTHR * 16); \
        wnext += wstep;                                                                                      \
    }
#define TOFF(T) ((((T) / 3) * PW + ((T) % 3)) * PITCH)
#define FRAG_READ(WF, XF, SLOT, KS, TOFFV)                                                                   \
    {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) WF[j_] = *(const u32x4*)(smem + wbase[KS] + ((SLOT) * B_BYTES + j_ * 4096)); \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) XF[i_] = *(const u32x4*)(smem + xbase[i_] + ((TOFFV) + (KS) * 32)); \
    }
#define FRAG_MFMA(WF, XF)                                                                                    \
    {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[j_]), __builtin_bit_cast(bf16x8, XF[i_]), acc[i_][j_], 0, 0, 0); \
    }
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (HP) {
        // half-tap ping-pong: same code for both groups, group 0 one phase behind; a phase = 2 k-steps (8 fragments, MT*NT*2 MFMAs)
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        u32x4 wf[2][NT], xf[2][MT];
        if (DMA) { GLDS_B(0); GLDS_B(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if (grp == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define HBAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define HPH(SLOT, TOFFV, H, FETCHSLOT)                                                                       \
    {                                                                                                        \
        if (DMA && (H) == 0) GLDS_B(FETCHSLOT);                                                              \
        FRAG_READ(wf[0], xf[0], SLOT, 2 * (H), TOFFV); FRAG_READ(wf[1], xf[1], SLOT, 2 * (H) + 1, TOFFV);    \
        if (DMA && (H) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        HBAR();                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        FRAG_MFMA(wf[0], xf[0]); FRAG_MFMA(wf[1], xf[1]);                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        HBAR();                                                                                              \
    }
#define HTAP(SLOT, TOFFV) { HPH(SLOT, TOFFV, 0, ((SLOT) + 2) % RING); HPH(SLOT, TOFFV, 1, 0); }
        t0 = __builtin_amdgcn_s_memtime();
        for (int g = 0; g < ngroups; ++g) {
            HTAP(0, TOFF(0)); HTAP(1, TOFF(1)); HTAP(2, TOFF(2)); HTAP(0, TOFF(3)); HTAP(1, TOFF(4)); HTAP(2, TOFF(5)); HTAP(0, TOFF(6)); HTAP(1, TOFF(7)); HTAP(2, TOFF(8));
            if (DMA && (g % 6) == 5) wnext = wglobal;
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (grp == 1) HBAR();
    } else if constexpr (!PP) {
        u32x4 wfA_[NT], xfA_[MT], wfB_[NT], xfB_[MT];
        if (DMA) { GLDS_B(0); GLDS_B(1); }
#define TAPP(TAPIDX, SLOT, TOFFV, TOFF_NEXT)                                                                 \
    {                                                                                                        \
        FRAG_MFMA(wfA_, xfA_);                                                                               \
        FRAG_READ(wfA_, xfA_, SLOT, 2, TOFFV);                                                               \
        FRAG_MFMA(wfB_, xfB_);                                                                               \
        FRAG_READ(wfB_, xfB_, SLOT, 3, TOFFV);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT + MT)) : "memory");                               \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if (DMA) GLDS_B(((SLOT) + 2) % RING);                                                                \
        FRAG_MFMA(wfA_, xfA_);                                                                               \
        FRAG_READ(wfA_, xfA_, ((SLOT) + 1) % RING, 0, TOFF_NEXT);                                            \
        FRAG_MFMA(wfB_, xfB_);                                                                               \
        FRAG_READ(wfB_, xfB_, ((SLOT) + 1) % RING, 1, TOFF_NEXT);                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NT + MT, 0);                                             \
    }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FRAG_READ(wfA_, xfA_, 0, 0, TOFF(0));
        FRAG_READ(wfB_, xfB_, 0, 1, TOFF(0));
        t0 = __builtin_amdgcn_s_memtime();
        for (int g = 0; g < ngroups; ++g) {
            TAPP(0, 0, TOFF(0), TOFF(1)); TAPP(1, 1, TOFF(1), TOFF(2)); TAPP(2, 2, TOFF(2), TOFF(3));
            TAPP(3, 0, TOFF(3), TOFF(4)); TAPP(4, 1, TOFF(4), TOFF(5)); TAPP(5, 2, TOFF(5), TOFF(6));
            TAPP(6, 0, TOFF(6), TOFF(7)); TAPP(7, 1, TOFF(7), TOFF(8)); TAPP(8, 2, TOFF(8), TOFF(0));
            if (DMA && (g % 6) == 5) wnext = wglobal;
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("" : "+v"(wfA_[0]), "+v"(wfB_[0]), "+v"(xfA_[0]), "+v"(xfB_[0]));
    } else {
        // ping-pong: waves w and w+4 share a SIMD (workgroup waves go round-robin over the 4 SIMDs); group 0 = waves 0-3, group 1 = waves 4-7
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        u32x4 wf[4][NT], xf[4][MT];
#define LOAD_TAP(SLOT, TOFFV)                                                                                \
    {                                                                                                        \
        FRAG_READ(wf[0], xf[0], SLOT, 0, TOFFV); FRAG_READ(wf[1], xf[1], SLOT, 1, TOFFV);                    \
        FRAG_READ(wf[2], xf[2], SLOT, 2, TOFFV); FRAG_READ(wf[3], xf[3], SLOT, 3, TOFFV);                    \
    }
#define MFMA_TAP()                                                                                           \
    {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        FRAG_MFMA(wf[0], xf[0]); FRAG_MFMA(wf[1], xf[1]); FRAG_MFMA(wf[2], xf[2]); FRAG_MFMA(wf[3], xf[3]);  \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    }
#define PHASE_END()                                                                                          \
    {                                                                                                        \
        if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
        // tiles 0 and 1 of the weight stream are in flight / resident before the loop (the ring is pre-filled anyway)
        if (DMA) { GLDS_B(0); GLDS_B(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if (grp == 0) {
            LOAD_TAP(0, TOFF(0));
            PHASE_END();
            t0 = __builtin_amdgcn_s_memtime();
            for (int g = 0; g < ngroups; ++g) {
                // tap T (slot S): MFMA(T) | barrier | DMA tile T+3 -> slot S (its last reader, group 1's LOAD(T), finished before that barrier);
                // LOAD(T+1) | barrier
#define PP0(S, TOFF_NEXT)                                                                                    \
    {                                                                                                        \
        MFMA_TAP();                                                                                          \
        PHASE_END();                                                                                         \
        if (DMA) GLDS_B(S);                                                                                  \
        LOAD_TAP(((S) + 1) % RING, TOFF_NEXT);                                                               \
        PHASE_END();                                                                                         \
    }
                PP0(0, TOFF(1)); PP0(1, TOFF(2)); PP0(2, TOFF(3)); PP0(0, TOFF(4)); PP0(1, TOFF(5)); PP0(2, TOFF(6)); PP0(0, TOFF(7)); PP0(1, TOFF(8)); PP0(2, TOFF(0));
                if (DMA && (g % 6) == 5) wnext = wglobal;
            }
            t1 = __builtin_amdgcn_s_memtime();
        } else {
            PHASE_END();
            for (int g = 0; g < ngroups; ++g) {
                // tap T (slot S): DMA tile T+2 -> slot S+2 (last reader: this group's LOAD(T-1), one barrier ago); LOAD(T) | barrier | MFMA(T) | barrier
#define PP1(S, TOFFV)                                                                                        \
    {                                                                                                        \
        if (DMA) GLDS_B(((S) + 2) % RING);                                                                   \
        LOAD_TAP(S, TOFFV);                                                                                  \
        PHASE_END();                                                                                         \
        MFMA_TAP();                                                                                          \
        PHASE_END();                                                                                         \
    }
                PP1(0, TOFF(0)); PP1(1, TOFF(1)); PP1(2, TOFF(2)); PP1(0, TOFF(3)); PP1(1, TOFF(4)); PP1(2, TOFF(5)); PP1(0, TOFF(6)); PP1(1, TOFF(7)); PP1(2, TOFF(8));
                if (DMA && (g % 6) == 5) wnext = wglobal;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * NTHR + tid] = s;
    if (tid == 0 && cyc) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int WAVES_M, int WAVES_N, int BN>
static void run(const char* name, const unsigned char* w, const unsigned char* a, float* out, unsigned long long* cyc, int ngroups, int grid) {
    constexpr int NBI = (BN * 128 + 512 * 16 - 1) / (512 * 16);
    const size_t lds = 3 * (size_t)NBI * 512 * 16 + 18 * 18 * 144;
    auto k = pp_kernel<MODE, WAVES_M, WAVES_N, BN>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, w, a, ngroups, out, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(grid);
    CK(hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double mc = 0; for (auto c : hc) mc += (double)c; mc /= grid;
    const double taps = 9.0 * ngroups, flop = (double)grid * 256.0 * BN * 64.0 * 2.0 * taps;
    printf("%-34s grid %4d: %8.1f us  %7.1f TFLOP/s (%.1f%% of 2500)  %.0f cycles/tap (ideal %d)  clock ~%.2f GHz\n", name, grid, best * 1e3, flop / best / 1e9, flop / best / 1e9 / 25.0,
           mc / taps, 2 * (256 / WAVES_M / 32) * (BN / WAVES_N / 32) * 4 * 32, mc / (best * 1e3) / 1e6 * 1.0);
}

int main(int argc, char** argv) {
    const int ngroups = argc > 1 ? atoi(argv[1]) : 240, grid = argc > 2 ? atoi(argv[2]) : 256;
    const size_t wbytes = (size_t)56 * 192 * 128 + (1 << 20), abytes = 1 << 20;
    std::vector<unsigned short> hw(wbytes / 2), ha(abytes / 2);
    srand(1);
    for (auto& v : hw) v = 0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15);
    for (auto& v : ha) v = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
    unsigned char *w, *a; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&a, abytes)); CK(hipMalloc(&out, (size_t)grid * 512 * 4)); CK(hipMalloc(&cyc, grid * 8));
    CK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(a, ha.data(), abytes, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 4, 2, 128>("half-tap pp        4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<5, 4, 2, 128>("half-tap pp + DMA  4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<5, 8, 1, 96>("half-tap pp + DMA  8x1 bn96", w, a, out, cyc, ngroups, grid);
        run<0, 4, 2, 128>("interleaved        4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<2, 4, 2, 128>("ping-pong          4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<1, 4, 2, 128>("interleaved + DMA  4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<3, 4, 2, 128>("ping-pong + DMA    4x2 bn128", w, a, out, cyc, ngroups, grid);
        run<0, 8, 1, 96>("interleaved        8x1 bn96", w, a, out, cyc, ngroups, grid);
        run<2, 8, 1, 96>("ping-pong          8x1 bn96", w, a, out, cyc, ngroups, grid);
        run<1, 8, 1, 96>("interleaved + DMA  8x1 bn96", w, a, out, cyc, ngroups, grid);
        run<3, 8, 1, 96>("ping-pong + DMA    8x1 bn96", w, a, out, cyc, ngroups, grid);
        run<2, 4, 2, 192>("ping-pong          4x2 bn192", w, a, out, cyc, ngroups, grid);
        run<3, 4, 2, 192>("ping-pong + DMA    4x2 bn192", w, a, out, cyc, ngroups, grid);
        run<2, 2, 4, 256>("ping-pong          2x4 bn256", w, a, out, cyc, ngroups, grid);
    }
    return 0;
}
