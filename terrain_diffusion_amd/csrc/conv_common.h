// Device helpers shared by the conv flavours (conv_igemm.hip, conv_glds.hip, conv_sb.hip, conv_s16.hip): element types, the fused
// prologue / epilogue arithmetic, the LDS swizzle and lane->pixel maps, the LDS-DMA statement.  gfx950 only.
#pragma once
#include "td_device.h"

// ablation hooks for tools/conv_bench.hip (-DTD_ABLATE_x); no-ops in the product build
#ifdef TD_ABLATE_BLOAD
#define TD_ABL_BLOAD(X)
#else
#define TD_ABL_BLOAD(X) X
#endif
#ifdef TD_ABLATE_BARRIER
#define TD_ABL_BARRIER(X)
#else
#define TD_ABL_BARRIER(X) X
#endif
#ifdef TD_ABLATE_BSTORE
#define TD_ABL_BSTORE(X)
#else
#define TD_ABL_BSTORE(X) X
#endif

// LDS rows are 128 bytes = eight 16-byte slots; slot index is XOR-ed with TD_SWZ(row) (row = patch pixel or cout-in-tile).
// ((row >> 1) & 7) is conflict-free for both the 16-row (16x16 MFMA) and the 32-row (32x32 MFMA) ds_read_b128 fragment reads.
#define TD_SWZ(r) (((r) >> 1) & 7)

namespace td {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-bit storage types of the engine: bf16 (default) and fp16 (WorldPipeline dtype='fp16', world_pipeline.py:365-370).  Same kernels, the
// element type only selects the vector typedefs and the MFMA opcode (v_mfma_f32_{32x32x16,16x16x32}_{bf16,f16}); accumulation is fp32 in both.
template <typename T> struct Half;
template <> struct Half<__bf16> {
    typedef bf16x8 x8; typedef bf16x4 x4;
    static __device__ __forceinline__ f32x16_ mfma32(x8 a, x8 b, f32x16_ c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half<_Float16> {
    typedef f16x8 x8; typedef f16x4 x4;
    static __device__ __forceinline__ f32x16_ mfma32(x8 a, x8 b, f32x16_ c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native 16-byte register piece (HIP's u32x4 struct defeats SROA -> scratch)

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int CHUNK = 32, PER16 = 4;
    static __device__ __forceinline__ float silu(float x) { return x / (1.f + expf(-x)) * (1.f / 0.596f); }
    static __device__ __forceinline__ float silu_scaled(float x, float s) { return silu(x * s); }   // mp_silu(s x), exact-fp32 mode
};
// 16-bit modes: mp_silu(s x) = (s x) / (1 + exp(-s x)) / 0.596 is evaluated as (x * k2) * rcp(1 + exp2(x * k1)), k1 = -s log2(e), k2 = s / 0.596
// (v_exp_f32 / v_rcp_f32, the scale folded into the two constants).  It is the SAME expression wherever the activation is applied -- patch
// staging of the consumer, second output of the producer, emb-scale epilogue -- so the choice of the place never changes a bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SiluK { float k1, k2; };
__device__ __forceinline__ SiluK silu_k(float s) { return SiluK{s * -1.4426950408889634f, s * (1.f / 0.596f)}; }
__device__ __forceinline__ float silu_k1(float x, SiluK k) { return (x * k.k2) * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * k.k1)); }
__device__ __forceinline__ f32x2 silu_k2(f32x2 x, SiluK k) {  // two elements: the multiplies and the add are v_pk_*_f32
    const f32x2 t = x * k.k1;
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return (x * k.k2) * r;
}
template <> struct Elem<_Float16> {
    static constexpr int CHUNK = 64, PER16 = 8;
    static __device__ __forceinline__ float silu(float x) { return silu_k1(x, silu_k(1.f)); }
    static __device__ __forceinline__ float silu_scaled(float x, float s) { return silu_k1(x, silu_k(s)); }
};
template <> struct Elem<__bf16> {
    static constexpr int CHUNK = 64, PER16 = 8;
    static __device__ __forceinline__ float silu(float x) { return silu_k1(x, silu_k(1.f)); }
    static __device__ __forceinline__ float silu_scaled(float x, float s) { return silu_k1(x, silu_k(s)); }
};

// mp_silu(s*x) on one 16-byte piece
template <typename T> __device__ __forceinline__ u32x4 xform_piece(u32x4 v, float s);
template <> __device__ __forceinline__ u32x4 xform_piece<float>(u32x4 v, float s) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = Elem<float>::silu(f[i] * s);
    return __builtin_bit_cast(u32x4, f);
}
template <> __device__ __forceinline__ u32x4 xform_piece<__bf16>(u32x4 v, float s) {
    bf16x8 h = __builtin_bit_cast(bf16x8, v);
    const SiluK k = silu_k(s);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 a = silu_k2(f32x2{(float)h[i], (float)h[i + 1]}, k);
        h[i] = (__bf16)a.x; h[i + 1] = (__bf16)a.y;
    }
    return __builtin_bit_cast(u32x4, h);
}
template <> __device__ __forceinline__ u32x4 xform_piece<_Float16>(u32x4 v, float s) {
    f16x8 h = __builtin_bit_cast(f16x8, v);
    const SiluK k = silu_k(s);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 a = silu_k2(f32x2{(float)h[i], (float)h[i + 1]}, k);
        h[i] = (_Float16)a.x; h[i + 1] = (_Float16)a.y;
    }
    return __builtin_bit_cast(u32x4, h);
}

__device__ __forceinline__ int src_pixel(int n, int y, int x, int Hs, int Ws, int resample) {
    if (resample == 1) { y *= 2; x *= 2; }
    else if (resample == 2) { y >>= 1; x >>= 1; }
    return (n * Hs + y) * Ws + x;
}

template <typename T> __device__ __forceinline__ f32x4 load4(const void* base, size_t idx);
template <> __device__ __forceinline__ f32x4 load4<float>(const void* base, size_t idx) {
    return *(const f32x4*)((const float*)base + idx);
}
template <> __device__ __forceinline__ f32x4 load4<_Float16>(const void* base, size_t idx) {
    f16x4 h = *(const f16x4*)((const _Float16*)base + idx);
    f32x4 f = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return f;
}
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const void* base, size_t idx) {
    bf16x4 h = *(const bf16x4*)((const __bf16*)base + idx);
    f32x4 f = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return f;
}

// Shared epilogue: transform 4 consecutive couts of one output pixel, store, return sum of squares of what was stored.
// `aux` = the 4 modulation values c[n][co..] (EPI_EMB_SILU) or the 4 residual values (EPI_RESIDUAL with p.res), fetched by the caller.
template <typename T>
__device__ __forceinline__ float epilogue4(const ConvParams& p, int n, int y, int x, int co, f32x4 v, float rn, f32x4 aux) {
    const int pix = (n * p.H + y) * p.W + x;
    if (p.epi == EPI_EMB_SILU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = Elem<T>::silu(v[k] * aux[k]);
    } else if (p.epi == EPI_RESIDUAL) {
        if (p.res) {
            const float s = p.res_scale * rn;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += s * aux[k];
        }
        if (p.clip > 0.f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(v[k], -p.clip), p.clip);
        }
    }
    float ss = 0.f;
    if (p.epi == EPI_DPM_STEP) {
        // v = F, the model output: one DPM-Solver++ step per (pixel, channel) in the arithmetic (and operation order) of dpm_step_kernel
        const size_t HW = (size_t)p.H * p.W, pl_ = (size_t)y * p.W + x;
        const SchedCoef k = p.dpm_k;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (co + q < p.Cout) {
                const size_t xi = ((size_t)n * p.Cout + (co + q)) * HW + pl_;
                const float xs = p.dpm_x[xi];
                float xn, m0;
                const float m1v = (k.order == 1 && !p.dpm_m2) ? 0.f : p.dpm_m1[xi];
                dpm_update(k, xs, v[q], m1v, k.order == 3 ? p.dpm_m2[xi] : 0.f, xn, m0);
                p.dpm_x[xi] = xn;
                if (p.dpm_m2) p.dpm_m2[xi] = m1v;   // history shifts: m2 <- m1 <- m0
                p.dpm_m1[xi] = m0;
                if (!k.last) ((T*)p.dpm_xin)[(size_t)pix * p.dpm_xin_cstride + co + q] = (T)(xn * k.c_in_next);
            }
        return 0.f;
    }
    if (p.out_f32) {
        float* o = (float*)p.out + (size_t)pix * p.out_cstride + co;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (co + k < p.Cout) { o[k] = v[k]; ss += v[k] * v[k]; }
    } else if (co < p.Cout) {  // Cout is a multiple of 4 whenever the output is a T tensor
        if constexpr (sizeof(T) == 4) {
            *(f32x4*)((float*)p.out + (size_t)pix * p.out_cstride + co) = v;
            ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if (p.out2) {
                f32x4 a;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = Elem<T>::silu_scaled(v[k], p.out2_scale);
                *(f32x4*)((float*)p.out2 + (size_t)pix * p.out_cstride + co) = a;
            }
        } else {
            typedef typename Half<T>::x4 hx4;
            hx4 h = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
            *(hx4*)((T*)p.out + (size_t)pix * p.out_cstride + co) = h;
#pragma unroll
            for (int k = 0; k < 4; ++k) { float f = (float)h[k]; ss += f * f; }
            if (p.out2) {  // from the ROUNDED value: bit-identical to applying the activation while staging the consumer's patch
                hx4 a;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = (T)Elem<T>::silu_scaled((float)h[k], p.out2_scale);
                *(hx4*)((T*)p.out2 + (size_t)pix * p.out_cstride + co) = a;
            }
        }
    }
    return ss;
}

// The partial sums of squares are independent loads: requested eight at a time (a rolled loop waits for each one -- up to 24 memory round
// trips in series in front of a kernel's first restage), added in ascending order as before.
__device__ __forceinline__ float pixel_rn(const float* sumsq, int nparts, size_t npix, int sp, float inv_c) {
    float s = 0.f;
    const float* b = sumsq + sp;
    int q = 0;
    // (round 5: sixteen at a time first -- ops on the 64 px x 16 cout flavour keep one plane per 16 couts, 36-48 planes at the deep levels, and every
    // round of this loop is one memory round trip in front of the consumer's first restage; the ascending order of the additions is unchanged)
    for (; q + 16 <= nparts; q += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = b[(size_t)(q + u) * npix];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += t[u];
    }
    for (; q + 8 <= nparts; q += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = b[(size_t)(q + u) * npix];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    if (q < nparts) {   // the last 1 ... 7 planes in ONE round trip (round 5; they used to be four at a time and then one by one: three dependent round trips
        float t[8];     // for the 6 planes of a 192-channel tensor); absent planes contribute an exact + 0.f (s >= 0), the order stays ascending
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = b[(size_t)(q + u < nparts ? q + u : nparts - 1) * npix];   // index-clamped: no branch around a load
#pragma unroll
        for (int u = 0; u < 8; ++u) s += q + u < nparts ? t[u] : 0.f;
    }
    return 1.f / (1e-4f + sqrtf(s * inv_c));  // mp_layers.py:9-12 with dim=1: x / (eps + ||x||_c / sqrt(C))
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// one 1-KiB LDS-DMA piece per wave: lane l copies 16 bytes from (uniform base + its own 32-bit offset) to LDS[lds_base + imm + 16*l].
// M0 is written in the statement that uses it (it is compiler-reserved and not preserved between statements).
// An SGPR written by the VALU (v_readfirstlane, v_readlane -- i.e. also the compiler's own reload of a SPILLED SGPR) needs five wait states before a VMEM
// instruction reads it, and the hazard recogniser does not look inside inline asm.  The default kernels are checked for this at the ISA level
// (tests/test_isa_contract.py, tools/isa_hazard_scan.py); where register pressure makes such reloads likely -- the persistent instantiation of the wide
// tile, the -DTD_TRACE builds -- the guarded forms carry the wait states themselves (TD_GLDS16G: s_add_u32 + s_nop 3 = five).
#ifdef TD_TRACE
#define TD_SGPR_GUARD "s_nop 3\n\t"
#define TD_SGPR_GUARD_LOAD "s_nop 4\n\t"
#else
#define TD_SGPR_GUARD "s_nop 0\n\t"
#define TD_SGPR_GUARD_LOAD ""
#endif
#define TD_GLDS16(VOFF, SBASE, LDS_BASE, IMM)                                                                 \
    asm volatile("s_add_u32 m0, %2, %3\n\t" TD_SGPR_GUARD "global_load_lds_dwordx4 %0, %1"                     \
                 ::"v"(VOFF), "s"(SBASE), "s"(LDS_BASE), "n"(IMM) : "memory", "scc")
#define TD_GLDS16G(VOFF, SBASE, LDS_BASE, IMM)                                                                \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1"                          \
                 ::"v"(VOFF), "s"(SBASE), "s"(LDS_BASE), "n"(IMM) : "memory", "scc")

// one dword per lane, global -> LDS[m0 base + 4 * lane]
#define TD_GLDS4(VOFF, SBASE, LDS_BASE, IMM)                                                                  \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1"                            \
                 ::"v"(VOFF), "s"(SBASE), "s"(LDS_BASE), "n"(IMM) : "memory", "scc")

// a wave-uniform pointer the compiler holds in vector registers, as an SGPR pair (operand of an inline-asm "s" constraint)
// (s_nop 4: an SGPR written by the VALU -- v_readfirstlane -- needs five wait states before a VMEM instruction reads it; the compiler's hazard recogniser
// does not look inside inline asm, and the first build's patch loads went out with a stale upper address half: memory fault)
template <typename P> __device__ __forceinline__ const P* td_uniform_ptr(const P* q) {
    const unsigned long long a = (unsigned long long)q;
    int lo = __builtin_amdgcn_readfirstlane((int)a), hi = __builtin_amdgcn_readfirstlane((int)(a >> 32));
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return (const P*)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// exact a / d for a < 2^32 / d with M = ceil(2^32 / d) (host: td_magic); d == 1 has no 32-bit M.  The workgroup-id decomposition of the conv
// kernels: a run-time integer division is ~40 scalar + vector instructions (v_rcp_iflag_f32 + corrections), and a kernel prologue had six of them
__device__ __forceinline__ unsigned td_udiv(unsigned a, unsigned d, unsigned M) { return d == 1 ? a : __umulhi(a, M); }
static inline unsigned td_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + d - 1) / d); }
// 16-byte load with an explicit GLOBAL address space (a pointer that went through an asm register pin loses hipcc's address-space inference and
// would be accessed with FLAT instructions, which count on lgkmcnt as well as vmcnt and collide with every LDS wait)
__device__ __forceinline__ u32x4 td_gld16(const void* q) { return *(const __attribute__((address_space(1))) u32x4*)q; }

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap: lanes 32-63 of a exchange with lanes 0-31 of b (both halves of a wave take part)
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

// One unit of the wide 16-bit epilogue (conv_glds.hip, conv_sb.hip): 8 accumulators of ONE pixel -- MFMA row groups 2m and 2m+1 of a 32x32
// block, i.e. couts c..c+3 (va) and c+8..c+11 (vb) on this lane, its partner lane l^32 holding c+4.. and c+12.. -- are transformed, rounded to
// T, and lane-pair transposed (v_permlane32_swap) into one 16-byte run per lane: o = 8 consecutive couts of the output, o2 = the consumer's
// mp_silu(scale * x) of the ROUNDED values (what its patch staging would compute), ss += sum of squares of the rounded values (pairwise
// v_pk_fma_f32 chain, then the two halves: a fixed order shared by every flavour).  Operands: ca/cb = the 8 modulation values (EPI_EMB_SILU),
// rw = the 16-byte residual run of this lane in the STORED layout (EPI_RESIDUAL), rs = its scale.  All flags are wave-uniform.
template <typename T>
__device__ __forceinline__ void epi_unit8(int epi, bool has_res, float clip, bool want_ss, bool want_o2, f32x4 va, f32x4 vb, f32x4 ca, f32x4 cb, u32x4 rw,
                                          float rs, SiluK k_o2, u32x4& o, u32x4& o2, float& ss) {
    typedef typename Half<T>::x4 hx4;
    f32x2 v[4] = {{va[0], va[1]}, {va[2], va[3]}, {vb[0], vb[1]}, {vb[2], vb[3]}};
    if (epi == EPI_EMB_SILU) {
        const f32x2 c[4] = {{ca[0], ca[1]}, {ca[2], ca[3]}, {cb[0], cb[1]}, {cb[2], cb[3]}};
        const SiluK k = silu_k(1.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = silu_k2(v[q] * c[q], k);
    } else if (epi == EPI_RESIDUAL) {
        if (has_res) {
            unsigned w0 = rw[0], w1 = rw[1], w2 = rw[2], w3 = rw[3];
            swap_halves(w0, w2); swap_halves(w1, w3);
            const hx4 ra = __builtin_bit_cast(hx4, u32x2{w0, w1}), rb = __builtin_bit_cast(hx4, u32x2{w2, w3});
            const f32x2 r[4] = {{(float)ra[0], (float)ra[1]}, {(float)ra[2], (float)ra[3]}, {(float)rb[0], (float)rb[1]}, {(float)rb[2], (float)rb[3]}};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] + rs * r[q];
        }
        if (clip > 0.f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = f32x2{__builtin_amdgcn_fmed3f(v[q].x, -clip, clip), __builtin_amdgcn_fmed3f(v[q].y, -clip, clip)};
        }
    }
    const hx4 ha = {(T)v[0].x, (T)v[0].y, (T)v[1].x, (T)v[1].y};
    const hx4 hb = {(T)v[2].x, (T)v[2].y, (T)v[3].x, (T)v[3].y};
    const f32x2 f[4] = {{(float)ha[0], (float)ha[1]}, {(float)ha[2], (float)ha[3]}, {(float)hb[0], (float)hb[1]}, {(float)hb[2], (float)hb[3]}};
    if (want_ss) {
        f32x2 s2 = f[0] * f[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) s2 = f[q] * f[q] + s2;
        ss += s2.x + s2.y;
    }
    {
        const u32x2 pa = __builtin_bit_cast(u32x2, ha), pb = __builtin_bit_cast(u32x2, hb);
        unsigned a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
        swap_halves(a0, b0); swap_halves(a1, b1);
        o = u32x4{a0, a1, b0, b1};
    }
    if (want_o2) {
        f32x2 g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = silu_k2(f[q], k_o2);
        const hx4 ga = {(T)g[0].x, (T)g[0].y, (T)g[1].x, (T)g[1].y}, gb = {(T)g[2].x, (T)g[2].y, (T)g[3].x, (T)g[3].y};
        const u32x2 qa = __builtin_bit_cast(u32x2, ga), qb = __builtin_bit_cast(u32x2, gb);
        unsigned c0 = qa[0], c1 = qa[1], d0 = qb[0], d1 = qb[1];
        swap_halves(c0, d0); swap_halves(c1, d1);
        o2 = u32x4{c0, c1, d0, d1};
    }
}

// Pixel owned by lane (l31) of the 32-pixel MFMA fragment that starts at tile-linear pixel q0 (a multiple of 32).
// ds_read_b128 is serviced in four fixed 16-lane groups, {0-3,12-15,20-27} and {4-11,16-19,28-31} for the lower half-wave (guide
// §LDS); with 128-byte rows swizzled by ((row >> 1) & 7) a group is conflict-free iff its 16 rows are distinct mod 16.  On a
// 16-wide tile each group therefore takes one whole tile row (16 consecutive patch rows, for every tap shift); the natural
// "lane = pixel" order would mix two tile rows 18 patch rows apart inside a group and cost 2 LDS cycles per group instead of 1.
template <int TW, int TPIX>
__device__ __forceinline__ void frag_pixel(int q0, int l31, int& img, int& ty, int& tx) {
#ifndef TD_NO_REMAP
#define TD_REMAP_ON 1
#else
#define TD_REMAP_ON 0
#endif
    if constexpr (TW == 16 && TD_REMAP_ON) {
        const bool g2 = (l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28;
        const int u = g2 ? (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16)) : (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12));
        img = q0 / TPIX;
        ty = (q0 % TPIX) / 16 + (g2 ? 1 : 0);
        tx = u;
    } else if constexpr (TW == 8 && TD_REMAP_ON) {
        // 8-wide tile, patch rows 12 pixels apart: the fragment is 4 tile rows x 8; group one takes the LEFT halves of the four rows
        // (patch rows p, p+12, p+24, p+36 (+0..3): residues p+{0..3}, p+{12..15}, p+{8..11}, p+{4..7} mod 16), group two the right halves
        const bool g2 = (l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28;
        const int u = g2 ? (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16)) : (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12));
        img = q0 / TPIX;
        ty = (q0 % TPIX) / 8 + (u >> 2);
        tx = (u & 3) + (g2 ? 4 : 0);
    } else {
        const int q = q0 + l31, r = q % TPIX;
        img = q / TPIX; ty = r / TW; tx = r % TW;
    }
}

}  // namespace td
