// Device helpers shared by the three conv flavours (conv_igemm.hip, conv_glds.hip, conv_pp.hip): element types, the fused
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
};
template <> struct Elem<_Float16> {
    static constexpr int CHUNK = 64, PER16 = 8;
    static __device__ __forceinline__ float silu(float x) {
        return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)) * (1.f / 0.596f);
    }
};
template <> struct Elem<__bf16> {
    static constexpr int CHUNK = 64, PER16 = 8;
    static __device__ __forceinline__ float silu(float x) {
        return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)) * (1.f / 0.596f);
    }
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
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)Elem<__bf16>::silu((float)h[i] * s);
    return __builtin_bit_cast(u32x4, h);
}
template <> __device__ __forceinline__ u32x4 xform_piece<_Float16>(u32x4 v, float s) {
    f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)Elem<_Float16>::silu((float)h[i] * s);
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
                for (int k = 0; k < 4; ++k) a[k] = Elem<T>::silu(v[k] * p.out2_scale);
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
                for (int k = 0; k < 4; ++k) a[k] = (T)Elem<T>::silu((float)h[k] * p.out2_scale);
                *(hx4*)((T*)p.out2 + (size_t)pix * p.out_cstride + co) = a;
            }
        }
    }
    return ss;
}

__device__ __forceinline__ float pixel_rn(const float* sumsq, int nparts, size_t npix, int sp, float inv_c) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += sumsq[(size_t)q * npix + sp];
    return 1.f / (1e-4f + sqrtf(s * inv_c));  // mp_layers.py:9-12 with dim=1: x / (eps + ||x||_c / sqrt(C))
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// one 1-KiB LDS-DMA piece per wave: lane l copies 16 bytes from (uniform base + its own 32-bit offset) to LDS[lds_base + imm + 16*l].
// M0 is written in the statement that uses it (it is compiler-reserved and not preserved between statements).
#define TD_GLDS16(VOFF, SBASE, LDS_BASE, IMM)                                                                 \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                          \
                 ::"v"(VOFF), "s"(SBASE), "s"(LDS_BASE), "n"(IMM) : "memory", "scc")

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap: lanes 32-63 of a exchange with lanes 0-31 of b (both halves of a wave take part)
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
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
