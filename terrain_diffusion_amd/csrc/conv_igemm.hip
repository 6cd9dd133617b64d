// Implicit-GEMM convolution for the EDMUnet2D conv stack, hand-written for gfx950 (CDNA4).
//
// Replaces the reference's `MPConv.forward` (terrain_diffusion/models/mp_layers.py:201-221) together with
// the element-wise work `UNetBlock.forward` wraps around it (terrain_diffusion/models/unet_block.py:116-156):
// pixel-norm + mp_silu on the way in (fused into LDS staging), emb-scale + mp_silu / residual mp_sum + clip
// on the way out (fused into the epilogue), mp_concat + the 1x1 skip conv as extra K-segments.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (segment, 128-byte channel chunk, tap).
//   * MFMA A operand = weights (rows = couts), B operand = activations (cols = pixels), so each lane ends up
//     with 4 consecutive couts of one pixel -> packed NHWC stores.
//   * Activations: a (TH+2)x(TW+2) halo patch of one channel chunk is staged ONCE in LDS and reused by all
//     9 taps (tap = shifted LDS address); weights stream per tap from a pre-packed, pre-swizzled slab.
//   * LDS rows are 128 B (64 bf16 / 32 fp32 channels); the eight 16-B slots of a row are XOR-swizzled with
//     (row & 7) so the 16-lane ds_read_b128 groups hit 16 distinct slots of the 256-B bank row.
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  fp32: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "td_device.h"

namespace td {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int CHUNK = 32, PER16 = 4;
    static __device__ __forceinline__ float silu(float x) { return x / (1.f + expf(-x)) * (1.f / 0.596f); }
};
template <> struct Elem<__bf16> {
    static constexpr int CHUNK = 64, PER16 = 8;
    static __device__ __forceinline__ float silu(float x) {
        return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)) * (1.f / 0.596f);
    }
};

// mp_silu(s*x) on one 16-byte piece
template <typename T> __device__ __forceinline__ uint4 xform_piece(uint4 v, float s);
template <> __device__ __forceinline__ uint4 xform_piece<float>(uint4 v, float s) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = Elem<float>::silu(f[i] * s);
    return __builtin_bit_cast(uint4, f);
}
template <> __device__ __forceinline__ uint4 xform_piece<__bf16>(uint4 v, float s) {
    bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)Elem<__bf16>::silu((float)h[i] * s);
    return __builtin_bit_cast(uint4, h);
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
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const void* base, size_t idx) {
    bf16x4 h = *(const bf16x4*)((const __bf16*)base + idx);
    f32x4 f = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return f;
}

// Shared epilogue: transform 4 consecutive couts of one output pixel, store, return sum of squares of what was stored.
template <typename T>
__device__ __forceinline__ float epilogue4(const ConvParams& p, int n, int y, int x, int co, f32x4 v, float rn) {
    const int pix = (n * p.H + y) * p.W + x;
    if (p.epi == EPI_EMB_SILU) {
        f32x4 c = *(const f32x4*)(p.cvec + (size_t)n * p.cvec_stride + co);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = Elem<T>::silu(v[k] * c[k]);
    } else if (p.epi == EPI_RESIDUAL) {
        if (p.res) {
            int sp = src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample);
            f32x4 r = load4<T>(p.res, (size_t)sp * p.res_cstride + co);
            float s = p.res_scale * rn;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += s * r[k];
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
        } else {
            bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(bf16x4*)((__bf16*)p.out + (size_t)pix * p.out_cstride + co) = h;
#pragma unroll
            for (int k = 0; k < 4; ++k) { float f = (float)h[k]; ss += f * f; }
        }
    }
    return ss;
}

__device__ __forceinline__ float pixel_rn(const float* sumsq, int nparts, size_t npix, int sp, float inv_c) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += sumsq[(size_t)q * npix + sp];
    return 1.f / (1e-4f + sqrtf(s * inv_c));  // mp_layers.py:9-12 with dim=1: x / (eps + ||x||_c / sqrt(C))
}

template <typename T, int TH, int TW, int NIMG, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_kernel(const ConvParams p) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int TPIX = TH * TW;
    constexpr int BM = NIMG * TPIX;
    constexpr int PH = TH + 2, PW = TW + 2, PPI = PH * PW, NPATCH = NIMG * PPI;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 16, NTL = WN / 16;
    constexpr int CHUNK = Elem<T>::CHUNK, PER16 = Elem<T>::PER16;
    constexpr int A_ITERS = (NPATCH * 8 + NTHR - 1) / NTHR;
    constexpr int B_ITERS = (BN * 8) / NTHR;
    static_assert(BM % (16 * WAVES_M) == 0 && BN % (16 * WAVES_N) == 0 && (BN * 8) % NTHR == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_a = smem;
    unsigned char* s_b = smem + NPATCH * 128;
    float* s_rn = (float*)(s_b + 2 * BN * 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;

    int bid = blockIdx.x;
    const int ntile = bid % p.n_ntiles; bid /= p.n_ntiles;
    const int mtiles = p.tiles_x * p.tiles_y * p.img_groups;
    const int mtile = bid % mtiles;
    const int ksp = bid / mtiles;
    const int txi = mtile % p.tiles_x, tyi = (mtile / p.tiles_x) % p.tiles_y, ig = mtile / (p.tiles_x * p.tiles_y);
    const int n0 = ig * NIMG, y0 = tyi * TH, x0 = txi * TW, co0 = ntile * BN;
    const int g0 = (int)((long)ksp * p.kgroups / p.ksplit), g1 = (int)((long)(ksp + 1) * p.kgroups / p.ksplit);

    // ---- per-thread staging coordinates (constant over the K loop): packed (n, y+1, x+1, interior) or -1
    int a_coord[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        int e = tid + it * NTHR, pp = e >> 3;
        int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
        int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
        bool ok = (pp < NPATCH) && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
        bool interior = py >= 1 && py <= TH && px >= 1 && px <= TW;
        a_coord[it] = ok ? ((n << 21) | (y << 11) | (x << 1) | (interior ? 1 : 0)) : -1;
    }

    // ---- per-pixel 1/(eps + rms) of the pixel-normed source, for the patch pixels
    const float* rn_sumsq = nullptr; int rn_parts = 0, rn_Hs = 0, rn_Ws = 0, rn_res = 0; float rn_invc = 0.f;
    if (p.seg[0].xform == 2) { rn_sumsq = p.seg[0].sumsq; rn_parts = p.seg[0].nparts; rn_Hs = p.seg[0].Hs; rn_Ws = p.seg[0].Ws; rn_res = p.seg[0].resample; rn_invc = p.seg[0].inv_c; }
    else if (p.res_sumsq) { rn_sumsq = p.res_sumsq; rn_parts = p.res_nparts; rn_Hs = p.res_Hs; rn_Ws = p.res_Ws; rn_res = p.res_resample; rn_invc = p.res_inv_c; }
    if (rn_sumsq) {
        const size_t npix = (size_t)p.N * rn_Hs * rn_Ws;
        for (int pp = tid; pp < NPATCH; pp += NTHR) {
            int img = pp / PPI, r = pp % PPI, py = r / PW, px = r % PW;
            int n = n0 + img, y = y0 + py - 1, x = x0 + px - 1;
            float rn = 0.f;
            if (n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W)
                rn = pixel_rn(rn_sumsq, rn_parts, npix, src_pixel(n, y, x, rn_Hs, rn_Ws, rn_res), rn_invc);
            s_rn[pp] = rn;
        }
    }

    // ---- MFMA operand addressing
    int base_pp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int q = wm * WM + i * 16 + lr;
        int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        base_pp[i] = img * PPI + (ty + 1) * PW + (tx + 1);
    }
    int nloc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) nloc[j] = wn * WN + j * 16 + lr;

    // fp32 mode accumulates each K-group (one chunk x taps, <= 288 products) in a fresh accumulator and adds it to the running
    // total afterwards: a single k-ordered fp32 chain over K ~ 14k costs ~3e-6 rel. error per conv (measured 2.9e-5 per forward).
    constexpr bool TWO_LEVEL = sizeof(T) == 4;
    f32x4 acc[MT][NTL];
    f32x4 tot[TWO_LEVEL ? MT : 1][TWO_LEVEL ? NTL : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (TWO_LEVEL) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // ---- locate first group: (segment, chunk) and its first kstep
    int seg = 0, chunk = 0, kstep = 0;
    {
        int g = 0;
        while (seg < p.nseg) {
            int nch = p.seg[seg].C / CHUNK;
            if (g0 < g + nch) { chunk = g0 - g; kstep += chunk * p.seg[seg].taps; break; }
            g += nch; kstep += nch * p.seg[seg].taps; ++seg;
        }
    }

    uint4 breg[B_ITERS];
    auto load_b = [&](int ks_) {
        const uint4* wsrc = (const uint4*)p.wpack + ((size_t)ks_ * p.CoutPad + co0) * 8;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) breg[i] = wsrc[tid + i * NTHR];
    };
    if (g0 < g1) load_b(kstep);
    int buf = 0;

    for (int g = g0; g < g1; ++g) {
        const ConvSeg& sg = p.seg[seg];
        const int taps = sg.taps;
        // ---------------- stage the activation patch of (seg, chunk)
        {
            uint4 av[A_ITERS];
            const T* src = (const T*)sg.src;
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                int c = a_coord[it];
                av[it] = uint4{0u, 0u, 0u, 0u};
                if (c >= 0 && (taps == 9 || (c & 1))) {
                    int n = c >> 21, y = (c >> 11) & 1023, x = (c >> 1) & 1023;
                    int sp = src_pixel(n, y, x, sg.Hs, sg.Ws, sg.resample);
                    int slot = (tid + it * NTHR) & 7;
                    av[it] = *(const uint4*)(src + (size_t)sp * sg.cstride + chunk * CHUNK + slot * PER16);
                }
            }
            __syncthreads();  // everyone is done reading s_a (previous group's last tap) and s_rn is written
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                int e = tid + it * NTHR, pp = e >> 3, slot = e & 7;
                if (pp < NPATCH) {
                    uint4 v = av[it];
                    if (sg.xform != 0 && a_coord[it] >= 0) {
                        float s = sg.scale;
                        if (sg.xform == 2) s *= s_rn[pp];
                        v = xform_piece<T>(v, s);
                    }
                    *(uint4*)(s_a + pp * 128 + ((slot ^ (pp & 7)) << 4)) = v;
                }
            }
        }
        // ---------------- taps
        for (int tap = 0; tap < taps; ++tap) {
            unsigned char* sb = s_b + buf * (BN * 128);
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) *(uint4*)(sb + (tid + i * NTHR) * 16) = breg[i];
            ++kstep;
            const bool more = (tap + 1 < taps) || (g + 1 < g1);
            if (more) load_b(kstep);  // next kstep's weights fly during this tap's MFMAs
            __syncthreads();
            const int dy = (taps == 9) ? (tap / 3 - 1) : 0, dx = (taps == 9) ? (tap % 3 - 1) : 0;
            const int doff = dy * PW + dx;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kg = ks * 4 + lg;
                uint4 wf[NTL], xf[MT];
#pragma unroll
                for (int j = 0; j < NTL; ++j) wf[j] = *(const uint4*)(sb + nloc[j] * 128 + ((kg ^ (nloc[j] & 7)) << 4));
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    int pp = base_pp[i] + doff;
                    xf[i] = *(const uint4*)(s_a + pp * 128 + ((kg ^ (pp & 7)) << 4));
                }
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NTL; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NTL; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, wf[j])[e], __builtin_bit_cast(f32x4, xf[i])[e], acc[i][j], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
        if constexpr (TWO_LEVEL) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // next group
        if (++chunk == sg.C / CHUNK) { chunk = 0; ++seg; }
    }
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = tot[i][j];
    }

    // ---------------- epilogue
    const size_t M = (size_t)p.N * p.H * p.W;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int q = wm * WM + i * 16 + lr;
        int img = q / TPIX, r = q % TPIX, ty = r / TW, tx = r % TW;
        int n = n0 + img, y = y0 + ty, x = x0 + tx;
        bool ok = n < p.N && y < p.H && x < p.W;
        float ss = 0.f;
        if (ok) {
            if (p.ksplit > 1) {
                size_t pix = ((size_t)n * p.H + y) * p.W + x;
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    int co = co0 + wn * WN + j * 16 + lg * 4;
                    *(f32x4*)(p.partial + ((size_t)ksp * M + pix) * p.CoutPad + co) = acc[i][j];
                }
            } else {
                float rn = (p.res_sumsq != nullptr) ? s_rn[base_pp[i]] : 1.f;
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    int co = co0 + wn * WN + j * 16 + lg * 4;
                    ss += epilogue4<T>(p, n, y, x, co, acc[i][j], rn);
                }
            }
        }
        if (p.out_sumsq && p.ksplit == 1) {
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (ok && lg == 0) {
                size_t pix = ((size_t)n * p.H + y) * p.W + x;
                p.out_sumsq[(size_t)(ntile * WAVES_N + wn) * M + pix] = ss;
            }
        }
    }
}

// Split-K tail: sums the fp32 partial slabs in fixed order and applies the same epilogue. One wave per pixel.
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
    const int lane = threadIdx.x & 63;
    const size_t M = (size_t)p.N * p.H * p.W;
    const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= M) return;
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), n = (int)(pix / ((size_t)p.W * p.H));
    float rn = 1.f;
    if (p.res_sumsq)
        rn = pixel_rn(p.res_sumsq, p.res_nparts, (size_t)p.N * p.res_Hs * p.res_Ws, src_pixel(n, y, x, p.res_Hs, p.res_Ws, p.res_resample), p.res_inv_c);
    float ss = 0.f;
    for (int co = lane * 4; co < p.CoutPad; co += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.ksplit; ++k) {
            f32x4 t = *(const f32x4*)(p.partial + ((size_t)k * M + pix) * p.CoutPad + co);
            v += t;
        }
        ss += epilogue4<T>(p, n, y, x, co, v, rn);
    }
    if (p.out_sumsq) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) p.out_sumsq[pix] = ss;
    }
}

// ------------------------------------------------------------------------------------------ host launcher
template <typename T, int TH, int TW, int NIMG, int BN>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t st) {
    constexpr int WAVES_M = 2, WAVES_N = 2;
    constexpr int NPATCH = NIMG * (TH + 2) * (TW + 2);
    size_t lds = (size_t)NPATCH * 128 + 2 * BN * 128 + NPATCH * 4;
    int grid = p.n_ntiles * p.tiles_x * p.tiles_y * p.img_groups * p.ksplit;
    auto kern = conv_igemm_kernel<T, TH, TW, NIMG, BN, WAVES_M, WAVES_N>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (p.ksplit > 1) {
        size_t M = (size_t)p.N * p.H * p.W;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<T>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p);
        e = hipGetLastError();
    }
    return e;
}

// Tile geometry is chosen by the plan builder and recorded in the params (tiles_x/tiles_y/img_groups/n_ntiles);
// `narrow` selects the 8x8x2 tile for feature maps narrower than 16, `bn` the cout tile.
hipError_t launch_conv(const ConvParams& p, bool is_bf16, bool narrow, int bn, hipStream_t st) {
    if (is_bf16) {
        if (!narrow) return bn == 128 ? launch_cfg<__bf16, 8, 16, 1, 128>(p, st) : launch_cfg<__bf16, 8, 16, 1, 64>(p, st);
        return bn == 128 ? launch_cfg<__bf16, 8, 8, 2, 128>(p, st) : launch_cfg<__bf16, 8, 8, 2, 64>(p, st);
    }
    if (!narrow) return bn == 128 ? launch_cfg<float, 8, 16, 1, 128>(p, st) : launch_cfg<float, 8, 16, 1, 64>(p, st);
    return bn == 128 ? launch_cfg<float, 8, 8, 2, 128>(p, st) : launch_cfg<float, 8, 8, 2, 64>(p, st);
}

}  // namespace td
