// Device-side parameter blocks shared by the host engine and the HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace td {

// One K-segment of an implicit-GEMM convolution: a source activation tensor (NHWC), how it is
// resampled onto the conv grid, and the element-wise transform fused into LDS staging.
struct ConvSeg {
    const void* src;      // NHWC activations, element type T of the kernel instantiation
    const float* sumsq;   // [nparts][Nimg*Hs*Ws] partial per-pixel sums of squares (pixel-norm), or null
    int C;                // channels taken from src (multiple of the K-chunk)
    int cstride;          // elements per pixel in src
    int Hs, Ws;           // source spatial dims
    int taps;             // 9 (3x3, pad 1) or 1 (1x1)
    int resample;         // 0 keep, 1 down (src[2y,2x]), 2 up (src[y/2,x/2])
    int xform;            // 0 none, 1 mp_silu(scale*x), 2 mp_silu(scale*rn(pixel)*x)
    int nparts;           // number of sumsq partials
    float scale;
    float inv_c;          // 1/C_total for the pixel norm
};

struct SchedCoef {        // one DPM-Solver++ step, fp32 scalars computed on the host
    float c_skip, c_out;  // x0 = c_skip*x + c_out*F
    float a, b0, inv_r0;  // order1: a*x - b0*m0 ; order2: ... - 0.5*b0*inv_r0*(m0-m1)
    float inv_r1, f01, inv_r01, c1, c2;  // order3 (dpmsolver.py:598-613): D1_0 = inv_r0 (m0-m1), D1_1 = inv_r1 (m1-m2), D1 = D1_0 + f01 (D1_0-D1_1),
                                         // D2 = inv_r01 (D1_0-D1_1); x = a*x - b0*m0 + c1*D1 - c2*D2
    float c_in_next;      // next model input scale (0 on the last step)
    int order;
    int last;
};

// One DPM-Solver++ (1st / 2nd order multistep) update of one value: x0 prediction m0 = c_skip x + c_out F, then the exponential-integrator step
// (dpmsolver.py:245-258, 472-482, 515-540).  Shared by dpm_step_kernel and the fused conv epilogue (EPI_DPM_STEP); the fused multiply-adds are
// spelled out so that both places round identically (the compiler's own contraction choices differ from context to context).
__device__ __forceinline__ void dpm_update(const SchedCoef& k, float xs, float f, float m1v, float m2v, float& xn, float& m0) {
    m0 = __builtin_fmaf(k.c_skip, xs, k.c_out * f);
    const float base = __builtin_fmaf(k.a, xs, -(k.b0 * m0));
    if (k.order == 1) xn = base;
    else if (k.order == 2) xn = __builtin_fmaf(-(0.5f * k.b0), k.inv_r0 * (m0 - m1v), base);
    else {  // third-order multistep update (dpmsolver.py:563-615)
        const float d10 = k.inv_r0 * (m0 - m1v), d11 = k.inv_r1 * (m1v - m2v), dd = d10 - d11;
        const float d1 = __builtin_fmaf(k.f01, dd, d10), d2 = k.inv_r01 * dd;
        xn = __builtin_fmaf(-k.c2, d2, __builtin_fmaf(k.c1, d1, base));
    }
}

enum { EPI_PLAIN = 0, EPI_EMB_SILU = 1, EPI_RESIDUAL = 2, EPI_DPM_STEP = 3 };

struct ConvParams {
    ConvSeg seg[3];
    int nseg;
    const void* wpack;    // packed weights [kstep][CoutPad][128 B], 16-B slots XOR-swizzled by (cout&7)
    int N, H, W;          // conv-resolution output dims
    int Cout, CoutPad;
    int tiles_x, tiles_y, img_groups, n_ntiles;
    int kgroups;          // total (segment, chunk) groups
    int ksplit;           // number of K splits (1 = direct epilogue)
    unsigned char kb[66]; // split-K slice s reduces K-groups [kb[s], kb[s+1]) (conv_set_kbounds; ksplit <= 64, kgroups <= 255)
    int epi;
    void* out;            // NHWC output (T, or float when out_f32)
    int out_cstride;
    int out_f32;
    const float* cvec;    // EPI_EMB_SILU: c[n*cvec_stride + co]
    int cvec_stride;
    const void* res;      // EPI_RESIDUAL: residual source (T) or null
    const float* res_sumsq;
    int res_cstride, res_Hs, res_Ws, res_resample, res_nparts;
    float res_inv_c;
    float res_scale;      // 0.7/sqrt(0.58) (conv weights carry 0.3/sqrt(0.58))
    float clip;           // <=0: no clip
    float* out_sumsq;     // [parts][N*H*W] or null; parts: CoutPad/32 (conv_glds / conv_sb), CoutPad/16 (conv_s16), n_ntiles*WAVES_N (conv_igemm), CoutPad/256 (split-K reduce)
    float* partial;       // split-K workspace [ksplit][N*H*W][CoutPad] fp32
    void* out2;           // optional second output, same layout as out: mp_silu(out2_scale * out) -- the consumer's activation, done once here
    float out2_scale;
    const void* zeros;    // >= 16 zero bytes in device memory: a readable page of zeros (was the halo source of the removed ping-pong flavour's LDS-DMA patch staging)
    // EPI_DPM_STEP (the U-Net's output conv inside the EDM sampler): the DPM-Solver++ update and the next step's input preconditioning run in
    // this conv's epilogue instead of a separate pass over F (dpmsolver.py:226-258, 454-561, 650-726) -- the conv result is the model output F
    float* dpm_x;         // sample, planar fp32 [N][Cout][H*W], updated in place
    float* dpm_m1;        // previous x0 prediction (multistep history), same layout, updated in place
    float* dpm_m2;        // the one before (third-order solver only), or null
    void* dpm_xin;        // next step's model input, NHWC T with dpm_xin_cstride elements per pixel: channels [0, Cout) = x_new * c_in_next
    int dpm_xin_cstride;
    SchedCoef dpm_k;
    int reverse;          // scheduling hint (speed only, never changes a bit): 1 = logical workgroup ids are walked backwards (the host alternates it
                          // from layer to layer: the producer's last-written, still cached rows are read first)
    int dma1x1;           // conv_glds: stream 1x1 segments by LDS-DMA when the launch qualifies (launch_glds_cfg decides)
    int persist;          // conv_glds_wide.hip: > 0 = persistent tile loop with this many workgroups (the stride of a workgroup's tile walk); 0 = one tile per workgroup
    // conv_sb.hip (small-batch flavour): the same weights in MFMA-fragment order [K-group][16-channel slice][tap][32-cout tile][lane][16 B]
    const void* wpack_sb;
    int sb_n3;            // number of leading 3x3 K-groups (every 3x3 segment precedes every 1x1 segment)
    int sb_order;         // workgroup order (speed only): 0 = cout tiles of a pixel tile adjacent, 1 = pixel tiles of a cout tile adjacent
    unsigned sb_d0, sb_m0, sb_d1, sb_m1, sb_m2, sb_m3, sb_grid8, sb_grid;   // (conv_glds.hip uses the same fields: d0 = pixel tiles, d1 = cout tiles, grid)
// launcher-made: first divisor of the workgroup-id decomposition, the magic multipliers (sb_udiv), grid / 8 (0 if 8 does not divide it)
};

// Split-K slice boundaries.  Uniform K-groups: the floor split s*kgroups/ksplit.  A mix of 3x3 and 1x1 groups (the decoder's conv_res1: 3x3 conv
// + 1x1 skip conv in one launch) is split by K-STEPS (a 3x3 group is 9 of them, a 1x1 group 1) when `weighted`: an even count of groups would give
// one slice nearly all of the work (12 3x3 + 6 1x1 groups | 18 1x1 groups = 114 | 18 steps).  Every slice keeps at least one group.
inline bool conv_set_kbounds(ConvParams& p, bool weighted, int chunk = 64) {
    if (p.ksplit < 1 || p.ksplit > 64 || p.kgroups > 255 || p.ksplit > p.kgroups) return false;
    int wt[256], n = 0, total = 0; bool mixed = false;
    for (int s = 0; s < p.nseg; ++s)
        for (int c = 0; c < p.seg[s].C / chunk && n < 256; ++c) { wt[n] = p.seg[s].taps; mixed = mixed || wt[n] != wt[0]; total += wt[n]; ++n; }
    if (n != p.kgroups) return false;
    p.kb[0] = 0; p.kb[p.ksplit] = (unsigned char)n;
    int g = 0, acc = 0;
    for (int s = 1; s < p.ksplit; ++s) {
        int b;
        if (!(weighted && mixed)) b = (int)((long)s * n / p.ksplit);
        else {
            const long target = (long)s * total;   // in units of 1/ksplit K-steps
            while (g < n && 2 * ((long)acc * p.ksplit - target) + (long)wt[g] * p.ksplit <= 0) { acc += wt[g]; ++g; }   // nearest prefix to the target
            b = g;
        }
        b = b < p.kb[s - 1] + 1 ? p.kb[s - 1] + 1 : b;
        b = b > n - (p.ksplit - s) ? n - (p.ksplit - s) : b;
        p.kb[s] = (unsigned char)b;
        if (weighted && mixed) { while (g < b) { acc += wt[g]; ++g; } }
    }
    return true;
}



}  // namespace td
