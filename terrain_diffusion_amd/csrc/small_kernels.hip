// Small / memory-bound kernels of the sampling path (gfx950): embeddings, attention core, scheduler step,
// portable-RNG noise field, overlap blend, layout conversion.  Each cites the reference lines it replaces.
#include "td_device.h"

namespace td {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mp_silu_acc(float x) { return x / (1.f + expf(-x)) * (1.f / 0.596f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ embeddings
// edm_unet.py:145-159 + mp_layers.py:96-131:
//   emb[row] = mp_silu( (noise_linear(posemb(t)) + sum_i w_i * e_i) / ||(1, w)|| ),
//   e_i = mp_silu(linear_i(cond_i)) for a "tensor" input, linear_i(cos(x*freqs_i + phases_i)*sqrt2) for a "float" input (no mp_silu).
// rows = (step, tile); t depends on step only, cond on tile only.  One workgroup per row, one wave per output.
struct EmbDesc {
    int n, row_len, feat_total;
    int type[8], dims[8], xoff[8], foff[8], woff[8], froff[8];
    float weight[8];
    float inv_norm;
};
__global__ __launch_bounds__(256) void emb_kernel(const float* __restrict__ t_steps, const float* __restrict__ cond, int n_tiles,
                                                  const float* __restrict__ freqs, int half, const float* __restrict__ w_noise,
                                                  const float* __restrict__ w_cond, const float* __restrict__ fourier, EmbDesc d, int emb_ch,
                                                  float* __restrict__ emb) {
    extern __shared__ float sh[];  // posemb[2*half] + cond features[feat_total]
    const int row = blockIdx.x, step = row / n_tiles, tile = row % n_tiles;
    const float t = t_steps[step];
    const int nd = 2 * half;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float y = t * freqs[i];
        sh[i] = sinf(y) * 1.41421356237309515f;
        sh[half + i] = cosf(y) * 1.41421356237309515f;
    }
    for (int c = 0; c < d.n; ++c) {
        const float* xr = cond + (size_t)tile * d.row_len + d.xoff[c];
        for (int i = threadIdx.x; i < d.dims[c]; i += blockDim.x) {
            float v;
            if (d.type[c] == 0) v = xr[i];
            else v = cosf(__fadd_rn(__fmul_rn(xr[0], fourier[d.froff[c] + i]), fourier[d.froff[c] + d.dims[c] + i])) * 1.41421356237309515f;  // mul then add, separately rounded like x.outer(freqs) + phases (|y| reaches ~50: an FMA would move cos by ~1e-6)
            sh[nd + d.foff[c] + i] = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // grid.y workgroups share a row's outputs in chunks of 64 (one workgroup per row took 0.4 ms for the 20 rows of a single-tile sampler call:
    // 192 dependent dot products per wave); every output is still one wave's sum in the same order, so the values do not change
    const int j_end = min(emb_ch, (int)(blockIdx.y + 1) * 64);
    for (int j = blockIdx.y * 64 + wave; j < j_end; j += blockDim.x / 64) {
        float a = 0.f;
        for (int k = lane; k < nd; k += 64) a += w_noise[(size_t)j * nd + k] * sh[k];
        a = wave_sum(a);
        for (int c = 0; c < d.n; ++c) {
            float b = 0.f;
            const float* w = w_cond + d.woff[c] + (size_t)j * d.dims[c];
            for (int k = lane; k < d.dims[c]; k += 64) b += w[k] * sh[nd + d.foff[c] + k];
            b = wave_sum(b);
            if (d.type[c] == 0) b = mp_silu_acc(b);
            a += d.weight[c] * b;
        }
        if (lane == 0) emb[(size_t)row * emb_ch + j] = mp_silu_acc(a * d.inv_norm);
    }
}

// unet_block.py:129-131: c = emb_linear(emb)*gain + 1 (gain folded into w).  grid (co tiles of 64, rows of 16, block index).
__global__ __launch_bounds__(256) void cvec_kernel(const float* __restrict__ emb, int rows, int emb_ch, const float* __restrict__ w_all,
                                                   const int* __restrict__ blk_woff, const int* __restrict__ blk_coff,
                                                   const int* __restrict__ blk_cout, int c_total, float* __restrict__ craw) {
    extern __shared__ float sh[];  // [16][emb_ch]
    const int blk = blockIdx.z, cout = blk_cout[blk];
    const int co_base = blockIdx.x * 64;
    if (co_base >= cout) return;
    const int r0 = blockIdx.y * 16, nr = min(16, rows - r0);
    for (int i = threadIdx.x; i < nr * emb_ch; i += blockDim.x) sh[i] = emb[(size_t)r0 * emb_ch + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* w = w_all + blk_woff[blk];
    for (int c = wave; c < 64; c += 4) {
        const int co = co_base + c;
        if (co >= cout) break;
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = lane; k < emb_ch; k += 64) {
            float wv = w[(size_t)co * emb_ch + k];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += wv * sh[r * emb_ch + k];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = wave_sum(acc[r]);
            if (lane == 0 && r < nr) craw[(size_t)(r0 + r) * c_total + blk_coff[blk] + co] = s + 1.f;
        }
    }
}

// The same product on the fp32 matrix cores (round 5): at the bench batch the scalar kernel above took 4.2 ms per sampler call (1280 rows x 14 k couts x 768:
// 27.5 GFLOP at 6.5 TFLOP/s -- 1.5 % of a configs[2] step for 0.01 % of its arithmetic), at batch 1 147 us (43 MB of weights at 290 GB/s).  Exact fp32 FMA
// chains (v_mfma_f32_16x16x4_f32); only the association of the 768 products of an output differs from the scalar kernel (a tree over lanes there, four
// interleaved chains here): ~1e-7 relative, far inside the fp32 mode's 5e-6.
// D[cout][row] = W[cout][k] E[row][k]: A operand = W (row = cout lane & 15), B operand = E (column = row lane & 15); the contraction index is taken in
// the order that makes every operand load a 16-byte piece: lane group g = lane >> 4 of super-step m holds k = 16 m + 4 g ... + 3, MFMA t of the
// super-step contracts {16 m + 4 g + t : g = 0..3} (any partition of K into fours is as good as any other as long as both operands agree).
// A wave: 16 couts x 64 rows (one A piece, four B pieces, 16 MFMAs per super-step, everything straight from global memory / L2: the four waves of a
// workgroup take four cout slices of the same 64 rows).  grid (cout tiles of 64, row tiles of 64, block index).
// U super-steps (16 U channels) are requested at a time, one block ahead of the MFMAs that use them: at batch 1 a wave streams its 48 KB of weights with
// 5 U KiB in flight (the single-tile call is bound by that stream: 43 MB of fp32 weights for 20 rows).  emb_ch % (16 U) == 0.
template <int U>
__global__ __launch_bounds__(256) void cvec_mfma_kernel(const float* __restrict__ emb, int rows, int emb_ch, const float* __restrict__ w_all,
                                                        const int* __restrict__ blk_woff, const int* __restrict__ blk_coff,
                                                        const int* __restrict__ blk_cout, int c_total, float* __restrict__ craw) {
    const int blk = blockIdx.z, cout = blk_cout[blk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int co0 = blockIdx.x * 64 + wave * 16;
    if (co0 >= cout) return;   // (couts are multiples of 16: a 16-cout slice is inside the block or outside it)
    const int r0 = blockIdx.y * 64;
    const float* wrow = w_all + blk_woff[blk] + (size_t)(co0 + li) * emb_ch + 4 * lg;
    const float* erow[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) { const int r = r0 + nb * 16 + li; erow[nb] = emb + (size_t)(r < rows ? r : rows - 1) * emb_ch + 4 * lg; }
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a_next[U], b_next[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a_next[u] = *(const f32x4*)(wrow + 16 * u);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) b_next[u][nb] = *(const f32x4*)(erow[nb] + 16 * u);
    }
    for (int k = 0; k < emb_ch; k += 16 * U) {
        f32x4 a[U], b[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = a_next[u];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) b[u][nb] = b_next[u][nb];
        }
        const int kn = k + 16 * U < emb_ch ? k + 16 * U : k;   // (the last block re-reads itself: the loads stay unconditional)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a_next[u] = *(const f32x4*)(wrow + kn + 16 * u);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) b_next[u][nb] = *(const f32x4*)(erow[nb] + kn + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b[u][nb][t], acc[nb], 0, 0, 0);
    }
    // D: row (cout) 4 (lane >> 4) + r, column (row of c) lane & 15: four consecutive couts of one row per lane
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int r = r0 + nb * 16 + li;
        if (r < rows) *(f32x4*)(craw + (size_t)r * c_total + blk_coff[blk] + co0 + 4 * lg) = acc[nb] + 1.f;
    }
}

// c /= sqrt(mean(c^2) + 1e-8) per (row, block).  grid (rows, blocks)
__global__ __launch_bounds__(256) void cvec_norm_kernel(float* __restrict__ c, const int* __restrict__ blk_coff, const int* __restrict__ blk_cout, int c_total) {
    __shared__ float red[4];
    const int blk = blockIdx.y, cout = blk_cout[blk];
    float* v = c + (size_t)blockIdx.x * c_total + blk_coff[blk];
    float s = 0.f;
    for (int i = threadIdx.x; i < cout; i += blockDim.x) s += v[i] * v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / sqrtf(tot / (float)cout + 1e-8f);
    for (int i = threadIdx.x; i < cout; i += blockDim.x) v[i] *= inv;
}

// ------------------------------------------------------------------------------------------------ attention core
// unet_block.py:102-108.  qkv is NHWC with channel = (head*64 + c)*3 + {q,k,v}; per-token unit-RMS norm of q,k,v over c,
// logits/sqrt(64), softmax over keys, out[token][head*64 + c].  One workgroup per (image, head); tokens <= 64.
template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int C) {
    __shared__ float sq[64][65], sk[64][65], sv[64][65], sw[64][65];
    const int n = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
    const size_t base = (size_t)n * tokens * 3 * C;
    for (int e = tid; e < tokens * 64; e += 256) {
        int tok = e >> 6, c = e & 63;
        const T* ptr = qkv + base + (size_t)tok * 3 * C + (head * 64 + c) * 3;
        sq[tok][c] = (float)ptr[0]; sk[tok][c] = (float)ptr[1]; sv[tok][c] = (float)ptr[2];
    }
    __syncthreads();
    // normalize(y, dim=2): x / (1e-4 + ||x||_c / sqrt(64)), for q, k and v of every token
    if (tid < 3 * 64) {
        int which = tid >> 6, tok = tid & 63;
        if (tok < tokens) {
            float (*m)[65] = which == 0 ? sq : (which == 1 ? sk : sv);
            float s = 0.f;
            for (int c = 0; c < 64; ++c) s += m[tok][c] * m[tok][c];
            float inv = 1.f / (1e-4f + sqrtf(s) * 0.125f);
            if (which == 1) inv *= 0.125f;  // k / sqrt(64)
            for (int c = 0; c < 64; ++c) m[tok][c] *= inv;
        }
    }
    __syncthreads();
    const int q = tid >> 2, part = tid & 3;  // 4 threads per query row
    float mx = -3.0e38f;
    if (q < tokens) {
        for (int k = part; k < tokens; k += 4) {
            float s = 0.f;
            for (int c = 0; c < 64; ++c) s += sq[q][c] * sk[k][c];
            sw[q][k] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
    float den = 0.f;
    if (q < tokens) {
        for (int k = part; k < tokens; k += 4) { float e = expf(sw[q][k] - mx); sw[q][k] = e; den += e; }
    }
    den += __shfl_xor(den, 1); den += __shfl_xor(den, 2);
    __syncthreads();
    if (q < tokens) {
        const float inv = 1.f / den;
        for (int c = part * 16; c < part * 16 + 16; ++c) {
            float s = 0.f;
            for (int k = 0; k < tokens; ++k) s += sw[q][k] * sv[k][c];
            out[((size_t)n * tokens + q) * C + head * 64 + c] = (T)(s * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------ layout / scheduler
// x: planar fp32 [N][C][HW]; xin: NHWC T [N][HW][cstride] = (x*scale, 1, 0...) — edm_unet.py:168 ones channel.
// `ones_idx` = in_channels: the bias-surrogate channel sits after ALL input channels (sample + conditioning image).
template <typename T>
__global__ void prep_input_kernel(const float* __restrict__ x, T* __restrict__ xin, int N, int C, int HW, int cstride, float scale, int ones_idx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < C; ++c) xin[i * cstride + c] = (T)(x[((size_t)n * C + c) * HW + p] * scale);
    xin[i * cstride + ones_idx] = (T)1.f;
}
// conditioning-image channels of the model input (constant over the solver steps): xin[.., c0 + c] = img[n][c][p]
template <typename T>
__global__ void write_cond_img_kernel(const float* __restrict__ img, T* __restrict__ xin, int N, int Cc, int HW, int cstride, int c0) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < Cc; ++c) xin[i * cstride + c0 + c] = (T)img[((size_t)n * Cc + c) * HW + p];
}

// F: NHWC fp32 [N][HW][fstride] -> planar [N][C][HW]
__global__ void unpack_output_kernel(const float* __restrict__ F, float* __restrict__ y, int N, int C, int HW, int fstride, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < C; ++c) y[((size_t)n * C + c) * HW + p] = F[i * fstride + c] * scale;
}

// One DPM-Solver++(2M) step (dpmsolver.py:245-258, 454-561, 650-726) fused with the next step's input
// preconditioning (dpmsolver.py:226-229).  x, m1 planar fp32 [N][C][HW]; F NHWC fp32.
template <typename T>
__global__ void dpm_step_kernel(float* __restrict__ x, float* __restrict__ m1, const float* __restrict__ F, T* __restrict__ xin,
                                int N, int C, int HW, int fstride, int cstride, SchedCoef k, const float* __restrict__ Fg, float gscale,
                                T* __restrict__ xin2, float* __restrict__ m2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < C; ++c) {
        size_t xi = ((size_t)n * C + c) * HW + p;
        float xs = x[xi];
        float f = F[i * fstride + c];
        if (Fg) { const float g = Fg[i * fstride + c]; f = g + gscale * (f - g); }  // autoguidance (sample_diffusion_base.py:107-110,155-160)
        float xn, m0;
        const float m1v = (k.order == 1 && !m2) ? 0.f : m1[xi];
        dpm_update(k, xs, f, m1v, k.order == 3 ? m2[xi] : 0.f, xn, m0);
        x[xi] = xn;
        if (m2) m2[xi] = m1v;   // third-order solver: history shifts m2 <- m1 <- m0
        m1[xi] = m0;
        if (!k.last) {
            xin[i * cstride + c] = (T)(xn * k.c_in_next);
            if (xin2) xin2[i * cstride + c] = (T)(xn * k.c_in_next);  // the guide model's input buffer
        }
    }
}

// Trig-flow consistency step (world_pipeline.py:1097-1129 / sample_diffusion_base.py:251-257):
// pre:  x_t = cos t * sample + sin t * sigma_d * z ; xin = x_t / sigma_d ;  post: out = cos t * x_t - sin t * sigma_d * (-F)
template <typename T>
__global__ void consistency_pre_kernel(const float* __restrict__ sample, const float* __restrict__ z, float* __restrict__ xt, T* __restrict__ xin,
                                       int N, int C, int HW, int cstride, float cos_t, float sin_t, float sigma_data, int ones_idx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < C; ++c) {
        size_t xi = ((size_t)n * C + c) * HW + p;
        float v = cos_t * sample[xi] + sin_t * (z[xi] * sigma_data);
        xt[xi] = v;
        xin[i * cstride + c] = (T)(v / sigma_data);
    }
    xin[i * cstride + ones_idx] = (T)1.f;
}
__global__ void consistency_post_kernel(const float* __restrict__ xt, const float* __restrict__ F, float* __restrict__ out, int N, int C, int HW,
                                        int fstride, float cos_t, float sin_t, float sigma_data) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW) return;
    int n = (int)(i / HW), p = (int)(i % HW);
    for (int c = 0; c < C; ++c) {
        size_t xi = ((size_t)n * C + c) * HW + p;
        float pred = -F[i * fstride + c];
        out[xi] = cos_t * xt[xi] - sin_t * sigma_data * pred;
    }
}

// ------------------------------------------------------------------------------------------------ DDIM + classifier-free guidance (configs[0])
// annotated_infinite_panorama.py:130-134 in one pass: pred = uncond + g (cond - uncond); x0 = (x - sqrt(1 - a_t) pred) / sqrt(a_t);
// out = sqrt(a_prev) x0 + sqrt(1 - a_prev) pred (DDIM, eta = 0; Song et al. 2021 eq. 12 as diffusers' DDIMScheduler.step runs it for the SD-v1.5
// config).  The four scalars are computed on the host from the schedule (terrain_diffusion_amd/pano.py).
__global__ void ddim_cfg_step_kernel(const float* __restrict__ x, const float* __restrict__ uncond, const float* __restrict__ cond, float* __restrict__ out, size_t n,
                                     float g, float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev, float sqrt_one_minus_aprev) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = uncond[i], e = u + g * (cond[i] - u);
    const float x0 = (x[i] - sqrt_one_minus_at * e) / sqrt_at;
    out[i] = sqrt_aprev * x0 + sqrt_one_minus_aprev * e;
}

// ------------------------------------------------------------------------------------------------ portable noise
// portable_rng.py:22-74: 64-bit LCG, XSH-RR output of the post-advance state, Marsaglia polar with rejection.
// Parallel form: thread t of a workgroup owns candidate pairs [round*R + t*PPT, +PPT) via LCG jump-ahead; accepted
// pairs are compacted in candidate order with a block prefix scan, so the output equals the sequential stream.
#define PCG_MULT 6364136223846793005ULL
#define PCG_INC 1442695040888963407ULL

__device__ __forceinline__ void lcg_jump(uint64_t n, uint64_t& mul, uint64_t& add) {
    uint64_t cm = PCG_MULT, ca = PCG_INC;
    mul = 1; add = 0;
    while (n) {
        if (n & 1) { mul *= cm; add = add * cm + ca; }
        ca = (cm + 1) * ca; cm *= cm; n >>= 1;
    }
}
__device__ __forceinline__ uint32_t pcg_out(uint64_t s) {
    uint32_t x = (uint32_t)(((s >> 18) ^ s) >> 27), rot = (uint32_t)(s >> 59);
    return (x >> rot) | (x << ((32u - rot) & 31u));
}

// one workgroup (256 threads) per noise tile; out[tile] has n floats
__global__ __launch_bounds__(256) void noise_tiles_kernel(const uint64_t* __restrict__ seeds, float* __restrict__ out, int64_t n) {
    constexpr int PPT = 8, NTH = 256, R = PPT * NTH;
    __shared__ int s_scan[NTH];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    float* o = out + (size_t)blockIdx.x * n;
    uint64_t jm, ja, rm, ra;
    lcg_jump((uint64_t)(2 * PPT) * tid, jm, ja);
    lcg_jump((uint64_t)(2 * R), rm, ra);
    uint64_t state = seeds[blockIdx.x] * jm + ja;  // state before this thread's first candidate pair
    if (tid == 0) s_base = 0;
    __syncthreads();
    const double inv = 1.0 / 4294967296.0;
    while (true) {
        const int base = s_base;  // accepted values so far
        if (base >= n) break;
        float v1o[PPT], v2o[PPT];
        unsigned accept = 0;
        uint64_t s = state;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            s = s * PCG_MULT + PCG_INC; uint32_t u1 = pcg_out(s);
            s = s * PCG_MULT + PCG_INC; uint32_t u2 = pcg_out(s);
            double v1 = 2.0 * ((double)u1 + 1.0) * inv - 1.0;
            double v2 = 2.0 * ((double)u2 + 1.0) * inv - 1.0;
            double r = v1 * v1 + v2 * v2;
            if (r > 0.0 && r < 1.0) {
                double f = sqrt(-2.0 * log(r) / r);
                v1o[j] = (float)(v1 * f); v2o[j] = (float)(v2 * f);
                accept |= 1u << j;
            }
        }
        state = state * rm + ra;
        const int cnt = __popc(accept);
        s_scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < NTH; off <<= 1) {  // Hillis-Steele inclusive scan
            int v = (tid >= off) ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        int64_t pos = base + 2 * (int64_t)(s_scan[tid] - cnt);
        const int total = s_scan[NTH - 1];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            if (accept & (1u << j)) {
                if (pos < n) o[pos] = v1o[j];
                if (pos + 1 < n) o[pos + 1] = v2o[j];
                pos += 2;
            }
        }
        __syncthreads();
        if (tid == 0) s_base = base + 2 * total;
        __syncthreads();
    }
}

// window gather: out[win][c][y][x] = tile(ty,tx)[c][yy][xx] (world_pipeline.py:66-115); tile_index maps (win, dy, dx) of the
// up-to-2x2 touched tiles to a slot in `tiles`.  scale multiplies the noise (e.g. sigma_0).
__global__ void noise_gather_kernel(const float* __restrict__ tiles, const int* __restrict__ tile_index, const int* __restrict__ origins,
                                    float* __restrict__ out, int C, int h, int w, int tile_h, int tile_w, float scale) {
    const int win = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * h * w) return;
    const int x = i % w, y = (i / w) % h, c = i / (w * h);
    const int gy = origins[2 * win] + y, gx = origins[2 * win + 1] + x;
    auto fdiv = [](int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; };
    const int ty = fdiv(gy, tile_h), tx = fdiv(gx, tile_w);
    const int ty0 = fdiv(origins[2 * win], tile_h), tx0 = fdiv(origins[2 * win + 1], tile_w);
    const int slot = tile_index[win * 4 + (ty - ty0) * 2 + (tx - tx0)];
    const int yy = gy - ty * tile_h, xx = gx - tx * tile_w;
    out[(size_t)win * C * h * w + i] = tiles[((size_t)slot * C + c) * tile_h * tile_w + yy * tile_w + xx] * scale;
}

// ------------------------------------------------------------------------------------------------ overlap blend
// sample_diffusion_base.py:164-168 / annotated_infinite_panorama.py:145-150.  Deterministic gather form: every canvas pixel sums its
// covering windows in ascending (row-window, col-window) order, which is the reference's loop order, so the fp32 sum is bit-identical.
// canvas: planar fp32 [(C+1)][Hc][Wc]; rowmap/colmap: per canvas row/col up to 4 covering window indices (-1 padded);
// tile_of: window grid -> slot in x (or -1 if that window is not present in this launch).
__global__ void blend_gather_kernel(const float* __restrict__ x, const float* __restrict__ wwin, float* __restrict__ canvas, int C, int Hc, int Wc,
                                    int size, const int* __restrict__ rowmap, const int* __restrict__ colmap, const int* __restrict__ row_start,
                                    const int* __restrict__ col_start, const int* __restrict__ tile_of, int n_wcols, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Hc * Wc) return;
    const int y = i / Wc, xq = i % Wc;
    float acc[8];
    for (int c = 0; c <= C; ++c) acc[c] = accumulate ? canvas[(size_t)c * Hc * Wc + i] : 0.f;
    for (int a = 0; a < 4; ++a) {
        const int ic = rowmap[y * 4 + a];
        if (ic < 0) continue;
        for (int b = 0; b < 4; ++b) {
            const int jc = colmap[xq * 4 + b];
            if (jc < 0) continue;
            const int slot = tile_of[ic * n_wcols + jc];
            if (slot < 0) continue;
            const int ly = y - row_start[ic], lx = xq - col_start[jc];
            const float w = wwin[ly * size + lx];
            for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(x[(((size_t)slot * C + c) * size + ly) * size + lx], w, acc[c]);   // spelled out: regions_gather_kernel must round identically
            acc[C] += w;
        }
    }
    for (int c = 0; c <= C; ++c) canvas[(size_t)c * Hc * Wc + i] = acc[c];
}

// Many equally sized regions of ONE window tensor in one launch (round 4: the lazy graph assembles the argument slices of a whole batch of
// windows at once -- td_blend_windows per slice cost six device allocations, six uploads and a stream synchronisation each, ~130 calls per
// batch of 64 latent windows, and left the GPU idle 43 % of a cascade step).  Region r is (C+1, h, w); its candidate windows are listed in
// desc[r][0 .. maxk) as (slot into `wins`, y of the window's first row relative to the region, x likewise), slot < 0 = end of list, in
// ascending (row, col) window order: the same per-pixel summation order and arithmetic as blend_gather_kernel, so the bits are the same.
__global__ void regions_gather_kernel(const float* const* __restrict__ wins, const int* __restrict__ desc, int maxk, const float* __restrict__ wwin,
                                      float* __restrict__ out, int C, int h, int w, int size) {
    const int r = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, xq = i % w;
    float acc[8];
    for (int c = 0; c <= C; ++c) acc[c] = 0.f;
    const int* d = desc + (size_t)r * maxk * 3;
    for (int k = 0; k < maxk; ++k) {
        const int slot = d[k * 3];
        if (slot < 0) break;
        const int ly = y - d[k * 3 + 1], lx = xq - d[k * 3 + 2];
        if (ly < 0 || ly >= size || lx < 0 || lx >= size) continue;
        const float wt = wwin[ly * size + lx];
        const float* x = wins[slot];
        for (int c = 0; c < C; ++c) acc[c] = __builtin_fmaf(x[((size_t)c * size + ly) * size + lx], wt, acc[c]);
        acc[C] += wt;
    }
    float* o = out + (size_t)r * (C + 1) * h * w;
    for (int c = 0; c <= C; ++c) o[(size_t)c * h * w + i] = acc[c];
}

// out[c] = canvas[c] / canvas[C] * scale   (normalise-on-read)
__global__ void blend_normalize_kernel(const float* __restrict__ canvas, float* __restrict__ out, int C, int HW, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float wsum = canvas[(size_t)C * HW + i];
    for (int c = 0; c < C; ++c) out[(size_t)c * HW + i] = canvas[(size_t)c * HW + i] / wsum * scale;
}

// window extraction from a normalised planar canvas into per-tile samples (phase input for multi-phase sampling)
__global__ void window_extract_kernel(const float* __restrict__ img, float* __restrict__ tiles, const int* __restrict__ origins, int C, int Hc, int Wc, int size) {
    const int win = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * size * size) return;
    const int x = i % size, y = (i / size) % size, c = i / (size * size);
    tiles[(size_t)win * C * size * size + i] = img[((size_t)c * Hc + origins[2 * win] + y) * Wc + origins[2 * win + 1] + x];
}

// explicit instantiations used by the engine
template __global__ void attn_kernel<float>(const float*, float*, int, int);
template __global__ void attn_kernel<__bf16>(const __bf16*, __bf16*, int, int);
template __global__ void attn_kernel<_Float16>(const _Float16*, _Float16*, int, int);
template __global__ void prep_input_kernel<float>(const float*, float*, int, int, int, int, float, int);
template __global__ void prep_input_kernel<__bf16>(const float*, __bf16*, int, int, int, int, float, int);
template __global__ void prep_input_kernel<_Float16>(const float*, _Float16*, int, int, int, int, float, int);
template __global__ void write_cond_img_kernel<float>(const float*, float*, int, int, int, int, int);
template __global__ void write_cond_img_kernel<__bf16>(const float*, __bf16*, int, int, int, int, int);
template __global__ void write_cond_img_kernel<_Float16>(const float*, _Float16*, int, int, int, int, int);
template __global__ void dpm_step_kernel<float>(float*, float*, const float*, float*, int, int, int, int, int, SchedCoef, const float*, float, float*, float*);
template __global__ void dpm_step_kernel<__bf16>(float*, float*, const float*, __bf16*, int, int, int, int, int, SchedCoef, const float*, float, __bf16*, float*);
template __global__ void dpm_step_kernel<_Float16>(float*, float*, const float*, _Float16*, int, int, int, int, int, SchedCoef, const float*, float, _Float16*, float*);
template __global__ void consistency_pre_kernel<float>(const float*, const float*, float*, float*, int, int, int, int, float, float, float, int);
template __global__ void consistency_pre_kernel<__bf16>(const float*, const float*, float*, __bf16*, int, int, int, int, float, float, float, int);
template __global__ void consistency_pre_kernel<_Float16>(const float*, const float*, float*, _Float16*, int, int, int, int, float, float, float, int);

}  // namespace td
