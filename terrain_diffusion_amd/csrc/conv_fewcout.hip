// 3x3 convolution with ONE to FOUR output channels and an fp32 output (round 6): the decoder model's output conv, 64 -> 1 channels at 512 x 512.
//
// On the MFMA flavours such a launch computes a whole 64-cout tile for its one real cout: 64 x the arithmetic, 149 us per batch-4 launch for 134 MB of input
// (0.9 TB/s) -- matrix-pipe-bound on zeros.  Here a thread owns one output pixel and does the 9 x C multiply-adds of each of its couts on the VALU with
// v_dot2c_f32_bf16 / v_dot2_f32_f16 (two channels per instruction, fp32 accumulator, no conversions), from a halo patch in LDS: 288 instructions per pixel,
// cout and 64 channels (first build: 576 conversions + 288 v_pk_fma_f32 -- 84 us, VALU-bound).  16 x 16 pixel tiles, 256 threads, one 64-channel chunk of the
// 18 x 18 patch at a time (144-byte pixel rows: 9 x 16 bytes, an odd number of 16-byte columns, so the 16 lanes a ds_read_b128 services together fall on 16
// different columns of the bank window), the chunk's weights next to it, un-swizzled, read by every lane at the same address (broadcast).
//
// Same parameter block and packed weight slab ([K-step][cout][128 B], slot (piece ^ TD_SWZ(cout))) as the other flavours.  fp32 accumulation in another
// order than theirs (two chains of channel-pair dot products per cout): results agree to fp32 rounding of the sums, not bit for bit; the flavour of a
// launch depends on its shape only.  One 3x3 segment, no transform at staging, no resampling, EPI_PLAIN, fp32 output rows of p.out_cstride floats.
#include "conv_common.h"

namespace td {

template <typename T> __device__ __forceinline__ float fewcout_dot2(unsigned a, unsigned b, float c);
template <> __device__ __forceinline__ float fewcout_dot2<__bf16>(unsigned a, unsigned b, float c) {
    typedef __bf16 bx2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bx2, a), __builtin_bit_cast(bx2, b), c, false);
}
template <> __device__ __forceinline__ float fewcout_dot2<_Float16>(unsigned a, unsigned b, float c) {
    typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(hx2, a), __builtin_bit_cast(hx2, b), c, false);
}

template <typename T, int CO>
__global__ __launch_bounds__(256) void conv_fewcout_kernel(const ConvParams p) {
    constexpr int TW = 16, TH = 16, PW = 18, NPATCH = PW * PW, PITCH = 144, CHUNK = 64;
    constexpr int W_BASE = NPATCH * PITCH;                        // weights of the chunk: [tap][cout][eight 16-byte pieces]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    unsigned b = blockIdx.x;
    const int txi = (int)(b % (unsigned)p.tiles_x); b /= (unsigned)p.tiles_x;
    const int tyi = (int)(b % (unsigned)p.tiles_y); const int n0 = (int)(b / (unsigned)p.tiles_y);
    const int y0 = tyi * TH, x0 = txi * TW;
    const ConvSeg& sg = p.seg[0];
    const T* src = (const T*)sg.src;
    const int cs = sg.cstride, nch = sg.C / CHUNK, H = p.H, W = p.W;
    const unsigned char* wpack = (const unsigned char*)p.wpack;
    const size_t wstep = (size_t)p.CoutPad * 128;
    const int ly = tid >> 4, lx = tid & 15;
    float acc[CO][2];   // two partial sums per cout (pieces 0 - 3 / 4 - 7 of a pixel): independent dependency chains
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c][0] = acc[c][1] = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        if (ch) __syncthreads();   // everybody is done with the previous chunk
        // ---- the chunk's 18 x 18 x 64-channel patch: 16-byte pieces, zero outside the image.  All of a thread's loads first, then its LDS writes (first
        // build: a rolled loop of load -> wait -> write, i.e. eleven HBM round trips in series per workgroup: 84 / 67 us per launch instead of ~45)
        constexpr int NIT = (NPATCH * 8 + 255) / 256;   // 11
        u32x4 pv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256, pp = e >> 3, q = e & 7, py = pp / PW, px = pp - py * PW;
            const int y = y0 + py - 1, x = x0 + px - 1;
            pv[it] = u32x4{0u, 0u, 0u, 0u};
            if (e < NPATCH * 8 && y >= 0 && y < H && x >= 0 && x < W) pv[it] = *(const u32x4*)(src + ((size_t)((n0 * H + y) * W + x) * cs + ch * CHUNK + q * 8));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            if (e < NPATCH * 8) *(u32x4*)(smem + (e >> 3) * PITCH + (e & 7) * 16) = pv[it];
        }
        // ---- its weights: [tap][cout][piece], the slab's slot swizzle undone
        for (int e = tid; e < 9 * CO * 8; e += 256) {
            const int q = e & 7, c = (e >> 3) % CO, tap = (e >> 3) / CO;
            *(u32x4*)(smem + W_BASE + ((tap * CO + c) * 8 + q) * 16) = *(const u32x4*)(wpack + (size_t)(ch * 9 + tap) * wstep + (size_t)c * 128 + ((q ^ TD_SWZ(c)) << 4));
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const unsigned char* xp = smem + ((ly + tap / 3) * PW + lx + tap % 3) * PITCH;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u32x4 xv = *(const u32x4*)(xp + q * 16);
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const u32x4 wv = *(const u32x4*)(smem + W_BASE + ((tap * CO + c) * 8 + q) * 16);
                    float a = acc[c][q >> 2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a = fewcout_dot2<T>(xv[i], wv[i], a);
                    acc[c][q >> 2] = a;
                }
            }
        }
    }
    const int y = y0 + ly, x = x0 + lx;
    if (y < H && x < W) {
        float* o = (float*)p.out + (size_t)((n0 * H + y) * W + x) * p.out_cstride;
#pragma unroll
        for (int c = 0; c < CO; ++c) if (c < p.Cout) o[c] = acc[c][0] + acc[c][1];
    }
}

template <typename T, int CO>
static hipError_t launch_fewcout_cfg(const ConvParams& p, hipStream_t st) {
    constexpr size_t LDS = (size_t)18 * 18 * 144 + (size_t)9 * CO * 128;
    const long long grid = (long long)p.tiles_x * p.tiles_y * p.N;
    if (grid <= 0 || grid >= ((long long)1 << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_fewcout_kernel<T, CO>), dim3((unsigned)grid), dim3(256), LDS, st, p);
    return hipGetLastError();
}

// dtype: 1 bf16, 2 fp16.  Tiles: tiles_x = ceil(W / 16), tiles_y = ceil(H / 16), one image per tile.
hipError_t launch_conv_fewcout(const ConvParams& p, int dtype, hipStream_t st) {
    if (p.nseg != 1 || p.seg[0].taps != 9 || p.seg[0].xform != 0 || p.seg[0].resample != 0 || p.seg[0].Hs != p.H || p.seg[0].Ws != p.W || (p.seg[0].C & 63) || p.seg[0].C < 64)
        return hipErrorInvalidValue;
    if (p.epi != EPI_PLAIN || !p.out_f32 || p.ksplit != 1 || p.res || p.out2 || p.out_sumsq || p.clip > 0.f || p.Cout < 1 || p.Cout > 4 || p.out_cstride < p.Cout)
        return hipErrorInvalidValue;
    if (p.tiles_x != (p.W + 15) / 16 || p.tiles_y != (p.H + 15) / 16) return hipErrorInvalidValue;
    const int co = p.Cout <= 1 ? 1 : p.Cout <= 2 ? 2 : 4;
    if (dtype == 2) return co == 1 ? launch_fewcout_cfg<_Float16, 1>(p, st) : co == 2 ? launch_fewcout_cfg<_Float16, 2>(p, st) : launch_fewcout_cfg<_Float16, 4>(p, st);
    return co == 1 ? launch_fewcout_cfg<__bf16, 1>(p, st) : co == 2 ? launch_fewcout_cfg<__bf16, 2>(p, st) : launch_fewcout_cfg<__bf16, 4>(p, st);
}

}  // namespace td
