// Output composition after the decoder: the memory-bound image operators behind WorldPipeline._compute_elev
// (terrain_diffusion/inference/world_pipeline.py:1276-1313) and laplacian_decode / laplacian_denoise
// (terrain_diffusion/data/laplacian_encoder.py:93-137) -- bilinear resize (align_corners=False, optional anti-aliasing), Gaussian blur with
// reflect padding, and the fused de-normalise + add + signed-square finish.  Resize and blur are ONE separable gather kernel pair driven by
// per-output-index tap tables (index + weight) that the host builds in the arithmetic torch / torchvision use; the kernels only apply them,
// horizontal pass first, then vertical, accumulating in tap order (the order torch's kernels use).
#include <hip/hip_runtime.h>

namespace td {

// tmp[c][y][xo] = sum_b wx[xo][b] * in[c][y][ix[xo][b]]          (one thread per output element)
__global__ void resample_rows_kernel(const float* __restrict__ in, float* __restrict__ tmp, int C, int Hin, int Win, int Wout,
                                     const int* __restrict__ ix, const float* __restrict__ wx, int Kx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * Hin * Wout) return;
    const int xo = (int)(i % Wout);
    const size_t row = i / Wout;  // c * Hin + y
    const float* src = in + row * Win;
    float acc = 0.f;
    for (int b = 0; b < Kx; ++b) {
        const float w = wx[xo * Kx + b];
        if (w != 0.f) acc += w * src[ix[xo * Kx + b]];
    }
    tmp[i] = acc;
}

// out[c][yo][xo] = sum_a wy[yo][a] * tmp[c][iy[yo][a]][xo]
__global__ void resample_cols_kernel(const float* __restrict__ tmp, float* __restrict__ out, int C, int Hin, int Hout, int Wout,
                                     const int* __restrict__ iy, const float* __restrict__ wy, int Ky) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * Hout * Wout) return;
    const int xo = (int)(i % Wout), yo = (int)((i / Wout) % Hout), c = (int)(i / ((size_t)Wout * Hout));
    const float* src = tmp + (size_t)c * Hin * Wout + xo;
    float acc = 0.f;
    for (int a = 0; a < Ky; ++a) {
        const float w = wy[yo * Ky + a];
        if (w != 0.f) acc += w * src[(size_t)iy[yo * Ky + a] * Wout];
    }
    out[i] = acc;
}

// world_pipeline.py:1300-1312 tail: elev_sqrt = residual_p + lowres_up (cropped), elev = sign(e) * e^2.
// residual_p is formed here from the packed decoder slice: (r0 / r1) * std + mean  (:1300)
__global__ void elev_finish_kernel(const float* __restrict__ packed /* (2,Hp,Wp) */, const float* __restrict__ lowres_up /* (Hp,Wp) */, float* __restrict__ out /* (h,w) */,
                                   int Hp, int Wp, int oi, int oj, int h, int w, float res_mean, float res_std) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w + oi, x = i % w + oj;
    const size_t p = (size_t)y * Wp + x;
    const float r = packed[p] / packed[(size_t)Hp * Wp + p] * res_std + res_mean;
    const float e = r + lowres_up[p];
    out[i] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * (e * e);
}

// residual_p + lowres_up over the whole padded window (input of the anti-aliased downsample inside laplacian_denoise)
__global__ void residual_plus_kernel(const float* __restrict__ packed, const float* __restrict__ lowres_up, float* __restrict__ out, int n, float res_mean, float res_std) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = packed[i] / packed[(size_t)n + i] * res_std + res_mean + lowres_up[i];
}

}  // namespace td
