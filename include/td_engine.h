/*
 * td_engine.h — C-ABI of the MI355X-native InfiniteDiffusion sampling engine (libtd_engine.so).
 *
 * The reference (xandergos/terrain-diffusion) is pure Python and has no FFI; its boundary for this path
 * is the Python operator surface listed in SURVEY.md §8b.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions: plain C, opaque handles, return 0 on success / negative code on failure with a message in
 * td_last_error().  One engine per GPU, single caller thread per handle (the reference's servers run
 * threaded=False: terrain_diffusion/inference/api.py:249).  Tensor arguments are raw pointers; unless a
 * parameter says "host", a pointer may be device OR host memory (detected with hipPointerGetAttributes;
 * host buffers are staged through the engine's stream).  Stream contract: the engine works on its own non-blocking HIP stream;
 * device buffers passed in must be complete before the call (synchronise the producing stream) and every result is complete when
 * the call returns.  All tensors are fp32 in the reference's layouts
 * (NCHW tiles, (C+1,H,W) canvases); bf16 exists only inside the engine.
 */
#ifndef TD_ENGINE_H
#define TD_ENGINE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct td_engine td_engine;
typedef struct td_unet td_unet;

enum { TD_OK = 0, TD_ERR_ARG = -1, TD_ERR_HIP = -2, TD_ERR_STATE = -3, TD_ERR_UNSUPPORTED = -4 };
enum { TD_DTYPE_F32 = 0, TD_DTYPE_BF16 = 1, TD_DTYPE_F16 = 2 };  /* storage type inside the engine; accumulation is always fp32 */

/* EDMUnet2D constructor arguments that shape the inference graph
 * (terrain_diffusion/models/edm_unet.py:17-37). */
typedef struct td_unet_config {
    int32_t image_size;            /* only used for block names / attention resolution test */
    int32_t in_channels, out_channels;
    int32_t model_channels;
    int32_t n_levels;
    int32_t channel_mults[8];
    int32_t layers_per_block[8];
    int32_t n_attn_resolutions;
    int32_t attn_resolutions[8];
    int32_t midblock_attention;
    float concat_balance;
    int32_t noise_emb_dims;        /* 0 -> model_channels */
    int32_t emb_channels;          /* 0 -> model_channels * max(mults) */
    /* conditional_inputs of the constructor, in order (edm_unet.py:95-99): type 0 = ["tensor", dim, w] (base model: one 58-vector),
     * type 1 = ["float", fourier_dims, w] (coarse model: five scalars through MPFourier, mp_layers.py:109-131); the decoder has none.
     * A sample's conditioning row is the concatenation of its inputs in this order (tensor: dim floats, float: 1 float). */
    int32_t n_cond;
    int32_t cond_type[8];
    int32_t cond_dims[8];
    float cond_weights[8];
} td_unet_config;

const char* td_last_error(void);
int td_version(void);
/* 16 hex digits identifying the kernel sources this library was built from (sha256 over the files of terrain_diffusion_amd/csrc and this header, stamped by
 * __graft_entry__.build()); bench.py ties committed rocprofv3 counter files to the library that is actually loaded with it.  No reference
 * counterpart (the reference has no native code). */
const char* td_build_id(void);

/* ---- engine ----------------------------------------------------------------------------------------------- */
int td_engine_create(int device_id, td_engine** out);
void td_engine_destroy(td_engine* e);
int td_engine_synchronize(td_engine* e);
/* raw hipStream_t the engine launches on (for timing with HIP events on the right stream) */
void* td_engine_stream(td_engine* e);
/* Caller-supplied stream: every later call enqueues on `hip_stream` (a hipStream_t created by the caller -- e.g. the
 * torch.cuda.Stream that also carries the caller's own kernels and its RCCL transfers; NOT the legacy NULL stream, which cannot be captured
 * into the engine's hipGraphs); NULL restores the engine's own stream.  The old stream is drained first.  By default every call still returns
 * with its results complete; with td_engine_set_option(e, "async", 1) a call whose DATA buffers are all device pointers (td_sample_edm*,
 * td_sample_consistency*, td_noise_patches, td_gather_regions, td_blend_windows, td_blend_normalize, td_resample2d, td_residual_plus,
 * td_elev_finish, td_climate_finish, td_ddim_cfg_step) only ENQUEUES its work, ordered with whatever else the caller puts on that stream.
 * The small host arrays such calls take (origins, descriptors, tap tables, timesteps) are copied into a pinned ring inside the call, so the
 * caller may reuse them on return; device buffers must stay valid in STREAM order (a torch tensor released on that stream is).  Per-call
 * device scratch comes from a stream-ordered pool, never from hipMalloc / hipFree in the steady state.  td_engine_synchronize() drains.
 * The reference gets the same ordering from torch's current-stream semantics (world_pipeline.py:941-949 runs model and scheduler ops on
 * one stream); round 4's cascade runs this way end to end (terrain_diffusion_amd/cascade_bench.py). */
int td_engine_set_stream(td_engine* e, void* hip_stream);
/* Engine options.  An UNKNOWN key is refused (TD_ERR_ARG, td_last_error names it): a misspelt key used to be stored and never read.
 * Behaviour:
 *   "graph"=0/1 (hipGraph capture of the sampler loops), "profile"=0/1, "async"=0/1 (see td_engine_set_stream),
 *   "batch_invariant"=0/1 (a window's result does not depend on the batch / GPU it rides in: no split-K, LDS-DMA conv flavour pinned),
 *   "solver_order"=1/2/3 and "lower_order_final"=0/1 (EDMDPMSolverMultistepScheduler.config), "fuse_solver"=0/1 (solver update in the
 *   output conv's epilogue), "dual_stream"=0/1 (default 1) + "dual_stream_min_batch" (default 32: batches at least that large run as two concurrent half-batch lanes), "plan_cache_mb", "plan_cache_max".
 * Plan builder (speed only; every one is part of the plan-cache key):
 *   "glds", "glds_min_wgs", "glds_bn64", "glds_round_aware", "glds_small_max_groups", "glds_dma1x1", "glds_tiny", "bn128_min_wgs",
 *   "splitk", "splitk_target_wgs", "splitk_weighted", "glds_splitk", "glds_splitk_from_groups", "glds_splitk_max", "glds_splitk_min_groups",
 *   "producer_act", "walk_alternate", "attn_mfma",
 *   "sb" (small-batch conv flavour), "sb_m4"=0/1 (its 128 px x 32 cout tile, off by default), "sb_target_wgs", "sb_order", "sb_max_glds_wgs", "sb_splitk", "sb_splitk_wgs", "sb_splitk_max",
 *   "s16"=0/1/2 (deep-level latency flavour, conv_s16.hip: never / where conv_sb would split K over workgroups and the 16-cout grid reaches
 *   "s16_min_wgs" workgroups / wherever conv_sb applies),
 *   "glds_wide"=0/1/2 (wide tile of the LDS-DMA flavour, conv_glds_wide.hip: never / pure-3x3 launches whose 256-pixel grid reaches "glds_wide_min_wgs"
 *   (384) workgroups / wherever it is legal), "glds_wide_tail"=0/1/2 (launches with an untransformed 1x1 tail on the wide tile: never / on its 64-cout tile (the decoder model) / always).
 *   "glds_wide_persist"=0/1 (persistent tile loop of the wide tile's 64-cout instantiation, staging split by wave: same bits, measured slower -- default 0),
 *   "fewcout"=1/0 (fp32-output 3x3 convs with <= 4 real output channels on maps >= 128 x 128 -- the decoder model's output conv -- on the VALU flavour
 *   conv_fewcout.hip instead of a 64-cout MFMA tile).
 * Test hooks that force a tile shape wherever it is legal: "glds_variant"=-1/0/1, "glds_bn"=0/64/96/128, "sb_mt"=0/1/2/4, "sb_nt"=0/1/2. */
int td_engine_set_option(td_engine* e, const char* key, int64_t value);

/* With option "profile"=1 the samplers run eagerly (no graph) with HIP events recorded on the engine stream around every
 * conv launch (and every other U-Net kernel); this reads/reset the accumulated kernel time and launch counts. */
int td_engine_profile_read(td_engine* e, double* conv_ms, int64_t* conv_launches, double* other_ms, int64_t* other_launches, int reset);

/* the LDS-DMA conv kernel family (td::conv_glds_kernel) alone: summed event time, algorithmic FLOP (2*pixels*Cout*K) and launches */
int td_engine_profile_read_glds(td_engine* e, double* ms, double* flop, int64_t* launches, int reset);
/* per-op breakdown of the same counters as text lines "label<TAB>ms<TAB>launches" (call before a resetting read) */
int td_engine_profile_dump(td_engine* e, char* buf, int64_t capacity);

/* ---- model ------------------------------------------------------------------------------------------------
 * Replaces EDMUnet2D(...) + load_state_dict (edm_unet.py:17-143; diffusers layout, SURVEY.md §8b face 3).
 * Weights are the reference's RAW fp32 parameters by state-dict name; the engine folds the magnitude-preserving
 * normalisation once (mp_layers.py:203-213) and packs them for the MFMA kernels. */
int td_unet_create(td_engine* e, const td_unet_config* cfg, int dtype, td_unet** out);
void td_unet_destroy(td_unet* u);
int td_unet_num_params(td_unet* u);
/* i-th expected parameter: name and shape (ndim<=4) — lets the caller validate a checkpoint */
int td_unet_param_info(td_unet* u, int i, const char** name, int32_t* ndim, int64_t shape[4]);
int td_unet_set_param(td_unet* u, const char* name, const float* host_data, int64_t numel);
/* prefolded=1: the parameters passed to td_unet_set_param already are W/(1e-4+||W||/sqrt(numel))*gain/sqrt(fan_in)
 * (emb_gain / out_gain folded in; the gain scalars are then ignored).  The Python host folds with the reference's own
 * fp32 torch arithmetic so the engine's weights equal the reference's bit-for-bit; prefolded=0 folds in fp64 here. */
int td_unet_set_prefolded(td_unet* u, int prefolded);
int td_unet_finalize(td_unet* u);

/* model(x, noise_labels=t, conditional_inputs=[...]) -> F      (edm_unet.py:161-184)
 * x: [n][in_channels][H][W], t: host [n], cond: [n][cond_row_len] (see td_unet_config), out: [n][out_channels][H][W] */
int td_unet_cond_row_len(td_unet* u);
int td_unet_forward(td_unet* u, int n, int H, int W, const float* x, const float* t_host, const float* cond, float* out);

/* Test/debug facility: after a td_unet_forward(n,H,W), copies the OUTPUT of the fused conv op `label` (e.g. "enc.512x512_block0.conv_res1",
 * the block output; "<block>.conv_res0" = y1) to host as NCHW fp32.  dims receives (n,C,h,w). */
int td_unet_read_activation(td_unet* u, int n, int H, int W, const char* label, float* out_host, int64_t capacity, int32_t dims[4]);

/* ---- portable noise (terrain_diffusion/inference/portable_rng.py:22-89, world_pipeline.py:58-115) ---------- */
uint64_t td_tile_seed(uint64_t base_seed, int64_t ty, int64_t tx);                 /* _tile_seed */
int td_standard_normal(td_engine* e, uint64_t seed, int64_t n, float* out);        /* standard_normal(seed, n) */
/* gaussian_noise_patch for `n_windows` windows of (channels,h,w) at origins[2*i]=(y0,x0) (host int64 pairs) */
int td_noise_patches(td_engine* e, uint64_t base_seed, int n_windows, const int64_t* origins_host, int h, int w,
                     int channels, int tile_h, int tile_w, float scale, float* out);

/* ---- schedule (terrain_diffusion/scheduler/dpmsolver.py:285-342): NOT exported.  The Karras sigma ladder is a few dozen fp32 scalars that
 * the reference computes with torch CPU ops; the host scheduler (terrain_diffusion_amd/scheduler.py) repeats those ops and is bit-exact.
 * A C restatement with libm powf differs in the last bits (2e-6), so round 1's td_schedule_karras was removed rather than ship an
 * export that disagrees with the product path.  The samplers below take the sigma ladder from the caller. */

/* ---- samplers ---------------------------------------------------------------------------------------------
 * Inner loop of sample_base_diffusion (terrain_diffusion/training/evaluation/sample_diffusion_base.py:147-162)
 * and _coarse_inference (world_pipeline.py:941-949) for a batch of independent tiles:
 *   for i in steps: F = model(c_in*x, atan(sigma_i/sigma_d), cond); x = DPMSolver++(2M).step(F, x)
 * x: in/out [n][C][H][W] (enter as noise*sigma_0, leave as the sigma=0 sample, NOT divided by sigma_data).
 * sigmas_host: n_steps+1 values (last = 0). */
int td_sample_edm(td_unet* u, int n, int H, int W, int n_steps, const float* sigmas_host, float sigma_data,
                  const float* cond, float* x);
/* Autoguidance (sample_diffusion_base.py:105-110,155-160): F = F_guide + scale * (F_main - F_guide) in every step; the guide is a second,
 * smaller EDMUnet2D taking the same inputs (configs/diffusion_base/30m/diffusion_128-3.cfg), created on the same engine with the same dtype. */
int td_sample_edm_guided(td_unet* u, td_unet* guide, float guidance_scale, int n, int H, int W, int n_steps, const float* sigmas_host,
                         float sigma_data, const float* cond, float* x);
/* One trig-flow consistency phase (sample_diffusion_base.py:248-257, world_pipeline.py:1097-1129):
 *   x_t = cos t*sample + sin t*sigma_d*z ; out = cos t*x_t + sin t*sigma_d*model(x_t/sigma_d, t, cond)
 * sample may be NULL (zeros, first phase). */
int td_sample_consistency(td_unet* u, int n, int H, int W, float t, float sigma_data, const float* sample, const float* z,
                          const float* cond, float* out);
/* Same two samplers for models whose input = [sample channels | conditioning-image channels] (in_channels = out_channels + cimg):
 * the coarse stage (world_pipeline.py:928-949: 6 sample + 5 noised-map channels) and the decoder (world_pipeline.py:1221-1239:
 * 1 sample + 4 upsampled-latent channels).  cond_img: [n][cimg_channels][H][W], constant over the steps; x / sample / z / out carry
 * out_channels channels. */
int td_sample_edm_img(td_unet* u, int n, int H, int W, int n_steps, const float* sigmas_host, float sigma_data, const float* cond,
                      const float* cond_img, int cimg_channels, float* x);
int td_sample_consistency_img(td_unet* u, int n, int H, int W, float t, float sigma_data, const float* sample, const float* z,
                              const float* cond, const float* cond_img, int cimg_channels, float* out);

/* ---- overlap blend (sample_diffusion_base.py:164-168; annotated_infinite_panorama.py:145-150) ----------------
 * canvas: (C+1, Hc, Wc) fp32, weighted sums + weight channel.  Adds window i (tiles[i] = [C][size][size]) at
 * (row_starts[wi[i]], col_starts[wj[i]]) with the linear weight window, summing in ascending (wi,wj) order
 * (the reference's loop order) so results are bit-reproducible.  accumulate=0 overwrites the canvas. */
int td_blend_windows(td_engine* e, float* canvas, int C, int Hc, int Wc, int size, int n_rows, const int32_t* row_starts_host,
                     int n_cols, const int32_t* col_starts_host, int n_tiles, const int32_t* wi_host, const int32_t* wj_host,
                     const float* tiles, int accumulate);

/* Many equally sized regions of ONE lazily evaluated window tensor in one launch: what the reference's `InfiniteTensor.__getitem__` does
 * slice by slice when a stage function is handed the argument slices of its windows (infinite_tensor call sites world_pipeline.py:982-992,
 * 1146-1201, 1259-1270; annotated_infinite_panorama.py:153-226).  out: [n_regions][C+1][h][w] device fp32 = (sum_w out_w * win, sum_w win)
 * over the windows listed for the region; desc_host: [n_regions][maxk][3] int32 = (slot into window_ptrs_host, row of the window's first
 * pixel relative to the region, column likewise), slot < 0 ends a region's list, windows in ascending (row, col) order (the reference's
 * summation order: results are bit-identical to td_blend_windows region by region); window_ptrs_host: device addresses of the raw window
 * outputs [C][size][size]. */
int td_gather_regions(td_engine* e, int C, int size, int n_regions, int h, int w, int maxk, const int32_t* desc_host, int n_windows,
                      const uint64_t* window_ptrs_host, float* out);
/* out[c] = canvas[c]/canvas[C]*scale, out: (C,Hc,Wc) */
int td_blend_normalize(td_engine* e, const float* canvas, int C, int Hc, int Wc, float scale, float* out);
/* linear_weight_window(size) (world_pipeline.py:117-124) -> out[size*size] */
int td_linear_weight_window(td_engine* e, int size, float* out);

/* ---- output composition after the decoder ---------------------------------------------------------
 * Separable gather: out[c][yo][xo] = sum_a wy[yo][a] * ( sum_b wx[xo][b] * in[c][iy[yo][a]][ix[xo][b]] ), rows first.  The tap tables
 * (host int32 / float arrays of Hout*Ky and Wout*Kx entries) carry bilinear resize with align_corners=False, its anti-aliased form and the
 * reflect-padded Gaussian blur -- torchvision.transforms.functional.resize / gaussian_blur as used by laplacian_encode / laplacian_decode
 * (terrain_diffusion/data/laplacian_encoder.py:62-137).  in/out: [C][H][W] fp32, host or device. */
int td_resample2d(td_engine* e, const float* in, int C, int Hin, int Win, int Hout, int Wout, const int32_t* iy_host, const float* wy_host, int Ky,
                  const int32_t* ix_host, const float* wx_host, int Kx, float* out);
/* (r0/r1)*std+mean + lowres_up over a packed (2,Hp,Wp) decoder slice -> (Hp,Wp): the `decoded` image inside laplacian_denoise
 * (laplacian_encoder.py:134-137 via world_pipeline.py:1300-1306).  Device buffers. */
int td_residual_plus(td_engine* e, const float* packed, const float* lowres_up, int Hp, int Wp, float res_mean, float res_std, float* out);
/* elev = sign(s) * s^2, s = (r0/r1)*std+mean + lowres_up, cropped to [oi,oi+h) x [oj,oj+w)  (world_pipeline.py:1300,1308-1312).  Device buffers. */
int td_elev_finish(td_engine* e, const float* packed, const float* lowres_up, int Hp, int Wp, int oi, int oj, int h, int w, float res_mean,
                   float res_std, float* out);
/* WorldPipeline._compute_climate, per-pixel half (world_pipeline.py:1333-1365): out (5, h, w) = [baseline + lapse * max(elev, 0), coarse channels 3, 4, 5
 * upsampled, lapse rate] for the pixel box that starts at (i1, j1); feats (5, Hs, Ws) = [sea-level baseline, lapse rate, coarse ch 3, 4, 5] on the coarse
 * grid whose first cell is (ci1, cj1) in units of S pixels; bilinear with clamped borders = torch grid_sample(align_corners=False, 'border').
 * Device buffers only. */
int td_climate_finish(td_engine* e, const float* feats, int Hs, int Ws, const float* elev, int i1, int j1, int h, int w, float S, int ci1, int cj1, float* out);

/* ---- panorama demo (BASELINE configs[0]) -------------------------------------------------------------------- */
/* One classifier-free-guided DDIM step (annotated_infinite_panorama.py:130-134: pred = uncond + g (cond - uncond); latent = scheduler.step(pred,
 * t, latent).prev_sample with diffusers' DDIMScheduler, eta = 0): out = sqrt(alpha_prev) (latent - sqrt(1 - alpha_t) pred) / sqrt(alpha_t) +
 * sqrt(1 - alpha_prev) pred over n elements.  alpha_t / alpha_prev = the schedule's cumulative alphas at this step's timestep and at the next one
 * (terrain_diffusion_amd.pano.DDIMSchedule).  Device buffers only; `out` may alias `latent`. */
int td_ddim_cfg_step(td_engine* e, const float* latent, const float* pred_uncond, const float* pred_cond, int64_t n, float guidance_scale, float alpha_t,
                     float alpha_prev, float* out);

/* ---- attention (MFMA flash kernel, terrain_diffusion_amd/csrc/attn_mfma.hip) -------------------------------------------------
 * out = softmax(scale * Q K^T) V per (batch, head); q: [B][H][Lq][D], k / v: [B][H][Lk][D], out: [B][H][Lq][D], fp32 at the boundary, bf16
 * operands / fp32 accumulation inside; any Lq, Lk >= 1, 1 <= D <= 160.  normalize=1 first scales every q, k, v row to unit RMS,
 * x / (1e-4 + ||x|| / sqrt(D)) -- UNetBlock.attn (terrain_diffusion/models/unet_block.py:102-108, with scale = 1 / sqrt(D)); normalize=0 with
 * scale = 1 / sqrt(D) is the plain scaled-dot-product attention of the SD-v1.5 U-Net in annotated_infinite_panorama.py:109-134.  The engine's
 * own U-Net uses the same kernel for its attention blocks in bf16 mode. */
int td_attention(td_engine* e, const float* q, const float* k, const float* v, int B, int H, int Lq, int Lk, int D, float scale, int normalize, float* out);

/* ---- synthetic conditioning map (terrain_diffusion/inference/synthetic_map.py:182-236, perlin_transform.py:41-45) ----
 * One channel of the coarse conditioning source over rows [i1, i1+rows) x cols [j1, j1+cols): gradient-noise FBm (x = row, y = col, scaled by
 * `frequency`; `octaves` octaves, lacunarity, gain; integer seed) pushed through the piecewise-linear quantile transfer src -> dst (n_quantiles
 * ascending knots each, ends clamped).  The generator has FastNoiseLite's structure but its own gradient set: values are parity-unpinned. */
int td_perlin_map(td_engine* e, int rows, int cols, int i1, int j1, int seed, float frequency, int octaves, float lacunarity, float gain,
                  const float* src_quantiles, const float* dst_quantiles, int n_quantiles, float* out);

#ifdef __cplusplus
}
#endif
#endif
