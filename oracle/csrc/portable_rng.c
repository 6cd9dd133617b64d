/*
 * ORACLE (test infrastructure, not product code): CPU restatement of the reference's
 * portable RNG and tile-seeded noise field.
 *
 * Follows (behaviour only, re-written from scratch):
 *   terrain_diffusion/inference/portable_rng.py:22-28   (_pcg64_next: 64-bit LCG, XSH-RR 64/32 taken
 *                                                        from the POST-advance state, state starts at seed)
 *   terrain_diffusion/inference/portable_rng.py:56-74   (_fill_standard_normal_impl: Marsaglia polar, f64 maths)
 *   terrain_diffusion/inference/portable_rng.py:31-42   (next_seed)
 *   terrain_diffusion/inference/world_pipeline.py:58-63 (_tile_seed)
 *   terrain_diffusion/inference/world_pipeline.py:66-115 (gaussian_noise_patch)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PCG_MULT 6364136223846793005ULL
#define PCG_INC 1442695040888963407ULL

static inline uint32_t pcg_step(uint64_t *state) {
    uint64_t s = *state * PCG_MULT + PCG_INC;
    *state = s;
    uint32_t x = (uint32_t)(((s >> 18) ^ s) >> 27);
    uint32_t rot = (uint32_t)(s >> 59);
    return (x >> rot) | (x << ((32u - rot) & 31u));
}

/* first n raw 32-bit outputs of the stream seeded with `seed` */
void orc_pcg_stream(uint64_t seed, uint32_t *out, int64_t n) {
    uint64_t s = seed;
    for (int64_t i = 0; i < n; ++i) out[i] = pcg_step(&s);
}

uint64_t orc_next_seed(uint64_t seed) {
    uint64_t s = seed;
    uint64_t lo = pcg_step(&s);
    uint64_t hi = pcg_step(&s);
    return (hi << 32) | lo;
}

/* fills out[0..n) (float32) with the reference's standard-normal stream for `seed` */
void orc_fill_standard_normal_f32(uint64_t seed, float *out, int64_t n) {
    uint64_t s = seed;
    const double inv = 1.0 / 4294967296.0;
    int64_t i = 0;
    while (i < n) {
        uint32_t u1 = pcg_step(&s);
        uint32_t u2 = pcg_step(&s);
        double v1 = 2.0 * ((double)u1 + 1.0) * inv - 1.0;
        double v2 = 2.0 * ((double)u2 + 1.0) * inv - 1.0;
        double r = v1 * v1 + v2 * v2;
        if (r > 0.0 && r < 1.0) {
            double f = sqrt(-2.0 * log(r) / r);
            out[i++] = (float)(v1 * f);
            if (i < n) out[i++] = (float)(v2 * f);
        }
    }
}

void orc_fill_standard_normal_f64(uint64_t seed, double *out, int64_t n) {
    uint64_t s = seed;
    const double inv = 1.0 / 4294967296.0;
    int64_t i = 0;
    while (i < n) {
        uint32_t u1 = pcg_step(&s);
        uint32_t u2 = pcg_step(&s);
        double v1 = 2.0 * ((double)u1 + 1.0) * inv - 1.0;
        double v2 = 2.0 * ((double)u2 + 1.0) * inv - 1.0;
        double r = v1 * v1 + v2 * v2;
        if (r > 0.0 && r < 1.0) {
            double f = sqrt(-2.0 * log(r) / r);
            out[i++] = v1 * f;
            if (i < n) out[i++] = v2 * f;
        }
    }
}

uint64_t orc_tile_seed(uint64_t base_seed, int64_t ty, int64_t tx) {
    /* Python big-int arithmetic: ((seed*G + (ty&M32)) mod 2^64 * G + (tx&M32)) mod 2^64.
       Reducing the first product mod 2^64 before the add is equivalent. */
    const uint64_t G = 0x9E3779B9ULL;
    uint64_t h = base_seed * G;
    h = h + ((uint64_t)ty & 0xFFFFFFFFULL);
    h = h * G + ((uint64_t)tx & 0xFFFFFFFFULL);
    return h;
}

static inline int64_t floordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

/* out is (channels, h, w) float32, C-contiguous */
int orc_gaussian_noise_patch(uint64_t base_seed, int64_t y0, int64_t x0, int64_t h, int64_t w,
                             int64_t channels, int64_t tile_h, int64_t tile_w, float *out) {
    int64_t ty0 = floordiv(y0, tile_h), ty1 = floordiv(y0 + h - 1, tile_h);
    int64_t tx0 = floordiv(x0, tile_w), tx1 = floordiv(x0 + w - 1, tile_w);
    int64_t tn = channels * tile_h * tile_w;
    float *tile = (float *)malloc(sizeof(float) * (size_t)tn);
    if (!tile) return -1;
    for (int64_t ty = ty0; ty <= ty1; ++ty) {
        int64_t tile_y0 = ty * tile_h;
        for (int64_t tx = tx0; tx <= tx1; ++tx) {
            int64_t tile_x0 = tx * tile_w;
            int64_t oy0 = y0 > tile_y0 ? y0 : tile_y0;
            int64_t oy1 = (y0 + h) < (tile_y0 + tile_h) ? (y0 + h) : (tile_y0 + tile_h);
            int64_t ox0 = x0 > tile_x0 ? x0 : tile_x0;
            int64_t ox1 = (x0 + w) < (tile_x0 + tile_w) ? (x0 + w) : (tile_x0 + tile_w);
            orc_fill_standard_normal_f32(orc_tile_seed(base_seed, ty, tx), tile, tn);
            for (int64_t c = 0; c < channels; ++c)
                for (int64_t y = oy0; y < oy1; ++y)
                    memcpy(out + (c * h + (y - y0)) * w + (ox0 - x0),
                           tile + (c * tile_h + (y - tile_y0)) * tile_w + (ox0 - tile_x0),
                           sizeof(float) * (size_t)(ox1 - ox0));
        }
    }
    free(tile);
    return 0;
}
