"""Time to first tile / time to second tile of a WorldPipeline -- the method of the reference's terrain_diffusion/evaluation/latency.py:19-127.

TTFT: a `get()` of a tile_size x tile_size box far away from everything computed before (nothing in the window caches: the whole dependency
cone -- coarse windows, latent windows of both phases, decoder windows -- is computed).  TTST: the adjacent box right after it (shares most of
the cone).  Boxes of successive runs are `separation` pixels apart (latency.py:16,63-70), the caches are emptied between runs (:94), every
measurement is bracketed by device synchronisation (:72-91).  Returns the reference's result keys.

The reference builds its world with `WorldPipeline.from_local_models(...)`; pass `world=` for that.  Without one, a world of the released
architectures with SYNTHETIC weights is built (no checkpoints exist offline) -- timing does not depend on the weight values."""
import math
import random
import time

import torch

SEPARATION = 200 * 256   # latency.py:16: far enough that no cached window is shared between runs


def _percentile(data, p):   # latency.py:108-111
    s = sorted(data)
    return s[int((len(s) - 1) * p / 100 + 0.5)]


def synthetic_world(device="cuda", seed=42, dtype="bf16", **kwargs):
    """WorldPipeline on the released 30m / 90m architectures with synthetic weights (bench.py / cascade_bench.py use the same models)."""
    from . import EDMUnet2D, WorldPipeline
    from .synthetic import synthetic_state_dict
    from .cascade_bench import COARSE_CONFIG, DECODER_CONFIG
    base_cfg = dict(image_size=512, in_channels=5, out_channels=5, model_channels=192, model_channel_mults=[1, 2, 3, 4], layers_per_block=3,
                    attn_resolutions=[8, 16], midblock_attention=True, concat_balance=0.5, conditional_inputs=[["tensor", 58, 1.0]], fourier_scale="pos")
    models = []
    for cfg, s in ((COARSE_CONFIG, 11), (base_cfg, 1234), (DECODER_CONFIG, 2468)):
        m = EDMUnet2D(**cfg, dtype=dtype, device=device)
        models.append(m.load_state_dict(synthetic_state_dict(m, seed=s)))
    return WorldPipeline.from_models(*models, seed=seed, dtype=dtype, device=device, **kwargs), models


def measure_latency(device="cuda", seed=42, onestep_latent=False, tile_size=512, grid_aligned=False, num_runs=100, decoder_tile_size=512,
                    decoder_tile_stride=384, max_batch_size=16, T=2, *, dtype="bf16", world=None, rng_seed=0):
    """latency.py:19-127 with the same arguments (+ dtype, an optional ready-made `world`, and a seed for the box positions)."""
    assert 2 ** round(math.log2(max_batch_size)) == max_batch_size
    owned = []
    if world is None:
        world, owned = synthetic_world(device=device, seed=seed, dtype=dtype, caching_strategy="direct", cache_limit=None, onestep_latent=onestep_latent,
                                       latents_batch_size=[2 ** i for i in range(round(math.log2(max_batch_size)) + 1)],
                                       decoder_tile_size=decoder_tile_size, decoder_tile_stride=decoder_tile_stride, T=T)
    world.to(device)
    world.bind("TEMP")
    torch.cuda.reset_peak_memory_stats()

    def sync():
        world.engine.synchronize()
        torch.cuda.synchronize()

    world.get(0, 0, tile_size, tile_size, with_climate=False)   # warm-up: plans, graphs, packed weights
    sync()
    rnd = random.Random(rng_seed)
    ttft, ttst = [], []
    for run in range(num_runs):
        if grid_aligned:
            base_i = ((run + 1) * SEPARATION // tile_size) * tile_size + rnd.randint(0, SEPARATION // (10 * tile_size)) * tile_size
            base_j = rnd.randint(0, SEPARATION // tile_size) * tile_size
        else:
            base_i = (run + 1) * SEPARATION + rnd.randint(0, SEPARATION // 10)
            base_j = rnd.randint(0, SEPARATION)
        sync()
        t0 = time.perf_counter()
        world.get(base_i, base_j, base_i + tile_size, base_j + tile_size, with_climate=False)
        sync()
        t1 = time.perf_counter()
        adj_j = base_j + tile_size
        world.get(base_i, adj_j, base_i + tile_size, adj_j + tile_size, with_climate=False)
        sync()
        t2 = time.perf_counter()
        ttft.append(t1 - t0)
        ttst.append(t2 - t1)
        world.empty_cache()
    peak = torch.cuda.max_memory_allocated() / (1024 * 1024)
    world.close()
    for m in owned:
        m.close()
    mean = lambda v: sum(v) / len(v)
    std = lambda v: (sum((t - mean(v)) ** 2 for t in v) / len(v)) ** 0.5
    return {"ttft_mean": mean(ttft), "ttst_mean": mean(ttst), "ttft_std": std(ttft), "ttst_std": std(ttst),
            "ttft_p5": _percentile(ttft, 5), "ttft_p50": _percentile(ttft, 50), "ttft_p95": _percentile(ttft, 95),
            "ttst_p5": _percentile(ttst, 5), "ttst_p50": _percentile(ttst, 50), "ttst_p95": _percentile(ttst, 95),
            "peak_vram_mb": peak, "num_runs": num_runs, "tile_size": tile_size, "dtype": dtype,
            "note": "peak_vram_mb counts torch's allocator only (the engine's weights and activations are hipMalloc'ed by the library)"}
