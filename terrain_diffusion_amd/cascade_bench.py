"""`python bench.py --workload cascade`: BASELINE configs[4] shapes on ONE GPU -- the coarse -> latent (two blended trig-flow phases) -> decoder
cascade through the lazy device-resident graph (WorldPipeline), then the output composition (elevation + climate), with the tile cache capped
so that windows are evicted and recomputed while the region streams.

A "step" = one fresh region of R x R decoded pixels requested in Q x Q-pixel `get()` calls (a server walking across the world); every step
uses another region, so nothing is served from cache across steps.  value = decoded MP / s.  Stage models: the released 30m/90m
architectures (coarse 128-ch single level, base 192-ch U-Net, decoder 64-ch U-Net at 512x512 / stride 384) with synthetic weights.
Algorithmic work per decoded MP: decoder 9 330 GFLOP + latent 5 910 + coarse 3 = 15.24 TFLOP.
"""
import json
import time

import torch

GFLOP_PER_MP = 15_243.0
PEAK_BF16_TFLOPS = 2500.0
COARSE_CONFIG = dict(image_size=16, in_channels=11, out_channels=6, model_channels=128, model_channel_mults=[1], layers_per_block=2, attn_resolutions=[],
                     midblock_attention=False, concat_balance=0.5, conditional_inputs=[["float", 64, 0.2]] * 5, fourier_scale="pos")
DECODER_CONFIG = dict(image_size=512, in_channels=5, out_channels=1, model_channels=64, model_channel_mults=[1, 2, 3, 4], layers_per_block=3, attn_resolutions=[],
                      midblock_attention=False, concat_balance=0.5, conditional_inputs=[], fourier_scale="pos")


def run_cascade(args, eng, dev, rank, world):
    import terrain_diffusion_amd as td
    from terrain_diffusion_amd.synthetic import synthetic_state_dict
    from bench import BASE_CONFIG
    dtype = "fp16" if args.dtype == "fp16" else args.dtype
    models = []
    for cfg, seed in ((COARSE_CONFIG, 11), (BASE_CONFIG, 1234), (DECODER_CONFIG, 2468)):
        m = td.EDMUnet2D(**cfg, dtype=dtype, device=dev)
        models.append(m.load_state_dict(synthetic_state_dict(m, seed=seed)))
    # region and request size in decoded pixels.  N = 1: 3 x 3 requests per step (the round-2 workload).  N > 1: ONE 6 x 6-request region of ONE
    # world (same seed on every rank) whose requests are dealt to the ranks in spatially compact shares (parallel.shard_requests: Z-curve order, so
    # a rank's decoder windows share latent / coarse windows); every rank holds the whole lazy graph, no window crosses a GPU boundary and there
    # is no data-path collective ("request-replica" mode).  Total work is fixed -> strong scaling; the N = 1 point of that workload is
    # `python bench.py --workload cascade --region 6144`.
    R, Q = (int(getattr(args, "region", 0)) or (3072 if world == 1 else 6144)), 1024
    from terrain_diffusion_amd.parallel import shard_requests
    cache = int(getattr(args, 'cache_mib', 100)) * 2 ** 20   # default 100 MiB = the reference's cache_limit (world_pipeline.py:311); one step produces ~150 MiB of windows -> streaming eviction
    world_p = td.WorldPipeline.from_models(*models, seed=4242, dtype=dtype, device=dev, cache_limit=cache, latents_batch_size=(1, 2, 4, 8, 16, 32, 64)).bind()

    split = {}

    def one_step(i):
        i0, j0 = 100_000 * (i + 1), -50_000 * (i + 1)
        out = None
        boxes = [(i0 + a, j0 + b, i0 + a + Q, j0 + b + Q) for a in range(0, R, Q) for b in range(0, R, Q)]
        mine = [bx for _, bx in shard_requests(boxes, world, rank)] if world > 1 else boxes
        split["requests_this_rank"] = len(mine); split["requests_per_step"] = len(boxes)
        for bx in mine:
            out = world_p.get(*bx)
        return out

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()
    # The requests run with the engine on torch's current stream and in enqueue-only mode (Engine.on_stream = td_engine_set_stream + option "async"):
    # sampler launches, region gathers, composition and the torch ops between them are ordered by ONE stream, and the host -- lazy-graph
    # bookkeeping, descriptor tables, Python -- runs ahead of the GPU instead of waiting for every call (--cascade-sync 1: the synchronous C-ABI
    # default, results complete on return of every call, as in rounds 2-3)
    import contextlib
    enqueue_only = not int(getattr(args, "cascade_sync", 0))
    with (eng.on_stream(torch.cuda.Stream(device=dev)) if enqueue_only else contextlib.nullcontext()):
        for i in range(args.warmup):
            one_step(-(i + 1))
        sync()
        for t in (world_p.coarse, world_p.latents, world_p.residual):
            t.windows_computed = 0
        ev0 = world_p.tile_store.evictions
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = one_step(i)
        sync()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out is None or bool(torch.isfinite(out["elev"]).all())
    per_rank = [split.get("requests_this_rank", 0)]
    if world > 1:   # the request split of the replica mode, rank by rank (make the N > 1 cascade line say who served what)
        pr = torch.zeros(world, dtype=torch.int64, device=dev); pr[rank] = per_rank[0]
        dist.all_reduce(pr)
        per_rank = [int(v) for v in pr.tolist()]
    # the regime between the two legs the driver line reports (batch 1 and batch 64): the base U-Net's forward at the batch sizes the latent stage
    # actually runs (latents_batch_size 1 ... 64), wall time per forward of eager launches on the engine stream
    sweep = {}
    if rank == 0:
        base = models[1]
        for nb in (1, 8, 32):
            x = td.standard_normal(7, (nb, 5, 64, 64), device=dev, as_torch=True); c = td.standard_normal(8, (nb, 58), device=dev, as_torch=True); tt_ = torch.full((nb,), 1.1)
            base(x, tt_, [c]); sync()
            t1 = time.perf_counter()
            for _ in range(5):
                base(x, tt_, [c])
            sync()
            ms_f = (time.perf_counter() - t1) / 5 * 1e3
            sweep[str(nb)] = {"ms_per_forward": round(ms_f, 3), "tflops": round(nb * 193.654 / ms_f, 1)}
    mp = R * R / 1e6
    value = args.steps * mp / dt
    # per-stage kernel time: one profiled (eager) request of fresh terrain
    eng.set_option("profile", 1)
    eng.profile_read(reset=True)
    world_p.get(7_000_000, 7_000_000, 7_000_000 + Q, 7_000_000 + Q)
    sync()
    ops = eng.profile_ops()
    conv_ms, conv_n, other_ms, other_n = eng.profile_read(reset=True)
    eng.set_option("profile", 0)
    import re
    by_level, mb_level = {}, {}
    for label, ms, n in ops:
        mres = re.search(r"\[(\d+x\d+) .* mb([0-9.]+)(?: mbs[0-9.]+)?\]", label)   # (round 5: the label also carries the strict figure, without the second output)
        if mres:
            a = mb_level.setdefault(mres.group(1), [0.0, 0.0])
            a[0] += float(mres.group(2)) * n; a[1] += ms
    for label, ms, n in ops:
        key = "coarse+latent 64x64 and below" if any(s in label for s in ("[64x64", "[32x32", "[16x16", "[8x8")) else "decoder levels " + label[label.find("[") + 1:].split(" ")[0] if "[" in label else "other"
        by_level[key] = by_level.get(key, 0.0) + ms
    tflops = value * GFLOP_PER_MP / 1e3
    result = {
        "metric": "terrain megapixels/sec (decoded) at fixed steps, 30m model", "value": round(value, 4), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] shapes on one GPU: coarse -> 2-phase latent -> decoder (512x512 / stride 384) cascade through the lazy device-resident "
                               f"graph + elevation/climate composition, {R}x{R} decoded pixels per step in {Q}x{Q} requests, window cache capped at {cache >> 20} MiB "
                               "(streaming eviction)",
                   "decoded_mp_per_step": mp, "parallelism": (f"{world} ranks, request-replica mode: one world, requests dealt in Z-curve order, no data-path collective" if world > 1 else "1 rank"), "windows_per_step": {k: round(t.windows_computed / args.steps, 1) for k, t in
                                                                   (("coarse", world_p.coarse), ("latent_final_phase", world_p.latents), ("decoder", world_p.residual))},
                   "cache_evictions_per_step": round((world_p.tile_store.evictions - ev0) / args.steps, 1),
                   "requests_per_step": split.get("requests_per_step"), "requests_per_rank": per_rank,
                   "engine_calls": "enqueue-only on one stream shared with torch (Engine.on_stream)" if enqueue_only else "synchronous (complete on return)"},
        "batch_sweep": {"what": "base U-Net forward (193.654 GFLOP per tile) at the latent stage's batch sizes, eager launches, wall ms per forward; "
                                "the full 1 ... 64 sweep with HBM bytes is profiles/r05_batch_sweep.txt", "by_batch": sweep},
        "roofline": {"bound": "mfma", "kernel": "td::conv_glds_kernel (all stages)", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "achieved": round(tflops, 2),
                     "frac": round(tflops / PEAK_BF16_TFLOPS, 4), "traffic": None,
                     "note": "end to end over the algorithmic 15.24 TFLOP per decoded MP (recomputed evicted windows are NOT counted as useful work)",
                     "one_request_conv_kernel_ms": round(conv_ms, 2), "one_request_other_kernel_ms": round(other_ms, 2),
                     "one_request_kernel_ms_by_resolution": {k: round(v, 2) for k, v in sorted(by_level.items())},
                     # achieved HBM GB/s of the conv path by resolution level: algorithmic megabytes of the level's launches (sources, residual, outputs and
                     # weights once each, `mb` of the engine's per-op profile) / their kernel time.  The decoder's 64-channel 512x512 level is the
                     # activation-bound one (~288 FLOP/B, below the 312 FLOP/B ridge of 2.5 PF over 8 TB/s)
                     "hbm_gbps_by_resolution": {k: round(v[0] / v[1], 1) for k, v in sorted(mb_level.items()) if v[1] > 0},
                     "hbm_gbps_decoder_512x512": round(mb_level["512x512"][0] / mb_level["512x512"][1], 1) if mb_level.get("512x512", [0, 0])[1] > 0 else None,
                     "hbm_peak_gbps": 8000.0},
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
