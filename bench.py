#!/usr/bin/env python
"""Headline benchmark: decoded terrain megapixels / second at fixed steps (BASELINE.json), terrain-diffusion-30m base model.

    python bench.py --gpus N --steps K --warmup W [--workload grid8|grid32|tiles|cascade] [--dtype bf16|fp32]

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
  grid8  (default at N=1, BASELINE configs[2]): an 8x8 grid of overlapping 64x64-latent windows (stride 32) on a 288x288 latent
         canvas, 20 EDM DPM-Solver++ steps, all 64 windows batched per solver step, overlap blend at the end: 5.308416 decoded MP.
  grid32 (default at N>1, BASELINE configs[3]): ONE 32x32 window grid (1056x1056 latent canvas, 71.37 decoded MP) sharded over the N
         ranks as a 2-D block mesh with the point-to-point seam exchange of window outputs over RCCL (terrain_diffusion_amd/parallel.py);
         total work is fixed -> "scaling": "strong".  The line carries the seam bytes and the exchange time.
  tiles  (BASELINE configs[1] batched): 64 INDEPENDENT single-tile jobs per step (no overlap): 16.78 decoded MP; N>1: every rank its own
         batch (weak).  The literal configs[1] number (ONE tile, latency-bound) is reported on every N=1 line as "latency_single_tile_ms".
  cascade (BASELINE configs[4] shapes on one GPU): coarse -> 2-phase latent -> decoder through the lazy InfiniteTensor graph with a capped
         tile cache; see run_cascade().
With --gpus N > 1 and no WORLD_SIZE in the environment bench.py launches its own N ranks (torch.distributed.run, 127.0.0.1).
Weights are synthetic (portable-RNG seeded, out_gain=1, emb_gain=0.5): there is no network for checkpoints.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import re
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_FORWARD = 193.654          # base model, one 64x64 tile (2 FLOP/MAC, convs+GEMMs+bmm)
WEIGHT_BYTES_BF16 = 253_688_037 * 2  # algorithmic minimum HBM bytes per forward at batch 1 (weights once)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PROFILE_JSON = os.path.join(ROOT, "profiles", "r06_hbm_traffic_and_mfma_util.json")   # stamped with the library build id it was collected with

# terrain-diffusion-30m base model (configs/diffusion_base/30m/diffusion_192-3.cfg:54-68)
BASE_CONFIG = dict(image_size=512, in_channels=5, out_channels=5, model_channels=192, model_channel_mults=[1, 2, 3, 4], layers_per_block=3,
                   attn_resolutions=[8, 16], midblock_attention=True, concat_balance=0.5, conditional_inputs=[["tensor", 58, 1.0]], fourier_scale="pos")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["grid8", "grid32", "tiles", "cascade"])
    ap.add_argument("--tiles-per-step", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"])
    ap.add_argument("--edm-steps", type=int, default=20)
    ap.add_argument("--region", type=int, default=0, help="cascade workload: side of the decoded region per step in pixels (default 3072 at N=1, 6144 at N>1)")
    ap.add_argument("--cascade-sync", type=int, default=0, help="cascade workload: 1 = every engine call synchronous (complete on return) instead of enqueue-only on one stream")
    ap.add_argument("--cache-mib", type=int, default=100, help="cascade workload: window-cache cap in MiB (default = the reference's cache_limit)")
    ap.add_argument("--engine-opts", default="", help="engine options for A/B runs, e.g. dual_stream=1,s16=0 (recorded in config.engine_opts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-tile latency leg (keeps rocprofv3 counter passes to the batched steps only)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks of this script on this node (one process per GPU, RCCL)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    workload = args.workload or ("grid8" if world == 1 else "grid32")
    # dry run of the N > 1 branch on a one-GPU box: TD_BENCH_ONE_GPU=1 puts every rank on cuda:0 and moves the seams through gloo / host
    # memory (RCCL refuses two ranks on one device); the line then says so in seam.backend -- it is a plumbing check, not a measurement
    one_gpu = os.environ.get("TD_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import terrain_diffusion_amd as td
    from terrain_diffusion_amd.engine import get_engine
    from terrain_diffusion_amd.synthetic import synthetic_state_dict, synthetic_cond_grid
    from terrain_diffusion_amd.sampling import _tile_starts, _process_cond_img

    dev = f"cuda:{local_rank}"
    eng = get_engine(dev)
    eng_opts = {}
    for kv in filter(None, args.engine_opts.split(",")):
        k_, v_ = kv.split("=")
        eng_opts[k_] = int(v_)
        eng.set_option(k_, int(v_))
    if workload == "cascade":
        from terrain_diffusion_amd.cascade_bench import run_cascade
        return run_cascade(args, eng, dev, rank, world)
    cfg = dict(BASE_CONFIG)
    model = td.EDMUnet2D(**cfg, dtype=args.dtype, device=dev)
    model.load_state_dict(synthetic_state_dict(model, seed=1234))
    sch = td.EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
    E = args.edm_steps
    if workload == "tiles":
        H = W = 64
        tiles_per_step = args.tiles_per_step
        mp_per_step = tiles_per_step * (64 * 8) ** 2 / 1e6
    elif workload == "grid32":
        H = W = 1056
        tiles_per_step, mp_per_step = 1024, (1056 * 8) ** 2 / 1e6
    else:
        H = W = 288
        tiles_per_step, mp_per_step = 64, (288 * 8) ** 2 / 1e6
    nt = len(_tile_starts(H, 64, 32))
    cond = synthetic_cond_grid(nt, nt, device=dev)
    kw = dict(cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5), steps=E, tile_size=64)
    cond58 = torch.cat([_process_cond_img(synthetic_cond_grid(1, 1, seed=0xC0DE + j, device=dev), torch.zeros(1, 5), torch.zeros(7), torch.ones(7), 0.0)
                        for j in range(max(1, args.tiles_per_step))])
    seam = {}
    strong = workload == "grid32"
    if strong and "batch_invariant=" not in args.engine_opts:
        # The sharded canvas must be bit-identical to the single-GPU canvas.  When every rank's window list is a whole number of max_batch = 64 batches
        # (world 1 / 2 / 4 / 8: 1024 / 512 / 256 / 128 windows per rank) every launch on every rank count has the same batch size, hence the same plan and
        # K order, and the DEFAULT plan already gives that (test_config3_default_plan_bit_identical_across_rank_counts: 1 rank vs 2 / 4 simulated ranks,
        # bit for bit, with and without the two sampler lanes).  Ragged shards (other rank counts) pin the plan with batch_invariant, which costs 7 %.
        from terrain_diffusion_amd.parallel import ShardPlan
        ragged = any(len(ws) % 64 for ws in ShardPlan(H, W, 64, world).windows)
        eng.set_option("batch_invariant", 1 if ragged else 0)
        eng_opts["batch_invariant"] = 1 if ragged else 0

    def one_step(i, b=None, wl=None):
        # different world regions each step (noise origins move), same as sampling successive regions of the world
        wl = wl or workload
        if wl == "tiles":
            b = b or tiles_per_step
            origins = [(4096 * j, 4096 * i) for j in range(b)]
            return td.sample_independent_tiles(model, sch, origins, cond58[:b], steps=E, noise_seed=42 + 5819 + rank)
        if wl == "grid32":
            from terrain_diffusion_amd.parallel import sample_base_diffusion_sharded
            return sample_base_diffusion_sharded(model, sch, (1, 5, H, W), cond, noise_seed=42 + 5819, noise_origin=(0, 4096 * i), max_batch=64, stats=seam, **kw)[0]
        return td.sample_base_diffusion(model, sch, (1, 5, H, W), cond, noise_seed=42 + 5819 + rank, noise_origin=(0, 4096 * i), **kw)

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    seam.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(args.warmup + i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(out).all())
    ms_per_step = dt / args.steps * 1e3
    weak_factor = 1 if (strong or world == 1) else world   # grid8 / tiles at N>1: every rank its own canvas / batch
    value = weak_factor * args.steps * mp_per_step / dt

    names = {"tiles": f"BASELINE configs[1] batched: terrain-diffusion-30m base U-Net, {tiles_per_step} independent single 64x64 latent tiles x {E} EDM "
                      "DPM-Solver++ steps per step (reference latents_batch_size pattern)",
             "grid32": f"BASELINE configs[3]: terrain-diffusion-30m base U-Net, 32x32 tile grid (1056x1056 latents) sharded over the ranks as a 2-D block mesh, "
                       f"seam exchange of window outputs over RCCL, {E} steps",
             "grid8": f"BASELINE configs[2]: terrain-diffusion-30m base U-Net, 8x8 tile grid (stride 32, 288x288 latents) with overlap blending, {E} steps, "
                      "64 windows batched per solver step"}
    result = {
        "metric": "terrain megapixels/sec (decoded) at fixed steps, 30m model", "value": round(value, 4), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": names[workload], "tiles_per_step": tiles_per_step, "edm_steps": E, "decoded_mp_per_step": mp_per_step,
                   "parallelism": (f"{world} ranks, 2-D block mesh {seam.get('mesh')}, point-to-point seam exchange (no all-reduce), two sampler lanes per rank, "
                                   f"{'batch-invariant plan (ragged shards)' if eng_opts.get('batch_invariant') else 'default plan (whole batches of 64 windows per rank)'}" if strong else
                                   f"{world} independent streams (one process per GPU, no data-path collective)")},
    }
    if strong:
        result["seam"] = {"bytes_total_per_step": seam.get("seam_bytes_total", 0), "bytes_sent_rank0_per_step": seam.get("seam_bytes_sent", 0),
                          "exchange_ms_per_step_rank0": round(seam.get("exchange_s", 0.0) / max(1, args.steps) * 1e3, 3),
                          "windows_rank0": seam.get("windows_this_rank"), "backend": ("gloo through host memory, all ranks on ONE GPU (dry run)" if one_gpu else "nccl (RCCL)") if world > 1 else "none (1 rank)"}
        # strong-scaling efficiency of THIS workload needs an N = 1 point of the same workload: the N = 1 line carries one, measured live in
        # the same run ("strong_scaling_anchor"); nothing is read from committed files

    if rank == 0:
        peak = PEAK_BF16_TFLOPS if args.dtype in ("bf16", "fp16") else PEAK_F32_TFLOPS
        flop_per_step = tiles_per_step * E * GFLOP_PER_FORWARD * 1e9 / (world if strong else 1)  # per GPU
        e2e_tflops = flop_per_step / (ms_per_step * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "td::conv_glds_kernel", "peak": peak, "unit": "TFLOP/s", "traffic": None,
                "end_to_end_achieved": round(e2e_tflops, 2), "end_to_end_frac": round(e2e_tflops / peak, 4)}
        if not args.no_kernel_profile and world == 1:
            # kernel level: HIP events on the engine's own stream around every conv launch (eager mode, no graph), one extra step
            # (one lane, whatever the timed region used: every kernel at the FULL batch, one after the other -- what `rocprofv3 --kernel-trace` of
            # `bench.py --engine-opts dual_stream=0` sees; two lanes are accounted for below)
            dual_on = "dual_stream=0" not in args.engine_opts   # engine default since round 6 (profiles/r06_dual_stream_ab.txt)
            eng.set_option("dual_stream", 0)
            eng.set_option("profile", 1)
            eng.profile_read(reset=True); eng.profile_read_glds(reset=True)
            one_step(10_000)
            sync()
            eng.set_option("dual_stream", 1 if dual_on else 0)
            g_ms, g_flop, g_n = eng.profile_read_glds(reset=True)
            ops_ = eng.profile_ops()
            sb_rows = [(float(re.search(r" gf([0-9.]+)", l_).group(1)), ms_, n_) for l_, ms_, n_ in ops_ if re.search(r" f[45]\w* bn", l_)]
            # algorithmic HBM bytes per launch of the roofline family (every source tensor once at its own resolution, residual, weights and outputs
            # once): with the optional pre-activated second output counted (what the launches are asked to write) and, strictly, without it
            g_rows = [(float(re.search(r" mb([0-9.]+)", l_).group(1)), float(re.search(r" mbs([0-9.]+)", l_).group(1)), n_) for l_, ms_, n_ in ops_ if re.search(r" f2\w* bn", l_)]
            conv_ms, conv_n, other_ms, other_n = eng.profile_read(reset=True)
            eng.set_option("profile", 0)
            roof.update({"all_conv_kernels_ms_per_step": round(conv_ms, 3), "all_conv_launches_per_step": conv_n,
                         "other_unet_kernel_ms_per_step": round(other_ms, 3)})
            if sb_rows:   # round 4: the launches whose grid does not fill the chip (the 8x8 level at batch 64) run on the small-batch flavour
                sb_ms = sum(ms_ for _, ms_, _ in sb_rows); sb_gf = sum(gf_ * n_ for gf_, _, n_ in sb_rows)
                roof["small_batch_kernel"] = {"kernel": "td::conv_sb_kernel", "launches_per_step": sum(n_ for _, _, n_ in sb_rows), "kernel_ms_per_step": round(sb_ms, 3),
                                              "achieved": round(sb_gf / sb_ms, 2) if sb_ms > 0 else None, "unit": "TFLOP/s"}
            if g_n > 0:   # dominant kernel family: the LDS-DMA implicit-GEMM conv (terrain_diffusion_amd/csrc/conv_glds.hip)
                ach = g_flop / (g_ms * 1e-3) / 1e12
                lanes = 2 if (dual_on and min(tiles_per_step, 64) >= 32) else 1
                share = g_ms / (conv_ms + other_ms)
                iso = {"achieved": round(ach, 2), "frac": round(ach / peak, 4), "avg_launch_us": round(g_ms / g_n * 1e3, 3), "kernel_ms_per_step": round(g_ms, 3)}
                roof.update({"launches_per_step": g_n, "flop_per_launch": round(g_flop / g_n), "share_of_unet_kernel_time": round(share, 4), "lanes": lanes})
                if g_rows:
                    nl_ = sum(n_ for _, _, n_ in g_rows)
                    roof["traffic_algorithmic"] = round(sum(mb_ * n_ for mb_, _, n_ in g_rows) / nl_ * 1e6)
                    roof["traffic_algorithmic_strict"] = round(sum(mbs_ * n_ for _, mbs_, n_ in g_rows) / nl_ * 1e6)
                if lanes == 1:
                    roof.update(iso)
                else:
                    # Two concurrent half-batch lanes (engine option dual_stream): kernels of the two lanes overlap pairwise in the timed region, so
                    # the time the kernel family occupies the GPU is the step's wall time x its share of the U-Net kernel time, not the sum of
                    # per-launch durations.  achieved = algorithmic FLOP of its launches / that time; avg_launch_us = that time / launches.
                    act_ms = ms_per_step * share
                    eff = g_flop / (act_ms * 1e-3) / 1e12
                    roof.update({"achieved": round(eff, 2), "frac": round(eff / peak, 4), "avg_launch_us": round(act_ms / g_n * 1e3, 3), "kernel_ms_per_step": round(act_ms, 3),
                                 "single_lane": iso,
                                 "lanes_note": "batches of >= 32 windows run as two concurrent half-batch lanes on two HIP streams (independent tiles: the other lane's kernels fill "
                                               "the CUs that a kernel's last partial round, launch gap and epilogue tail leave idle). achieved / frac / avg_launch_us = the kernel family's "
                                               "algorithmic FLOP over the wall time it occupies in the TIMED region (step time x its share of U-Net kernel time; avg_launch_us = that time / "
                                               "the launches of a one-lane step). single_lane = the family's launches at the full batch of 64 timed one by one with HIP events on the engine's "
                                               "stream (extra step, option dual_stream = 0): the figure that `rocprofv3 --kernel-trace` of `bench.py --engine-opts dual_stream=0` reproduces "
                                               "(profiles/r06_bench_grid8_kernel_trace_summary.csv). Traced under rocprofv3 with two lanes, kernels overlap pairwise and their durations "
                                               "sum to more than the wall time."})
            else:
                roof.update({"achieved": roof["end_to_end_achieved"], "frac": roof["end_to_end_frac"]})
            if workload in ("grid8", "tiles") and tiles_per_step == 64 and args.dtype == "bf16":
                # HBM bytes per launch come from rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled per the MI355X guide's gfx950
                # correction, + WRITE_SIZE); PMC collection needs rocprofv3, so they are not re-measured here -- but they are only reported when the
                # counter file was collected with the very library that is loaded now (td_build_id stamp), never stale
                from terrain_diffusion_amd._lib import lib as _lib
                bid = _lib().td_build_id().decode()
                roof["library_build_id"] = bid
                if not os.path.exists(PROFILE_JSON):
                    roof["traffic_note"] = f"no counter file {os.path.relpath(PROFILE_JSON, ROOT)}"
                else:
                    pj = json.load(open(PROFILE_JSON))
                    if pj.get("library_build_id") != bid:
                        roof["traffic_note"] = (f"{os.path.relpath(PROFILE_JSON, ROOT)} was collected with library build {pj.get('library_build_id')}, the loaded library is "
                                                f"{bid}: stale counters are not reported (re-run tools/collect_profiles.sh)")
                    else:
                        tj = pj["kernels"]
                        ks_ = [v for k_, v in tj.items() if "conv_glds_kernel" in k_ and v.get("dispatches") and "hbm_read_bytes_per_launch" in v]
                        n_ = sum(v["dispatches"] for v in ks_)
                        if n_:
                            roof["traffic"] = round(sum(v["dispatches"] * (v["hbm_read_bytes_per_launch"] + v.get("hbm_write_bytes_per_launch", 0)) for v in ks_) / n_)
                            roof["traffic_source"] = os.path.relpath(PROFILE_JSON, ROOT)
                            if roof.get("traffic_algorithmic_strict"):
                                roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["traffic_algorithmic"], 3)
                                roof["traffic_over_algorithmic_strict"] = round(roof["traffic"] / roof["traffic_algorithmic_strict"], 3)
        else:
            roof.update({"achieved": roof["end_to_end_achieved"], "frac": roof["end_to_end_frac"]})
        result["roofline"] = roof
        if world == 1 and not args.no_latency:
            # BASELINE configs[1] as written: ONE tile through the same path (batch 1: weight-streaming / launch-latency bound, not MFMA bound)
            one_step(20_000, 1, "tiles"); sync()
            l0 = time.perf_counter()
            for r_ in range(3):
                one_step(20_001 + r_, 1, "tiles")
            sync()
            lat = (time.perf_counter() - l0) / 3
            result["latency_single_tile_ms"] = round(lat * 1e3, 3)
            result["single_tile_mp_per_s"] = round(0.262144 / lat, 3)
            hbm = E * WEIGHT_BYTES_BF16 / lat / 1e9
            result["roofline_single_tile"] = {"bound": "hbm", "achieved": round(hbm, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(hbm / PEAK_HBM_GBS, 4),
                                              "kernel": "td::conv_sb_kernel (small-batch flavour: K split over the waves of a workgroup) at the 64x64 / 32x32 levels, td::conv_s16_kernel "
                                                        "(64 px x 16 couts, no split-K over workgroups) at the 16x16 level, conv_sb + td::conv_splitk_reduce_kernel at the 8x8 level",
                                              "note": f"algorithmic bytes = {E} forwards x {WEIGHT_BYTES_BF16} B of bf16 weights (activations of one tile are negligible); "
                                                      "measured HBM bytes per forward: profiles/r06_batch1_hbm_traffic.json"}

        if world == 1 and not args.no_latency and workload == "grid8" and args.dtype == "bf16":
            # N = 1 point of the STRONG-scaling workload the driver runs at N > 1 (grid32, BASELINE configs[3]): one full step, timed live in this
            # run with the same engine options the sharded run uses, so that the 1 -> N curve has a same-run, same-node anchor
            from terrain_diffusion_amd.parallel import sample_base_diffusion_sharded
            # (the sharded run's options: the default plan -- 1024 windows = 16 whole batches of 64 on one rank -- and the engine's two sampler lanes)
            try:
                one_step(30_000, 64, "tiles"); sync()          # plans and graphs of the two 32-window lanes
                nt32 = len(_tile_starts(1056, 64, 32))
                cond32 = synthetic_cond_grid(nt32, nt32, device=dev)
                a0 = time.perf_counter()
                sample_base_diffusion_sharded(model, sch, (1, 5, 1056, 1056), cond32, noise_seed=42 + 5819, noise_origin=(0, 4096 * 31_000), max_batch=64, stats={}, **kw)
                sync()
                adt = time.perf_counter() - a0
                mp32 = (1056 * 8) ** 2 / 1e6
                result["strong_scaling_anchor"] = {"workload": "grid32 (BASELINE configs[3]) on ONE rank: 32x32 windows, 1056x1056 latents, 20 steps, default plan (16 whole batches of 64 windows), two sampler lanes",
                                                   "value": round(mp32 / adt, 4), "unit": "MP/s", "ms_per_step": round(adt * 1e3, 2), "steps_timed": 1,
                                                   "ms_per_64_window_batch": round(adt * 1e3 / 16, 2),
                                                   "note": "divide an N > 1 driver line's value by N x this to get the strong-scaling efficiency of that workload. A grid32 step is 16 batches "
                                                           "of 64 windows for 71.37 MP (0.0697 MP per window: every window overlaps its neighbours by half), a grid8 step one batch for 5.31 MP "
                                                           "(0.0829 MP per window: a quarter of the 8x8 canvas is border that one window covers alone) -- at equal time per batch grid32 reads "
                                                           "0.84x grid8's MP/s"}
            finally:
                pass

        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline = the oracle (CPU restatement pinned to the reference) on this host's cores, bounded sample
            # (the only place bench.py touches oracle/: the checker is what is timed here, never the product path)
            from oracle import tiling
            from oracle.unet import OracleUnet, synth_state_dict
            om = OracleUnet(cfg, synth_state_dict(cfg, seed=1234))
            ocond = tiling.synthetic_cond_grid(1, 1)
            run = lambda k: tiling.sample_base_diffusion_tiled(om, (1, 5, 64, 64), ocond, steps=k, tile_size=64)
            # pick the host thread count that is fastest for this workload (256 threads on the GPU node's host is ~150x slower
            # than 32 because of oversubscription); one solver step per candidate, method of evaluation/latency.py:72-91
            ncpu = os.cpu_count() or 1
            best_t, best_dt = None, None
            for nthreads in [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]:
                torch.set_num_threads(nthreads)
                run(1)
                c0 = time.perf_counter(); run(1); d = time.perf_counter() - c0
                if best_dt is None or d < best_dt:
                    best_t, best_dt = nthreads, d
                if d > 20:
                    break
            torch.set_num_threads(best_t)
            n_sample_steps = max(2, min(E, int(15.0 / max(best_dt, 1e-3))))
            c0 = time.perf_counter()
            run(n_sample_steps)
            cdt = time.perf_counter() - c0
            per_tile = cdt / n_sample_steps * E
            overlap = tiles_per_step * 0.262144 / mp_per_step   # windows' area / canvas area (1 for independent tiles)
            result["cpu_baseline"] = {"value": round(0.262144 / per_tile / overlap, 6), "unit": "MP/s", "cores": best_t, "kind": "port",
                                      "host_cpus": ncpu,
                                      "sample": f"oracle (torch fp32 CPU restatement pinned to the reference), 1 window x {n_sample_steps} of {E} solver steps "
                                                f"timed ({cdt:.1f} s on {best_t} threads, best of 8..128), scaled to {E} steps"
                                                + ("" if workload == "tiles" else f" and to the {tiles_per_step} overlapping windows of the canvas")}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
