"""Sharding of the tiled sampler across the GPUs of one node (one process per GPU, torch.distributed: RCCL on GPUs, gloo in tests).

The reference is single-process; its scaling mechanism is the algorithm itself: inside a phase every window is independent
(docs/index.html "Embarrassingly parallel"), and between phases / at the final blend a canvas pixel needs the windows that
overlap it (sample_diffusion_base.py:164-168).  So:

  * the window grid is cut into a 2-D block mesh, one block per rank; a rank samples only its own windows;
  * each rank OWNS the canvas region that starts at its first window's origin (up to the next block's first origin);
  * seam exchange: a rank sends the raw outputs of the windows that reach into a neighbour's region (its last window row /
    column / corner) to that neighbour — point-to-point isend/irecv over xGMI, ~80 KB per window, no all-reduce;
  * every rank blends its own region with the deterministic gather kernel in ascending (row, col) window order — the
    reference's loop order — so the result is bit-identical to the single-GPU result when the engine runs in batch-invariant
    mode (fixed kernel flavour, no split-K), and within fp32/bf16 rounding otherwise.
Exchanging window outputs instead of partial accumulator strips costs ~2.5x the bytes (still KBs) and buys a
canonical summation order independent of the GPU count.
"""
import math

import torch
import torch.distributed as dist


from .geometry import tile_starts as _tile_starts  # noqa: E402


def mesh_shape(world, n_rows, n_cols):
    """(pr, pc) with pr*pc == world, blocks as square as possible, never more parts than windows along an axis."""
    best = None
    for pr in range(1, world + 1):
        if world % pr:
            continue
        pc = world // pr
        if pr > n_rows or pc > n_cols:
            continue
        score = abs(math.log((n_rows / pr) / (n_cols / pc)))
        if best is None or score < best[0]:
            best = (score, pr, pc)
    if best is None:
        raise ValueError(f"cannot place {world} ranks on a {n_rows}x{n_cols} window grid")
    return best[1], best[2]


def _cuts(n, parts):
    return [(i * n) // parts for i in range(parts + 1)]


class ShardPlan:
    """Static description of who samples which window, who owns which canvas region and which window outputs cross seams."""

    def __init__(self, H, W, tile_size, world, stride=None, extended=False):
        """extended=False: a rank's region is the canvas area it OWNS (the regions tile the canvas).  extended=True: the region is the bounding
        box of the rank's own windows -- what it needs of an intermediate phase's blended canvas to cut the next phase's window inputs from
        (multi-phase samplers); extended regions overlap, and `needed` / `sends` grow by the windows that reach into the wider box."""
        self.H, self.W, self.size, self.world = H, W, tile_size, world
        self.extended = bool(extended)
        stride = stride or tile_size // 2
        self.stride = stride
        self.h_starts, self.w_starts = _tile_starts(H, tile_size, stride), _tile_starts(W, tile_size, stride)
        nr, nc = len(self.h_starts), len(self.w_starts)
        self.pr, self.pc = mesh_shape(world, nr, nc)
        self.row_cuts, self.col_cuts = _cuts(nr, self.pr), _cuts(nc, self.pc)
        self.owner = {}
        self.windows = [[] for _ in range(world)]
        for br in range(self.pr):
            for bc in range(self.pc):
                r = br * self.pc + bc
                for ic in range(self.row_cuts[br], self.row_cuts[br + 1]):
                    for jc in range(self.col_cuts[bc], self.col_cuts[bc + 1]):
                        self.owner[(ic, jc)] = r
                        self.windows[r].append((ic, jc))
        # owned canvas regions: from the origin of the block's first window to the origin of the next block's first window
        self.regions = []
        for br in range(self.pr):
            for bc in range(self.pc):
                y0 = self.h_starts[self.row_cuts[br]] if br > 0 else 0
                y1 = self.h_starts[self.row_cuts[br + 1]] if br + 1 < self.pr else H
                x0 = self.w_starts[self.col_cuts[bc]] if bc > 0 else 0
                x1 = self.w_starts[self.col_cuts[bc + 1]] if bc + 1 < self.pc else W
                if extended:
                    y1 = self.h_starts[self.row_cuts[br + 1] - 1] + tile_size
                    x1 = self.w_starts[self.col_cuts[bc + 1] - 1] + tile_size
                    y0, x0 = self.h_starts[self.row_cuts[br]], self.w_starts[self.col_cuts[bc]]
                self.regions.append((y0, min(y1, H), x0, min(x1, W)))
        # windows intersecting each region, and the seam traffic (src -> dst: windows of src that dst's region needs)
        self.needed = []
        for r, (y0, y1, x0, x1) in enumerate(self.regions):
            need = [(ic, jc) for ic, hs in enumerate(self.h_starts) for jc, ws in enumerate(self.w_starts)
                    if hs < y1 and hs + tile_size > y0 and ws < x1 and ws + tile_size > x0]
            self.needed.append(need)
        self.sends = {(s, d): [w for w in self.needed[d] if self.owner[w] == s] for s in range(world) for d in range(world) if s != d}
        self.sends = {k: v for k, v in self.sends.items() if v}

    def seam_bytes(self, channels=5):
        return {k: len(v) * channels * self.size * self.size * 4 for k, v in self.sends.items()}


def exchange_windows(plan: ShardPlan, rank, my_tiles, group=None, seam_comm=None):
    """my_tiles: [len(plan.windows[rank]), C, S, S] outputs of this rank's windows (device of the backend).  Returns
    {(ic, jc): tile} for every window this rank's region needs (local ones included).
    seam_comm (seam.SeamComm): send the windows through the C-ABI's td_seam_exchange_windows (include/td_seam.h: one grouped ncclSend/ncclRecv on
    torch's current stream = the engine's stream inside the sharded samplers) instead of torch.distributed; same windows, same bits."""
    if seam_comm is not None:
        if seam_comm.rank != rank or seam_comm.world != plan.world:
            raise ValueError(f"seam_comm is rank {seam_comm.rank} of {seam_comm.world}, the exchange is for rank {rank} of {plan.world}")
        return seam_comm.exchange_windows(plan, my_tiles.contiguous())
    local_index = {w: i for i, w in enumerate(plan.windows[rank])}
    have = {w: my_tiles[local_index[w]] for w in plan.needed[rank] if plan.owner[w] == rank}
    ops, recv_bufs = [], {}
    # gloo moves host memory only: device tiles are staged through the host (CPU tests, and the one-GPU dry run of bench.py's N > 1 branch);
    # RCCL ("nccl") sends the device buffers themselves over xGMI
    host_stage = my_tiles.is_cuda and dist.get_backend(group) == "gloo"
    xdev = torch.device("cpu") if host_stage else my_tiles.device
    for (s, d), wins in sorted(plan.sends.items()):
        if s == rank:
            buf = torch.stack([my_tiles[local_index[w]] for w in wins]).contiguous().to(xdev)
            ops.append(dist.P2POp(dist.isend, buf, d, group))
        elif d == rank:
            buf = torch.empty((len(wins),) + tuple(my_tiles.shape[1:]), dtype=my_tiles.dtype, device=xdev)
            recv_bufs[s] = (buf, wins)
            ops.append(dist.P2POp(dist.irecv, buf, s, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if my_tiles.is_cuda:
            # RCCL's wait() orders the transfer before later work on torch's CURRENT stream.  The sharded samplers put the engine on that very
            # stream (_engine_stream below: td_engine_set_stream), so the blend that follows is ordered behind the transfers by the stream itself
            # and the host never waits; only a caller that left the engine on its own stream still needs the synchronisation.
            from .engine import engine_on_current_stream
            if not engine_on_current_stream(my_tiles.device):
                torch.cuda.current_stream(my_tiles.device).synchronize()
    for s, (buf, wins) in recv_bufs.items():
        if host_stage:
            buf = buf.to(my_tiles.device)
        for i, w in enumerate(wins):
            have[w] = buf[i]
    return have


_EXCHANGE_STREAMS = {}


def _engine_stream(model):
    """Context manager for the engine-backed sharded samplers: the engine, torch's tensor glue and the RCCL seam exchange all run on ONE
    side stream per device (torch's default stream is the legacy NULL stream, which the engine's captured graphs cannot use), so that
    sample -> exchange -> blend is ordered by the stream and not by host synchronisations.  Engine calls stay synchronous inside (results are
    complete on return, as everywhere else); leaving the context drains the stream."""
    import contextlib
    dev = torch.device(getattr(model, "device", "cpu"))
    if dev.type != "cuda":
        return contextlib.nullcontext()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    from .engine import engine_on_current_stream
    if engine_on_current_stream(dev):
        return contextlib.nullcontext()   # the caller already put the engine on torch's current stream (Engine.on_stream): keep it, and its mode
    if idx not in _EXCHANGE_STREAMS:
        _EXCHANGE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return model.engine.on_stream(_EXCHANGE_STREAMS[idx], asynchronous=False)


def blend_region(plan: ShardPlan, rank, have, blend_fn, normalize_fn, channels, scale):
    """Blends this rank's owned canvas region from the window outputs in `have` (ascending window order) -> (C, h, w)."""
    y0, y1, x0, x1 = plan.regions[rank]
    wins = sorted(have)
    tiles = torch.stack([have[w] for w in wins]).contiguous()
    canvas = torch.zeros((channels + 1, y1 - y0, x1 - x0), dtype=torch.float32, device=tiles.device)
    blend_fn(canvas, tiles, wins, [h - y0 for h in plan.h_starts], [w - x0 for w in plan.w_starts], plan.size)
    return normalize_fn(canvas, scale)


def engine_fns(model, scheduler, plan, cond_inputs, *, cond_means, cond_stds, noise_level, histogram_raw, steps, channels, noise_seed, noise_origin,
               max_batch):
    """(sample_fn, blend_fn, normalize_fn) backed by the HIP engine."""
    from . import sampling as _s

    def sample_fn(windows):
        outs = []
        cond = torch.as_tensor(cond_inputs, dtype=torch.float32)
        for b0 in range(0, len(windows), max_batch):
            chunk = windows[b0:b0 + max_batch]
            origins = [(noise_origin[0] + plan.h_starts[ic], noise_origin[1] + plan.w_starts[jc]) for ic, jc in chunk]
            c58 = _s._tile_conditioning(cond, chunk, histogram_raw, cond_means, cond_stds, noise_level)
            outs.append(_s.sample_independent_tiles(model, scheduler, origins, c58, steps=steps, tile_size=plan.size, channels=channels,
                                                    noise_seed=noise_seed, return_raw=True))
        return torch.cat(outs)

    def blend_fn(canvas, tiles, wins, hs, ws, size):
        _s.blend_windows(model.engine, canvas, tiles, wins, hs, ws, size, accumulate=False)

    def normalize_fn(canvas, scale):
        return _s.blend_normalize(model.engine, canvas, scale)

    return sample_fn, blend_fn, normalize_fn


def sample_base_diffusion_sharded(model, scheduler, shape, cond_inputs, *, cond_means, cond_stds, noise_level=0.0, histogram_raw, steps=15,
                                  tile_size=64, noise_seed=42 + 5819, noise_origin=(0, 0), max_batch=64, group=None, gather_to=None,
                                  sample_fn=None, blend_fn=None, normalize_fn=None, stats=None, seam_comm=None, _on_engine_stream=False):
    """Sharded sample_base_diffusion (terrain_diffusion/training/evaluation/sample_diffusion_base.py:115-168).
    Returns (region_tensor (C,h,w), (y0,y1,x0,x1)) for this rank, or the assembled (1,C,H,W) on rank `gather_to`.
    sample_fn / blend_fn / normalize_fn default to the HIP engine; tests inject CPU stand-ins to exercise the plumbing under gloo.
    seam_comm: a seam.SeamComm -> the seam exchange goes through the C-ABI (td_seam_exchange_windows) instead of torch.distributed."""
    if sample_fn is None and model is not None and not _on_engine_stream:
        # engine path: everything below runs on the engine's side stream (see _engine_stream); the flag travels as an argument, not as global
        # state, so concurrent callers (threads, devices) cannot see each other's
        with _engine_stream(model):
            return sample_base_diffusion_sharded(model, scheduler, shape, cond_inputs, cond_means=cond_means, cond_stds=cond_stds, noise_level=noise_level,
                                                 histogram_raw=histogram_raw, steps=steps, tile_size=tile_size, noise_seed=noise_seed, noise_origin=noise_origin,
                                                 max_batch=max_batch, group=group, gather_to=gather_to, stats=stats, seam_comm=seam_comm, _on_engine_stream=True)
    B, C_, H, W = shape
    assert B == 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    plan = ShardPlan(H, W, tile_size, world)
    sd = float(scheduler.config.sigma_data)
    if sample_fn is None:
        sample_fn, blend_fn, normalize_fn = engine_fns(model, scheduler, plan, cond_inputs, cond_means=cond_means, cond_stds=cond_stds, noise_level=noise_level,
                                                       histogram_raw=histogram_raw, steps=steps, channels=C_, noise_seed=noise_seed, noise_origin=noise_origin,
                                                       max_batch=max_batch)
    my_tiles = sample_fn(plan.windows[rank])
    if stats is not None:   # seam accounting for bench.py: bytes this rank sends / receives and the wall time of the exchange itself
        import time
        if my_tiles.is_cuda:
            torch.cuda.synchronize(my_tiles.device)
        t0 = time.perf_counter()
    have = exchange_windows(plan, rank, my_tiles, group, seam_comm) if world > 1 else {w: my_tiles[i] for i, w in enumerate(plan.windows[rank])}
    if stats is not None:
        if my_tiles.is_cuda:
            torch.cuda.synchronize(my_tiles.device)
        sb = plan.seam_bytes(C_)
        stats["exchange_s"] = stats.get("exchange_s", 0.0) + (time.perf_counter() - t0)
        stats["seam_bytes_sent"] = sum(v for (s_, d_), v in sb.items() if s_ == rank)
        stats["seam_bytes_received"] = sum(v for (s_, d_), v in sb.items() if d_ == rank)
        stats["seam_bytes_total"] = sum(sb.values())
        stats["mesh"] = [plan.pr, plan.pc]
        stats["windows_this_rank"] = len(plan.windows[rank])
    region = blend_region(plan, rank, have, blend_fn, normalize_fn, C_, 1.0 / sd)
    if gather_to is None:
        return region, plan.regions[rank]
    if world == 1:
        return region[None]
    # assemble on one rank (optional; production keeps the canvas sharded like the reference keeps tiles in a tile store)
    if rank == gather_to:
        full = torch.empty((C_, H, W), dtype=torch.float32, device=region.device)
        for r, (y0, y1, x0, x1) in enumerate(plan.regions):
            if r == rank:
                full[:, y0:y1, x0:x1] = region
            else:
                buf = torch.empty((C_, y1 - y0, x1 - x0), dtype=torch.float32, device=region.device)
                dist.recv(buf, r, group)
                full[:, y0:y1, x0:x1] = buf
        return full[None]
    dist.send(region.contiguous(), gather_to, group)
    return None


def consistency_engine_fns(model, plan, cond_inputs, *, cond_means, cond_stds, noise_level, histogram_raw, channels, noise_seed, noise_origin, max_batch, sigma_data):
    """(step_fn, blend_fn, normalize_fn) backed by the HIP engine for the multi-phase sampler: step_fn(windows, k, t, prev_tiles) runs ONE
    trig-flow consistency step (world_pipeline.py:1097-1129) on the given windows; prev_tiles is None in the first phase."""
    from . import sampling as _s
    from . import noise as _noise
    from ._lib import lib, check
    from .engine import ptr
    dev = model.device

    def step_fn(windows, k, t, prev_tiles):
        cond = torch.as_tensor(cond_inputs, dtype=torch.float32)
        outs = []
        for b0 in range(0, len(windows), max_batch):
            chunk = windows[b0:b0 + max_batch]
            origins = [(noise_origin[0] + plan.h_starts[ic], noise_origin[1] + plan.w_starts[jc]) for ic, jc in chunk]
            z = _noise.gaussian_noise_patches(noise_seed + k, origins, plan.size, plan.size, channels=channels, tile_h=64, tile_w=64, device=dev)
            c58 = _s._tile_conditioning(cond, chunk, histogram_raw, cond_means, cond_stds, noise_level).to(dev).contiguous()
            prev = None if prev_tiles is None else prev_tiles[b0:b0 + max_batch].contiguous()
            out = torch.empty_like(z)
            check(lib().td_sample_consistency(model._h, z.shape[0], plan.size, plan.size, float(t), float(sigma_data), ptr(prev), ptr(z), ptr(c58), ptr(out)))
            outs.append(out)
        return torch.cat(outs)

    def blend_fn(canvas, tiles, wins, hs, ws, size):
        _s.blend_windows(model.engine, canvas, tiles, wins, hs, ws, size, accumulate=False)

    def normalize_fn(canvas, scale):
        return _s.blend_normalize(model.engine, canvas, scale)

    return step_fn, blend_fn, normalize_fn


def sample_base_consistency_sharded(model, scheduler, shape, cond_inputs, *, cond_means, cond_stds, noise_level=0.0, histogram_raw, intermediate_t=0.0,
                                    tile_size=64, noise_seed=42 + 5819, noise_origin=(0, 0), max_batch=64, group=None, gather_to=None,
                                    step_fn=None, blend_fn=None, normalize_fn=None, stats=None, seam_comm=None, _on_engine_stream=False):
    """Sharded sample_base_consistency (sample_diffusion_base.py:171-268; the latent stage's blended trig-flow phases, world_pipeline.py:1133-1203).
    T phases = T seam exchanges: in every phase a rank runs one consistency step on ITS windows, then receives the outputs of the
    neighbours' windows it needs and blends
      * after an intermediate phase: the bounding box of its own windows (ShardPlan(extended=True)) -- the next phase cuts every window's input
        sample out of that box, so no blended canvas ever crosses a seam, only window outputs do (canonical per-pixel summation order);
      * after the last phase: the region it owns.
    In batch-invariant engine mode the assembled canvas is bit-identical to the one-rank sampler.  Returns like sample_base_diffusion_sharded."""
    if step_fn is None and model is not None and not _on_engine_stream:
        with _engine_stream(model):   # engine path: sample -> exchange -> blend of every phase on the engine's side stream
            return sample_base_consistency_sharded(model, scheduler, shape, cond_inputs, cond_means=cond_means, cond_stds=cond_stds, noise_level=noise_level,
                                                   histogram_raw=histogram_raw, intermediate_t=intermediate_t, tile_size=tile_size, noise_seed=noise_seed,
                                                   noise_origin=noise_origin, max_batch=max_batch, group=group, gather_to=gather_to, stats=stats,
                                                   seam_comm=seam_comm, _on_engine_stream=True)
    B, C_, H, W = shape
    assert B == 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    own, ext = ShardPlan(H, W, tile_size, world), ShardPlan(H, W, tile_size, world, extended=True)
    sd = float(scheduler.config.sigma_data)
    t0 = float(torch.atan(torch.tensor(float(scheduler.config.sigma_max), dtype=torch.float32) / sd))
    t_scalars = (t0, float(torch.tensor(intermediate_t, dtype=torch.float32))) if intermediate_t > 0 else (t0,)
    if step_fn is None:
        step_fn, blend_fn, normalize_fn = consistency_engine_fns(model, own, cond_inputs, cond_means=cond_means, cond_stds=cond_stds, noise_level=noise_level,
                                                                 histogram_raw=histogram_raw, channels=C_, noise_seed=noise_seed, noise_origin=noise_origin,
                                                                 max_batch=max_batch, sigma_data=sd)
    mine = own.windows[rank]
    prev_tiles, region = None, None
    for k, t in enumerate(t_scalars):
        last = k == len(t_scalars) - 1
        plan = own if last else ext
        out = step_fn(mine, k, t, prev_tiles)
        have = exchange_windows(plan, rank, out, group, seam_comm) if world > 1 else {w: out[i] for i, w in enumerate(mine)}
        if stats is not None:
            stats["exchanges"] = stats.get("exchanges", 0) + (1 if world > 1 else 0)
            stats["seam_bytes_total"] = stats.get("seam_bytes_total", 0) + sum(plan.seam_bytes(C_).values())
        region = blend_region(plan, rank, have, blend_fn, normalize_fn, C_, (1.0 / sd) if last else 1.0)
        if not last:   # next phase's window inputs, cut from this rank's own blended box
            y0, _, x0, _ = ext.regions[rank]
            prev_tiles = torch.stack([region[:, own.h_starts[ic] - y0:own.h_starts[ic] - y0 + tile_size, own.w_starts[jc] - x0:own.w_starts[jc] - x0 + tile_size]
                                      for ic, jc in mine]).contiguous()
    if gather_to is None:
        return region, own.regions[rank]
    if world == 1:
        return region[None]
    if rank == gather_to:
        full = torch.empty((C_, H, W), dtype=torch.float32, device=region.device)
        for r, (y0, y1, x0, x1) in enumerate(own.regions):
            if r == rank:
                full[:, y0:y1, x0:x1] = region
            else:
                buf = torch.empty((C_, y1 - y0, x1 - x0), dtype=torch.float32, device=region.device)
                dist.recv(buf, r, group)
                full[:, y0:y1, x0:x1] = buf
        return full[None]
    dist.send(region.contiguous(), gather_to, group)
    return None


def shard_requests(boxes, world=None, rank=None):
    """Request-replica mode for WorldPipeline / the cascade on a multi-GPU node (BASELINE configs[4]): every rank holds the whole lazy graph
    (weights are small, windows are recomputed where needed) and serves a contiguous, spatially compact share of the request boxes.  Boxes are
    ordered along a Z-curve of their centres first, so that a rank's requests share upstream coarse / latent windows (the cascade's cost is
    dominated by the latent stage: neighbouring decoder windows reuse it) and no window ever has to cross a GPU boundary -- the cascade needs no
    data-path collective.  Returns the list of (index, box) this rank serves; the union over ranks is every box exactly once."""
    world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else world
    rank = (dist.get_rank() if dist.is_initialized() else 0) if rank is None else rank

    def z(ci, cj):
        ci, cj = int(ci) + (1 << 30), int(cj) + (1 << 30)
        v = 0
        for b in range(31):
            v |= ((ci >> b) & 1) << (2 * b + 1) | ((cj >> b) & 1) << (2 * b)
        return v
    order = sorted(range(len(boxes)), key=lambda k: z((boxes[k][0] + boxes[k][2]) // 2, (boxes[k][1] + boxes[k][3]) // 2))
    lo, hi = (rank * len(order)) // world, ((rank + 1) * len(order)) // world
    return [(k, boxes[k]) for k in order[lo:hi]]
