"""Worker of tests/test_gpu_seam.py (one GPU; run as a child process so that a hanging RCCL call can be killed by PID): the C-ABI seam exchange
(include/td_seam.h) executed on hardware with a world-1 RCCL communicator.  With one GPU every message goes to the rank itself — RCCL's local
copy path, not xGMI — so this checks the library's plumbing (communicator, grouped ncclSend/ncclRecv on the caller's stream, message cuts,
slot layout), not link bandwidth.  Stages print one line each; the test asserts all of them.

  A  one message to self, on a side stream, bit-exact
  B  23 messages of ragged sizes to self in ONE group, absolute addresses (base NULL)
  C  2-, 4- and 8-rank plans (owned and extended regions) with every simulated rank's sends and receives posted to self: each rank ends up with exactly the windows its
     region needs, bit-exact (the cuts of sender and receiver pair up on hardware as tests/test_seam_cpu.py says they do on paper)
  D  engine sampling -> td_seam_exchange on the ENGINE's stream -> blend, 2 and 4 simulated ranks, batch-invariant mode: the assembled canvas is
     bit-identical to the unsharded sampler (the same claim test_sharded_sampling_simulated_ranks makes for the in-memory exchange)
  E  SeamComm.exchange_windows at world 1 (nothing crosses a seam): returns the rank's own windows, no RCCL call hangs on an empty exchange
  F  (measurement) the cost of POSTING one exchange of BASELINE configs[3] on 8 ranks, busiest rank, looped back on this GPU
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake(w, size=64):
    g = torch.Generator().manual_seed(1000 * w[0] + w[1])
    return torch.randn(5, size, size, generator=g)


def simulated_exchange(comm, cplan, tiles):
    """tiles[r]: rank r's own-order window outputs on the device.  Posts the messages of EVERY simulated rank on the world-1 communicator
    (peer 0 = self, absolute addresses) as one group and returns recv[r] (RECVS slot order)."""
    from terrain_diffusion_amd import seam
    world = cplan.world
    wb = tiles[0][0].numel() * tiles[0].element_size()
    recv = [torch.empty((len(cplan.windows_of(r, seam.RECVS)[0]),) + tuple(tiles[0].shape[1:]), dtype=tiles[0].dtype, device=tiles[0].device) for r in range(world)]
    msgs = [cplan.messages(r, wb) for r in range(world)]
    sends, recvs = [], []
    for s in range(world):
        for d in range(world):
            if s == d:
                continue
            sends += [(0, tiles[s].data_ptr() + off, n) for peer, off, n in msgs[s][0] if peer == d]
            recvs += [(0, recv[d].data_ptr() + off, n) for peer, off, n in msgs[d][1] if peer == s]
    comm.exchange(None, sends, None, recvs)
    return recv


def have_of(cplan, r, tiles, recv):
    from terrain_diffusion_amd import seam
    own, _ = cplan.windows_of(r, seam.OWN)
    local = {w: i for i, w in enumerate(own)}
    need, owners = cplan.windows_of(r, seam.NEEDED)
    have = {w: tiles[r][local[w]] for w, o in zip(need, owners) if o == r}
    for k, w in enumerate(cplan.windows_of(r, seam.RECVS)[0]):
        have[w] = recv[r][k]
    assert sorted(have) == sorted(need)
    return have


def main():
    from terrain_diffusion_amd import seam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    comm = seam.SeamComm.create(dev)
    assert comm.info() == (1, 0, 0), comm.info()

    # A
    side = torch.cuda.Stream()
    a = torch.randn(5, 64, 64, device=dev)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        comm.exchange(a, [(0, 0, a.numel() * 4)], b, [(0, 0, a.numel() * 4)])
    side.synchronize()
    assert torch.equal(a, b)
    print("SEAM_A_OK", flush=True)

    # B
    g = torch.Generator().manual_seed(5)
    src = torch.randn(1 << 20, generator=g).to(dev)
    dst = torch.zeros_like(src)
    sends, recvs, pos = [], [], 0
    for k in range(23):
        n = 4 * (1000 + 37 * k * k)
        sends.append((0, src.data_ptr() + pos, n))
        recvs.append((0, dst.data_ptr() + pos, n))
        pos += n + 64   # gaps stay zero
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        comm.exchange(None, sends, None, recvs)
    side.synchronize()
    want = torch.zeros_like(src)
    for _, p, n in sends:
        o = (p - src.data_ptr()) // 4
        want[o:o + n // 4] = src[o:o + n // 4]
    assert torch.equal(dst, want)
    print("SEAM_B_OK", flush=True)

    # C
    for world, (H, W) in ((2, (160, 224)), (4, (288, 288)), (8, (352, 608))):
        for extended in (False, True):
            cplan = seam.CShardPlan(H, W, 64, world, extended=extended)
            tiles = [torch.stack([fake(w) for w in cplan.windows_of(r, seam.OWN)[0]]).to(dev) for r in range(world)]
            torch.cuda.synchronize()
            recv = simulated_exchange(comm, cplan, tiles)
            torch.cuda.synchronize()
            for r in range(world):
                for w, t in have_of(cplan, r, tiles, recv).items():
                    assert torch.equal(t.cpu(), fake(w)), (world, extended, r, w)
    print("SEAM_C_OK", flush=True)

    # D
    import terrain_diffusion_amd as td
    from terrain_diffusion_amd.engine import get_engine
    from terrain_diffusion_amd.parallel import ShardPlan, _engine_stream, blend_region, engine_fns
    from oracle import tiling
    from oracle.unet import synth_state_dict, tiny_config
    cfg = tiny_config(64, 1)
    m = td.EDMUnet2D(**cfg, dtype="bf16", device=dev).load_state_dict(synth_state_dict(cfg, seed=77))
    eng = get_engine(dev)
    eng.set_option("batch_invariant", 1)
    sch = td.EDMDPMSolverMultistepScheduler()
    H, W, S, steps = 40, 56, 16, 5
    cond = tiling.synthetic_cond_grid(len(tiling.tile_starts(H, S, S // 2)), len(tiling.tile_starts(W, S, S // 2)))
    kw = dict(cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5))
    ref = td.sample_base_diffusion(m, sch, (1, 5, H, W), cond, steps=steps, tile_size=S, noise_seed=7, **kw)
    for world in (2, 4):
        plan = ShardPlan(H, W, S, world)
        cplan = comm.plan_for(plan)
        fns = engine_fns(m, sch, plan, cond, steps=steps, channels=5, noise_seed=7, noise_origin=(0, 0), max_batch=64, **kw)
        with _engine_stream(m):   # engine, torch glue and the RCCL group on ONE stream: nothing below synchronises before the blend
            from terrain_diffusion_amd.engine import engine_on_current_stream
            assert engine_on_current_stream(dev)
            tiles = [fns[0](plan.windows[r]).contiguous() for r in range(world)]
            recv = simulated_exchange(comm, cplan, tiles)
            full = torch.empty((5, H, W), device=dev)
            for r in range(world):
                y0, y1, x0, x1 = plan.regions[r]
                full[:, y0:y1, x0:x1] = blend_region(plan, r, have_of(cplan, r, tiles, recv), fns[1], fns[2], 5, 1.0 / 0.5)
        torch.cuda.synchronize()
        assert torch.equal(full[None], ref), world
    eng.set_option("batch_invariant", 0)
    print("SEAM_D_OK", flush=True)

    # E
    plan1 = ShardPlan(H, W, S, 1)
    mine = torch.stack([fake(w, S) for w in plan1.windows[0]]).to(dev)
    have = comm.exchange_windows(plan1, mine)
    assert sorted(have) == sorted(plan1.needed[0]) and all(torch.equal(have[w].cpu(), fake(w, S)) for w in have)
    print("SEAM_E_OK", flush=True)

    # F (a measurement, not a check): what one exchange of BASELINE configs[3] costs to POST -- the 32x32 window grid on 8 ranks, the busiest rank's
    # sends looped back to itself (same message count and sizes as on 8 GPUs; the copies are local, so the stream time is not the xGMI time)
    import time
    cplan = seam.CShardPlan(1056, 1056, 64, 8)
    wb = 5 * 64 * 64 * 4
    r = max(range(8), key=lambda k: len(cplan.messages(k, wb)[0]))
    s_msgs, r_msgs = cplan.messages(r, wb)
    mine = torch.randn(len(cplan.windows_of(r, seam.OWN)[0]), 5, 64, 64, device=dev)
    back = torch.empty(sum(n for _, _, n in s_msgs) // 4, device=dev)
    sends = [(0, off, n) for _, off, n in s_msgs]
    recvs, pos = [], 0
    for _, _, n in s_msgs:
        recvs.append((0, pos, n))
        pos += n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        comm.exchange(mine, sends, back, recvs)
    torch.cuda.synchronize()
    host = []
    e0.record()
    for it in range(20):
        t0 = time.perf_counter()
        comm.exchange(mine, sends, back, recvs)
        host.append(time.perf_counter() - t0)
    e1.record()
    torch.cuda.synchronize()
    print(f"SEAM_F configs[3] on 8 ranks, rank {r} of mesh {cplan.pr}x{cplan.pc}: {len(s_msgs)} sends + {len(r_msgs)} receives per exchange "
          f"({sum(n for _, _, n in s_msgs) / 1e6:.2f} MB out, {sum(n for _, _, n in r_msgs) / 1e6:.2f} MB in); looped back on one GPU ({len(sends)} + {len(recvs)} messages in one group): "
          f"host enqueue {1e6 * sorted(host)[len(host) // 2]:.0f} us median, stream time {1e3 * e0.elapsed_time(e1) / 20:.0f} us per exchange", flush=True)
    comm.close()


if __name__ == "__main__":
    main()
