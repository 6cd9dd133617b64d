"""torchrun worker of tests/test_gpu_bench_config.py::test_nccl_seam_exchange_two_gpus: the RCCL seam exchange on real device tensors.
Every rank fabricates deterministic 'window outputs' (a function of the window index only), exchanges them with exchange_windows over the
nccl backend and checks that every window its region needs arrived bit-exactly; then the sharded sampler runs on a tiny model in
batch-invariant mode and the gathered canvas must equal the single-GPU canvas bit for bit.  With the argument `capi`
(test_gpu_seam.py::test_capi_seam_exchange_two_gpus) both are repeated with the seam exchange going through the C-ABI (libtd_seam.so: its own RCCL
communicator, one grouped ncclSend/ncclRecv on the engine's stream); the two transports are separate tests so that a failure names its transport."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from terrain_diffusion_amd.parallel import ShardPlan, exchange_windows, sample_base_diffusion_sharded
    dev = torch.device("cuda", local)
    plan = ShardPlan(160, 224, 64, world)

    def fake(w):
        g = torch.Generator().manual_seed(1000 * w[0] + w[1])
        return torch.randn(5, 64, 64, generator=g)
    mine = torch.stack([fake(w) for w in plan.windows[rank]]).to(dev)
    have = exchange_windows(plan, rank, mine)
    assert sorted(have) == sorted(plan.needed[rank])
    for w, t in have.items():
        assert torch.equal(t.cpu(), fake(w)), (rank, w)

    import terrain_diffusion_amd as td
    from terrain_diffusion_amd.engine import get_engine
    from oracle import tiling
    from oracle.unet import synth_state_dict, tiny_config
    cfg = tiny_config(64, 1)
    m = td.EDMUnet2D(**cfg, dtype="bf16", device=dev).load_state_dict(synth_state_dict(cfg, seed=77))
    get_engine(dev).set_option("batch_invariant", 1)
    sch = td.EDMDPMSolverMultistepScheduler()
    H, W = 40, 56
    cond = tiling.synthetic_cond_grid(len(tiling.tile_starts(H, 16, 8)), len(tiling.tile_starts(W, 16, 8)))
    kw = dict(cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5), steps=5, tile_size=16)
    full = sample_base_diffusion_sharded(m, sch, (1, 5, H, W), cond, gather_to=0, **kw)
    if rank == 0:
        single = td.sample_base_diffusion(m, sch, (1, 5, H, W), cond, **kw)
        assert torch.equal(full, single), float((full - single).abs().max())
        print("NCCL_EXCHANGE_OK", flush=True)
    if "capi" in sys.argv[1:]:
        # the same exchange and the same sharded sampler through the C-ABI transport (include/td_seam.h: td_seam_exchange_windows on torch's current stream)
        from terrain_diffusion_amd.seam import SeamComm
        sc = SeamComm.create(dev)
        have_c = exchange_windows(plan, rank, mine, seam_comm=sc)
        torch.cuda.synchronize()
        assert sorted(have_c) == sorted(plan.needed[rank])
        for w, t in have_c.items():
            assert torch.equal(t.cpu(), fake(w)), ("c-abi", rank, w)
        full_c = sample_base_diffusion_sharded(m, sch, (1, 5, H, W), cond, gather_to=0, seam_comm=sc, **kw)
        if rank == 0:
            assert torch.equal(full_c, single)
            print("CAPI_EXCHANGE_OK", flush=True)
        sc.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
