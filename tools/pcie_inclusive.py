"""The default bench workload (BASELINE configs[2]) timed twice in one process: inputs and result resident in HBM (what bench.py's `value` is) and
with the boundary handing over HOST buffers — conditioning grid on the host, canvas copied back to pageable host memory inside the timed region.
Prints both rates; DESIGN.md quotes the second as the PCIe-inclusive rate (it is never `value`)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrain_diffusion_amd as td  # noqa: E402
from bench import BASE_CONFIG  # noqa: E402
from terrain_diffusion_amd.engine import get_engine  # noqa: E402
from terrain_diffusion_amd.sampling import _tile_starts  # noqa: E402
from terrain_diffusion_amd.synthetic import synthetic_cond_grid, synthetic_state_dict  # noqa: E402


def main(steps=3):
    dev = "cuda:0"
    eng = get_engine(dev)
    model = td.EDMUnet2D(**dict(BASE_CONFIG), dtype="bf16", device=dev)
    model.load_state_dict(synthetic_state_dict(model, seed=1234))
    sch = td.EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
    H = W = 288
    nt = len(_tile_starts(H, 64, 32))
    kw = dict(cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5), steps=20, tile_size=64)
    cond_dev = synthetic_cond_grid(nt, nt, device=dev)
    cond_host = cond_dev.cpu()

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()

    def run(host):
        for i in range(1):
            td.sample_base_diffusion(model, sch, (1, 5, H, W), cond_host if host else cond_dev, noise_origin=(0, 4096 * i), **kw)
        sync()
        t0 = time.perf_counter()
        nbytes = 0
        for i in range(steps):
            out = td.sample_base_diffusion(model, sch, (1, 5, H, W), cond_host if host else cond_dev, noise_origin=(0, 4096 * (1 + i)), **kw)
            if host:
                out = out.cpu()
                nbytes += out.numel() * out.element_size()
        sync()
        dt = (time.perf_counter() - t0) / steps
        return (H * 8) ** 2 / 1e6 / dt, dt * 1e3, nbytes // max(steps, 1)

    a = run(False)
    b = run(True)
    a2 = run(False)
    print(f"resident in HBM: {a[0]:.3f} MP/s ({a[1]:.2f} ms per step); again after the host run: {a2[0]:.3f} MP/s ({a2[1]:.2f} ms)")
    print(f"host buffers at the boundary (conditioning grid {cond_host.numel() * 4} B in, canvas {b[2]} B out to pageable memory, inside the timed region): "
          f"{b[0]:.3f} MP/s ({b[1]:.2f} ms per step) = {100 * (b[1] / min(a[1], a2[1]) - 1):+.2f} % on the step")


if __name__ == "__main__":
    main()
