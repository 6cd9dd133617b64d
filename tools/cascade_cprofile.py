"""cProfile of one warm cascade step (9 requests of 1024^2): where the HOST time goes."""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from terrain_diffusion_amd.cascade_bench import COARSE_CONFIG, DECODER_CONFIG
from terrain_diffusion_amd.synthetic import synthetic_state_dict
from bench import BASE_CONFIG
dev = "cuda:0"
models = []
for cfg, seed in ((COARSE_CONFIG, 11), (BASE_CONFIG, 1234), (DECODER_CONFIG, 2468)):
    m = td.EDMUnet2D(**cfg, dtype="bf16", device=dev)
    models.append(m.load_state_dict(synthetic_state_dict(m, seed=seed)))
wp = td.WorldPipeline.from_models(*models, seed=4242, dtype="bf16", device=dev, cache_limit=100 * 2 ** 20, latents_batch_size=(1, 2, 4, 8, 16, 32, 64)).bind()
Q, R = 1024, 3072
def step(i):
    i0, j0 = 100_000 * (i + 1), -50_000 * (i + 1)
    for a in range(0, R, Q):
        for b in range(0, R, Q):
            wp.get(i0 + a, j0 + b, i0 + a + Q, j0 + b + Q)
step(-1); step(-2); torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); step(0); torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(f"one step under cProfile: {dt * 1e3:.0f} ms"); print(s.getvalue()[:9000])
