"""Which batches does the cascade actually run?  One warm 3x3-request step of the cascade bench with every sampler call logged (model, batch, H x W)."""
import sys, os, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from terrain_diffusion_amd import _lib
from terrain_diffusion_amd.cascade_bench import COARSE_CONFIG, DECODER_CONFIG
from terrain_diffusion_amd.synthetic import synthetic_state_dict
from bench import BASE_CONFIG
dev = "cuda:0"
models = []
for cfg, seed in ((COARSE_CONFIG, 11), (BASE_CONFIG, 1234), (DECODER_CONFIG, 2468)):
    m = td.EDMUnet2D(**cfg, dtype="bf16", device=dev)
    models.append(m.load_state_dict(synthetic_state_dict(m, seed=seed)))
names = {models[0]._h.value: "coarse", models[1]._h.value: "base", models[2]._h.value: "decoder"}
wp = td.WorldPipeline.from_models(*models, seed=4242, dtype="bf16", device=dev, cache_limit=100 * 2 ** 20, latents_batch_size=(1, 2, 4, 8, 16, 32, 64)).bind()
L = _lib.lib()
log = []
for fn in ("td_sample_consistency", "td_sample_consistency_img", "td_sample_edm_img", "td_sample_edm"):
    orig = getattr(L, fn)
    def wrap(*a, _o=orig, _f=fn):
        t0 = time.perf_counter(); r = _o(*a); torch.cuda.synchronize(); log.append((_f, names.get(a[0].value if hasattr(a[0], "value") else a[0], "?"), int(a[1]), int(a[2]), (time.perf_counter() - t0) * 1e3)); return r
    setattr(L, fn, wrap)
Q, R = 1024, 3072
def step(i):
    i0, j0 = 100_000 * (i + 1), -50_000 * (i + 1)
    for a in range(0, R, Q):
        for b in range(0, R, Q):
            wp.get(i0 + a, j0 + b, i0 + a + Q, j0 + b + Q)
step(-1); log.clear()
t0 = time.perf_counter(); step(0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
agg = collections.defaultdict(lambda: [0, 0.0])
for f, m, n, h, ms in log: a = agg[(m, f.replace("td_sample_", ""), n, h)]; a[0] += 1; a[1] += ms
print(f"one step (9 requests of 1024^2): {dt * 1e3:.0f} ms wall with per-call synchronisation, {len(log)} sampler calls")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {k[0]:8s} {k[1]:16s} batch {k[2]:3d}  {k[3]:4d}px  calls {v[0]:3d}  total {v[1]:7.1f} ms  ({v[1] / v[0]:.2f} ms per call, {v[1] / v[0] / k[2]:.3f} per window)")
