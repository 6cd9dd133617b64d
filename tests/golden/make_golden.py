"""Generates the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES.

Runs only in the build container (needs /root/reference).  The reference has no tests or golden
vectors of its own, so every fixture here is an output of the reference code on
seeded inputs; inputs are reproducible on any machine from the portable RNG (oracle/rng.py), so only
outputs (and tiny inputs) are stored.  Fixtures are data: no reference source text is stored.

    python tests/golden/make_golden.py            # all
    python tests/golden/make_golden.py rng unet   # subsets

Shim: `diffusers` / `numba` are not installed; they are replaced by ~30 lines of serialisation-only
stand-ins so the reference's arithmetic modules import unchanged.
"""
import ast
import functools
import inspect
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------- reference import shim
def install_shim():
    class _Cfg(dict):
        __getattr__ = dict.get

    class ConfigMixin:
        def register_to_config(self, **kw):
            self.__dict__.setdefault("_cfg", _Cfg()).update(kw)
        config = property(lambda self: self._cfg)

    def register_to_config(init):
        @functools.wraps(init)
        def wrap(self, *a, **kw):
            ba = inspect.signature(init).bind(self, *a, **kw)
            ba.apply_defaults()
            ConfigMixin.register_to_config(self, **{k: v for k, v in ba.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return wrap

    class ModelMixin(torch.nn.Module):
        pass

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    _mod("diffusers", ConfigMixin=ConfigMixin)
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.models")
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.utils")
    _mod("diffusers.utils.torch_utils",
         randn_tensor=lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, device=device, dtype=dtype))
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin, SchedulerOutput=SchedulerOutput)
    _mod("numba", njit=lambda *a, **k: a[0] if a and callable(a[0]) else (lambda f: f))
    sys.path.insert(0, REF)


def extract_functions(path, names, namespace):
    """exec only the named pure FunctionDefs of a reference file whose module import needs absent deps."""
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------- fixtures
def gen_rng():
    from terrain_diffusion.inference import portable_rng as pr
    ns = extract_functions(os.path.join(REF, "terrain_diffusion/inference/world_pipeline.py"),
                           {"_tile_seed", "gaussian_noise_patch"},
                           {"np": np, "torch": torch, "fill_standard_normal": pr.fill_standard_normal})
    out = {}
    seeds = [1, 42, 123, 0xFFFFFFFFFFFFFFFF, 785323394005271306]
    out["stream_seeds"] = np.array(seeds, dtype=np.uint64)
    streams = []
    for s in seeds:
        st, o = s, []
        for _ in range(64):
            st, u = pr._pcg64_next(st)
            o.append(u)
        streams.append(o)
    out["streams"] = np.array(streams, dtype=np.uint32)
    out["next_seed_in"] = np.array([1, 42, 2 ** 63 + 5], dtype=np.uint64)
    out["next_seed_out"] = np.array([pr.next_seed(int(s)) for s in out["next_seed_in"]], dtype=np.uint64)
    out["normal_seed123_n4097"] = pr.standard_normal(123, (4097,))
    out["normal_seed7_f64_n33"] = pr.standard_normal(7, (33,), dtype=np.float64)
    ts = [(42, 0, 0), (42, -1, 3), (42, 5, -7), (0xFFFFFFFFFFFFFFFF, -2 ** 31, 2 ** 31 - 1), (1234567890123, 100000, -100000),
          (42 + 5819, 17, 17), (2 ** 64 + 5, 1, 1), (-3, 2, 2)]
    out["tile_seed_in"] = np.array([[a & 0xFFFFFFFFFFFFFFFF, b & 0xFFFFFFFFFFFFFFFF, c & 0xFFFFFFFFFFFFFFFF] for a, b, c in ts], dtype=np.uint64)
    out["tile_seed_in_signed"] = np.array([[b, c] for _, b, c in ts], dtype=np.int64)
    out["tile_seed_out"] = np.array([ns["_tile_seed"](*t) for t in ts], dtype=np.uint64)
    gnp = ns["gaussian_noise_patch"]
    # latent-stage call shape (world_pipeline.py:1090-1093): window straddling 4 noise tiles, negative coords
    out["patch_latent_m32_32"] = gnp(42, -32, 32, 64, 64, channels=5, tile_h=64, tile_w=64)
    out["patch_latent_aligned"] = gnp(42 + 5819, 128, -64, 64, 64, channels=5, tile_h=64, tile_w=64)
    # coarse-stage call shape (world_pipeline.py:928-938): multiples of 48 on a 64-tile grid, 6 channels
    out["patch_coarse_48_m96"] = gnp(43, 48, -96, 64, 64, channels=6, tile_h=64, tile_w=64)
    # ragged / small windows and non-square tiles
    out["patch_small"] = gnp(7, -3, 61, 7, 9, channels=2, tile_h=16, tile_w=32)
    out["patch_1x1"] = gnp(7, -1, -1, 1, 1, channels=1, tile_h=8, tile_w=8)
    # decoder-stage call shape (world_pipeline.py:1229-1232): (1,512,512) on a 512 grid at multiples of 384 — store a strided sample
    dec = gnp(42 + 5819, 384, -384, 512, 512, channels=1, tile_h=512, tile_w=512)
    out["patch_decoder_stride37"] = dec.ravel()[::37].copy()
    out["patch_decoder_sum"] = np.array([dec.astype(np.float64).sum(), np.abs(dec.astype(np.float64)).sum()])
    save("rng", **out)


def gen_geometry():
    from terrain_diffusion.training.evaluation import _linear_weight_window, _tile_starts
    pano = extract_functions(os.path.join(REF, "annotated_infinite_panorama.py"), {"linear_kernel", "build_timestep_ranges", "tiled_gaussian_noise"},
                             {"np": np, "torch": torch, "LATENT_CHANNELS": 4, "LATENT_TILE": 64})
    wp = extract_functions(os.path.join(REF, "terrain_diffusion/inference/world_pipeline.py"), {"linear_weight_window", "normalize_tensor"},
                           {"np": np, "torch": torch})
    out = {}
    for s in (4, 16, 64, 512):
        out[f"lww_{s}"] = _linear_weight_window(s, torch.device("cpu"), torch.float32)[0, 0].numpy()
        assert torch.equal(wp["linear_weight_window"](s, torch.device("cpu"), torch.float32), torch.from_numpy(out[f"lww_{s}"]))
    # SD-demo noise (annotated_infinite_panorama.py:57-73): windows straddling 256-column noise tiles, negative columns
    out["pano_noise_s1234_x0"] = pano["tiled_gaussian_noise"](1234, 0, 64)
    out["pano_noise_s1234_xm300"] = pano["tiled_gaussian_noise"](1234, -300, 64)
    out["pano_noise_s7_x224_w96_c5"] = pano["tiled_gaussian_noise"](7, 224, 96, channels=5)
    out["pano_kernel_64"] = pano["linear_kernel"](64, 64).numpy()
    out["pano_kernel_8x512"] = pano["linear_kernel"](8, 512).numpy()
    cases = [(64, 64, 32), (288, 64, 32), (1056, 64, 32), (100, 64, 32), (65, 64, 32), (10, 64, 32), (96, 64, 32), (130, 64, 48), (512, 512, 384), (33, 16, 8)]
    out["tile_starts_cases"] = np.array(cases, dtype=np.int64)
    starts = [_tile_starts(*c) for c in cases]
    out["tile_starts_len"] = np.array([len(s) for s in starts], dtype=np.int64)
    out["tile_starts_flat"] = np.array([v for s in starts for v in s], dtype=np.int64)
    # DDIM-style descending timesteps (50 of 1000, leading spacing as diffusers: 981, 961, ..., 1)
    ts = torch.arange(49, -1, -1) * 20 + 1
    ranges = pano["build_timestep_ranges"](ts, (400, 600, 750, 900))
    out["ddim_timesteps"] = ts.numpy()
    out["phase_len"] = np.array([len(r) for r in ranges], dtype=np.int64)
    out["phase_flat"] = torch.cat(ranges).numpy()
    ts4 = torch.tensor([751, 501, 251, 1])
    r4 = pano["build_timestep_ranges"](ts4, (400, 600, 750, 900))
    out["phase4_len"] = np.array([len(r) for r in r4], dtype=np.int64)
    out["phase4_flat"] = torch.cat(r4).numpy()
    save("geometry", **out)


def gen_schedule():
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from oracle import rng
    out = {}
    for n in (4, 12, 20, 32):
        sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
        sch.set_timesteps(n)
        out[f"sigmas_{n}"] = sch.sigmas.numpy()
        out[f"timesteps_{n}"] = sch.timesteps.numpy()
        # full step trace with a synthetic deterministic "model": F = tanh(0.3*x_in) - 0.2*cos(cn)
        x = torch.from_numpy(rng.standard_normal(900 + n, (2, 5, 8, 8))) * sch.sigmas[0]
        trace, orders = [], []
        for t, sigma in zip(sch.timesteps, sch.sigmas):
            xin = sch.precondition_inputs(x, sigma)
            cn = sch.trigflow_precondition_noise(sigma.view(-1))
            F_ = torch.tanh(0.3 * xin) - 0.2 * torch.cos(cn)
            before = sch.lower_order_nums
            x = sch.step(F_, t, x).prev_sample
            trace.append(x.numpy().copy())
        out[f"trace_{n}"] = np.stack(trace)
        out[f"trigflow_t_{n}"] = sch.trigflow_precondition_noise(sch.sigmas[:-1]).numpy()
    save("schedule", **out)


def gen_schedule3():
    """scheduler traces of the reference with solver_order = 3 (dpmsolver.py:563-615 third-order multistep update, order rule :688-715) and 1."""
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from oracle import rng
    out = {}
    for order in (3, 1):
        for n in (6, 20):
            sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, solver_order=order)
            sch.set_timesteps(n)
            x = torch.from_numpy(rng.standard_normal(950 + n, (2, 5, 8, 8))) * sch.sigmas[0]
            trace = []
            for t, sigma in zip(sch.timesteps, sch.sigmas):
                xin = sch.precondition_inputs(x, sigma)
                cn = sch.trigflow_precondition_noise(sigma.view(-1))
                F_ = torch.tanh(0.3 * xin) - 0.2 * torch.cos(cn)
                x = sch.step(F_, t, x).prev_sample
                trace.append(x.numpy().copy())
            out[f"trace_order{order}_{n}"] = np.stack(trace)
    save("schedule3", **out)


def _ref_model(cfg, sd):
    from terrain_diffusion.models.edm_unet import EDMUnet2D
    m = EDMUnet2D(**cfg)
    msd = m.state_dict()
    from oracle.unet import param_shapes
    ours = param_shapes(cfg)
    ref_names = {k for k in msd if not k.startswith("logvar_")}
    assert ref_names == set(ours) | {"noise_fourier.freqs"}, (ref_names ^ (set(ours) | {"noise_fourier.freqs"}))
    for k, shp in ours.items():
        assert tuple(msd[k].shape) == tuple(shp), (k, msd[k].shape, shp)
    assert torch.equal(msd["noise_fourier.freqs"], sd["noise_fourier.freqs"]), "positional freqs restatement differs"
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("logvar_") for k in missing), (missing, unexpected)
    return m.eval()


def gen_unet():
    from oracle import rng
    from oracle.unet import BASE_CONFIG, tiny_config, synth_state_dict
    out = {}
    # --- tiny model: every block variant (enc/dec, skip conv, down/up, mid attention), B=2, 16x16
    cfg = tiny_config(64, 1)
    sd = synth_state_dict(cfg, seed=77)
    m = _ref_model(cfg, sd)
    x = torch.from_numpy(rng.standard_normal(7, (2, 5, 16, 16)))
    t = torch.tensor([1.2, 0.3])
    cond = torch.from_numpy(rng.standard_normal(8, (2, 58)))
    taps = {}
    hooks = []
    for name, mod in list(m.enc.items()):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, n="enc." + name: taps.__setitem__(n, o.detach().numpy().copy())))
    for name, mod in list(m.dec.items()):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, n="dec." + name: taps.__setitem__(n, o.detach().numpy().copy())))
    with torch.no_grad():
        y = m(x, noise_labels=t, conditional_inputs=[cond])
        emb = m.compute_embeddings(t, [cond])
    out["tiny_out"] = y.numpy()
    out["tiny_emb"] = emb.numpy()
    for k, v in taps.items():
        out["tiny_tap:" + k] = v
    # --- tiny model with 2 layers per block and attention at an encoder level too (attn_resolutions hits res name 128)
    cfg2 = tiny_config(64, 2, attn_resolutions=[128])
    sd2 = synth_state_dict(cfg2, seed=78)
    m2 = _ref_model(cfg2, sd2)
    x2 = torch.from_numpy(rng.standard_normal(9, (1, 5, 32, 32)))
    with torch.no_grad():
        out["tiny2_out"] = m2(x2, noise_labels=torch.tensor([0.9]), conditional_inputs=[torch.from_numpy(rng.standard_normal(10, (1, 58)))]).numpy()
    # --- full-size base model (configs/diffusion_base/30m/diffusion_192-3.cfg:54-68), one forward, B=1, 64x64
    cfgb = dict(BASE_CONFIG)
    sdb = synth_state_dict(cfgb, seed=1234)
    mb = _ref_model(cfgb, sdb)
    xb = torch.from_numpy(rng.standard_normal(7, (1, 5, 64, 64)))
    cb = torch.from_numpy(rng.standard_normal(8, (1, 58)))
    with torch.no_grad():
        yb = mb(xb, noise_labels=torch.tensor([1.1]), conditional_inputs=[cb])
    out["base_out"] = yb.numpy()
    print("base forward rms", float(yb.pow(2).mean().sqrt()))
    save("unet", **out)


def gen_sampling():
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from terrain_diffusion.training.evaluation import sample_diffusion_base as sdb_mod
    from oracle import rng, tiling
    from oracle.unet import BASE_CONFIG, tiny_config, synth_state_dict
    out = {}
    cfg = tiny_config(64, 1)
    sd = synth_state_dict(cfg, seed=77)
    m = _ref_model(cfg, sd)
    sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)

    # _process_cond_img on NaN-free input
    cond_img = torch.from_numpy(rng.standard_normal(31, (2, 7, 4, 4)))
    means = torch.tensor([0.1, -0.2, 0.3, 0.0, 1.0, -1.0, 0.0])
    stds = torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0, 3.0, 1.0])
    out["cond58"] = sdb_mod._process_cond_img(cond_img.clone(), torch.tensor([[0.1, 0.2, 0.3, 0.4, 0.5]]).expand(2, -1), means, stds, torch.full((2,), 0.25)).numpy()
    out["cond58_zero"] = sdb_mod._process_cond_img(cond_img[:1].clone(), torch.zeros(1, 5), torch.zeros(7), torch.ones(7), torch.tensor(0.0)).numpy()

    # reference uses torch.randn for the initial field; substitute the portable absolute-coordinate field
    # (what the pipeline does via gaussian_noise_patch) so CPU and GPU see identical noise.
    real_randn = torch.randn

    def run_tiled(model, H, W, steps, tile, seed):
        def fake_randn(shape, generator=None, device=None, dtype=None):
            assert tuple(shape) == (1, 5, H, W)
            return tiling.initial_noise_field(seed, H, W, 5)
        torch.randn = fake_randn
        try:
            cond = tiling.synthetic_cond_grid(len(tiling.tile_starts(H, tile, tile // 2)), len(tiling.tile_starts(W, tile, tile // 2)))
            return sdb_mod.sample_base_diffusion(model, sch, (1, 5, H, W), cond, cond_means=torch.zeros(7), cond_stds=torch.ones(7),
                                                 noise_level=torch.tensor(0.0), histogram_raw=torch.zeros(1, 5), steps=steps, tile_size=tile)
        finally:
            torch.randn = real_randn

    # tiny model, 3x3 tiles of 16 (stride 8) on a 32x32 canvas, 6 steps (N<15 order rule) and 16 steps
    out["tiny_grid3_steps6"] = run_tiled(m, 32, 32, 6, 16, 42 + 5819).numpy()
    out["tiny_grid3_steps16"] = run_tiled(m, 32, 32, 16, 16, 42 + 5819).numpy()
    # ragged canvas: 40x24 -> starts [0,8,16,24] x [0,8] (last start clamped)
    out["tiny_ragged_40x24_steps5"] = run_tiled(m, 40, 24, 5, 16, 99).numpy()

    # consistency sampler, 2 phases, explicit noise (reference accepts `noise=`)
    noise = [tiling.initial_noise_field(42 + 5819 + k, 32, 32, 5) for k in range(2)]
    cond = tiling.synthetic_cond_grid(3, 3)
    out["tiny_consistency_2phase"] = sdb_mod.sample_base_consistency(
        m, sch, (1, 5, 32, 32), cond, cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0),
        histogram_raw=torch.zeros(1, 5), intermediate_t=float(np.arctan(0.35 / 0.5)), tile_size=16, noise=noise).detach().numpy()

    # BASELINE config 2: full-size base model, single 64x64 tile, 20 steps
    cfgb = dict(BASE_CONFIG)
    mb = _ref_model(cfgb, synth_state_dict(cfgb, seed=1234))
    import time
    t0 = time.time()
    out["base_tile_steps20"] = run_tiled(mb, 64, 64, 20, 64, 42 + 5819).numpy()
    print(f"reference base tile x20 steps: {time.time() - t0:.1f}s on {torch.get_num_threads()} threads")
    save("sampling", **out)


def gen_guided():
    """autoguidance (sample_diffusion_base.py:105-110,155-160): main + guide model (different widths, like 192 vs 128 in the 30m configs)."""
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from terrain_diffusion.training.evaluation import sample_diffusion_base as sdb_mod
    from oracle import tiling
    from oracle.unet import tiny_config, synth_state_dict
    cfg_m, cfg_g = tiny_config(128, 1), tiny_config(64, 1)
    m = _ref_model(cfg_m, synth_state_dict(cfg_m, seed=81))
    g = _ref_model(cfg_g, synth_state_dict(cfg_g, seed=82))
    sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
    real_randn = torch.randn
    out = {}
    for key, (H, W, steps, tile, scale) in {"guided_grid3_steps6_s2": (32, 32, 6, 16, 2.0), "guided_ragged_24x40_steps5_s1p5": (24, 40, 5, 16, 1.5)}.items():
        torch.randn = lambda shape, generator=None, device=None, dtype=None: tiling.initial_noise_field(42 + 5819, H, W, 5)
        try:
            cond = tiling.synthetic_cond_grid(len(tiling.tile_starts(H, tile, tile // 2)), len(tiling.tile_starts(W, tile, tile // 2)))
            out[key] = sdb_mod.sample_base_diffusion(m, sch, (1, 5, H, W), cond, cond_means=torch.zeros(7), cond_stds=torch.ones(7), noise_level=torch.tensor(0.0),
                                                     histogram_raw=torch.zeros(1, 5), steps=steps, tile_size=tile, guide_model=g, guidance_scale=scale).numpy()
        finally:
            torch.randn = real_randn
    save("guided", **out)


def gen_compose():
    """postprocessing.local_baseline_temperature_torch and the Laplacian pyramid's pure helper pad_linear_extrapolation, run from the
    reference's own source (AST-extracted: the modules import matplotlib / torchvision, which are absent here)."""
    import torch.nn.functional as F
    from oracle import rng
    ns = {"torch": torch, "F": F, "np": np}
    extract_functions(os.path.join(REF, "terrain_diffusion", "inference", "postprocessing.py"), ["local_baseline_temperature_torch"], ns)
    extract_functions(os.path.join(REF, "terrain_diffusion", "data", "laplacian_encoder.py"), ["pad_linear_extrapolation"], ns)
    T = torch.from_numpy(rng.standard_normal(501, (40, 52))) * 8 + 12
    e = torch.from_numpy(rng.standard_normal(502, (40, 52))) * 600 + 150
    out = {"lbt_T": T.numpy(), "lbt_e": e.numpy()}
    for win, thr in ((15, 0.02), (3, 0.3)):
        ts, beta = ns["local_baseline_temperature_torch"](T, e, win=win, fallback_threshold=thr)
        out[f"lbt_sea_w{win}"], out[f"lbt_beta_w{win}"] = ts.numpy()[0], beta.numpy()[0]
    x = torch.from_numpy(rng.standard_normal(503, (2, 7, 9)))
    out["ple_in"], out["ple_out"] = x.numpy(), ns["pad_linear_extrapolation"](x).numpy()
    save("compose", **out)


def gen_stages():
    """EDMUnet2D in its other two pipeline roles: coarse model (5 'float' conditional inputs through MPFourier,
    11 -> 6 channels) and decoder (no conditional inputs, 5 -> 1 channels), full-size configs at 64x64 input."""
    from oracle import rng
    from oracle.unet import COARSE_CONFIG, DECODER_CONFIG, synth_state_dict
    out = {}
    cfgc = dict(COARSE_CONFIG)
    mc = _ref_model(cfgc, synth_state_dict(cfgc, seed=4321))
    x = torch.from_numpy(rng.standard_normal(41, (2, 11, 64, 64)))
    conds = [torch.from_numpy(rng.standard_normal(50 + i, (2,))) for i in range(5)]
    with torch.no_grad():
        out["coarse_out"] = mc(x, noise_labels=torch.tensor([1.3, 0.4]), conditional_inputs=conds).numpy()
        out["coarse_emb"] = mc.compute_embeddings(torch.tensor([1.3, 0.4]), conds).numpy()
    cfgd = dict(DECODER_CONFIG)
    md = _ref_model(cfgd, synth_state_dict(cfgd, seed=2468))
    xd = torch.from_numpy(rng.standard_normal(43, (1, 5, 64, 64)))
    with torch.no_grad():
        out["decoder_out"] = md(xd, noise_labels=torch.tensor([1.5]), conditional_inputs=[]).numpy()
    print("coarse rms", float(np.sqrt((out["coarse_out"] ** 2).mean())), "decoder rms", float(np.sqrt((out["decoder_out"] ** 2).mean())))
    save("stages", **out)


def extract_methods(path, cls, names, namespace):
    """exec the named methods of class `cls` of a reference file as plain functions (first argument = self)."""
    tree = ast.parse(open(path).read())
    body = [m for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls for m in n.body if isinstance(m, ast.FunctionDef) and m.name in names]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), namespace)
    return namespace


def gen_stage_glue():
    """Per-window glue of the coarse and decoder stages: the reference's own WorldPipeline._coarse_inference /
    _pool_coarse_conditioning / _decoder_inference bodies (world_pipeline.py:909-959, 997-1015, 1209-1242) run against a stub `self`
    carrying reference EDMUnet2D models with the synthetic weights (coarse: the full-size coarse config; decoder: the full-size
    decoder config on a 64-pixel tile, stride 48) and the deterministic synthetic map of oracle/stages.py."""
    from terrain_diffusion.inference import portable_rng as pr
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from oracle.unet import COARSE_CONFIG, DECODER_CONFIG, synth_state_dict
    from oracle import stages
    wp = os.path.join(REF, "terrain_diffusion/inference/world_pipeline.py")
    ns = extract_functions(wp, {"_tile_seed", "gaussian_noise_patch", "linear_weight_window"},
                           {"np": np, "torch": torch, "fill_standard_normal": pr.fill_standard_normal, "MOCK": False})
    extract_methods(wp, "WorldPipeline", {"_coarse_inference", "_pool_channel", "_pool_coarse_conditioning", "_decoder_inference"}, ns)
    means = [0.3, -0.2, 0.1, 0.0, 0.4, -0.1]; stds = [1.5, 0.8, 1.2, 0.9, 1.1, 0.7]; snr = [0.5, 0.4, 0.6, 0.3, 0.8]
    mc = _ref_model(dict(COARSE_CONFIG), synth_state_dict(COARSE_CONFIG, seed=4321))
    md = _ref_model(dict(DECODER_CONFIG), synth_state_dict(DECODER_CONFIG, seed=2468))
    self = types.SimpleNamespace(kwargs=dict(coarse_means=means, coarse_stds=stds, elev_coarse_pool_mode="max", p5_coarse_pool_mode="min"),
                                 _dtype=None, device="cpu", seed=1234, log_mode="quiet", coarse_model=mc, decoder_model=md, latent_compression=8,
                                 _conditioning_model_input=stages.synthetic_coarse_map)
    self._pool_channel = lambda *a: ns["_pool_channel"](self, *a)
    self._pool_coarse_conditioning = lambda *a: ns["_pool_coarse_conditioning"](self, *a)
    out = {"coarse_means": np.array(means, np.float32), "coarse_stds": np.array(stds, np.float32), "cond_snr": np.array(snr, np.float32)}
    sched = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80, sigma_data=0.5)
    t_cond = torch.atan(torch.tensor(snr))
    cond_inputs = [v.detach().view(-1) for v in torch.log(torch.tan(t_cond) / 8.0)]
    with torch.no_grad():
        for name, ctx, pool in [("coarse_ctx_0_1_m2_pool1", (0, 1, -2), 1), ("coarse_ctx_0_m1_0_pool2", (0, -1, 0), 2)]:
            w = ns["linear_weight_window"](64 // pool, "cpu", torch.float32)
            out[name] = ns["_coarse_inference"](self, ctx, sched, w, t_cond, cond_inputs, pool_size=pool).numpy()
        pin = torch.from_numpy(pr.standard_normal(77, (6, 16, 16)))
        out["pool_in"] = pin.numpy()
        out["pool4_max_min"] = ns["_pool_coarse_conditioning"](self, pin, 4).numpy()
        # decoder: tile 64, stride 48, latent window (6, 8, 8) = packed un-normalised sums (weight channel last)
        T, S = 64, 48
        lat = torch.from_numpy(pr.standard_normal(78, (6, T // 8, T // 8)))
        lat[-1] = lat[-1].abs() + 0.5
        out["decoder_latents_in"] = lat.numpy()
        t_list = [torch.atan(sched.sigmas[0] / sched.config.sigma_data)] if hasattr(sched, "sigmas") else None
        sched.set_timesteps(20)
        t_list = [torch.atan(sched.sigmas[0] / sched.config.sigma_data)]
        wd = ns["linear_weight_window"](T, "cpu", torch.float32)
        out["decoder_ctx_0_2_m1"] = ns["_decoder_inference"](self, (0, 2, -1), lat.clone(), sched, wd, t_list, T, S).numpy()
        t2 = t_list + [torch.arctan(torch.tensor(0.065) / 0.5)]   # the commented-out second phase of wp.py:1253: exercises i > 0
        out["decoder_ctx_0_2_m1_two_phases"] = ns["_decoder_inference"](self, (0, 2, -1), lat.clone(), sched, wd, t2, T, S).numpy()
    save("stage_glue", **out)


def gen_latent_glue():
    """WorldPipeline._process_latent_conditioning (incl. its NaN handling and portable-RNG fill) and _latent_inference
    (world_pipeline.py:1018-1131) run from the reference's own method bodies against a stub `self` with a tiny reference base model."""
    from terrain_diffusion.inference import portable_rng as pr
    from terrain_diffusion.models.mp_layers import mp_concat
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from oracle.unet import tiny_config, synth_state_dict
    wp = os.path.join(REF, "terrain_diffusion/inference/world_pipeline.py")
    ns = extract_functions(wp, {"_tile_seed", "gaussian_noise_patch", "linear_weight_window"},
                           {"np": np, "torch": torch, "fill_standard_normal": pr.fill_standard_normal, "standard_normal": pr.standard_normal,
                            "mp_concat": mp_concat, "MOCK": False})
    extract_methods(wp, "WorldPipeline", {"_process_latent_conditioning", "_latent_inference"}, ns)
    cfg = tiny_config(64, 1)
    mb = _ref_model(cfg, synth_state_dict(cfg, seed=77))
    self = types.SimpleNamespace(_dtype=None, device="cpu", seed=1234, log_mode="quiet", base_model=mb, torch_compile=False)
    self._process_latent_conditioning = lambda *a, **k: ns["_process_latent_conditioning"](self, *a, **k)
    means = torch.tensor([14.99, 11.65, 15.87, 619.26, 833.12, 69.40, 0.66]); stds = torch.tensor([21.72, 21.78, 10.40, 452.29, 738.09, 34.59, 0.47])
    hist = torch.tensor([[0.1, 0.3, 0.2, 0.25, 0.15]])
    out = {"cond_means": means.numpy(), "cond_stds": stds.numpy(), "histogram_raw": hist.numpy()}
    # ---- conditioning vectors
    c1 = torch.from_numpy(pr.standard_normal(90, (1, 7, 4, 4))) * stds.view(1, -1, 1, 1) + means.view(1, -1, 1, 1)
    c1[0, 0, 1, 2] = float("nan"); c1[0, 3, 1, 1] = float("nan"); c1[0, 6, 0, 0] = float("nan")
    out["plc_in_n1"] = c1.numpy().copy()
    out["plc_out_n1"] = ns["_process_latent_conditioning"](self, c1.clone(), hist, means, stds, torch.tensor(0.0), seed_offset=3 * 65536 - 2).numpy()
    c3 = torch.from_numpy(pr.standard_normal(91, (3, 7, 4, 4))) * stds.view(1, -1, 1, 1) + means.view(1, -1, 1, 1)
    c3[0, 2, 1, 1] = float("nan"); c3[1, 0, 0, 0] = float("nan"); c3[2, 2, 1, 2] = float("nan"); c3[2, 5, 2, 2] = float("nan"); c3[2, 4, 0, 0] = float("nan")
    out["plc_in_n3"] = c3.numpy().copy()
    out["plc_out_n3"] = ns["_process_latent_conditioning"](self, c3.clone(), hist.expand(3, -1), means, stds, torch.zeros(3), seed_offset=7).numpy()
    # ---- latent windows: phase 0 (no previous sample) and a later phase (packed previous sums), two window contexts
    sched = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80, sigma_data=0.5)
    sched.set_timesteps(20)
    w = ns["linear_weight_window"](64, "cpu", torch.float32)
    ctxs = [(0, 2, -3), (0, -1, 0)]
    conds = []
    for k in range(2):   # packed coarse windows (7,4,4): 6 value channels * weight, weight
        v = torch.from_numpy(pr.standard_normal(92 + k, (6, 4, 4))) * stds[:6].view(-1, 1, 1) + means[:6].view(-1, 1, 1)
        ww = torch.from_numpy(pr.standard_normal(95 + k, (1, 4, 4))).abs() + 0.5
        conds.append(torch.cat([v * ww, ww], dim=0))
    conds[1][2, 1, 1] = float("nan")
    out["latent_cond_windows"] = torch.stack(conds).numpy()
    t0 = torch.atan(sched.sigmas[0] / sched.config.sigma_data)
    with torch.no_grad():
        o0 = ns["_latent_inference"](self, ctxs, None, [c.clone() for c in conds], t0, sched, w, hist, means, stds, seed_offset=5819)
        out["latent_phase0"] = torch.stack(o0).numpy()
        t1 = torch.arctan(torch.tensor(0.35) / 0.5)
        o1 = ns["_latent_inference"](self, ctxs, [o.clone() for o in o0], [c.clone() for c in conds], t1, sched, w, hist, means, stds, seed_offset=5820)
        out["latent_phase1_from_phase0_windows"] = torch.stack(o1).numpy()
    save("latent_glue", **out)


def gen_bounded_twins():
    """The bounded decoder / coarse samplers (training/evaluation/sample_diffusion_decoder.py:44-211, sample_coarse.py:29-125) run from the
    reference's own modules on full-size decoder / coarse architectures at small canvases.  The reference draws the coarse sampler's noises with
    torch.randn / torch.randn_like; they are replaced by portable-RNG tensors (recorded by seed) so that any implementation can replay them."""
    from terrain_diffusion.scheduler.dpmsolver import EDMDPMSolverMultistepScheduler
    from terrain_diffusion.training.evaluation import sample_diffusion_decoder as sdd, sample_coarse as sc
    from oracle import rng
    from oracle.unet import COARSE_CONFIG, DECODER_CONFIG, synth_state_dict
    out = {}
    md = _ref_model(dict(DECODER_CONFIG), synth_state_dict(DECODER_CONFIG, seed=2468))
    sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
    # decoder: batch of 2, canvas 40 x 56, tiles of 32 with stride 24 (ragged last tile), 4 conditioning channels
    noise = torch.from_numpy(rng.standard_normal(901, (2, 1, 40, 56)))
    cond = torch.from_numpy(rng.standard_normal(902, (2, 4, 40, 56)))
    # (finding: the reference's decoder DIFFUSION sampler sets the timesteps once, outside its tile loop, so the scheduler's step index runs off
    # the sigma table on the second tile -- IndexError at dpmsolver.py:512.  It only works for one tile; the goldens are single-tile.)
    sq = torch.from_numpy(rng.standard_normal(908, (2, 1, 40, 40)))
    csq = torch.from_numpy(rng.standard_normal(909, (2, 4, 40, 40)))
    out["dec_diffusion_b2_40x40_steps6"] = sdd.sample_decoder_diffusion_tiled(md, sch, csq, sq * 80.0, num_steps=6).numpy()
    # conditioning image at half resolution (nearest upsampling inside the sampler), one tile = whole canvas
    cond_lo = torch.from_numpy(rng.standard_normal(903, (2, 4, 16, 16)))
    noise32 = torch.from_numpy(rng.standard_normal(904, (2, 1, 32, 32)))
    out["dec_diffusion_b2_32x32_condlo_steps4"] = sdd.sample_decoder_diffusion_tiled(md, sch, cond_lo, noise32 * 80.0, num_steps=4).numpy()
    sch.set_timesteps(20)
    out["dec_consistency_b2_40x56_t32_s24_1step"] = sdd.sample_decoder_consistency_tiled(md, sch, cond, noise, 32, 24).numpy()
    out["dec_consistency_b2_40x56_t32_s24_3step"] = sdd.sample_decoder_consistency_tiled(md, sch, cond, noise, 32, 24, intermediate_t=[float(np.arctan(0.35 / 0.5)), 0.2]).numpy()
    # coarse: 5 conditioning channels at 64 x 64, ONE tile (the same step-index bug as above makes a second tile fail), 5 steps; noises pinned
    mc = _ref_model(dict(COARSE_CONFIG), synth_state_dict(COARSE_CONFIG, seed=4321))
    cimg = torch.from_numpy(rng.standard_normal(905, (1, 5, 64, 64)))
    snr = torch.tensor([[0.5, 0.4, 0.6, 0.3, 0.8]])
    real_randn, real_randn_like = torch.randn, torch.randn_like
    calls = []

    def fake_randn(*shape, generator=None, device=None, dtype=None, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        calls.append(shape)
        return torch.from_numpy(rng.standard_normal(907 + len(calls) - 1, shape))
    torch.randn = fake_randn
    torch.randn_like = lambda t, **kw: torch.from_numpy(rng.standard_normal(906, tuple(t.shape)))
    try:
        out["coarse_64x64_steps5"] = sc.sample_coarse_tiled(mc, sch, cimg, snr, steps=5).numpy()
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    assert calls == [(1, 6, 64, 64)], calls
    save("bounded_twins", **out)


ALL = dict(schedule3=gen_schedule3, bounded_twins=gen_bounded_twins, compose=gen_compose, guided=gen_guided, latent_glue=gen_latent_glue, stage_glue=gen_stage_glue, stages=gen_stages, rng=gen_rng, geometry=gen_geometry, schedule=gen_schedule, unet=gen_unet, sampling=gen_sampling)

if __name__ == "__main__":
    assert os.path.isdir(REF), "golden generation needs the reference checkout"
    install_shim()
    torch.manual_seed(0)
    for k in (sys.argv[1:] or list(ALL)):
        ALL[k]()
