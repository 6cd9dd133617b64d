"""WorldPipeline facade, synthetic map source, persistence and wire formats on the GPU (SURVEY.md 8b face 2, 8f-2/3/4)."""
import numpy as np
import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def td():
    import terrain_diffusion_amd as t
    assert torch.cuda.is_available()
    return t


@pytest.fixture(scope="module")
def models(td):
    from oracle.unet import COARSE_CONFIG, DECODER_CONFIG, synth_state_dict, tiny_config
    bcfg = tiny_config(64, 1)
    ms = [td.EDMUnet2D(**c, dtype="fp32").load_state_dict(synth_state_dict(c, seed=s)) for c, s in ((COARSE_CONFIG, 1), (bcfg, 2), (DECODER_CONFIG, 3))]
    yield ms
    for m in ms:
        m.close()


def _world(td, models, **kw):
    return td.WorldPipeline.from_models(*models, seed=4242, decoder_tile_size=64, decoder_tile_stride=48, latents_batch_size=16, **kw).bind()


def test_perlin_map_kernel_vs_numpy_twin(td):
    from oracle import synthmap
    from terrain_diffusion_amd.engine import get_engine
    from terrain_diffusion_amd.synthetic_map import make_synthetic_map_factory
    f = make_synthetic_map_factory(get_engine("cuda"), frequency_mult=[1.5, 3, 3, 3, 3], seed=77)
    for ch in (0, 1, 4):
        fr, o, l, g = f.params[ch]
        src, dst = f.stats["noise_quantile_tables"][ch].astype(np.float32), f.stats["data_quantile_tables"][ch].astype(np.float32)
        got = f._channel(ch, -37, 1200, 50, 70).cpu().numpy()
        ref = synthmap.fbm_map(50, 70, -37, 1200, f.seeds[ch], fr, o, l, g, src, dst)
        scale = np.abs(dst).max()
        assert np.abs(got - ref).max() < 2e-4 * scale, ch      # fast sin/cos intrinsics on the device: ~1e-6 in the noise, stretched by the transfer
    raw = f.sample_raw(10, 20, 42, 84)
    fin = f.finalize(raw).cpu().numpy()
    s = f.stats
    assert np.allclose(fin, synthmap.finalize(raw.cpu().numpy(), s["a_temp_std"], s["b_temp_std"], s["temp_std_p1"], s["temp_std_p99"]), rtol=1e-5, atol=1e-3)
    full = f(10, 20, 42, 84)
    assert full.shape == (5, 32, 64) and torch.isfinite(full).all()
    # the field is a function of absolute coordinates: overlapping requests agree
    assert torch.equal(f.sample_raw(0, 0, 16, 16)[:, 4:, 4:], f.sample_raw(4, 4, 16, 16))


def test_world_pipeline_get_matches_oracle_composition(td, models):
    """get() = _compute_elev + _compute_climate over the pipeline's own stage tensors: compared with the CPU oracle's composition of host
    copies of those tensors (oracle/compose.py), for a box that is not aligned to anything; deterministic; seed-dependent."""
    from oracle import compose
    w = _world(td, models)
    box = (-21, 13, 75, 141)
    out = w.get(*box)
    elev, clim = out["elev"], out["climate"]
    assert elev.is_cuda and elev.shape == (96, 128) and clim.shape == (5, 96, 128) and torch.isfinite(elev).all() and torch.isfinite(clim).all()

    class Host:
        def __init__(self, t):
            self.t = t

        def __getitem__(self, idx):
            return torch.as_tensor(self.t[idx]).cpu()
    ref_e = compose.compute_elev(Host(w.residual), Host(w.latents), *box, 8, w.kwargs["residual_mean"], w.kwargs["residual_std"])
    ref_c = compose.compute_climate(Host(w.coarse), *box, ref_e, 8)
    assert rel_rms(elev.cpu().numpy(), ref_e.numpy()) < 1e-5 and rel_rms(clim.cpu().numpy(), ref_c.numpy()) < 1e-5
    again = w.get(*box)
    assert torch.equal(again["elev"], elev) and torch.equal(again["climate"], clim)
    n0 = w.residual.windows_computed
    w.empty_cache()
    w.engine.set_option("batch_invariant", 1)
    try:
        a = w.get(*box)["elev"]
        w.get(-200, -180, -150, -100, with_climate=False)   # another region first: the windows of `box` are then recomputed in other batches
        w.empty_cache()
        w.get(-21, 13, 27, 77, with_climate=False)          # (a sub-box is NOT the crop of the box: the Laplacian re-encode sees another extent,
        b = w.get(*box)["elev"]                             #  as in the reference; the same box after a flush must give the same bits)
        assert torch.equal(b, a)
    finally:
        w.engine.set_option("batch_invariant", 0)
    assert w.residual.windows_computed > n0
    assert w.change_seed(4243) and not w.change_seed(4243)
    assert not torch.equal(w.get(*box, with_climate=False)["elev"], elev)
    w.close()


def test_world_pipeline_persistent_store_and_wire_format(td, models, tmp_path):
    from terrain_diffusion_amd import wire
    path = str(tmp_path / "world")
    w = _world_indirect(td, models, path)
    box = (0, 0, 64, 96)
    out = w.get(*box)
    assert w.tile_store.params["seed"] == 4242
    payload, hdr = wire.binary_payload(out["elev"], out["climate"])
    assert len(payload) == 64 * 96 * 2 + 64 * 96 * 16 and hdr == {"X-Height": "64", "X-Width": "96"}
    e16, c4 = wire.parse_payload(payload, 64, 96)
    assert np.array_equal(e16, np.clip(np.floor(out["elev"].cpu().numpy()), -32768, 32767).astype(np.int16))
    assert np.array_equal(c4, out["climate"][:4].cpu().numpy())
    w.close()
    # reopen: every window comes from disk, nothing is recomputed, bytes identical
    calls = []
    w2 = _world_indirect(td, models, path)
    f0 = w2.residual.f
    w2.residual.f = lambda *a, **k: (calls.append(1), f0(*a, **k))[1]
    out2 = w2.get(*box)
    assert not calls and torch.equal(out2["elev"], out["elev"]) and torch.equal(out2["climate"], out["climate"])
    w2.close()


def _world_indirect(td, models, path):
    return td.WorldPipeline.from_models(*models, seed=4242, decoder_tile_size=64, decoder_tile_stride=48, latents_batch_size=16,
                                        caching_strategy="indirect").bind(path)


def test_save_pretrained_round_trip(td, models, tmp_path):
    """save_pretrained writes the reference's layout (world_pipeline.py:500-518: config.json + coarse_model/ base_model/ decoder_model/, each a
    config.json + one safetensors state dict under the reference's parameter names); from_pretrained on that directory rebuilds a pipeline
    that serves the same bytes."""
    import json
    import os
    from safetensors.torch import load_file
    w = _world(td, models)
    box = (5, -9, 69, 71)
    ref = w.get(*box)
    root = str(tmp_path / "pipe")
    w.save_pretrained(root)
    assert sorted(os.listdir(root)) == ["base_model", "coarse_model", "config.json", "decoder_model"]
    cfg = json.load(open(os.path.join(root, "config.json")))
    assert cfg["_class_name"] == "WorldPipeline" and cfg["decoder_tile_size"] == 64 and "seed" not in cfg
    sd = load_file(os.path.join(root, "base_model", "diffusion_pytorch_model.safetensors"))
    assert set(models[1].expected_parameters()) <= set(sd) and all(v.dtype == torch.float32 for v in sd.values())
    w2 = td.WorldPipeline.from_pretrained(root, seed=4242, latents_batch_size=16, dtype=None).bind()
    got = w2.get(*box)
    assert torch.equal(got["elev"], ref["elev"]) and torch.equal(got["climate"], ref["climate"])
    w2.close()
    for m in (w2.coarse_model, w2.base_model, w2.decoder_model):
        m.close()
    w.close()
