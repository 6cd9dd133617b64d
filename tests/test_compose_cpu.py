"""Output composition, CPU side: the oracle against what the reference's own helpers produce, and the product's tap tables
(terrain_diffusion_amd/composition.py -- what the HIP gather kernels apply) against torch's F.interpolate / conv2d, which is what
torchvision's resize / gaussian_blur call.  No engine call here."""
import numpy as np
import torch
import torch.nn.functional as F

from conftest import rel_rms


def _apply(x, taps_y, taps_x):
    (iy, wy), (ix, wx) = taps_y, taps_x
    tmp = (x[:, ix] * wx[None]).sum(-1)             # rows first, like the kernels
    return (tmp[iy] * wy[..., None]).sum(1)


def test_oracle_helpers_match_reference_golden(golden):
    from oracle import compose
    g = golden("compose")
    T, e = torch.from_numpy(g["lbt_T"]), torch.from_numpy(g["lbt_e"])
    for win, thr in ((15, 0.02), (3, 0.3)):
        sea, beta = compose.local_baseline_temperature(T, e, win=win, fallback_threshold=thr)
        assert np.array_equal(sea.numpy(), g[f"lbt_sea_w{win}"]) and np.array_equal(beta.numpy(), g[f"lbt_beta_w{win}"])
    assert np.array_equal(compose.pad_linear_extrapolation(torch.from_numpy(g["ple_in"])).numpy(), g["ple_out"])


def test_product_local_baseline_matches_reference_golden(golden):
    from terrain_diffusion_amd.composition import local_baseline_temperature
    g = golden("compose")
    T, e = torch.from_numpy(g["lbt_T"]), torch.from_numpy(g["lbt_e"])
    for win, thr in ((15, 0.02), (3, 0.3)):
        sea, beta = local_baseline_temperature(T, e, win=win, fallback_threshold=thr)
        assert np.array_equal(sea.numpy(), g[f"lbt_sea_w{win}"]) and np.array_equal(beta.numpy(), g[f"lbt_beta_w{win}"])


def test_tap_tables_reproduce_torch_operators():
    from terrain_diffusion_amd import composition as cp
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 53)).astype(np.float32)
    xt = torch.from_numpy(x)[None, None]
    for (ho, wo) in ((296, 424), (74, 53), (40, 60)):        # up x8, up x2 in one axis, mixed
        ref = F.interpolate(xt, size=(ho, wo), mode="bilinear", align_corners=False)[0, 0].numpy()
        got = _apply(x, cp.bilinear_taps(37, ho), cp.bilinear_taps(53, wo))
        assert np.abs(got - ref).max() < 2e-5, (ho, wo)   # non-dyadic scales: fp32 source coordinates differ from torch's in the last bits
    big = rng.standard_normal((320, 416)).astype(np.float32)
    for (ho, wo) in ((40, 52), (7, 9), (100, 416)):          # anti-aliased shrink x8, x46, and one axis only
        ref = F.interpolate(torch.from_numpy(big)[None, None], size=(ho, wo), mode="bilinear", align_corners=False, antialias=True)[0, 0].numpy()
        ty = cp.bilinear_aa_taps(320, ho)
        tx = cp.bilinear_aa_taps(416, wo) if wo < 416 else cp.bilinear_taps(416, wo)
        assert rel_rms(_apply(big, ty, tx), ref) < 1e-5, (ho, wo)
    from oracle import compose
    for sigma in (5, 2):
        ref = compose.tf_gaussian_blur(torch.from_numpy(x), sigma).numpy()
        got = _apply(x, cp.gaussian_taps(37, sigma), cp.gaussian_taps(53, sigma))
        assert np.abs(got - ref).max() < 2e-6, sigma


def test_laplacian_round_trip_and_denoise_properties():
    """size-independent properties of the pyramid (laplacian_encoder.py): decode(encode(x)) == x; denoise leaves the residual untouched and
    returns a low band of the same shape; a constant image has zero residual."""
    from oracle import compose
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((64, 64)).astype(np.float32)).cumsum(0).cumsum(1) * 0.01
    res, low = compose.laplacian_encode(x, 8, 5)
    assert low.shape == (8, 8) and torch.allclose(compose.laplacian_decode(res, low), x, atol=1e-5)
    r2, low2 = compose.laplacian_denoise(res, low, 5)
    assert r2 is res and low2.shape == low.shape
    c_res, c_low = compose.laplacian_encode(torch.full((64, 64), 3.5), 8, 5)
    assert c_res.abs().max() < 1e-5 and torch.allclose(c_low, torch.full((8, 8), 3.5), atol=1e-5)


def test_latent_conditioning_of_a_batch_of_windows_equals_the_per_window_loop():
    """round 4: the latent stage evaluates the conditioning of all windows of a batch at once (process_latent_conditioning_windows).  The
    reference calls _process_latent_conditioning once per window with a batch of ONE (world_pipeline.py:1080-1088), where its batch-dimension
    NaN fill covers the whole window; the vectorised form must reproduce that loop bit for bit, NaNs and infinities included."""
    import numpy as np
    import torch
    from terrain_diffusion_amd.sampling import process_latent_conditioning, process_latent_conditioning_windows
    from terrain_diffusion_amd.pipeline import LATENT_COND_MEAN, LATENT_COND_STD
    g = torch.Generator().manual_seed(5)
    x = torch.randn(9, 7, 4, 4, generator=g) * torch.tensor(LATENT_COND_STD).view(1, -1, 1, 1) + torch.tensor(LATENT_COND_MEAN).view(1, -1, 1, 1)
    x[1, 0, 2, 3] = float("nan"); x[2, 3, 1, 1] = float("nan"); x[4, 2:6] = float("nan"); x[5, 1, 0, 0] = float("inf"); x[7] = float("nan")
    hist = torch.randn(1, 5, generator=g)
    loop = torch.cat([process_latent_conditioning(x[i:i + 1].clone(), hist, LATENT_COND_MEAN, LATENT_COND_STD, 0.0, seed=77, seed_offset=i * 65536 + 3) for i in range(9)])
    vec = process_latent_conditioning_windows(x.clone(), hist, LATENT_COND_MEAN, LATENT_COND_STD, 0.0)
    assert vec.shape == loop.shape == (9, 58)
    assert torch.equal(vec, loop), float((vec - loop).abs().max())
