"""Synthetic coarse conditioning source -- the engine-side counterpart of make_synthetic_map_factory
(terrain_diffusion/inference/synthetic_map.py:182-271): five FBm channels (elevation, temperature, temperature std, precipitation,
precipitation CV) with the reference's frequencies / octaves / per-channel seeds, each pushed through a 64-knot quantile transfer
(perlin_transform.py:41-45), then the reference's `finalize` arithmetic (lapse rate, cold stretch, temperature-std baseline, precipitation-CV
damping) and sign(x) sqrt(|x|) on elevation.

What is NOT the reference's: (i) the noise values -- pyfastnoiselite is absent, the HIP kernel has FastNoiseLite's Perlin/FBm structure with its
own gradient set; (ii) the quantile tables and the four regression constants -- the reference derives them from ETOPO / WorldClim rasters
(synthetic_map.py:45-136, data/global/*.tif, not shipped and not downloadable here), so DEFAULT_STATS below are stand-ins with plausible
marginals (documented in DESIGN.md).  A stats JSON in the reference's cache format (synthetic_map.py:159-180) can be passed to use real ones.
PARITY UNPINNED for both; the finalize arithmetic and the transfer are restated from the reference and tested against numpy.
"""
import ctypes as C
import json

import numpy as np
import torch

from ._lib import lib, check
from .engine import ptr

MAP_CONFIGS = [(0.05, 4, 2.0, 0.5), (0.05, 2, 2.0, 0.5), (0.05, 4, 2.0, 0.5), (0.05, 4, 2.0, 0.5), (0.05, 4, 2.0, 0.5)]   # synthetic_map.py:222-228


def _default_targets():
    q = np.linspace(1e-4, 1 - 1e-4, 64)
    from scipy.stats import norm
    z = norm.ppf(q)
    elev = np.where(z < 0.35, -4200 + 1500 * z, 40 + 900 * np.clip(z - 0.35, 0, None) ** 1.8)          # ~64 % ocean, long mountain tail  (metres)
    temp = 14 + 11 * z - 1.5 * z ** 2                                                # sea-level temperature, deg C
    tstd = 250 * z                                                                   # de-trended seasonality (deg C x 100)
    precip = np.clip(900 * np.exp(0.75 * z) - 150, 0, None)                          # mm / year
    pcv = np.clip(65 + 28 * z, 5, None)
    return [np.maximum.accumulate(a + np.arange(64) * 1e-6) for a in (elev, temp, tstd, precip, pcv)]


def default_stats(noise_quantiles):
    return dict(a_temp_std=-12.0, b_temp_std=700.0, temp_std_p1=-450.0, temp_std_p99=900.0,
                noise_quantile_tables=[np.asarray(nq, np.float64) for nq in noise_quantiles], data_quantile_tables=_default_targets())


class SyntheticMapFactory:
    """callable(j1, i1, j2, i2) -> (5, i2-i1, j2-j1) float32 device tensor, like sample_full_synthetic_map (synthetic_map.py:262-265);
    .sample_raw / .finalize as on the reference's factory object (:267-268)."""

    def __init__(self, engine, seed, frequency_mult=(1.0, 1.0, 1.0, 1.0, 1.0), drop_water_pct=0.0, stats_json=None):
        self.engine = engine
        self.device = torch.device("cuda", engine.device_id)
        self.seeds = [((int(seed) or 1) + i + 1) & 0x7FFFFFFF for i in range(5)]                    # synthetic_map.py:183
        self.params = [(f * float(m), o, l, g) for (f, o, l, g), m in zip(MAP_CONFIGS, frequency_mult)]
        if stats_json is not None:
            d = json.load(open(stats_json)) if isinstance(stats_json, str) else stats_json
            self.stats = dict(a_temp_std=float(d["a_temp_std"]), b_temp_std=float(d["b_temp_std"]), temp_std_p1=float(d["temp_std_p1"]), temp_std_p99=float(d["temp_std_p99"]),
                              noise_quantile_tables=[np.asarray(t, np.float64) for t in d["noise_quantile_tables"]],
                              data_quantile_tables=[np.asarray(t, np.float64) for t in d["data_quantile_tables"]])
        else:
            # noise quantiles: measured on this generator the way _compute_map_stats does (1024 x 1024 samples at stride 32, fixed seeds 1..5)
            ident = np.linspace(-1.0, 1.0, 64).astype(np.float32)
            nqs = []
            for ch, (f, o, l, g) in enumerate(self.params):
                raw = self._channel(ch, 0, 0, 1024, 1024, seed=ch + 1, src=ident, dst=ident, scale=32).cpu().numpy().ravel()
                nqs.append(_build_quantiles(raw, 64, 1e-4))
            self.stats = default_stats(nqs)

    def _channel(self, ch, i1, j1, rows, cols, seed=None, src=None, dst=None, scale=1):
        f, o, l, g = self.params[ch]
        src = np.ascontiguousarray((self.stats["noise_quantile_tables"][ch] if src is None else src), dtype=np.float32)
        dst = np.ascontiguousarray((self.stats["data_quantile_tables"][ch] if dst is None else dst), dtype=np.float32)
        out = torch.empty((rows, cols), dtype=torch.float32, device=self.device)
        vp = lambda a: C.c_void_p(a.ctypes.data)
        check(lib().td_perlin_map(self.engine._h, rows, cols, int(i1), int(j1), int(self.seeds[ch] if seed is None else seed), float(f * scale), int(o), float(l), float(g),
                                  vp(src), vp(dst), len(src), ptr(out)))
        return out

    def sample_raw(self, i1, j1, i2, j2):
        """(5, i2-i1, j2-j1) transferred noise before `finalize`, in the REFERENCE's orientation (synthetic_map.py:207-218):
        x = arange(i1, i2), y = arange(j1, j2), np.meshgrid(x, y) in 'xy' order, values flattened row-major and reshaped to (i2-i1, j2-j1).
        For a square request that is out[r, c] = noise(x = i1 + c, y = j1 + r) -- the transpose of the natural order, which WorldPipeline
        undoes with its (i, j) -> (j, i) argument swap (world_pipeline.py:901), so that world cell (I, J) always sees noise(J, I) whatever
        window asks for it.  For a non-square request the reference's reshape scrambles rows; reproduced as written (the pipeline only asks
        for square windows).  `_channel` is the kernel's natural order K[r][c] = noise(i1 + r, j1 + c): out = K^T flattened and reshaped."""
        n, m = i2 - i1, j2 - j1
        return torch.stack([self._channel(ch, i1, j1, n, m).t().contiguous().reshape(n, m) for ch in range(5)])

    def finalize(self, raw):
        """synthetic_map.py:232-252, same arithmetic on device tensors."""
        s = self.stats
        elev, temp, tstd, precip, pcv = (raw[k].float() for k in range(5))
        lapse = (-6.5 + 0.0015 * precip).clamp(-9.8, -4.0) / 1000
        temp = (temp + lapse * torch.clamp(elev, min=0)).clamp(-10, 40)
        temp = torch.where(temp > 20, temp, (temp - 20) * 1.25 + 20)
        p1, p99, a, b = s["temp_std_p1"], s["temp_std_p99"], s["a_temp_std"], s["b_temp_std"]
        t = (tstd - p1) / (p99 - p1)
        baseline = torch.clamp(-(a * temp + b), min=p1)
        tstd = t * (p99 - baseline) + baseline
        tstd = torch.clamp(tstd + (a * temp + b), min=20)
        pcv = pcv * torch.clamp((185 - 0.04111 * precip) / 185, min=0)
        return torch.stack([elev, temp, tstd, precip, pcv])

    def __call__(self, i1, j1, i2, j2):
        m = self.finalize(self.sample_raw(i1, j1, i2, j2))
        m[0] = torch.sign(m[0]) * torch.sqrt(torch.abs(m[0]))
        return m


def _build_quantiles(values, n_quantiles=64, eps=1e-4):
    """perlin_transform.py:3-39: strictly increasing empirical quantiles."""
    v = np.asarray(values, np.float64).ravel()
    v = v[~np.isnan(v)]
    vq = np.quantile(v, np.linspace(eps, 1.0 - eps, n_quantiles))
    d = np.diff(vq)
    md = d[d > 0].min() if np.any(d > 0) else 1e-10
    for i in range(1, len(vq)):
        if vq[i] <= vq[i - 1]:
            vq[i] = vq[i - 1] + md * 0.1
    return vq


def make_synthetic_map_factory(engine, frequency_mult=(1.0, 1.0, 1.0, 1.0, 1.0), seed=None, drop_water_pct=0.0, stats_json=None):
    return SyntheticMapFactory(engine, seed if seed else 1, frequency_mult, drop_water_pct, stats_json)
