"""WorldPipeline on the engine: the reference's public object (terrain_diffusion/inference/world_pipeline.py:276-819, 1367-1384) with the three
EDMUnet2D stages, the scheduler loops, the noise field, the overlap blend and the output composition running in HIP.

Same constructor keywords, the same kwargs dictionary, the same public methods (from_local_models / from_pretrained / save_pretrained, to,
bind, get, coarse / latents / residual, empty_cache, close, rebuild, change_seed, set_cond_snr, set_custom_conditioning_import, context
manager), the same window geometry, seeds and channel arithmetic (pipeline.py stage builders, pinned to the reference's own
_coarse/_latent/_decoder_inference outputs).  Differences, all deliberate:
  * windows stay in HBM by default (device_resident=True: DeviceWindowTensor + DeviceTileStore; `cache_limit` bounds it in bytes); with
    caching_strategy='indirect' the windows go through a persistent host store (wire.FileTileStore, or HDF5TileStore when h5py exists);
  * all missing windows of a request are batched through the U-Net (the reference batches only the latent stage);
  * `dtype` accepts 'bf16' / 'fp16' / None(fp32 validation mode) and selects the engine's storage type; torch_compile is accepted and ignored;
  * the synthetic conditioning source is synthetic_map.SyntheticMapFactory (noise values and quantile tables are this package's own: the
    reference's need pyfastnoiselite and ETOPO / WorldClim rasters, absent here) unless `conditioning_fn` is given.
"""
import json
import os

import numpy as np
import torch

from .engine import get_engine
from .infinite_tensor import DeviceTileStore, MemoryTileStore
from .noise import next_seed
from .scheduler import EDMDPMSolverMultistepScheduler
from .unet import EDMUnet2D
from . import composition, pipeline as stages

DEFAULT_COARSE_MEANS = [-37.67916460232751, 2.22578822145657, 18.030293275011356, 333.8442390481231, 1350.1259248456176, 52.444339366764396]
DEFAULT_COARSE_STDS = [39.68515115440358, 3.0981253981231522, 8.940333096712806, 322.25238547630295, 856.3430083394657, 30.982620765341043]


class WorldPipeline:
    config_name = "config.json"
    COARSE_MODEL_FOLDER, BASE_MODEL_FOLDER, DECODER_MODEL_FOLDER = "coarse_model", "base_model", "decoder_model"
    ignore_for_config = ["seed", "latents_batch_size", "log_mode", "cache_limit", "caching_strategy", "torch_compile", "dtype"]

    def __init__(self, seed=None, latents_batch_size=(1, 2, 4, 8, 16), native_resolution=90.0, *, T=2, log_mode="info", torch_compile=False, dtype=None,
                 latent_compression=8, frequency_mult=None, drop_water_pct=0.5, cond_snr=None, coarse_pooling=1, elev_coarse_pool_mode="avg",
                 p5_coarse_pool_mode="avg", residual_mean=0.0, residual_std=1.1678, coarse_means=None, coarse_stds=None, caching_strategy="direct",
                 cache_limit=100 * 1024 * 1024, onestep_latent=False, decoder_tile_size=512, decoder_tile_stride=384, device="cuda", device_resident=True,
                 conditioning_fn=None, synthetic_stats_json=None, **deprecated_kwargs):
        if T not in (1, 2):
            raise ValueError(f"T must be 1 or 2, got {T}")
        self.T = T
        self.seed = (int(seed) & 0xFFFFFFFFFFFFFFFF) if seed is not None else next_seed(None)
        self._batch_sizes = [latents_batch_size] if isinstance(latents_batch_size, int) else sorted(latents_batch_size)
        self.latents_batch_size = self._batch_sizes[-1]
        self.native_resolution, self.latent_compression, self.log_mode = native_resolution, latent_compression, log_mode
        self.torch_compile = False  # accepted for drop-in compatibility: the engine replaces the compiler
        self.caching_strategy, self.cache_limit, self.onestep_latent = caching_strategy, cache_limit, onestep_latent
        self.decoder_tile_size, self.decoder_tile_stride = decoder_tile_size, decoder_tile_stride
        self.kwargs = {
            "latent_compression": latent_compression, "log_mode": log_mode,
            "frequency_mult": frequency_mult if frequency_mult is not None else [1.5, 3, 3, 3, 3],
            "drop_water_pct": drop_water_pct, "cond_snr": cond_snr if cond_snr is not None else [0.3, 0.1, 1.0, 0.1, 1.0],
            "coarse_pooling": coarse_pooling, "elev_coarse_pool_mode": elev_coarse_pool_mode, "p5_coarse_pool_mode": p5_coarse_pool_mode,
            "histogram_raw": deprecated_kwargs.get("histogram_raw", [0.0] * 5) or [0.0] * 5,
            "residual_mean": residual_mean, "residual_std": residual_std,
            "coarse_means": coarse_means if coarse_means is not None else list(DEFAULT_COARSE_MEANS),
            "coarse_stds": coarse_stds if coarse_stds is not None else list(DEFAULT_COARSE_STDS),
        }
        self._config = dict(native_resolution=native_resolution, T=T, latent_compression=latent_compression, frequency_mult=self.kwargs["frequency_mult"],
                            drop_water_pct=drop_water_pct, cond_snr=self.kwargs["cond_snr"], coarse_pooling=coarse_pooling, elev_coarse_pool_mode=elev_coarse_pool_mode,
                            p5_coarse_pool_mode=p5_coarse_pool_mode, residual_mean=residual_mean, residual_std=residual_std, coarse_means=self.kwargs["coarse_means"],
                            coarse_stds=self.kwargs["coarse_stds"], onestep_latent=onestep_latent, decoder_tile_size=decoder_tile_size, decoder_tile_stride=decoder_tile_stride)
        self.dtype = {"bf16": "bf16", "fp16": "fp16", None: "fp32", "fp32": "fp32"}[dtype]
        self._device_spec, self.device_resident = device, bool(device_resident)
        self.engine = get_engine(device)
        self.device = torch.device("cuda", self.engine.device_id)
        self.coarse_model = self.base_model = self.decoder_model = None
        self.tile_store = None
        self._store_path = None
        self.synthetic_map_factory = None
        self._conditioning_fn, self._synthetic_stats_json = conditioning_fn, synthetic_stats_json
        self.coarse = self.latents = self.residual = None
        self.custom_conditioning_imports, self.custom_conditioning_import_origins, self.custom_conditioning_default_values = {}, {}, {}

    # ------------------------------------------------------------------ construction / persistence (world_pipeline.py:470-565)
    @classmethod
    def from_local_models(cls, coarse_model_path, base_model_path, decoder_model_path, **kwargs):
        p = cls(**kwargs)
        p.coarse_model = EDMUnet2D.from_pretrained(coarse_model_path, dtype=p.dtype, device=p._device_spec)
        p.base_model = EDMUnet2D.from_pretrained(base_model_path, dtype=p.dtype, device=p._device_spec)
        p.decoder_model = EDMUnet2D.from_pretrained(decoder_model_path, dtype=p.dtype, device=p._device_spec)
        return p

    @classmethod
    def from_models(cls, coarse_model, base_model, decoder_model, **kwargs):
        """Already-constructed engine models (tests / synthetic weights)."""
        p = cls(**kwargs)
        p.coarse_model, p.base_model, p.decoder_model = coarse_model, base_model, decoder_model
        return p

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, token=None, **kwargs):
        """Directory in the reference's layout: config.json + coarse_model/ base_model/ decoder_model/ (each config.json + *.safetensors).
        Hub ids are not resolvable offline."""
        root = pretrained_model_name_or_path
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root}: not a local directory (no network: HuggingFace Hub ids cannot be resolved here)")
        with open(os.path.join(root, cls.config_name)) as f:
            config = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        config.update(kwargs)
        return cls.from_local_models(os.path.join(root, cls.COARSE_MODEL_FOLDER), os.path.join(root, cls.BASE_MODEL_FOLDER),
                                     os.path.join(root, cls.DECODER_MODEL_FOLDER), **config)

    def save_config(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(dict(self._config, _class_name="WorldPipeline"), f, indent=2, sort_keys=True)

    def save_pretrained(self, save_directory, **kwargs):
        """world_pipeline.py:500-518: the pipeline's config.json + one sub-folder per model (config.json + safetensors)."""
        self.save_config(save_directory)
        for model, folder in ((self.coarse_model, self.COARSE_MODEL_FOLDER), (self.base_model, self.BASE_MODEL_FOLDER), (self.decoder_model, self.DECODER_MODEL_FOLDER)):
            if model is not None:
                model.save_pretrained(os.path.join(save_directory, folder), **kwargs)

    def to(self, device):
        """world_pipeline.py:568-585 moves the three torch modules.  Here the models live in the engine of the GPU the pipeline was constructed
        on (weights packed for its kernels at load time): `to` of that device is a no-op, anything else is refused loudly instead of being
        silently ignored -- construct the pipeline with `device=` (one process per GPU, parallel.shard_requests) to use another GPU."""
        if isinstance(device, (torch.dtype,)):
            raise TypeError("WorldPipeline.to(dtype): the storage type is fixed at construction (dtype='bf16' | 'fp16' | 'fp32')")
        d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if d.type != "cuda":
            raise RuntimeError(f"WorldPipeline.to({device!r}): terrain_diffusion_amd runs on MI355X only, there is no CPU path")
        if d.index is not None and d.index != self.engine.device_id:
            raise RuntimeError(f"WorldPipeline.to({device!r}): this pipeline's models are resident on cuda:{self.engine.device_id}; "
                               "construct a pipeline with device=... for another GPU")
        return self

    # ------------------------------------------------------------------ bind / rebuild (world_pipeline.py:588-740)
    def _params(self):
        return {"seed": self.seed, "kwargs": self.kwargs}

    def bind(self, hdf5_file=None, mode="a", compression="gzip", compression_opts=4, on_param_mismatch="stored"):
        """world_pipeline.py:587-621.  hdf5_file='TEMP' makes a temporary world that close() removes.  When the world file already records other
        parameters than this pipeline's (seed / kwargs), the reference asks on the console whether to overwrite them; a library cannot, so
        `on_param_mismatch` decides: 'stored' (the reference's default answer: keep the stored world and adopt ITS seed and kwargs, with a
        warning that lists the differences), 'overwrite' (the reference's 'y') or 'error'."""
        if on_param_mismatch not in ("stored", "overwrite", "error"):
            raise ValueError("on_param_mismatch must be 'stored', 'overwrite' or 'error'")
        if self.caching_strategy == "direct":
            self._init_tile_store(None, None)
        else:
            if hdf5_file is None:
                raise ValueError("hdf5_file is required when caching_strategy='indirect'")
            self._is_temp_file = str(hdf5_file).upper() == "TEMP"
            if self._is_temp_file:
                import tempfile
                hdf5_file = tempfile.mkdtemp(prefix="terrain_")   # FileTileStore keeps a directory; with h5py the file lives inside it
                self._temp_dir = hdf5_file
                try:
                    import h5py  # noqa: F401
                    hdf5_file = os.path.join(hdf5_file, "world.h5")
                except ImportError:
                    pass
            self._store_path = hdf5_file
            self._init_tile_store(hdf5_file, mode, compression, compression_opts)
            current = json.loads(json.dumps(self._params()))
            stored = getattr(self.tile_store, "params", None)
            if stored is None:
                self.tile_store.params = current
            elif stored != current:
                diffs = {k: (stored.get(k), current[k]) for k in current if stored.get(k) != current[k]}
                if on_param_mismatch == "error":
                    raise ValueError(f"world file {hdf5_file!r} was made with other parameters (stored, current): {diffs}")
                if on_param_mismatch == "overwrite":
                    self.tile_store.params = current
                else:
                    import warnings
                    warnings.warn(f"world file {hdf5_file!r} was made with other parameters; keeping the STORED world (stored, current): {diffs}")
                    self.seed, self.kwargs = stored["seed"], stored["kwargs"]
        self._init_conditioning()
        self._build_hierarchy()
        return self

    def _init_tile_store(self, path, mode, compression=None, compression_opts=None):
        if self.caching_strategy == "direct":
            self.tile_store = DeviceTileStore(cache_size_bytes=self.cache_limit) if self.device_resident else MemoryTileStore(cache_size_bytes=self.cache_limit)
            return
        try:
            from .infinite_tensor import HDF5TileStore
            self.tile_store = HDF5TileStore(path, mode=mode, compression=compression, compression_opts=compression_opts, cache_size_tiles=100)
        except ImportError:
            from .wire import FileTileStore
            self.tile_store = FileTileStore(path, mode=mode, cache_size_tiles=100)

    def _init_conditioning(self):
        if self._conditioning_fn is None:
            from .synthetic_map import make_synthetic_map_factory
            self.synthetic_map_factory = make_synthetic_map_factory(self.engine, seed=self.seed & 0x7FFFFFFF, frequency_mult=self.kwargs["frequency_mult"],
                                                                    drop_water_pct=self.kwargs["drop_water_pct"], stats_json=self._synthetic_stats_json)

    def _build_hierarchy(self):
        resident = self.device_resident and self.caching_strategy == "direct"
        kw = dict(tile_store=self.tile_store, device_resident=resident)
        bs = tuple(self._batch_sizes)   # allowed batch sizes; the stages cut their missing windows greedily into these
        sch = EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80, sigma_data=0.5)
        self.coarse = stages.build_coarse_stage(self.coarse_model, sch, seed=self.seed, cond_map_fn=self._conditioning_model_input,
                                                coarse_means=self.kwargs["coarse_means"], coarse_stds=self.kwargs["coarse_stds"], cond_snr=self.kwargs["cond_snr"],
                                                coarse_pooling=self.kwargs["coarse_pooling"], elev_coarse_pool_mode=self.kwargs["elev_coarse_pool_mode"],
                                                p5_coarse_pool_mode=self.kwargs["p5_coarse_pool_mode"], batch_size=bs, **kw)
        self.latents = stages.build_latent_stage(self.base_model, seed=self.seed, coarse=self.coarse, histogram_raw=[self.kwargs["histogram_raw"]], T=self.T,
                                                 onestep_latent=self.onestep_latent, batch_size=bs, **kw)
        self.residual = stages.build_decoder_stage(self.decoder_model, self.latents, seed=self.seed, tile_size=self.decoder_tile_size,
                                                   tile_stride=self.decoder_tile_stride, latent_compression=self.latent_compression, batch_size=tuple(b for b in bs if b <= 4) or (1,), **kw)

    def rebuild(self):
        if self.tile_store is None:
            return
        if self.caching_strategy == "direct":
            self._init_tile_store(None, None)
        else:
            self.tile_store.close()
            self._init_tile_store(self._store_path, "w")
            self.tile_store.params = json.loads(json.dumps(self._params()))
        self._init_conditioning()
        self._build_hierarchy()

    def change_seed(self, seed=None):
        new_seed = (int(seed) & 0xFFFFFFFFFFFFFFFF) if seed is not None else next_seed(None)
        if new_seed == self.seed:
            return False
        self.seed = new_seed
        self.rebuild()
        return True

    def set_cond_snr(self, cond_snr):
        if len(cond_snr) != 5:
            raise ValueError("cond_snr must contain exactly 5 values.")
        self.kwargs["cond_snr"] = [float(x) for x in cond_snr]
        self.rebuild()

    # ------------------------------------------------------------------ conditioning source (world_pipeline.py:779-907)
    def set_custom_conditioning_import(self, channel, values, origin_i, origin_j, default_value=None):
        values = np.asarray(values, dtype=np.float32)
        if values.ndim != 2:
            raise ValueError("Custom conditioning import must be a 2-D array.")
        channel = int(channel)
        self.custom_conditioning_imports[channel] = values.copy()
        self.custom_conditioning_import_origins[channel] = (int(origin_i), int(origin_j))
        if default_value is None:
            self.custom_conditioning_default_values.pop(channel, None)
        else:
            self.custom_conditioning_default_values[channel] = float(default_value)
        self.rebuild()

    def _sample_custom_conditioning_channel(self, channel, ci0, ci1, cj0, cj1):
        imp, default = self.custom_conditioning_imports.get(channel), self.custom_conditioning_default_values.get(channel)
        if imp is None and default is None:
            return None, None
        h, w = ci1 - ci0, cj1 - cj0
        values = np.full((h, w), 0.0 if default is None else float(default), dtype=np.float32)
        mask = np.full((h, w), default is not None, dtype=bool)
        if imp is not None:
            si0, sj0 = self.custom_conditioning_import_origins[channel]
            oi0, oi1, oj0, oj1 = max(ci0, si0), min(ci1, si0 + imp.shape[0]), max(cj0, sj0), min(cj1, sj0 + imp.shape[1])
            if oi0 < oi1 and oj0 < oj1:
                values[oi0 - ci0:oi1 - ci0, oj0 - cj0:oj1 - cj0] = imp[oi0 - si0:oi1 - si0, oj0 - sj0:oj1 - sj0]
                mask[oi0 - ci0:oi1 - ci0, oj0 - cj0:oj1 - cj0] = True
        return (values, mask) if mask.any() else (None, None)

    def _conditioning_model_input(self, ci0, ci1, cj0, cj1):
        """(5, ci1-ci0, cj1-cj0) conditioning for the coarse U-Net (world_pipeline.py:884-907), incl. the reference's (i, j) -> (j, i) swap."""
        if self._conditioning_fn is not None:
            return torch.as_tensor(self._conditioning_fn(ci0, ci1, cj0, cj1), dtype=torch.float32)
        f = self.synthetic_map_factory
        if not self.custom_conditioning_imports:
            return f(cj0, ci0, cj1, ci1)
        raw = f.sample_raw(cj0, ci0, cj1, ci1).clone()
        for ch in range(raw.shape[0]):
            values, mask = self._sample_custom_conditioning_channel(ch, ci0, ci1, cj0, cj1)
            if values is not None:
                m = torch.from_numpy(mask).to(raw.device)
                raw[ch][m] = torch.from_numpy(values).to(raw.device)[m]
        raw[0] = torch.sign(raw[0]) * torch.sqrt(torch.abs(raw[0]))
        return raw.float()

    # ------------------------------------------------------------------ output (world_pipeline.py:1276-1384)
    def _compute_elev(self, i1, j1, i2, j2, residual_map, scale):
        return composition.compute_elev(self.engine, residual_map, self.latents, i1, j1, i2, j2, scale, self.kwargs["residual_mean"], self.kwargs["residual_std"])

    def _compute_climate(self, i1, j1, i2, j2, elev, scale):
        return composition.compute_climate(self.coarse, i1, j1, i2, j2, elev, scale, engine=self.engine)

    def get(self, i1, j1, i2, j2, with_climate=True):
        """{'elev': (H, W) metres, 'climate': (5, H, W) or None} for the pixel box [i1,i2) x [j1,j2); device tensors."""
        i1, j1, i2, j2 = int(i1), int(j1), int(i2), int(j2)
        if i2 <= i1 or j2 <= j1:
            raise ValueError(f"empty box [{i1},{i2}) x [{j1},{j2})")
        if self.residual is None:
            raise RuntimeError("bind() the pipeline first")
        elev = self._compute_elev(i1, j1, i2, j2, self.residual, scale=self.latent_compression)
        climate = self._compute_climate(i1, j1, i2, j2, elev, scale=self.latent_compression) if with_climate else None
        return {"elev": elev, "climate": climate}

    def empty_cache(self):
        if self.tile_store is None:
            return
        for t in (self.coarse, self.latents, self.residual):
            if t is not None:
                t.clear_cache()

    def close(self):
        if self.tile_store is not None and hasattr(self.tile_store, "close"):
            self.tile_store.close()
        if getattr(self, "_is_temp_file", False) and getattr(self, "_temp_dir", None):   # world_pipeline.py:711-713
            import shutil
            shutil.rmtree(self._temp_dir, ignore_errors=True)
            self._temp_dir = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()
        return False
