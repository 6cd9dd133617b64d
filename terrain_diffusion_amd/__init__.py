"""MI355X-native InfiniteDiffusion sampling engine — Python host side over the C-ABI (include/td_engine.h).

Drop-in surface for the hot path of xandergos/terrain-diffusion: EDMUnet2D, the EDM
DPM-Solver++ scheduler, the bounded tiled samplers, the portable tile-seeded noise field and the
InfiniteTensor operator layer.  The compute runs in hand-written HIP kernels (csrc/); there is no CPU
fallback — importing is cheap, the first call that needs the engine raises if the HIP library or a GPU is missing.
"""
from ._lib import TdError, LIB_PATH, EXPORTS  # noqa: F401
from .unet import EDMUnet2D  # noqa: F401
from .scheduler import EDMDPMSolverMultistepScheduler  # noqa: F401
from .sampling import sample_base_diffusion, sample_base_consistency, sample_independent_tiles, _tile_starts, _linear_weight_window, _process_cond_img  # noqa: F401
from .sampling import sample_decoder_diffusion_tiled, sample_decoder_consistency_tiled, sample_coarse_tiled  # noqa: F401
from .noise import gaussian_noise_patch, gaussian_noise_patches, standard_normal, next_seed, _tile_seed  # noqa: F401
from .world_pipeline import WorldPipeline  # noqa: F401
from .infinite_tensor import InfiniteTensor, TensorWindow, MemoryTileStore, DeviceTileStore, HDF5TileStore  # noqa: F401
