"""REST wire formats of the reference's servers (terrain_diffusion/inference/api.py:73-100; API_README.md:67-83) and tile-store persistence
(world_pipeline.py:625-674), so that a server built on this engine speaks the same bytes.

  * /terrain binary payload: elevation as int16 little-endian, floor()ed and clamped to [-32768, 32767], H*W*2 bytes, followed (when climate
    is present) by the first four climate channels as float32 little-endian INTERLEAVED per pixel (H, W, 4); headers X-Height / X-Width.
  * persistent world file: window outputs keyed by (tensor_id, window index) plus one attribute WORLD_PIPELINE_PARAMS = json.dumps({'seed',
    'kwargs'}, sort_keys=True).  The reference stores it in HDF5 through the third-party infinite_tensor.HDF5TileStore; h5py is not
    installed here, so FileTileStore keeps the same records in a directory of .npy files + params.json (HDF5TileStore delegates to h5py when
    it is importable).  Format parity with the reference's HDF5 layout is unpinned (third-party, absent).
"""
import json
import os
import re

import numpy as np
import torch

from .infinite_tensor import MemoryTileStore


def elev_to_int16(elev):
    arr = torch.as_tensor(elev).detach().cpu().numpy().astype(np.float32, copy=False)
    return np.clip(np.floor(arr), -32768, 32767).astype("<i2", copy=False)


def binary_payload(elev, climate=None):
    """-> (bytes, {"X-Height": h, "X-Width": w})  (api.py:80-100)"""
    e16 = elev_to_int16(elev)
    h, w = e16.shape
    payload = e16.tobytes()
    if climate is not None and climate.shape[0] >= 4:
        c = torch.as_tensor(climate)[:4].detach().cpu().numpy().astype("<f4", copy=False)
        payload += np.ascontiguousarray(np.transpose(c, (1, 2, 0))).tobytes()
    return payload, {"X-Height": str(h), "X-Width": str(w)}


def parse_payload(payload, height, width):
    """inverse of binary_payload (client side, API_README.md:67-83) -> (elev int16 (H,W), climate float32 (4,H,W) or None)"""
    n = height * width
    elev = np.frombuffer(payload[:2 * n], dtype="<i2").reshape(height, width)
    rest = payload[2 * n:]
    clim = np.frombuffer(rest, dtype="<f4").reshape(height, width, 4).transpose(2, 0, 1) if len(rest) == 16 * n else None
    return elev, clim


class FileTileStore(MemoryTileStore):
    """Persistent tile store: an LRU of `cache_size_tiles` windows in memory in front of a directory of .npy records; survives restarts.
    put() writes through; get() falls back to disk.  `params` mirrors the WORLD_PIPELINE_PARAMS attribute (world_pipeline.py:625-664)."""
    ATTR_KEY = "WORLD_PIPELINE_PARAMS"

    def __init__(self, path, mode="a", compression=None, compression_opts=None, cache_size_tiles=100):
        super().__init__(cache_size_bytes=None)
        self.path, self.cache_size_tiles = path, cache_size_tiles
        if os.path.exists(path) and not os.path.isdir(path):
            raise ValueError(f"{path!r} is a regular file: FileTileStore keeps its records in a DIRECTORY of .npy files (an HDF5 world file of the "
                             "reference needs h5py, which this environment does not have)")
        if mode == "w" and os.path.isdir(path):   # truncate: only this store's own records, never a user's other files
            for f in os.listdir(path):
                if f.endswith(".npy") or f == "params.json":
                    os.remove(os.path.join(path, f))
        os.makedirs(path, exist_ok=True)

    @staticmethod
    def _fname(key):
        tid, ctx = key
        return re.sub(r"[^A-Za-z0-9_.-]", "_", str(tid)) + "__" + "_".join(("m%d" % -c) if c < 0 else str(c) for c in ctx) + ".npy"

    def get(self, key):
        t = super().get(key)
        if t is not None:
            return t
        f = os.path.join(self.path, self._fname(key))
        if not os.path.exists(f):
            return None
        t = torch.from_numpy(np.load(f))
        self._remember(key, t)
        return t

    def _remember(self, key, t):
        super().put(key, t)
        while len(self._d) > self.cache_size_tiles:
            self._d.popitem(last=False)
            self.evictions += 1

    def put(self, key, t):
        t = torch.as_tensor(t).detach().cpu()
        np.save(os.path.join(self.path, self._fname(key)), t.numpy())
        self._remember(key, t)

    def clear(self, tensor_id=None):
        super().clear(tensor_id)
        pre = None if tensor_id is None else re.sub(r"[^A-Za-z0-9_.-]", "_", str(tensor_id)) + "__"
        for f in os.listdir(self.path):
            if f.endswith(".npy") and (pre is None or f.startswith(pre)):
                os.remove(os.path.join(self.path, f))

    # ---- WORLD_PIPELINE_PARAMS (world_pipeline.py:625-664)
    @property
    def params(self):
        f = os.path.join(self.path, "params.json")
        return json.load(open(f))[self.ATTR_KEY] if os.path.exists(f) else None

    @params.setter
    def params(self, value):
        with open(os.path.join(self.path, "params.json"), "w") as fh:
            json.dump({self.ATTR_KEY: value}, fh, sort_keys=True)

    def close(self):
        MemoryTileStore.clear(self)
