"""Window-grid geometry shared by the bounded samplers and the multi-GPU sharding (one definition for both)."""


def tile_starts(length, tile_size, stride):
    """Origins of the windows that cover [0, length): a regular ladder 0, stride, 2*stride, ... while the window still fits, plus one
    final window flush with the end when the ladder stops short.  Same result as training/evaluation/__init__.py:16-22 for every
    (length, tile, stride) — pinned by tests/golden/geometry.npz."""
    last = max(0, int(length) - int(tile_size))      # origin of the window that ends exactly at `length`
    step = max(1, int(stride))
    starts = [k * step for k in range(last // step + 1)]
    if starts[-1] < last:
        starts.append(last)
    return starts
