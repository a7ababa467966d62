"""Image-space split of one frame over N ranks (new; the reference is single-GPU).

The frame is cut into horizontal stripes of `stripe_rows` rows; rank r owns stripes r, r+N, r+2N, ...
(okrt_set_tile). Each rank's framebuffer holds its rows packed in order. RNG seeds depend only on
global pixel coordinates (RaytracerCS.hlsl:430), so the union of the tiles is bit-identical to a
single-GPU frame. Tiles are padded to the largest one so a plain gather works.
"""
import numpy as np


def owned_rows(height, stripe_rows, rank, world):
    """Global row indices owned by `rank`, in local (packed) order."""
    rows = []
    n_stripes = (height + stripe_rows - 1) // stripe_rows
    for s in range(rank, n_stripes, world):
        rows.extend(range(s * stripe_rows, min((s + 1) * stripe_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def lightest_rank(height, stripe_rows, world):
    """The rank that owns the fewest rows of the striped split (the last one among equals): the natural ROOT of the gather — it
    finishes its own frames first, so the root-side work (own-tile copy, receives, de-interleave) starts earliest there.
    1080 rows in 8-row stripes over 8 ranks: ranks 0-6 own 136 rows, rank 7 owns 128."""
    rows = [len(owned_rows(height, stripe_rows, r, world)) for r in range(world)]
    least = min(rows)
    return max(r for r in range(world) if rows[r] == least)


def max_owned_rows(height, stripe_rows, world):
    return max(len(owned_rows(height, stripe_rows, r, world)) for r in range(world))


def deinterleave(tiles, height, stripe_rows):
    """tiles: list (len = world) of [>=rows_r, W, C] arrays/tensors -> the full [H, W, C] frame."""
    world = len(tiles)
    first = tiles[0]
    is_torch = hasattr(first, "new_zeros")
    full = first.new_zeros((height,) + tuple(first.shape[1:])) if is_torch else np.zeros((height,) + first.shape[1:], first.dtype)
    for r, t in enumerate(tiles):
        rows = owned_rows(height, stripe_rows, r, world)
        if is_torch:
            import torch
            full[torch.as_tensor(rows, device=first.device)] = t[: len(rows)]
        else:
            full[rows] = t[: len(rows)]
    return full
