"""Host-side description of the screen-tile sharding (mirror of ScreenDev / local_pixel_to_xy in csrc/cap_device.h).

The image is cut into 8x8 tiles numbered row-major; shard s of n owns the tiles t with t % n == s.  A shard's tile
buffer holds its tiles back to back, 64 pixels each (row-major inside the tile), padded to the same length
ceil(tiles / n) * 64 on every shard so that one fixed-size gather moves all of them.
"""
import numpy as np

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def padded_pixels(width, height, shard_count):
    tx, ty = tile_grid(width, height)
    return ((tx * ty + shard_count - 1) // shard_count) * TILE * TILE


def pixel_table(width, height, shard_index, shard_count):
    """-> (x, y, valid) arrays of length padded_pixels: the pixel each tile-buffer slot of this shard maps to."""
    tx, ty = tile_grid(width, height)
    n = padded_pixels(width, height, shard_count)
    pl = np.arange(n, dtype=np.int64)
    lt, w = pl >> 6, pl & 63
    gt = lt * shard_count + shard_index
    x = (gt % tx) * TILE + (w & 7)
    y = (gt // tx) * TILE + (w >> 3)
    valid = (gt < tx * ty) & (x < width) & (y < height)
    return x, y, valid


def extract(image, shard_index, shard_count):
    """Row-major image [H, W, C] -> this shard's tile buffer [padded_pixels, C] (zeros in padding)."""
    h, w = image.shape[:2]
    x, y, valid = pixel_table(w, h, shard_index, shard_count)
    out = np.zeros((x.size,) + image.shape[2:], image.dtype)
    out[valid] = image[y[valid], x[valid]]
    return out


def assemble(buffers, width, height):
    """List of per-shard tile buffers (the gather result) -> row-major image."""
    n = len(buffers)
    img = np.zeros((height, width) + buffers[0].shape[1:], buffers[0].dtype)
    for s, b in enumerate(buffers):
        x, y, valid = pixel_table(width, height, s, n)
        img[y[valid], x[valid]] = b[valid]
    return img
