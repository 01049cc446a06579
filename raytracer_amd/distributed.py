"""Image tiles over ranks: one process per GPU, tiles t with t % world == rank, one all-gather.

This replaces the scanline work queue of the reference (Source/render_work_queue.cpp:59-80) at
node scale.  The scene is replicated on every GPU; each rank renders its interleaved 16x16-pixel
tiles into a compact buffer (768 bytes per tile), the buffers are exchanged with ONE
all_gather_into_tensor (RCCL over xGMI with backend "nccl"; "gloo" on CPU tensors in the tests),
and rank 0 scatters the tiles into the row-major framebuffer.  Nothing else crosses ranks.
"""
import numpy as np

from . import abi

TILE = abi.RT_TILE
TILE_BYTES = TILE * TILE * 3


def rect_of(params):
    if params.rect_w:
        return params.rect_x0, params.rect_y0, params.rect_w, params.rect_h
    return 0, 0, params.width, params.height


def tile_grid(params):
    _, _, w, h = rect_of(params)
    tiles_x, tiles_y = -(-w // TILE), -(-h // TILE)
    return tiles_x, tiles_y, tiles_x * tiles_y


def tile_position(t, tiles_x):
    """(row, column) of tile number t: rows of tiles are rotated against each other (include/rt_amd.h)."""
    row, place = divmod(t, tiles_x)
    return row, (place - row) % tiles_x


def rank_tiles(params, rank, world):
    """Global tile indices rank `rank` renders, in buffer order."""
    _, _, total = tile_grid(params)
    return list(range(rank, total, world))


def stride_tiles(params, world):
    """Tiles per rank in the gathered buffer (the largest share; smaller shares are padded)."""
    _, _, total = tile_grid(params)
    return -(-total // world)


def tile_buffer_from_image(rgb, params, rank, world):
    """Compact tile buffer (stride_tiles x 16 x 16 x 3, uint8) holding rank's tiles of a rectangle image."""
    _, _, w, h = rect_of(params)
    tiles_x, _, _ = tile_grid(params)
    out = np.zeros((stride_tiles(params, world), TILE, TILE, 3), np.uint8)
    for slot, t in enumerate(rank_tiles(params, rank, world)):
        ty, tx = tile_position(t, tiles_x)
        block = rgb[ty * TILE:min((ty + 1) * TILE, h), tx * TILE:min((tx + 1) * TILE, w)]
        out[slot, :block.shape[0], :block.shape[1]] = block
    return out


def assemble_tiles_host(gathered, params, world):
    """numpy form of the device scatter (rt_assemble_tiles_device): gathered is (world, stride, 16, 16, 3)."""
    _, _, w, h = rect_of(params)
    tiles_x, _, total = tile_grid(params)
    img = np.zeros((h, w, 3), np.uint8)
    for t in range(total):
        rank, slot = t % world, t // world
        ty, tx = tile_position(t, tiles_x)
        hh, ww = min(TILE, h - ty * TILE), min(TILE, w - tx * TILE)
        img[ty * TILE:ty * TILE + hh, tx * TILE:tx * TILE + ww] = gathered[rank, slot, :hh, :ww]
    return img


def all_gather_tiles(local, world):
    """One all_gather_into_tensor of equally sized uint8 tile buffers (a torch tensor, CPU or GPU)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local.reshape(-1))
    return out
