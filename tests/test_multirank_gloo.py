"""The multi-rank path (tile interleave -> one all-gather -> assemble) on CPU: world_size 2, gloo.

Each rank renders its tiles with the CPU oracle (there is no GPU here), exchanges the compact tile
buffers with the same collective the GPU path uses, and rank 0 assembles the frame, which must
equal the single-rank render.  The GPU form of the same flow is tests/test_gpu_multirank.py.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, rect, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import checkers
    from raytracer_amd import api, distributed

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    oracle = checkers.load_oracle()
    fi = api.FileInfo(scene_path(name))
    cam = fi.camera()
    prm = fi.params(rect_x0=rect[0], rect_y0=rect[1], rect_w=rect[2], rect_h=rect[3], tile_rank=rank, tile_world=world)
    assert api.rank_tile_count(prm) == len(distributed.rank_tiles(prm, rank, world))
    mine = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, prm, threads=2)["rgb"]   # only this rank's tiles are filled
    local = torch.from_numpy(distributed.tile_buffer_from_image(mine, prm, rank, world).reshape(-1))
    gathered = distributed.all_gather_tiles(local, world)
    if rank == 0:
        stride = distributed.stride_tiles(prm, world)
        img = distributed.assemble_tiles_host(gathered.numpy().reshape(world, stride, 16, 16, 3), prm, world)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,rect", [("scene4-specular", (250, 170, 100, 70)), ("scene6", (300, 220, 40, 33))])
def test_two_ranks_assemble_the_single_rank_image(tmp_path, oracle, name, rect):
    import torch.multiprocessing as mp
    import checkers
    from raytracer_amd import api
    out_path = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, rect, out_path), nprocs=2, join=True)
    fi = api.FileInfo(scene_path(name))
    prm = fi.params(rect_x0=rect[0], rect_y0=rect[1], rect_w=rect[2], rect_h=rect[3])
    want = checkers.cpu_render(oracle, "rto_", fi.scene_desc, fi.camera(), prm)["rgb"]
    assert np.array_equal(np.load(out_path), want)


def test_tile_layout_helpers():
    from raytracer_amd import abi, distributed
    p = abi.RenderParams(width=100, height=37, max_bounce_count=1)
    tiles_x, tiles_y, total = distributed.tile_grid(p)
    assert (tiles_x, tiles_y, total) == (7, 3, 21)
    seen = sorted(t for r in range(4) for t in distributed.rank_tiles(p, r, 4))
    assert seen == list(range(21)) and distributed.stride_tiles(p, 4) == 6
    img = np.random.default_rng(0).integers(0, 255, (37, 100, 3), dtype=np.uint8)
    gathered = np.stack([distributed.tile_buffer_from_image(img, p, r, 4) for r in range(4)])
    assert np.array_equal(distributed.assemble_tiles_host(gathered, p, 4), img)


def test_tile_numbers_are_a_rotation_of_every_row():
    """Tile number t sits in row t // tiles_x at column (t % tiles_x - row) mod tiles_x: every position once, and the
    tiles of a rank on diagonals where the number of ranks divides the tiles per row."""
    from raytracer_amd import abi, distributed
    for w, h, world in ((3840, 2160, 8), (112, 48, 7), (100, 37, 3), (16, 16, 2), (640, 480, 4)):
        p = abi.RenderParams(width=w, height=h)
        tiles_x, tiles_y, total = distributed.tile_grid(p)
        seen = {distributed.tile_position(t, tiles_x) for t in range(total)}
        assert seen == {(r, c) for r in range(tiles_y) for c in range(tiles_x)}
        if tiles_x % world == 0 and tiles_y > 1:
            column_of_rank0 = [{c for (r, c) in (distributed.tile_position(t, tiles_x) for t in distributed.rank_tiles(p, 0, world)) if r == row} for row in (0, 1)]
            assert column_of_rank0[0] != column_of_rank0[1]      # the next row is shifted: no columns of one rank
