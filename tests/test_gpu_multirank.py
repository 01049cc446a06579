"""Tile-interleaved rendering on the GPU: the per-rank tile buffers of a 2- and 3-way split, laid out as the
all-gather delivers them, assemble (rt_assemble_tiles_device) into the image a single rank renders."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_split_frames_assemble_to_the_single_rank_frame(gpu, world):
    import torch
    from raytracer_amd import distributed
    fi = gpu.FileInfo(scene_path("scene6"))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    rect = dict(rect_x0=200, rect_y0=150, rect_w=250, rect_h=170)     # partial tiles on the right and bottom edges
    whole = scene.render(cam, fi.params(**rect))["rgb"]
    stride = distributed.stride_tiles(fi.params(**rect), world)
    gathered = torch.zeros(world * stride * 768, dtype=torch.uint8, device="cuda:0")
    for rank in range(world):
        prm = fi.params(tile_rank=rank, tile_world=world, **rect)
        part = gathered[rank * stride * 768:(rank + 1) * stride * 768]
        scene.render_tiles_device(cam, prm, part.data_ptr())
        host = scene.render(cam, prm)["rgb"]                          # rt_render with a split fills only this rank's tiles
        mask = np.zeros(whole.shape[:2], bool)
        tiles_x = distributed.tile_grid(prm)[0]
        for t in distributed.rank_tiles(prm, rank, world):
            ty, tx = divmod(t, tiles_x)
            mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
        assert np.array_equal(host[mask], whole[mask]) and not host[~mask].any()
    frame = torch.zeros(whole.size, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    gpu.assemble_tiles_device(gathered.data_ptr(), stride, fi.params(tile_world=world, **rect), frame.data_ptr())
    assert np.array_equal(frame.cpu().numpy().reshape(whole.shape), whole)
    scene.close()
