"""rt_render_tiles_noalloc (the entry point that allocates nothing) against rt_render_tiles_device, through the C ABI:
two calls in a row on one scene, the second with a smaller input; tile buffers must be identical byte for byte.
Also: the device's free memory is the same after the call as before it, and a share of more than RT_NOALLOC_MAX_TILES tiles is
refused with a clear error."""
import numpy as np
import pytest

from conftest import scene_path
from raytracer_amd import abi

pytestmark = pytest.mark.gpu


def _both(gpu, scene, cam, prm):
    import torch
    n = gpu.rank_tile_count(prm) * 768
    want = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
    got = torch.full((n,), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    scene.render_tiles_device(cam, prm, want.data_ptr())
    free_before = torch.cuda.mem_get_info(0)[0]              # hipMemGetInfo: free device memory
    gpu.render_tiles_noalloc(scene, cam, prm, got.data_ptr())
    assert torch.cuda.mem_get_info(0)[0] == free_before      # the call took no device memory (and gave none back)
    torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy()


def test_no_alloc_entry_point_matches_the_allocating_one(gpu):
    fi = gpu.FileInfo(scene_path("scene6"))          # triangles, spheres, an ellipsoid, two lights, maxdepth 5
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    first = fi.params(rect_x0=256, rect_y0=176, rect_w=128, rect_h=128)             # 64 tiles: the maximum
    second = fi.params(rect_x0=300, rect_y0=200, rect_w=50, rect_h=37)              # smaller, ragged tiles; the array keeps the first call's contents
    for prm in (first, second, first):
        want, got = _both(gpu, scene, cam, prm)
        assert want.any() and np.array_equal(want, got)
    # a rank's share of a bigger rectangle
    split = fi.params(rect_x0=0, rect_y0=0, rect_w=640, rect_h=480, tile_rank=3, tile_world=19)
    assert gpu.rank_tile_count(split) <= abi.RT_NOALLOC_MAX_TILES
    want, got = _both(gpu, scene, cam, split)
    assert np.array_equal(want, got)
    scene.close()

    fi7 = gpu.FileInfo(scene_path("scene7"))         # the walk of a triangle hierarchy
    cam7 = fi7.camera()
    scene7 = gpu.Scene(fi7.scene_desc)
    for prm in (fi7.params(rect_x0=280, rect_y0=200, rect_w=112, rect_h=96), fi7.params(rect_x0=300, rect_y0=220, rect_w=40, rect_h=24, max_bounce_count=2)):
        want, got = _both(gpu, scene7, cam7, prm)
        assert want.any() and np.array_equal(want, got)
    scene7.close()


def test_no_alloc_entry_point_refuses_more_than_its_maximum(gpu):
    import torch
    fi = gpu.FileInfo(scene_path("scene1"))
    scene = gpu.Scene(fi.scene_desc)
    prm = fi.params(rect_x0=0, rect_y0=0, rect_w=144, rect_h=128)                   # 72 tiles
    buf = torch.zeros(gpu.rank_tile_count(prm) * 768, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(gpu.RtError, match="more than 64 tiles"):
        gpu.render_tiles_noalloc(scene, fi.camera(), prm, buf.data_ptr())
    scene.close()
