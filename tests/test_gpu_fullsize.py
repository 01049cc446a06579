"""Properties at the benchmark's full size (scene7 at 3840x2160), where the oracle is far too slow:
the pre-tested scan equals the full-test scan on a centred region, a 4-way tile split assembles to the
single-rank frame, and rendering is deterministic."""
import hashlib

import numpy as np
import pytest

from conftest import scene_path
from raytracer_amd import abi

pytestmark = pytest.mark.gpu


def test_scene7_4k_properties(gpu):
    import torch
    from raytracer_amd import distributed
    fi = gpu.FileInfo(scene_path("scene7"))
    fi.set_size(3840, 2160)
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    whole = scene.render(cam, fi.params())
    again = scene.render(cam, fi.params())
    assert hashlib.sha256(whole["rgb"].tobytes()).digest() == hashlib.sha256(again["rgb"].tobytes()).digest()
    for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(whole["stats"], f) == getattr(again["stats"], f), f
    st = whole["stats"]
    assert st.samples == 3840 * 2160 * 16 and st.closest_tests[0] == st.closest_rays * 100000
    assert st.shadow_rays == st.hits      # one point light

    # the centre of the frame through the full test only
    rect = dict(rect_x0=1600, rect_y0=896, rect_w=640, rect_h=368)
    slow = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER, **rect), want_ids=True)
    fast = scene.render(cam, fi.params(**rect), want_ids=True)
    assert np.array_equal(slow["ids"], fast["ids"]) and np.array_equal(slow["rgb"], fast["rgb"])
    assert np.array_equal(fast["rgb"], whole["rgb"][896:896 + 368, 1600:1600 + 640])

    # four ranks' tile buffers, as the all-gather lays them out, assemble to the same frame
    world = 4
    stride = distributed.stride_tiles(fi.params(), world)
    gathered = torch.zeros(world * stride * 768, dtype=torch.uint8, device="cuda:0")
    for rank in range(world):
        scene.render_tiles_device(cam, fi.params(tile_rank=rank, tile_world=world), gathered[rank * stride * 768:].data_ptr())
    frame = torch.zeros(whole["rgb"].size, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    gpu.assemble_tiles_device(gathered.data_ptr(), stride, fi.params(tile_world=world), frame.data_ptr())
    assert np.array_equal(frame.cpu().numpy().reshape(whole["rgb"].shape), whole["rgb"])
    scene.close()


@pytest.mark.gpu
def test_scene7_whole_frames_fast_path_equals_full_test(gpu):
    """Whole frames (no crop: the launch shapes of the bench), pre-tests on and off: same bytes, ids and ray counters.

    The counters matter: one lost shadow-ray block changes a pixel by a few levels at most, and
    16 samples per pixel can hide it; the blocked count cannot.
    """
    fi = gpu.FileInfo(scene_path("scene7"))
    scene = gpu.Scene(fi.scene_desc)
    for (w, h, depth) in ((1280, 720, 3), (1920, 1080, 1)):
        fi.set_size(w, h)
        cam = fi.camera()
        fast = scene.render(cam, fi.params(max_bounce_count=depth), want_ids=True)
        slow = scene.render(cam, fi.params(max_bounce_count=depth, flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
        again = scene.render(cam, fi.params(max_bounce_count=depth), want_ids=True)
        for other in (slow, again):
            for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
                assert getattr(fast["stats"], f) == getattr(other["stats"], f), (w, h, f)
            assert np.array_equal(fast["ids"], other["ids"])
            assert np.array_equal(fast["rgb"], other["rgb"])
    scene.close()


@pytest.mark.gpu
def test_synthetic_1m_triangles_4k_properties(gpu, oracle):
    """The largest configuration: 1 M triangles + 10 k ellipsoids, 8 point lights, 3840x2160, maxdepth 2 (raytracer_amd/synth.py,
    a legal `.test` file that goes through the ordinary parser).  Its triangle hierarchy has 3907 top nodes and 245 super nodes,
    so the scans walk from the super nodes with two rays per lane: the path smaller scenes only reach through RT_FLAG_DEEP_WALK.
    * whole frames are deterministic (bytes and ray counters);
    * a centred crop in the default mode equals the brute-force scan (RT_FLAG_NO_FILTER) on ids, bytes and ray counters, and equals
      the same region of the whole frame;
    * a small crop equals the CPU oracle on ids, bytes and ray counters, and the unmodified reference on bytes where it was built."""
    import checkers
    from raytracer_amd import synth
    fi = gpu.FileInfo(synth.ensure_scene_file())
    assert (fi.width, fi.height, fi.max_recursion_depth) == (3840, 2160, 2)
    d = fi.scene_desc
    assert (d.n_triangles, d.n_ellipsoids, d.n_spheres, d.n_point_lights) == (1_000_000, 10_000, 0, 8)
    cam = fi.camera()
    scene = gpu.Scene(d)
    whole = scene.render(cam, fi.params())
    again = scene.render(cam, fi.params())
    assert hashlib.sha256(whole["rgb"].tobytes()).digest() == hashlib.sha256(again["rgb"].tobytes()).digest()
    for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(whole["stats"], f) == getattr(again["stats"], f), f
    st = whole["stats"]
    assert st.samples == 3840 * 2160 * 16 and st.shadow_rays == 8 * st.hits and st.hits > st.samples // 4 and st.shadow_blocked > 0
    assert st.closest_tests[0] == st.closest_rays * 1_000_000 and st.closest_tests[2] == st.closest_rays * 10_000

    rect = dict(rect_x0=1824, rect_y0=1026, rect_w=192, rect_h=108)
    fast = scene.render(cam, fi.params(**rect), want_ids=True)
    slow = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER, **rect), want_ids=True)
    assert np.array_equal(fast["ids"], slow["ids"]) and np.array_equal(fast["rgb"], slow["rgb"])
    for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(fast["stats"], f) == getattr(slow["stats"], f), f
    assert np.array_equal(fast["rgb"], whole["rgb"][1026:1026 + 108, 1824:1824 + 192])
    kinds = fast["ids"][fast["ids"] < 0xFFFFFFFE] >> 30
    assert (kinds == 0).any() and (kinds == 2).any()          # triangles and ellipsoids are both hit in the crop

    small = dict(rect_x0=1908, rect_y0=1072, rect_w=24, rect_h=16)
    g = scene.render(cam, fi.params(**small), want_ids=True)
    o = checkers.cpu_render(oracle, "rto_", d, cam, fi.params(**small), want_ids=True)
    assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["rgb"], o["rgb"])
    for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(g["stats"], f) == getattr(o["stats"], f), f
    if checkers.have_ref():
        r = checkers.cpu_render(checkers.load_ref(), "ref_", d, cam, fi.params(**small))
        assert np.array_equal(g["rgb"], r["rgb"])
    scene.close()
