"""GPU render path (through the C ABI) against the CPU oracle on the same parsed scenes.

Bar: the RGB8 image and the per-sample primary hit ids are bit-identical to the oracle's.
(The task's floor is: ids exact except documented grazing/tie samples, image within 1 LSB on
>= 99.9 % of pixels; the HIP path is written to reproduce every rounding, so the tests ask for
equality.)
"""
import numpy as np
import pytest

import checkers
from conftest import scene_path

pytestmark = pytest.mark.gpu

# (scene, rectangle or None for the whole 640x480 image)
CASES = [
    ("scene1", None),
    ("scene2", None),
    ("scene3", None),
    ("scene4-diffuse", None),
    ("scene4-specular", None),
    ("scene4-emission", None),
    ("scene5", (160, 120, 320, 160)),
    ("scene6", None),
    ("scene7", (304, 208, 32, 16)),
]


def _render_both(gpu, oracle, name, rect, **overrides):
    fi = gpu.FileInfo(scene_path(name))
    cam = fi.camera()
    prm = fi.params(**overrides)
    if rect:
        prm.rect_x0, prm.rect_y0, prm.rect_w, prm.rect_h = rect
    scene = gpu.Scene(fi.scene_desc)
    g = scene.render(cam, prm, want_ids=True)
    o = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, prm, want_ids=True)
    scene.close()
    return g, o


@pytest.mark.parametrize("mode", ["default", "walk_all", "deep_walk"])
@pytest.mark.parametrize("name,rect", CASES, ids=[c[0] for c in CASES])
def test_image_and_ids_match_oracle(gpu, oracle, name, rect, mode):
    # default: kinds of at most 64 slots are scanned by brute force; walk_all: every kind through its box hierarchy;
    # deep_walk: likewise, walked the way big scenes are (from the super nodes, two rays per lane)
    from raytracer_amd import abi
    flags = {"default": 0, "walk_all": abi.RT_FLAG_WALK_ALL, "deep_walk": abi.RT_FLAG_WALK_ALL | abi.RT_FLAG_DEEP_WALK}[mode]
    g, o = _render_both(gpu, oracle, name, rect, flags=flags)
    id_diff = int((g["ids"] != o["ids"]).sum())
    px_diff = int((g["rgb"] != o["rgb"]).any(axis=2).sum())
    max_lsb = int(np.abs(g["rgb"].astype(int) - o["rgb"].astype(int)).max())
    assert id_diff == 0, "%d of %d sample ids differ" % (id_diff, g["ids"].size)
    assert px_diff == 0, "%d pixels differ (max %d LSB)" % (px_diff, max_lsb)
    gs, os_ = g["stats"], o["stats"]
    for field in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(gs, field) == getattr(os_, field), field
    assert list(gs.closest_tests) == list(os_.closest_tests)


@pytest.mark.parametrize("name", ["scene4-specular", "scene6", "scene7"])
def test_pretest_does_not_change_results(gpu, name):
    """The conservative ray-triangle pre-test must be invisible: same image and ids as the
    build that sends every test through the full test (RT_FLAG_NO_FILTER)."""
    from raytracer_amd import abi
    fi = gpu.FileInfo(scene_path(name))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    a = scene.render(cam, fi.params(flags=abi.RT_FLAG_WALK_ALL), want_ids=True)
    b = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
    c = scene.render(cam, fi.params(), want_ids=True)
    scene.close()
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(c["ids"], b["ids"])
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(c["rgb"], b["rgb"])
    assert b["stats"].full_tests == 0
    if fi.scene_desc.n_triangles:
        assert 0 < a["stats"].full_tests


@pytest.mark.parametrize("name", ["scene6", "scene7"])
def test_chunking_does_not_change_results(gpu, name):
    """A frame traced as many small chunks (alternating between the two lanes, arenas reused in
    stream order, last chunk partial) equals the frame traced in the default way."""
    fi = gpu.FileInfo(scene_path(name))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    whole = scene.render(cam, fi.params(), want_ids=True)                      # 1200 tiles: two chunks, one per lane
    for chunk_tiles in (1200, 101, 7 if name == "scene6" else 173):
        part = scene.render(cam, fi.params(chunk_tiles=chunk_tiles), want_ids=True)
        assert np.array_equal(part["rgb"], whole["rgb"]) and np.array_equal(part["ids"], whole["ids"])
        for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(part["stats"], f) == getattr(whole["stats"], f), (chunk_tiles, f)
    scene.close()


@pytest.mark.parametrize("name", ["scene5", "scene6"])   # 1000 spheres / 34 triangles, 3 spheres, an ellipsoid; a directional and a point light each
@pytest.mark.parametrize("factor", [1.0005, 0.9, 3.0])
def test_directional_light_that_is_not_of_unit_length(gpu, factor, name):
    """The parser normalises the light's direction, a caller of the C ABI need not: the pre-tests
    of the shadow scans assume unit length and must step aside (same results as the full tests)."""
    from raytracer_amd import abi
    fi = gpu.FileInfo(scene_path(name))
    assert fi.scene_desc.has_directional_light
    for i in range(3):
        fi.scene_desc.directional_light[i] *= factor
    cam = fi.camera()
    rect = dict(rect_x0=160, rect_y0=120, rect_w=320, rect_h=160)
    scene = gpu.Scene(fi.scene_desc)
    a = scene.render(cam, fi.params(flags=abi.RT_FLAG_WALK_ALL, **rect), want_ids=True)
    b = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER, **rect), want_ids=True)
    scene.close()
    assert np.array_equal(a["ids"], b["ids"])
    assert np.array_equal(a["rgb"], b["rgb"])
    for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(a["stats"], f) == getattr(b["stats"], f), f
    assert a["stats"].shadow_blocked > 0


def test_scene7_large_region_against_the_unmodified_reference(gpu):
    """A quarter of scene7's frame (the region with the model in it) against the reference itself: image and primary ids.
    The oracle, being scalar, is kept to a 32x16 crop above; the reference's AVX path on all host threads does 320x240 in seconds.
    Also the brute-force mode on the same region in its three forms: pre-tests first (the default), the test as written
    (RT_FLAG_PLAIN_TEST) and the test in exact stages (RT_FLAG_EXACT_STAGES)."""
    if not checkers.have_ref():
        pytest.skip("oracle/_ref/librt_ref.so was not built (needs the reference sources)")
    from raytracer_amd import abi
    fi = gpu.FileInfo(scene_path("scene7"))
    cam = fi.camera()
    rect = dict(rect_x0=160, rect_y0=120, rect_w=320, rect_h=240)
    scene = gpu.Scene(fi.scene_desc)
    g = scene.render(cam, fi.params(**rect), want_ids=True)
    staged = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER, **rect), want_ids=True)
    plain = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_PLAIN_TEST, **rect), want_ids=True)
    exact = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_EXACT_STAGES, **rect), want_ids=True)
    scene.close()
    r = checkers.cpu_render(checkers.load_ref(), "ref_", fi.scene_desc, cam, fi.params(**rect), want_ids=True)
    assert (g["ids"] < 0xFFFFFFFE).mean() > 0.2                      # the model covers a good part of the region
    for other in (g, staged, plain, exact):
        assert np.array_equal(other["ids"], r["ids"])
        assert np.array_equal(other["rgb"], r["rgb"])
    for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(g["stats"], f) == getattr(staged["stats"], f) == getattr(plain["stats"], f) == getattr(exact["stats"], f), f
    # the default form of the brute-force scan decides almost every pair before the test as written
    steps = [list(v) for v in staged["stats"].brute_wave_steps]
    assert steps[0][0] > 0 and steps[0][3] < 0.01 * steps[0][0] and steps[1][0] > steps[1][1] > steps[1][2] >= steps[1][3]
