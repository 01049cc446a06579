"""Edge cases of the render path, GPU against the CPU oracle (image, ids, ray counters), each in the
default mode, with every kind walked through its box hierarchy, and walked the way big scenes are.

Empty scenes, no bounces, the deepest recursion the library takes, images and rectangles that are
not whole tiles, degenerate primitives, an eye inside a sphere (distances below the tolerance,
negative ones included: Source/ray_tracing.cpp:202), an eye far outside the scene box (the box
records are rebuilt for it) and back.
"""
import numpy as np
import pytest

import checkers
from raytracer_amd import abi

pytestmark = pytest.mark.gpu

MODES = {"default": 0, "walk_all": abi.RT_FLAG_WALK_ALL, "deep_walk": abi.RT_FLAG_WALK_ALL | abi.RT_FLAG_DEEP_WALK}

HEADER = """size {w} {h}
maxdepth {depth}
output edge.png
camera {cam} 0 1 0 40
ambient .1 .1 .1
point 4 6 8 .7 .7 .7
directional 0 -1 -1 .3 .3 .2
diffuse .4 .3 .2
specular .3 .3 .3
shininess 15
"""

FLOOR = """maxverts 16
vertex -6 -1 -6
vertex 6 -1 -6
vertex 6 -1 6
vertex -6 -1 6
tri 0 2 1
tri 0 3 2
"""

SPHERES = """sphere 0 0 0 1
sphere 1.2 0.3 -0.5 0.8
sphere -1.5 0 0.7 0.6
pushTransform
translate 0.5 1.2 0.4
scale 0.9 0.4 0.6
sphere 0 0 0 1
popTransform
"""


def write(tmp_path, name, text):
    path = tmp_path / (name + ".test")
    path.write_text(text)
    return str(path)


def compare(gpu, oracle, path, rect=None, cam_from=None, **overrides):
    fi = gpu.FileInfo(path)
    cam = (cam_from or fi).camera()
    scene = gpu.Scene(fi.scene_desc)
    out = None
    for mode, flags in MODES.items():
        prm = fi.params(flags=flags, **overrides)
        if rect:
            prm.rect_x0, prm.rect_y0, prm.rect_w, prm.rect_h = rect
        g = scene.render(cam, prm, want_ids=True)
        if out is None:
            out = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, prm, want_ids=True)
        assert np.array_equal(g["ids"], out["ids"]), mode
        assert np.array_equal(g["rgb"], out["rgb"]), mode
        for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(g["stats"], f) == getattr(out["stats"], f), (mode, f)
    scene.close()
    return out


def test_scene_without_primitives(gpu, oracle, tmp_path):
    o = compare(gpu, oracle, write(tmp_path, "empty", HEADER.format(w=40, h=24, depth=3, cam="0 2 9 0 0 0")))
    assert o["stats"].closest_rays == 0 and not o["rgb"].any()


@pytest.mark.parametrize("depth", [0, 1, 62])
def test_recursion_depths(gpu, oracle, tmp_path, depth):
    text = HEADER.format(w=48, h=32, depth=depth, cam="0 2 9 0 0 0") + FLOOR + SPHERES
    o = compare(gpu, oracle, write(tmp_path, "depth%d" % depth, text))
    if depth == 0:
        assert o["stats"].closest_rays == 0
    else:
        assert o["stats"].hits > 0 and o["stats"].shadow_blocked > 0


@pytest.mark.parametrize("w,h,rect", [(37, 23, None), (1, 1, None), (130, 70, (113, 53, 17, 17)), (64, 64, (15, 16, 33, 1))])
def test_sizes_that_are_not_whole_tiles(gpu, oracle, tmp_path, w, h, rect):
    text = HEADER.format(w=w, h=h, depth=3, cam="0 2 9 0 0 0") + FLOOR + SPHERES
    compare(gpu, oracle, write(tmp_path, "size%dx%d" % (w, h), text), rect=rect)


def test_degenerate_primitives(gpu, oracle, tmp_path):
    text = HEADER.format(w=64, h=40, depth=3, cam="0 2 9 0 0 0") + FLOOR + """vertex 1 1 1
vertex 1 1 1
vertex 2 2 2
vertex 0 0 0
vertex 1 0 0
vertex 2 0 0
tri 4 4 4
tri 4 5 6
tri 7 8 9
sphere 0.3 0.3 2 0
sphere -2 1 1 0.000001
pushTransform
translate -1 0.5 2
scale 0.000001 0.5 0.5
sphere 0 0 0 1
popTransform
""" + SPHERES
    o = compare(gpu, oracle, write(tmp_path, "degenerate", text))
    assert o["stats"].hits > 0


def test_eye_inside_a_sphere(gpu, oracle, tmp_path):
    # every primary ray starts inside the big sphere: its test returns the near root, behind the eye
    text = HEADER.format(w=64, h=40, depth=4, cam="0 0.2 0.5 0 0 -3") + FLOOR + "sphere 0 0 0 3\n" + SPHERES
    o = compare(gpu, oracle, write(tmp_path, "inside", text))
    assert o["stats"].hits > 0


def test_eye_far_outside_the_scene_box_and_back(gpu, oracle, tmp_path):
    body = FLOOR + SPHERES
    near = write(tmp_path, "near", HEADER.format(w=64, h=40, depth=3, cam="0 2 9 0 0 0") + body)
    far = write(tmp_path, "far", HEADER.format(w=64, h=40, depth=3, cam="300 900 2500 0 0 0").replace("0 1 0 40\n", "0 1 0 0.5\n") + body)
    fi_near, fi_far = gpu.FileInfo(near), gpu.FileInfo(far)
    scene = gpu.Scene(fi_near.scene_desc)
    results = []
    for fi in (fi_near, fi_far, fi_near):     # the far eye makes the library rebuild its box records for a larger bound
        cam = fi.camera()
        g = scene.render(cam, fi_near.params(flags=abi.RT_FLAG_WALK_ALL), want_ids=True)
        o = checkers.cpu_render(oracle, "rto_", fi_near.scene_desc, cam, fi_near.params(), want_ids=True)
        assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["rgb"], o["rgb"])
        results.append(g)
    scene.close()
    assert np.array_equal(results[0]["rgb"], results[2]["rgb"]) and results[1]["stats"].hits > 0


def test_coincident_and_collinear_primitives(gpu, oracle, tmp_path):
    # the host ordering finds no extent to split along (all centroids equal), or only one axis (a row of spheres)
    text = HEADER.format(w=48, h=32, depth=3, cam="0 2 9 0 0 0") + FLOOR
    text += "tri 0 2 1\n" * 300 + "sphere 0.5 0.2 0.3 0.7\n" * 200
    text += "".join("sphere %g 1.5 -1 0.05\n" % (-3 + 0.02 * i) for i in range(300))
    text += "pushTransform\ntranslate -1 0.5 1\nscale 0.5 0.3 0.2\n" + "sphere 0 0 0 1\n" * 150 + "popTransform\n"
    o = compare(gpu, oracle, write(tmp_path, "coincident", text))
    assert o["stats"].hits > 0


def test_many_lights_and_triangles_outside_the_staged_range(gpu, oracle, tmp_path):
    """Forty point lights and a directional one (two groups of light bits), 700 small triangles (three tiles of the brute-force
    scans: fewer than the shadow scan's refill interval) and two triangles outside the range the staged form of the test is
    proven for, a huge one and a tiny one (their tiles go through the test as written).  Default mode, the walks, brute force
    with the staged test and with the test as written, all against the oracle."""
    rng = np.random.default_rng(5)
    text = "size 56 36\nmaxdepth 2\noutput edge.png\ncamera 0 2 9 0 0 0 0 1 0 40\nambient .05 .05 .05\n"
    text += "directional 0 -1 -1 .02 .02 .02\n"
    for i in range(40):
        a = 2.0 * np.pi * i / 40
        text += "point %.6f %.6f %.6f .02 .02 .02\n" % (7.0 * np.cos(a), 5.0 + (i % 3), 7.0 * np.sin(a))
    text += "diffuse .4 .3 .2\nspecular .3 .3 .3\nshininess 15\n" + FLOOR.replace("maxverts 16", "maxverts 4000")
    n_vertices = 4
    for _ in range(700):
        c = rng.uniform([-3.0, -0.5, -3.0], [3.0, 2.5, 3.0])
        for _ in range(3):
            v = c + rng.uniform(-0.25, 0.25, 3)
            text += "vertex %.6f %.6f %.6f\n" % (v[0], v[1], v[2])
        text += "tri %d %d %d\n" % (n_vertices, n_vertices + 1, n_vertices + 2)
        n_vertices += 3
    text += "vertex -9000 -1.5 -9000\nvertex 0 -1.5 9000\nvertex 9000 -1.5 -9000\ntri %d %d %d\n" % (n_vertices, n_vertices + 1, n_vertices + 2)
    n_vertices += 3
    text += "vertex 0.5 0.5 0.5\nvertex 0.500001 0.5 0.5\nvertex 0.5 0.500001 0.5\ntri %d %d %d\n" % (n_vertices, n_vertices + 1, n_vertices + 2)
    fi = gpu.FileInfo(write(tmp_path, "many_lights", text))
    assert fi.scene_desc.n_point_lights == 40 and fi.scene_desc.n_triangles == 704
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    o = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, fi.params(), want_ids=True)
    assert o["stats"].shadow_rays == 41 * o["stats"].hits and 0 < o["stats"].shadow_blocked < o["stats"].shadow_rays
    modes = dict(MODES, brute_force=abi.RT_FLAG_NO_FILTER, brute_force_as_written=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_PLAIN_TEST,
                 brute_force_exact_stages=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_EXACT_STAGES)
    for mode, flags in modes.items():
        g = scene.render(cam, fi.params(flags=flags), want_ids=True)
        assert np.array_equal(g["ids"], o["ids"]), mode
        assert np.array_equal(g["rgb"], o["rgb"]), mode
        for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(g["stats"], f) == getattr(o["stats"], f), (mode, f)
    scene.close()
