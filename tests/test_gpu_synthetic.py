"""Stress of the pre-tests on random triangle soups with rotated ellipsoids, spheres and eight point lights
(raytracer_amd/synth.py, the generator of the benchmark's synthetic scene, at a small size).

* the pre-tested scan must be invisible: same image and ids as RT_FLAG_NO_FILTER (every test
  through the full test), whole frame;
* GPU == oracle on a crop (image, ids, ray counters);
* where the reference build is present, GPU == reference on a larger crop.
"""
import numpy as np
import pytest

import checkers
from test_oracle_golden import padding_hits
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu

CASES = [
    dict(n_triangles=20000, n_ellipsoids=200, width=192, height=108, maxdepth=3, seed=3),
    dict(n_triangles=3000, n_ellipsoids=40, width=160, height=96, maxdepth=5, seed=11, edge=1.5),     # big triangles: weak pre-tests
    dict(n_triangles=50000, n_ellipsoids=0, width=128, height=72, maxdepth=2, seed=5, edge=0.02),     # tiny triangles
    dict(n_triangles=64, n_ellipsoids=3000, width=256, height=144, maxdepth=3, seed=7),                                   # ellipsoid bounds, closest and shadow
    dict(n_triangles=64, n_ellipsoids=2000, width=192, height=108, maxdepth=2, seed=8, ell_scale=(0.01, 0.05)),          # small far ellipsoids: the noise term of their bounds matters
    dict(n_triangles=64, n_ellipsoids=1000, width=192, height=108, maxdepth=2, seed=9, ell_scale=(0.002, 0.02)),         # so small that the bounds give up (keep-all records)
    dict(n_triangles=0, n_ellipsoids=700, width=160, height=96, maxdepth=3, seed=10, ell_scale=(0.3, 1.2)),              # no triangles at all; big overlapping ellipsoids
    dict(n_triangles=64, n_ellipsoids=0, n_spheres=3000, width=256, height=144, maxdepth=3, seed=12),                    # sphere bounds, closest and shadow
    dict(n_triangles=0, n_ellipsoids=0, n_spheres=1500, width=192, height=108, maxdepth=2, seed=13, sphere_radius=(0.002, 0.03)),   # small far spheres: the noise term of their bounds
    dict(n_triangles=0, n_ellipsoids=0, n_spheres=900, width=160, height=96, maxdepth=5, seed=14, sphere_radius=(0.4, 1.5)),        # overlapping spheres: origins inside spheres, distances below the tolerance
    dict(n_triangles=5000, n_ellipsoids=500, n_spheres=1003, width=192, height=108, maxdepth=3, seed=15),                # all three kinds; a last Sphere8 with padding lanes
]


@pytest.mark.parametrize("kw", CASES, ids=["soup20k", "bigtris", "tinytris", "ellipsoids", "small_ellipsoids", "tiny_ellipsoids", "only_ellipsoids",
                                            "spheres", "small_spheres", "big_spheres", "all_kinds"])
def test_pretests_are_invisible_and_match_oracle(gpu, oracle, tmp_path, kw):
    fi = gpu.FileInfo(synth.ensure_scene_file(directory=str(tmp_path), **kw))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    a = scene.render(cam, fi.params(flags=abi.RT_FLAG_WALK_ALL), want_ids=True)     # every kind through its hierarchy
    b = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
    c = scene.render(cam, fi.params(), want_ids=True)                                # small kinds by brute force
    d = scene.render(cam, fi.params(flags=abi.RT_FLAG_WALK_ALL | abi.RT_FLAG_DEEP_WALK), want_ids=True)   # the walk of big scenes: four levels, two rays per lane
    for other in (a, c, d):
        assert np.array_equal(other["ids"], b["ids"])
        assert np.array_equal(other["rgb"], b["rgb"])
        for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(other["stats"], f) == getattr(b["stats"], f), f
    assert a["stats"].hits > 0 and a["stats"].shadow_blocked > 0
    assert a["stats"].full_tests < 0.02 * (sum(a["stats"].closest_tests) + sum(a["stats"].shadow_tests))

    rect = dict(rect_x0=fi.width // 2 - 10, rect_y0=fi.height // 2 - 4, rect_w=20, rect_h=8)
    g = scene.render(cam, fi.params(**rect), want_ids=True)
    o = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, fi.params(**rect), want_ids=True)
    assert np.array_equal(g["ids"], o["ids"])
    assert np.array_equal(g["rgb"], o["rgb"])
    for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert getattr(g["stats"], f) == getattr(o["stats"], f), f

    if checkers.have_ref():
        rect = dict(rect_x0=fi.width // 4, rect_y0=fi.height // 4, rect_w=fi.width // 2, rect_h=fi.height // 2)
        g = scene.render(cam, fi.params(**rect), want_ids=True)
        r = checkers.cpu_render(checkers.load_ref(), "ref_", fi.scene_desc, cam, fi.params(**rect))
        # a ray exactly through the origin hits the zero-padding lanes of the last Sphere8, whose material the reference reads past its vector
        d = fi.scene_desc
        pad = padding_hits(g["ids"], dict(triangles=d.n_triangles, spheres=d.n_spheres, ellipsoids=d.n_ellipsoids))
        ok = ~pad.any(axis=2)
        assert pad.sum() <= 1 and np.array_equal(g["rgb"][ok], r["rgb"][ok])
    scene.close()


def test_long_ray_lists_brute_force_equals_default(gpu, tmp_path):
    """Ray lists long enough for the brute-force shadow scan to work in ranges of its full size (16 x 256 rays per workgroup,
    lanes refilled from the range as rays are blocked, the triangle list walked round and round): 23 M samples on a dense soup, eight lights.
    Whole frame, brute force against the default mode: ids, bytes, ray counters."""
    fi = gpu.FileInfo(synth.ensure_scene_file(directory=str(tmp_path), n_triangles=20000, n_ellipsoids=200, width=1600, height=900, maxdepth=2, seed=21, edge=0.8))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    fast = scene.render(cam, fi.params(), want_ids=True)
    slow = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
    plain = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_PLAIN_TEST), want_ids=True)
    assert fast["stats"].shadow_rays == 8 * fast["stats"].hits and fast["stats"].hits > 3_200_000      # 12 500 blocks of 256 rays per light
    assert 0 < fast["stats"].shadow_blocked < fast["stats"].shadow_rays
    for other in (slow, plain):
        assert np.array_equal(other["ids"], fast["ids"])
        assert np.array_equal(other["rgb"], fast["rgb"])
        for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(other["stats"], f) == getattr(fast["stats"], f), f
    # The brute-force shadow scan takes the hit points in an order that differs from run to run (cells filled by atomics), so
    # which rays share a wave differs too: a result that depended on a ray's wave mates once showed as one blocked ray more in
    # 92 M, in one run out of three.  The blocked count is exact where the 8-bit image can hide one lost light.
    for again in range(6):
        r = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER | (abi.RT_FLAG_EXACT_STAGES if again % 2 else 0)))
        assert r["stats"].shadow_blocked == fast["stats"].shadow_blocked, again
        assert np.array_equal(r["rgb"], fast["rgb"])
    scene.close()
