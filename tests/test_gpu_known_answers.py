"""Device routines one at a time (rt_debug_eval) against known answers captured from the unmodified reference.

The same fixtures pin the CPU oracle in test_oracle_known_answers.py; here the GPU forms of fp_pow, fp_min / fp_max,
the primary-ray direction, the scene-box slab test, the Triangle8 and Sphere8 tests (maths.cpp:9-159,
ray_tracing.cpp:460-473, :265-297, :129-165, :169-206) are compared with them bit for bit, so that a compiler or
code change that alters one routine is named by a test instead of showing up as "some pixel differs".
Also: the staged form of the triangle test and the in-range division against the plain forms on random and edge operands.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from raytracer_amd import abi
from test_oracle_known_answers import KA, POW_TABLE, same_bits

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(np.finfo(np.float32).max)


def test_fp_pow(gpu):
    x = np.array([t[0] for t in POW_TABLE], np.float32)
    y = np.array([t[1] for t in POW_TABLE], np.float32)
    out = gpu.debug_eval(abi.RT_DEBUG_FP_POW, x, y, 1)
    assert [int(v) for v in out.view(np.uint32)] == [t[2] for t in POW_TABLE]
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_FP_POW, KA["pow_x"], KA["pow_y"], 1), KA["pow_out"])


def test_fp_min_max_nan_order(gpu):
    mm = KA["mm_in"]
    a, b = np.repeat(mm, len(mm)), np.tile(mm, len(mm))
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_FP_MIN, a, b, 1), KA["min_out"].reshape(-1))
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_FP_MAX, a, b, 1), KA["max_out"].reshape(-1))


@pytest.mark.parametrize("name", ["scene1", "scene6"])
def test_primary_rays(gpu, name):
    parsed = json.load(open(os.path.join(ROOT, "tests", "golden", "parsed.json")))
    cam = np.frombuffer(bytes.fromhex(parsed[name]["camera_desc_hex"]), np.float32)      # eye, i, j, k, half_world: 14 floats
    a = np.concatenate([cam[:14], np.array([320.0, 240.0], np.float32)])
    s = KA["ray_%s_sample" % name]
    b = np.stack([KA["ray_%s_px" % name].astype(np.float32), KA["ray_%s_py" % name].astype(np.float32),
                  (s % 4).astype(np.float32) * 0.25 + 2.0, (s // 4).astype(np.float32) * 0.25 + 2.0], axis=1)
    out = gpu.debug_eval(abi.RT_DEBUG_PRIMARY_RAY, a, b, 3).reshape(-1, 3)
    assert same_bits(out, KA["ray_%s_dir" % name])
    if name == "scene1":   # the two bit patterns tabulated from the reference
        two = gpu.debug_eval(abi.RT_DEBUG_PRIMARY_RAY, a, np.array([[0, 0, 2.0, 2.0], [320, 240, 2.75, 2.75]], np.float32), 3)
        assert [int(v) for v in two.view(np.uint32)] == [0x3E987578, 0x3F305FE8, 0xBF292A70, 0x3F2A5D29, 0x3F06F6D7, 0xBF074404]


def test_scene_box_slab_test(gpu):
    b = np.concatenate([KA["aabb_o"], KA["aabb_d"]], axis=1)
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_SCENE_BOX, KA["aabb_box"], b, 2).reshape(-1, 2), KA["aabb_out"])


def _lanes_of_batches(rec8, words, o, d):
    """(batches, words*8) AoSoA records + one ray per batch -> per-lane items (batches*8, words), rays repeated per lane."""
    items = rec8.reshape(len(rec8), words, 8).transpose(0, 2, 1).reshape(-1, words)
    rays = np.repeat(np.concatenate([o, d], axis=1), 8, axis=0)
    return np.ascontiguousarray(items), np.ascontiguousarray(rays)


@pytest.mark.parametrize("op", [abi.RT_DEBUG_TRIANGLE, abi.RT_DEBUG_TRIANGLE_STAGED])
def test_triangle8_batches(gpu, op):
    items, rays = _lanes_of_batches(KA["tri8"], 9, KA["tri_o"], KA["tri_d"])
    out = gpu.debug_eval(op, items, rays, 1).reshape(-1, 8)
    unsure = np.isnan(out) & ~np.isnan(KA["tri_out"])
    # the fixture holds crossings exactly on edges and vertices; the staged form hands those to the plain form
    assert unsure.sum() <= (64 if op == abi.RT_DEBUG_TRIANGLE_STAGED else 0)
    assert same_bits(np.where(unsure, KA["tri_out"], out), KA["tri_out"])
    assert (out == FLT_MAX).any() and (out != FLT_MAX).any()


def test_sphere8_batches(gpu):
    items, rays = _lanes_of_batches(KA["sph8"], 4, KA["sph_o"], KA["sph_d"])
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_SPHERE, items, rays, 1).reshape(-1, 8), KA["sph_out"])


def _random_pairs(rng, n, scale):
    """Rays and triangles such that a good share of the rays hit: triangles around points along the rays."""
    o = rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = rng.uniform(-0.2, 3.0, (n, 1)).astype(np.float32) * scale
    centre = o + t * d
    e1 = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.001, 1.0, (n, 1)).astype(np.float32) * scale
    e2 = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.001, 1.0, (n, 1)).astype(np.float32) * scale
    w = rng.uniform(-0.3, 1.3, (n, 2)).astype(np.float32)        # barycentric position of the ray's crossing: inside, outside, near the edges
    a = centre - w[:, :1] * e1 - w[:, 1:] * e2
    return np.concatenate([a, e1, e2], axis=1).astype(np.float32), np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0, 1e-3, 1e5])
def test_staged_triangle_test_equals_the_plain_one(gpu, scale):
    """2 M random pairs per scale (hits, near misses, behind, parallel); 1e5 is outside the staged form's range (plain form runs)."""
    rng = np.random.default_rng(int(scale * 1000) % 9973)
    n = 1 << 21
    tri, ray = _random_pairs(rng, n, scale)
    k = n // 16
    tri[:k, 3:6] = tri[:k, 6:9] * rng.uniform(0.5, 2.0, (k, 1)).astype(np.float32)       # degenerate: parallel edges
    ray[k:2 * k, 3:] = tri[k:2 * k, 3:6] / np.linalg.norm(tri[k:2 * k, 3:6], axis=1, keepdims=True)   # rays in the triangle's plane direction
    tri[2 * k:2 * k + 64] = 0.0                                                            # zero padding lanes
    # exact edge crossings: ray through the vertex a, through a + a_to_b, and along an edge's plane
    to_a = tri[3 * k:4 * k, 0:3] - ray[3 * k:4 * k, :3]
    ray[3 * k:4 * k, 3:] = to_a / np.linalg.norm(to_a, axis=1, keepdims=True)
    plain = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE, tri, ray, 1)
    staged = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE_STAGED, tri, ray, 1)
    unsure = np.isnan(staged) & ~np.isnan(plain)
    hits = (plain != FLT_MAX) & ~np.isnan(plain)
    assert scale < 1.0 or hits.sum() > n // 10       # (triangles of size 1e-3 have |d.n| below the tolerance: the reference calls every ray parallel to them)
    assert scale > 100.0 or scale < 0.01 or unsure[4 * k:].mean() < 1e-3              # (large / tiny coordinates are outside the staged form's range)
    # the fall-back is rare on ordinary pairs (the constructed ones above court it)
    print("unsure: constructed %.4f, ordinary %.6f" % (unsure[:4 * k].mean(), unsure[4 * k:].mean()))
    assert same_bits(np.where(unsure, plain, staged), plain)


def test_in_range_division_is_the_ieee_division(gpu):
    """16 M operand pairs over the range tri_head can meet (|den| in [1e-4, 2^28], |num| in [2^-100, 2^41], same sign), mantissa edge cases included."""
    rng = np.random.default_rng(5)
    n = 1 << 24
    sign = np.where(rng.integers(0, 2, n) == 1, -1.0, 1.0).astype(np.float32)
    den = (np.exp2(rng.uniform(np.log2(1e-4), 28.0, n)).astype(np.float32)) * sign
    num = (np.exp2(rng.uniform(-100.0, 41.0, n)).astype(np.float32)) * sign
    # mantissas of all ones / one bit set / just above a power of two, where correction steps are most delicate
    edge = np.array([0x7FFFFF, 0x000000, 0x000001, 0x7FFFFE, 0x400000, 0x3FFFFF, 0x555555, 0x2AAAAA], np.uint32)
    m = 1 << 20
    den_bits = den[:m].view(np.uint32)
    den[:m] = ((den_bits & np.uint32(0xFF800000)) | edge[rng.integers(0, len(edge), m)]).view(np.float32)
    num_bits = num[m:2 * m].view(np.uint32)
    num[m:2 * m] = ((num_bits & np.uint32(0xFF800000)) | edge[rng.integers(0, len(edge), m)]).view(np.float32)
    den = np.where(np.abs(den) < 1e-4, np.float32(1e-4) * sign, den).astype(np.float32)
    want = gpu.debug_eval(abi.RT_DEBUG_DIVIDE, num, den, 1)
    assert same_bits(want, (num.astype(np.float64) / den.astype(np.float64)).astype(np.float32))    # the device's own IEEE division is correctly rounded
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_DIVIDE_IN_RANGE, num, den, 1), want)
