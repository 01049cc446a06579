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


# ---- ellipsoid routines (ray_tracing.cpp:105-126, :208-263, :336-344, :376-401, :432-454, :556-579) ------------------------
from test_oracle_known_answers import KA_ELL  # noqa: E402


def _ellipsoid_lanes():
    """(batches, 96) Mat3x4x8 pairs + one ray per batch -> per-lane records (batches*8, 24) and rays repeated per lane."""
    k = KA_ELL
    n = len(k["o"])
    inv = k["inv8"].reshape(n, 12, 8).transpose(0, 2, 1).reshape(-1, 12)
    fwd = k["fwd8"].reshape(n, 12, 8).transpose(0, 2, 1).reshape(-1, 12)
    rays = np.repeat(np.concatenate([k["o"], k["d"]], axis=1), 8, axis=0)
    return np.ascontiguousarray(inv), np.ascontiguousarray(fwd), np.ascontiguousarray(rays)


def _lanes(x):      # (batches, 3 * 8) stored as x[8] y[8] z[8] -> (batches * 8, 3)
    return x.reshape(len(x), 3, 8).transpose(0, 2, 1).reshape(-1, 3)


def test_ellipsoid_routines(gpu):
    """Mat3x4AVX * Vec3AVX, transform_direction, ellipsoid_is_absent and intersect_with_unit_sphere, one by one."""
    inv, fwd, rays = _ellipsoid_lanes()
    out = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_PARTS, inv, rays).reshape(-1, 8)
    assert same_bits(out[:, 0:3], _lanes(KA_ELL["o2"]))
    assert same_bits(out[:, 3:6], _lanes(KA_ELL["d2"]))
    assert np.array_equal(out[:, 6], KA_ELL["absent"].reshape(-1))
    assert same_bits(out[:, 7], KA_ELL["t2"].reshape(-1))


def test_ellipsoid8_batches(gpu):
    """The closest-hit body as written and in stages against the reference's distances."""
    inv, fwd, rays = _ellipsoid_lanes()
    rec = np.concatenate([inv, fwd], axis=1)
    want = KA_ELL["dist"].reshape(-1)
    assert same_bits(gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID, rec, rays), want)
    staged = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_STAGED, rec, rays).reshape(-1, 2)
    hit = want < FLT_MAX
    assert same_bits(staged[hit, 0], want[hit]) and (staged[hit, 1] == 2).all()
    assert (staged[~hit, 0] == FLT_MAX).all()         # (misses of a regular record as written are +inf or NaN: not below FLT_MAX either)
    assert set(np.unique(staged[:, 1])) == {0.0, 1.0, 2.0}


def test_ellipsoid_shadow_bodies(gpu):
    k = KA_ELL
    a = np.concatenate([k["shadow_inv"], k["shadow_light"], k["shadow_point"][:, None]], axis=1)
    b = np.concatenate([k["shadow_start"], k["shadow_dir"]], axis=1)
    assert np.array_equal(gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_SHADOW, a, b), k["shadow_blocked"])
    staged = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_SHADOW_STAGED, a, b).reshape(-1, 2)
    assert np.array_equal(staged[:, 0], k["shadow_blocked"])
    assert (staged[:, 1] == 0).sum() > 100 and (staged[:, 1] == 1).sum() > 100


def _random_ellipsoid_pairs(rng, n, extent, scale_lo, scale_hi, graze):
    """Ellipsoid records (inverse rows, rows) and rays aimed near them; `graze`: share of rays aimed at the silhouette."""
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(0, 2 * np.pi, n)
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -axis[:, 2], axis[:, 1], axis[:, 2], -axis[:, 0], -axis[:, 1], axis[:, 0]
    R = np.eye(3)[None] + np.sin(ang)[:, None, None] * K + (1 - np.cos(ang))[:, None, None] * (K @ K)
    sc = np.exp(rng.uniform(np.log(scale_lo), np.log(scale_hi), (n, 3)))
    c = rng.uniform(-extent, extent, (n, 3))
    A = R * sc[:, None, :]                                   # R @ diag(sc)
    Ainv = (R / sc[:, None, :]).transpose(0, 2, 1)           # diag(1/sc) @ R^T
    t_inv = -np.einsum("nij,nj->ni", Ainv, c)
    rec = np.concatenate([np.concatenate([Ainv, t_inv[:, :, None]], axis=2).reshape(n, 12),
                          np.concatenate([A, c[:, :, None]], axis=2).reshape(n, 12)], axis=1).astype(np.float32)
    o = rng.uniform(-extent, extent, (n, 3))
    inside = rng.random(n) < 0.05
    o[inside] = c[inside] + rng.uniform(-0.3, 0.3, (inside.sum(), 3)) * sc[inside].min(axis=1, keepdims=True)
    # a point on (or near) the surface, seen from o: unit vector u in ellipsoid space, scaled around 1 for grazing rays
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    radial = np.where(rng.random(n) < graze, 1.0 + rng.normal(size=n) * 1e-4, rng.uniform(0.0, 2.5, n))
    target = c + np.einsum("nij,nj->ni", A, u * radial[:, None])
    # grazing: direction tangent to the surface at that point = orthogonal (in ellipsoid space) to u
    tangent = np.cross(u, rng.normal(size=(n, 3)))
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    far = target - np.einsum("nij,nj->ni", A, tangent) * rng.uniform(1.0, 30.0, (n, 1))
    use_tangent = rng.random(n) < graze
    o = np.where(use_tangent[:, None], far, o)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[rng.random(n) < 0.1] *= -1.0
    return rec, np.concatenate([o, d], axis=1).astype(np.float32), c.astype(np.float32)


@pytest.mark.parametrize("case", ["ordinary", "small_far", "grazing", "thin", "huge", "out_of_range"])
def test_staged_ellipsoid_tests_equal_the_plain_ones(gpu, case):
    """2 M random pairs per case: closest-hit body and shadow body, staged against as written.

    A stage may call a pair decided only when the form as written returns nothing below FLT_MAX (closest hit) or "not blocked";
    where the staged form returns a distance it must be the same bits."""
    extent, lo, hi, graze = {"ordinary": (10, 0.05, 3.0, 0.2), "small_far": (300, 0.01, 0.3, 0.3), "grazing": (20, 0.05, 2.0, 1.0),
                             "thin": (10, 0.001, 3.0, 0.3), "huge": (3000, 5.0, 500.0, 0.3), "out_of_range": (1e5, 0.1, 10.0, 0.2)}[case]
    rng = np.random.default_rng(sum(map(ord, case)))
    n = 1 << 21
    rec, ray, centre = _random_ellipsoid_pairs(rng, n, extent, lo, hi, graze)
    rec[:64] = 0.0                                           # padding lanes
    rec[64:128, 0:4] = 0.0                                   # singular inverse transforms: not regular
    plain = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID, rec, ray)
    staged = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_STAGED, rec, ray).reshape(-1, 2)
    stage = staged[:, 1]
    hit = plain < FLT_MAX                                    # (false for NaN)
    assert same_bits(staged[hit, 0], plain[hit])
    assert (stage[hit] >= 2).all()
    decided = stage < 2
    assert not (plain[decided] < FLT_MAX).any() and (staged[decided, 0] == FLT_MAX).all()
    assert same_bits(staged[stage == 3, 0], plain[stage == 3])
    share = [float((stage == s).mean()) for s in range(4)]
    print("closest hit: stage 0 %.3f, stage 1 %.3f, tail %.3f, as written %.3f; hits %.3f" % (*share, hit.mean()))
    if case == "out_of_range":
        assert share[3] > 0.9
    else:
        # (huge: some of the origins the construction places lie beyond the 4096 the staged form is proven for)
        assert share[3] < (0.1 if case == "huge" else 0.001) and hit.mean() > 0.05 and share[0] > 0.05

    # shadow body: point lights behind, before and inside the ellipsoid, and directional lights
    point = (rng.random(n) < 0.7).astype(np.float32)
    along = rng.choice([0.5, 0.97, 1.0, 1.03, 2.0, 5.0], n)
    reach = np.linalg.norm(centre - ray[:, :3], axis=1)
    light_pos = ray[:, :3] + ray[:, 3:] * (reach * along)[:, None]
    light = np.where(point[:, None] > 0, light_pos, -ray[:, 3:]).astype(np.float32)
    a = np.concatenate([rec[:, :12], light, point[:, None]], axis=1).astype(np.float32)
    # the direction as the reference derives it from the light (normalised difference) is not bit-equal to ray[:, 3:]; any direction is a valid input
    plain_s = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_SHADOW, a, ray)
    staged_s = gpu.debug_eval(abi.RT_DEBUG_ELLIPSOID_SHADOW_STAGED, a, ray).reshape(-1, 2)
    assert np.array_equal(staged_s[:, 0], plain_s)
    print("shadow: stage 0 %.3f, blocked %.3f" % ((staged_s[:, 1] == 0).mean(), plain_s.mean()))
    assert case == "out_of_range" or (0.02 < plain_s.mean() < 0.9 and (staged_s[:, 1] == 0).mean() > 0.05)


@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0, 1e-3, 1e5])
def test_pretest_form_of_the_triangle_test_equals_the_plain_one(gpu, scale):
    """The default form of the brute-force scans (sign test, two conservative pre-tests, then the test as written) on the pairs of
    test_staged_triangle_test_equals_the_plain_one: a step may call a pair decided only where the test as written returns no distance
    (FLT_MAX, or NaN); where the form returns a distance it is the test's own.  Likewise the form for rays of one origin."""
    rng = np.random.default_rng(int(scale * 1000) % 9973 + 1)
    n = 1 << 21
    tri, ray = _random_pairs(rng, n, scale)
    k = n // 16
    tri[:k, 3:6] = tri[:k, 6:9] * rng.uniform(0.5, 2.0, (k, 1)).astype(np.float32)       # degenerate: parallel edges
    ray[k:2 * k, 3:] = tri[k:2 * k, 3:6] / np.linalg.norm(tri[k:2 * k, 3:6], axis=1, keepdims=True)   # rays in the triangle's plane direction
    tri[2 * k:2 * k + 64] = 0.0                                                            # zero padding lanes
    to_a = tri[3 * k:4 * k, 0:3] - ray[3 * k:4 * k, :3]                                   # rays through the vertex a
    ray[3 * k:4 * k, 3:] = to_a / np.linalg.norm(to_a, axis=1, keepdims=True)
    ray[4 * k:4 * k + 32, 3] = np.nan                                                      # a NaN and an infinity in a ray: never a hit either way
    ray[4 * k + 32:4 * k + 64, 0] = np.inf
    plain = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE, tri, ray)
    form = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE_PRETEST, tri, ray).reshape(-1, 3)
    hit = plain < FLT_MAX                                                                  # (false for NaN)
    step = form[:, 1]
    assert same_bits(form[step == 3, 0], plain[step == 3])                                 # the test as written ran: its result, bit for bit
    assert (step[hit] == 3).all()                                                          # no hit was decided away ...
    assert form[hit, 2].all()                                                              # ... by the form for rays of one origin either
    assert (form[step < 3, 0] == FLT_MAX).all() and not hit[step < 3].any()
    share = [float((step == q).mean()) for q in range(4)]
    print("decided by: sign test %.3f, first pre-test %.3f, second %.3f, as written %.3f; hits %.3f" % (*share, hit.mean()))
    if scale in (1.0, 30.0, 3000.0):
        # the constructed pairs aim at the triangles: most reach the test as written; of those it rejects, the pre-tests must have taken a good part
        assert share[0] > 0.02 and share[1] + share[2] > 0.1 and share[3] < hit.mean() + 0.12


def _pairs_near_edges(rng, n, scale):
    """Rays whose crossing of the triangle's plane lies within 1e-8 .. 1e-2 (barycentric) of an edge or a vertex, on either side,
    a third of them at grazing angles (1e-6 .. 1e-1 off the plane): where a conservative pre-test is most likely to err."""
    e1 = rng.normal(size=(n, 3)) * rng.uniform(0.01, 1.0, (n, 1)) * scale
    e2 = rng.normal(size=(n, 3)) * rng.uniform(0.01, 1.0, (n, 1)) * scale
    a = rng.uniform(-scale, scale, (n, 3))
    delta = 10.0 ** rng.uniform(-8.0, -2.0, n) * rng.choice([-1.0, 1.0], n)
    along = rng.uniform(-0.1, 1.1, n)
    which = rng.integers(0, 5, n)
    u = np.where(which == 0, delta, np.where(which == 1, along, np.where(which == 2, along, np.where(which == 3, delta, 1.0 + delta))))
    v = np.where(which == 0, along, np.where(which == 1, delta, np.where(which == 2, 1.0 - along + delta, np.where(which == 3, delta * rng.uniform(-1, 1, n), -delta))))
    p = a + u[:, None] * e1 + v[:, None] * e2                       # the crossing: near edge u = 0, v = 0, u + v = 1, vertex a, vertex b
    normal = np.cross(e1, e2)
    normal /= np.maximum(np.linalg.norm(normal, axis=1, keepdims=True), 1e-30)
    d = rng.normal(size=(n, 3))
    grazing = rng.random(n) < 1.0 / 3.0
    in_plane = e1 * rng.uniform(-1, 1, (n, 1)) + e2 * rng.uniform(-1, 1, (n, 1))
    in_plane /= np.maximum(np.linalg.norm(in_plane, axis=1, keepdims=True), 1e-30)
    tilt = 10.0 ** rng.uniform(-6.0, -1.0, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    d = np.where(grazing[:, None], in_plane + tilt * normal, d)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = 10.0 ** rng.uniform(-3.0, 1.0, (n, 1)) * scale
    o = p - t * d
    return np.concatenate([a, e1, e2], axis=1).astype(np.float32), np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0])
def test_pretests_keep_every_hit_near_edges_and_at_grazing_angles(gpu, scale):
    """4 M pairs per scale aimed at the edges and vertices of their triangles, a third of them grazing: the sign test and the two
    conservative pre-tests (for rays with origins of their own and for rays of one origin) may decide a pair only where the test as
    written returns no distance, and the staged form of the test returns the test's own bits or declares itself unsure."""
    rng = np.random.default_rng(int(scale) + 77)
    n = 1 << 22
    tri, ray = _pairs_near_edges(rng, n, scale)
    plain = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE, tri, ray)
    form = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE_PRETEST, tri, ray).reshape(-1, 3)
    staged = gpu.debug_eval(abi.RT_DEBUG_TRIANGLE_STAGED, tri, ray)
    hit = plain < FLT_MAX
    step = form[:, 1]
    assert 0.2 < hit.mean() < 0.8                                   # the pairs straddle the edges
    assert (step[hit] == 3).all() and form[hit, 2].all()
    assert same_bits(form[step == 3, 0], plain[step == 3])
    assert (form[step < 3, 0] == FLT_MAX).all()
    unsure = np.isnan(staged) & ~np.isnan(plain)
    assert same_bits(np.where(unsure, plain, staged), plain)
    print("hits %.3f; decided before the test as written %.3f; staged form unsure %.4f" % (hit.mean(), (step < 3).mean(), unsure.mean()))
