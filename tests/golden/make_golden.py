#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Runs only where oracle/_ref/librt_ref.so exists (it is built from the reference's own sources,
see oracle/Makefile).  Everything written here is data: inputs and the reference's outputs.

    python tests/golden/make_golden.py

Outputs
  tests/golden/render_<scene>.npz : crops of the reference's image and per-sample primary hit ids,
                                    per-sample colours for one crop, SHA-256 of the whole 640x480 image
  tests/golden/parsed.json        : SHA-256 of every array parse_input_file produces, per scene
  tests/golden/known_answers.npz  : random inputs and the reference's outputs for fp_pow, fp_min/max,
                                    ray_direction_through_pixel, the AABB test, Triangle8 / Sphere8 tests
  tests/golden/known_answers_ellipsoid.npz : the same for the ellipsoid routines (closest-hit body, shadow body)

    python tests/golden/make_golden.py ellipsoids     # only the last file
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import checkers  # noqa: E402
from raytracer_amd import abi  # noqa: E402

SCENES = ["scene1", "scene2", "scene3", "scene4-diffuse", "scene4-emission", "scene4-specular", "scene5", "scene6", "scene7"]
# crops (x0, y0, w, h) chosen to cover every primitive kind, shadows and reflections in each scene
CROPS = {
    "scene1": [(288, 208, 64, 48), (0, 0, 32, 32)],
    "scene2": [(280, 200, 80, 64), (100, 300, 48, 32)],
    "scene3": [(260, 180, 96, 64), (96, 240, 48, 48)],
    "scene4-diffuse": [(260, 180, 96, 64)],
    "scene4-emission": [(260, 180, 96, 64)],
    "scene4-specular": [(260, 180, 96, 64), (400, 260, 48, 48)],
    "scene5": [(288, 216, 64, 32), (64, 64, 32, 32)],
    "scene6": [(272, 200, 96, 80), (0, 400, 64, 48), (310, 232, 16, 16)],
    "scene7": [(308, 216, 24, 16), (240, 300, 16, 8)],
}
FULL_HASH = ["scene1", "scene2", "scene3", "scene4-diffuse", "scene4-emission", "scene4-specular", "scene5", "scene6"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ellipsoid_batch(rng, zero_from=8):
    """One Mat3x4x8 pair: M = translate * rotate * scale per lane, inverse via float64; lanes >= zero_from are padding (zeros)."""
    inv8 = np.zeros((3, 4, 8), np.float32)
    fwd8 = np.zeros((3, 4, 8), np.float32)
    centres, radii = np.zeros((8, 3)), np.zeros(8)
    for j in range(min(8, zero_from)):
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 2 * np.pi)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        sc = np.exp(rng.uniform(np.log(0.05), np.log(3.0), 3))
        c = rng.uniform(-4, 4, 3)
        M = np.eye(4)
        M[:3, :3] = R @ np.diag(sc)
        M[:3, 3] = c
        fwd8[:, :, j] = M[:3].astype(np.float32)
        inv8[:, :, j] = np.linalg.inv(M)[:3].astype(np.float32)
        centres[j], radii[j] = c, sc.max()
    return inv8, fwd8, centres, radii


def ellipsoid_known_answers(ref):
    """Inputs and the reference's outputs for the ellipsoid routines -> known_answers_ellipsoid.npz."""
    rng = np.random.default_rng(20240808)
    nb = 384
    ka = {k: [] for k in ("inv8", "fwd8", "o", "d", "o2", "d2", "absent", "t2", "dist")}
    checked_ids = 0
    for b in range(nb):
        inv8, fwd8, centres, radii = ellipsoid_batch(rng, zero_from=5 if b % 4 == 0 else 8)
        lane = b % (5 if b % 4 == 0 else 8)
        if b % 3 == 0:   # origin inside the ellipsoid
            o = (centres[lane] + rng.uniform(-0.2, 0.2, 3) * 0.05).astype(np.float32)
        else:
            o = rng.uniform(-8, 8, 3).astype(np.float32)
        # aim at the chosen lane: through it, grazing it (offset close to its largest radius), past it
        offset = rng.standard_normal(3)
        offset *= radii[lane] * (0.3, 0.97, 1.03, 1.6)[b % 4] / np.linalg.norm(offset)
        d = (centres[lane] + offset - o)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if b % 7 == 0:
            d = -d       # the ellipsoid behind the ray
        out = [np.zeros(24, np.float32), np.zeros(24, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32)]
        ref.ref_ellipsoid8(C.byref(checkers.vec3(o)), C.byref(checkers.vec3(d)), inv8.ctypes.data, fwd8.ctypes.data, *[x.ctypes.data for x in out])
        # the closest-hit body is inline in intersect(Ray, Scene, int): check the composed distances against that function's choice
        n_real = 5 if b % 4 == 0 else 8
        mats = np.zeros((n_real, 10), np.float32)
        mats[:, 8] = np.arange(1, n_real + 1)       # emission blue = index + 1
        desc = abi.SceneDesc()
        desc.ellipsoid8_inverse_transforms = inv8.ctypes.data_as(abi.c_float_p)
        desc.ellipsoid8_transforms = fwd8.ctypes.data_as(abi.c_float_p)
        desc.ellipsoid_materials = mats.ctypes.data_as(abi.c_float_p)
        desc.n_ellipsoid8, desc.n_ellipsoids = 1, n_real
        col = (C.c_float * 3)()
        ref.ref_trace(C.byref(desc), C.byref(checkers.vec3(o)), C.byref(checkers.vec3(d)), 1, C.byref(col))
        dist = out[4]
        want = 0
        if (dist < np.float32(3.4028234663852886e38)).any():
            want = int(np.argmin(dist)) + 1     # first lane among equal distances: the reference scans lanes with a strict <
        if want <= n_real:                      # (a hit on a padding lane reads a material past the vector)
            assert int(col[2]) == want, (b, list(col), dist)
            checked_ids += 1
        for k, v in zip(("inv8", "fwd8", "o", "d", "o2", "d2", "absent", "t2", "dist"), (inv8.reshape(-1), fwd8.reshape(-1), o, d, *out)):
            ka[k].append(v.copy())
    ka = {k: np.array(v, np.float32) for k, v in ka.items()}
    print("ellipsoid closest-hit bodies: %d batches, %d lanes hit, %d absent, choice checked against intersect() on %d batches"
          % (nb, int((ka["dist"] < 3e38).sum()), int(ka["absent"].sum()), checked_ids))

    # shadow body: path_is_blocked on one ellipsoid at a time (the other lanes of the batch are padding)
    n = 3072
    sh = {k: [] for k in ("inv", "start", "light", "point", "dir", "blocked")}
    for i in range(n):
        inv8, fwd8, centres, radii = ellipsoid_batch(rng, zero_from=1)
        point = i % 3 != 0
        start = rng.uniform(-8, 8, 3).astype(np.float32)
        offset = rng.standard_normal(3)
        offset *= radii[0] * (0.3, 0.97, 1.03, 1.6)[i % 4] / np.linalg.norm(offset)
        through = centres[0] + offset
        if point:
            # the light beyond the ellipsoid, in front of it, or inside it
            light = (start + (through - start) * (2.0, 0.5, 1.0, 3.0, 0.97)[i % 5]).astype(np.float32)
        else:
            light = (start - through)
            light = (light / np.linalg.norm(light)).astype(np.float32)      # direction toward which the light travels = -ray direction
            if i % 8 == 0:
                light *= np.float32(0.5)                                     # not of unit length
        dir_out = (C.c_float * 3)()
        blocked = ref.ref_ellipsoid8_blocks(C.byref(checkers.vec3(start)), C.byref(checkers.vec3(light)), 1 if point else 0, inv8.ctypes.data, C.byref(dir_out))
        for k, v in zip(("inv", "start", "light", "point", "dir", "blocked"),
                        (inv8[:, :, 0].reshape(-1), start, light, np.float32(1.0 if point else 0.0), np.array(list(dir_out), np.float32), np.float32(blocked))):
            sh[k].append(v)
    for k, v in sh.items():
        ka["shadow_" + k] = np.array(v, np.float32)
    print("ellipsoid shadow bodies: %d, %d blocked" % (n, int(ka["shadow_blocked"].sum())))
    np.savez_compressed(os.path.join(HERE, "known_answers_ellipsoid.npz"), **ka)


def main():
    ref = checkers.load_ref()
    if sys.argv[1:] == ["ellipsoids"]:
        ellipsoid_known_answers(ref)
        return
    parsed = {}
    for name in SCENES:
        path = os.path.join(HERE, "scenes", name + ".test")
        err = C.c_char_p()
        h = ref.ref_parse(path.encode(), C.byref(err))
        assert h, err.value
        desc = ref.ref_parse_scene(h).contents
        cam11 = (C.c_float * 11)()
        ref.ref_parse_camera(h, C.byref(cam11))
        cam = abi.CameraDesc()
        ref.ref_camera_setup(C.byref(cam11), C.byref(cam))
        W, H, depth = ref.ref_parse_width(h), ref.ref_parse_height(h), ref.ref_parse_max_depth(h)
        arrays = checkers.desc_arrays(desc)
        parsed[name] = {
            "width": W, "height": H, "max_recursion_depth": depth, "output": ref.ref_parse_filename(h).decode(),
            "camera11_hex": bytes(cam11).hex(), "camera_desc_hex": bytes(cam).hex(),
            "arrays_sha256": {k: sha(v) for k, v in arrays.items()},
            "counts": {"triangles": int(desc.n_triangles), "spheres": int(desc.n_spheres), "ellipsoids": int(desc.n_ellipsoids),
                       "point_lights": int(desc.n_point_lights), "directional": int(desc.has_directional_light)},
        }
        out = {"crops": np.array(CROPS[name], np.int32)}
        for i, (x0, y0, w, hh) in enumerate(CROPS[name]):
            prm = abi.RenderParams(width=W, height=H, max_bounce_count=depth, rect_x0=x0, rect_y0=y0, rect_w=w, rect_h=hh)
            r = checkers.cpu_render(ref, "ref_", desc, cam, prm, want_ids=True, want_colours=(i == 0), threads=8)
            out["rgb_%d" % i] = r["rgb"]
            out["ids_%d" % i] = r["ids"]
            if i == 0:
                out["colours_0"] = r["colours"]
        if name in FULL_HASH:
            prm = abi.RenderParams(width=W, height=H, max_bounce_count=depth)
            r = checkers.cpu_render(ref, "ref_", desc, cam, prm, threads=8)
            out["full_sha256"] = np.frombuffer(bytes.fromhex(sha(r["rgb"])), np.uint8)
        np.savez_compressed(os.path.join(HERE, "render_%s.npz" % name), **out)
        print(name, "done")
        ref.ref_parse_free(h)
    with open(os.path.join(HERE, "parsed.json"), "w") as f:
        json.dump(parsed, f, indent=1, sort_keys=True)

    # ---- known answers -------------------------------------------------------------------------
    rng = np.random.default_rng(20240807)
    ka = {}
    n = 4096
    x = np.concatenate([rng.random(n - 8, dtype=np.float32), np.array([0, 1, 1e-30, 1e-3, 0.5, 0.25, 0.999, 0.9], np.float32)])
    y = np.concatenate([(rng.random(n - 8, dtype=np.float32) * 64).astype(np.float32), np.array([20, 20, 1, 1, 0, 20, 20, 40], np.float32)])
    ka["pow_x"], ka["pow_y"] = x, y
    ka["pow_out"] = np.array([ref.ref_fp_pow(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    mm = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 1e-40], np.float32)
    ka["mm_in"] = mm
    ka["min_out"] = np.array([[ref.ref_fp_min(float(a), float(b)) for b in mm] for a in mm], np.float32)
    ka["max_out"] = np.array([[ref.ref_fp_max(float(a), float(b)) for b in mm] for a in mm], np.float32)

    # primary rays + AABB for scene1's and scene6's cameras
    for name in ("scene1", "scene6"):
        cam = abi.CameraDesc.from_buffer_copy(bytes.fromhex(parsed[name]["camera_desc_hex"]))
        px = rng.integers(0, 640, 256).astype(np.uint32)
        py = rng.integers(0, 480, 256).astype(np.uint32)
        s = rng.integers(0, 16, 256)
        dirs = np.zeros((256, 3), np.float32)
        for i in range(256):
            out3 = (C.c_float * 3)()
            ref.ref_ray_direction_through_pixel(int(px[i]), int(py[i]), float(s[i] % 4) * 0.25 + 2.0, float(s[i] // 4) * 0.25 + 2.0,
                                                C.byref(cam), 320.0, 240.0, C.byref(out3))
            dirs[i] = list(out3)
        ka["ray_%s_px" % name], ka["ray_%s_py" % name], ka["ray_%s_sample" % name], ka["ray_%s_dir" % name] = px, py, s.astype(np.uint32), dirs
    bb = np.array([-1.5, 2.0, -0.25, 0.75, -3.0, 0.0], np.float32)
    o = (rng.random((512, 3), dtype=np.float32) * 8 - 4).astype(np.float32)
    d = rng.standard_normal((512, 3)).astype(np.float32)
    d[:16, 0] = 0.0   # axis-parallel rays: 1/0 = inf and 0*inf = NaN exercise the minss/maxss operand order
    d[16:32, 1] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d = d.astype(np.float32)
    aabb = np.zeros((512, 2), np.float32)
    for i in range(512):
        out2 = (C.c_float * 2)()
        ref.ref_intersect_aabb(C.byref(checkers.vec3(o[i])), C.byref(checkers.vec3(d[i])), C.byref((C.c_float * 6)(*bb)), C.byref(out2))
        aabb[i] = list(out2)
    ka["aabb_box"], ka["aabb_o"], ka["aabb_d"], ka["aabb_out"] = bb, o, d, aabb

    # Triangle8 / Sphere8: random batches, rays aimed at them (hits, edge hits, misses, padding lanes)
    nb = 192
    tri8 = np.zeros((nb, 72), np.float32)
    tri_o = np.zeros((nb, 3), np.float32)
    tri_d = np.zeros((nb, 3), np.float32)
    tri_out = np.zeros((nb, 8), np.float32)
    for b in range(nb):
        a = rng.random((8, 3), dtype=np.float32) * 4 - 2
        ab = rng.random((8, 3), dtype=np.float32) * 2 - 1
        ac = rng.random((8, 3), dtype=np.float32) * 2 - 1
        if b % 4 == 0:
            a[5:], ab[5:], ac[5:] = 0, 0, 0   # zero padding lanes
        t8 = np.concatenate([a.T.reshape(-1), ab.T.reshape(-1), ac.T.reshape(-1)]).astype(np.float32)
        lane = b % 8
        u, v = rng.random(2)
        if b % 3 == 0:
            v = 1.0 - u   # on the a_to_b / a_to_c edge: alpha + beta == 1 up to rounding
        target = a[lane] + np.float32(u) * ab[lane] + np.float32(v) * ac[lane] * (1.0 if b % 5 else 1.3)
        oo = (rng.random(3) * 10 - 5).astype(np.float32)
        dd = (target - oo).astype(np.float32)
        dd = (dd / np.linalg.norm(dd)).astype(np.float32)
        out8 = (C.c_float * 8)()
        ref.ref_intersect_triangle8(C.byref(checkers.vec3(oo)), C.byref(checkers.vec3(dd)), t8.ctypes.data, C.byref(out8))
        tri8[b], tri_o[b], tri_d[b], tri_out[b] = t8, oo, dd, list(out8)
    ka["tri8"], ka["tri_o"], ka["tri_d"], ka["tri_out"] = tri8, tri_o, tri_d, tri_out

    sph8 = np.zeros((nb, 32), np.float32)
    sph_o = np.zeros((nb, 3), np.float32)
    sph_d = np.zeros((nb, 3), np.float32)
    sph_out = np.zeros((nb, 8), np.float32)
    for b in range(nb):
        c = rng.random((8, 3), dtype=np.float32) * 6 - 3
        r = rng.random(8, dtype=np.float32) * 1.5 + 0.05
        if b % 4 == 0:
            c[6:], r[6:] = 0, 0
        s8 = np.concatenate([c.T.reshape(-1), r]).astype(np.float32)
        lane = b % 8
        if b % 3 == 0:
            oo = (c[lane] + (rng.random(3) - 0.5) * r[lane] * 0.8).astype(np.float32)   # origin inside the sphere
        else:
            oo = (rng.random(3) * 12 - 6).astype(np.float32)
        target = c[lane] + (rng.random(3) * 2 - 1) * r[lane] * (0.7 if b % 5 else 1.2)
        dd = (target - oo).astype(np.float32)
        dd = (dd / np.linalg.norm(dd)).astype(np.float32)
        out8 = (C.c_float * 8)()
        ref.ref_intersect_sphere8(C.byref(checkers.vec3(oo)), C.byref(checkers.vec3(dd)), s8.ctypes.data, C.byref(out8))
        sph8[b], sph_o[b], sph_d[b], sph_out[b] = s8, oo, dd, list(out8)
    ka["sph8"], ka["sph_o"], ka["sph_d"], ka["sph_out"] = sph8, sph_o, sph_d, sph_out
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **ka)
    print("known answers done")
    ellipsoid_known_answers(ref)


if __name__ == "__main__":
    main()
