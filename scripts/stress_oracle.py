#!/usr/bin/env python3
"""Random small scenes: the GPU path (default mode) against the CPU oracle on bytes, ids and ray counters,
or with a third argument `ref` against the unmodified reference (oracle/_ref/librt_ref.so, where it was
built) on bytes: it has no ids or counters to offer, and pixels with a sample on a zero-padding lane
of the last Sphere8 are left out (the reference reads a material past the end of its vector there).
GPU box:  python scripts/stress_oracle.py FIRST_SEED COUNT [ref]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checkers
from raytracer_amd import api, synth


def run(first, count, against_ref=False):
    """Returns the number of scenes that differ."""
    api.init(0)
    oracle = checkers.load_ref() if against_ref else checkers.load_oracle()
    d = tempfile.mkdtemp()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        kw = dict(n_triangles=int(rng.choice([0, 100, 1500])), n_ellipsoids=int(rng.choice([0, 20, 300])), n_spheres=int(rng.choice([0, 9, 300])),
                  n_lights=int(rng.integers(1, 4)), width=int(rng.integers(24, 72)), height=int(rng.integers(16, 48)), maxdepth=int(rng.integers(1, 6)), seed=seed,
                  extent=float(rng.choice([1.0, 10.0, 300.0, 3000.0, 30000.0])), edge=float(rng.choice([0.05, 0.3, 1.5])),
                  ell_scale=(float(rng.choice([0.01, 0.1, 0.3])), float(rng.choice([0.4, 1.2]))), sphere_radius=(float(rng.choice([0.01, 0.1, 0.3])), float(rng.choice([0.4, 1.5]))))
        kw["edge"] *= kw["extent"] / 10.0
        if kw["n_triangles"] + kw["n_ellipsoids"] + kw["n_spheres"] == 0:
            kw["n_triangles"] = 100
        more = np.random.default_rng(seed + 1_000_000)      # (a stream of its own: the parameters above stay what they were per seed)
        if more.random() < 0.4:
            kw["directional"] = tuple(float(v) for v in np.round(more.normal(size=3) + np.array([0.0, -1.0, 0.0]), 3))
        if more.random() < 0.3:
            kw["attenuation"] = (1.0, 0.05, 0.01)
        if more.random() < 0.3:                              # an eye inside the cloud (rays start among the primitives), a wide or a narrow view
            extent = kw.get("extent", 10.0)
            kw["eye"] = tuple(float(v) for v in np.round(more.uniform(-0.8, 0.8, 3) * extent, 3))
            kw["fov"] = float(more.choice([20.0, 70.0, 110.0]))
        if against_ref:
            # whole batches of 8 only: a zero-padding lane can be hit at any bounce (a sphere of radius 0 at the origin is
            # "hit" by a ray that passes within sqrt(eps) of its distance from it), and the reference then shades with
            # whatever lies behind its material vector; the ids the library reports cover the first bounce only
            for key in ("n_triangles", "n_ellipsoids", "n_spheres"):
                kw[key] = (kw[key] + 7) // 8 * 8
        fi = api.FileInfo(synth.ensure_scene_file(directory=d, **kw))
        scene = api.Scene(fi.scene_desc)
        cam = fi.camera()
        g = scene.render(cam, fi.params(), want_ids=True)
        if against_ref:
            o = checkers.cpu_render(oracle, "ref_", fi.scene_desc, cam, fi.params(), threads=16)
            ids = g["ids"]
            kind, index = ids >> 30, ids & 0x3FFFFFFF
            sd = fi.scene_desc
            limit = np.choose(np.minimum(kind, 2), [sd.n_triangles, sd.n_spheres, sd.n_ellipsoids])
            ok = ~((ids < 0xFFFFFFFE) & (index >= limit)).any(axis=2)
            same = np.array_equal(g["rgb"][ok], o["rgb"][ok])
            o["ids"] = ids
        else:
            o = checkers.cpu_render(oracle, "rto_", fi.scene_desc, cam, fi.params(), want_ids=True, threads=16)
            same = np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["rgb"], o["rgb"]) and all(
                getattr(g["stats"], f) == getattr(o["stats"], f) for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"))
        if not same:
            bad += 1
            print("MISMATCH seed %d %s: ids %d pixels %d" % (seed, kw, int((g["ids"] != o["ids"]).sum()), int((g["rgb"] != o["rgb"]).any(axis=2).sum())), flush=True)
        scene.close()
        if (seed - first) % 10 == 9:
            print("seed %d done, %d mismatches so far" % (seed, bad), flush=True)
    print("stress: %d scenes, %d mismatches" % (count, bad))
    return bad


def main():
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2]), len(sys.argv) > 3 and sys.argv[3] == "ref") else 0)


if __name__ == "__main__":
    main()
