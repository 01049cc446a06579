#!/usr/bin/env python3
"""Random scenes (raytracer_amd/synth.py with random parameters): the default mode and the forced walks
must give the bytes, ids and ray counters of the brute-force mode (staged triangle test), and so must brute force with the test as written.  GPU box:  python scripts/stress_parity.py FIRST_SEED COUNT"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import abi, api, synth


def run(first, count):
    """Returns the number of (scene, mode) pairs that differ."""
    api.init(0)
    d = tempfile.mkdtemp()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        kw = dict(n_triangles=int(rng.choice([0, 200, 3000, 20000, 60000])), n_ellipsoids=int(rng.choice([0, 30, 800, 4000])),
                  n_spheres=int(rng.choice([0, 9, 700, 3000])), n_lights=int(rng.integers(1, 5)), width=int(rng.integers(40, 200)), height=int(rng.integers(30, 120)),
                  maxdepth=int(rng.integers(1, 6)), seed=seed, extent=float(rng.choice([1.0, 10.0, 300.0, 3000.0, 30000.0])), edge=float(rng.choice([0.01, 0.15, 1.5])),
                  ell_scale=(float(rng.choice([0.002, 0.05, 0.3])), float(rng.choice([0.4, 1.2]))), sphere_radius=(float(rng.choice([0.002, 0.05, 0.3])), float(rng.choice([0.4, 1.5]))))
        kw["edge"] *= kw["extent"] / 10.0
        if kw["n_triangles"] + kw["n_ellipsoids"] + kw["n_spheres"] == 0:
            kw["n_spheres"] = 50
        more = np.random.default_rng(seed + 1_000_000)      # (a stream of its own: the parameters above stay what they were per seed)
        if more.random() < 0.4:
            kw["directional"] = tuple(float(v) for v in np.round(more.normal(size=3) + np.array([0.0, -1.0, 0.0]), 3))
        if more.random() < 0.3:
            kw["attenuation"] = (1.0, 0.05, 0.01)
        if more.random() < 0.3:                              # an eye inside the cloud (rays start among the primitives), a wide or a narrow view
            extent = kw.get("extent", 10.0)
            kw["eye"] = tuple(float(v) for v in np.round(more.uniform(-0.8, 0.8, 3) * extent, 3))
            kw["fov"] = float(more.choice([20.0, 70.0, 110.0]))
        fi = api.FileInfo(synth.ensure_scene_file(directory=d, **kw))
        scene = api.Scene(fi.scene_desc)
        cam = fi.camera()
        ref = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
        for name, flags in (("default", 0), ("walk_all", abi.RT_FLAG_WALK_ALL), ("deep_walk", abi.RT_FLAG_WALK_ALL | abi.RT_FLAG_DEEP_WALK),
                            ("brute force, test as written", abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_PLAIN_TEST),
                            ("brute force, exact stages", abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_EXACT_STAGES)):
            g = scene.render(cam, fi.params(flags=flags), want_ids=True)
            same = np.array_equal(g["ids"], ref["ids"]) and np.array_equal(g["rgb"], ref["rgb"]) and all(
                getattr(g["stats"], f) == getattr(ref["stats"], f) for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"))
            if not same:
                bad += 1
                print("MISMATCH seed %d mode %s %s: ids %d pixels %d blocked %d vs %d" % (
                    seed, name, kw, int((g["ids"] != ref["ids"]).sum()), int((g["rgb"] != ref["rgb"]).any(axis=2).sum()),
                    g["stats"].shadow_blocked, ref["stats"].shadow_blocked), flush=True)
        scene.close()
        if (seed - first) % 10 == 9:
            print("seed %d done, %d mismatches so far" % (seed, bad), flush=True)
    print("stress: %d scenes, %d mismatches" % (count, bad))
    return bad


def main():
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)


if __name__ == "__main__":
    main()
