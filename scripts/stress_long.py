#!/usr/bin/env python3
"""Random scenes with LONG ray lists (millions of hit points, up to eight lights): the brute-force mode in its default form
must give the bytes, ids and ray counters of the default mode.  What only long launches reach is exercised here: the shadow scan
with two rays per lane and lanes refilled from ranges of 16 capacities, reflected rays taken in the order of their directions,
hit points taken in the order of their cells.  GPU box:  python scripts/stress_long.py FIRST_SEED COUNT"""
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
        w, h = [(1280, 720), (1600, 900), (1920, 1080)][int(rng.integers(0, 3))]
        kw = dict(n_triangles=int(rng.choice([2048, 5000, 20000, 40000])), n_ellipsoids=int(rng.choice([0, 100, 1000])), n_spheres=int(rng.choice([0, 0, 50])),
                  n_lights=int(rng.choice([1, 2, 5, 8])), width=w, height=h, maxdepth=int(rng.integers(1, 4)), seed=seed,
                  edge=float(rng.choice([0.15, 0.8, 2.0])))
        more = np.random.default_rng(seed + 1_000_000)      # (a stream of its own: the parameters above stay what they were per seed)
        if more.random() < 0.4:
            kw["directional"] = tuple(float(v) for v in np.round(more.normal(size=3) + np.array([0.0, -1.0, 0.0]), 3))
        if more.random() < 0.3:
            kw["attenuation"] = (1.0, 0.05, 0.01)
        if more.random() < 0.3:                              # an eye inside the cloud (rays start among the primitives), a wide or a narrow view
            extent = kw.get("extent", 10.0)
            kw["eye"] = tuple(float(v) for v in np.round(more.uniform(-0.8, 0.8, 3) * extent, 3))
            kw["fov"] = float(more.choice([20.0, 70.0, 110.0]))
        path = synth.ensure_scene_file(directory=d, **kw)
        fi = api.FileInfo(path)
        scene = api.Scene(fi.scene_desc)
        cam = fi.camera()
        ref = scene.render(cam, fi.params(), want_ids=True)
        modes = [("brute force", abi.RT_FLAG_NO_FILTER), ("brute force again", abi.RT_FLAG_NO_FILTER)]
        if seed % 3 == 0:
            modes.append(("brute force, exact stages", abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_EXACT_STAGES))
        for name, flags in modes:
            g = scene.render(cam, fi.params(flags=flags), want_ids=True)
            same = np.array_equal(g["ids"], ref["ids"]) and np.array_equal(g["rgb"], ref["rgb"]) and all(
                getattr(g["stats"], f) == getattr(ref["stats"], f) for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"))
            if not same:
                bad += 1
                print("MISMATCH seed %d mode %s %s: ids %d pixels %d blocked %d vs %d" % (
                    seed, name, kw, int((g["ids"] != ref["ids"]).sum()), int((g["rgb"] != ref["rgb"]).any(axis=2).sum()),
                    g["stats"].shadow_blocked, ref["stats"].shadow_blocked), flush=True)
        print("seed %d: %d hits, %d shadow rays (%d blocked), %d lights, %d triangles; %d mismatches so far" % (
            seed, ref["stats"].hits, ref["stats"].shadow_rays, ref["stats"].shadow_blocked, kw["n_lights"], kw["n_triangles"], bad), flush=True)
        scene.close()
        os.remove(path)
    print("stress_long: %d scenes, %d mismatches" % (count, bad))
    return bad


def main():
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)


if __name__ == "__main__":
    main()
