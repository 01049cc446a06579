#!/usr/bin/env python3
"""Random small scenes: the GPU path (default mode) against the CPU oracle on bytes, ids and ray counters.
GPU box:  python scripts/stress_oracle.py FIRST_SEED COUNT"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checkers
from raytracer_amd import api, synth


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    api.init(0)
    oracle = checkers.load_oracle()
    d = tempfile.mkdtemp()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        kw = dict(n_triangles=int(rng.choice([0, 100, 1500])), n_ellipsoids=int(rng.choice([0, 20, 300])), n_spheres=int(rng.choice([0, 9, 300])),
                  n_lights=int(rng.integers(1, 4)), width=int(rng.integers(24, 72)), height=int(rng.integers(16, 48)), maxdepth=int(rng.integers(1, 6)), seed=seed,
                  extent=float(rng.choice([1.0, 10.0, 300.0])), edge=float(rng.choice([0.05, 0.3, 1.5])),
                  ell_scale=(float(rng.choice([0.01, 0.1, 0.3])), float(rng.choice([0.4, 1.2]))), sphere_radius=(float(rng.choice([0.01, 0.1, 0.3])), float(rng.choice([0.4, 1.5]))))
        kw["edge"] *= kw["extent"] / 10.0
        if kw["n_triangles"] + kw["n_ellipsoids"] + kw["n_spheres"] == 0:
            kw["n_triangles"] = 100
        fi = api.FileInfo(synth.ensure_scene_file(directory=d, **kw))
        scene = api.Scene(fi.scene_desc)
        cam = fi.camera()
        g = scene.render(cam, fi.params(), want_ids=True)
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
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
