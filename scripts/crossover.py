#!/usr/bin/env python3
"""Frame time of small synthetic scenes, hierarchy walk against brute force, per primitive kind.

    python scripts/crossover.py KIND N [N ...]      KIND = triangles | ellipsoids | spheres

Run once with RT_AMD_BRUTE_BELOW=0 (always walk) and once with a huge value (never walk): the
size where the two meet is BRUTE_FORCE_BELOW in render.hip.
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, synth


def main():
    kind, sizes = sys.argv[1], [int(a) for a in sys.argv[2:]]
    api.init(0)
    d = tempfile.mkdtemp()
    for n in sizes:
        kw = dict(n_triangles=0, n_ellipsoids=0, n_spheres=0, width=1920, height=1080, maxdepth=3, seed=21, edge=1.0,
                  ell_scale=(0.3, 1.0), sphere_radius=(0.3, 1.0))
        kw["n_" + kind] = n
        fi = api.FileInfo(synth.ensure_scene_file(directory=d, **kw))
        scene = api.Scene(fi.scene_desc)
        cam, prm = fi.camera(), fi.params()
        best = 1e9
        for it in range(4):
            t0 = time.perf_counter()
            out = scene.render(cam, prm)
            dt = time.perf_counter() - t0
            if it:
                best = min(best, out["stats"].device_ms if hasattr(out["stats"], "device_ms") else dt * 1e3)
        st = out["stats"]
        print("%s n=%d  brute_below=%s  %.2f ms  rays %d + %d  child tests %s  leaf tests %s  closest %.2f ms shadow %.2f ms" % (
            kind, n, os.environ.get("RT_AMD_BRUTE_BELOW", "default"), best, st.closest_rays, st.shadow_rays,
            list(st.pair_pretests), list(st.queued_pairs), st.closest_ms, st.shadow_ms), flush=True)
        scene.close()


if __name__ == "__main__":
    main()
