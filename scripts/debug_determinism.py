"""Render the same frame several times and report where the images and counters differ (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (before the library: torch ships its own HIP runtime)
from raytracer_amd import api

api.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "scene7"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
depth = int(sys.argv[4]) if len(sys.argv) > 4 else None
fi = api.FileInfo(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", name + ".test"))
fi.set_size(w, h)
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
kw = {} if depth is None else {"max_bounce_count": depth}
runs = [scene.render(cam, fi.params(**kw), want_ids=True) for _ in range(3)]
for i, r in enumerate(runs):
    s = r["stats"]
    print(i, "hits", s.hits, "closest", s.closest_rays, "shadow", s.shadow_rays, "blocked", s.shadow_blocked, "full", s.full_tests, "queued", list(s.queued_pairs))
for i in (1, 2):
    d = np.argwhere((runs[0]["rgb"] != runs[i]["rgb"]).any(axis=-1))
    di = np.argwhere(runs[0]["ids"] != runs[i]["ids"])
    print("run 0 vs", i, ": pixels differing", len(d), "first", d[:8].tolist(), "| primary ids differing", len(di), di[:4].tolist())
    for y, x in d[:4]:
        print("   ", (int(y), int(x)), runs[0]["rgb"][y, x].tolist(), runs[i]["rgb"][y, x].tolist())
if os.environ.get("RT_DBG_SAVE"):
    np.save(os.environ["RT_DBG_SAVE"], runs[0]["rgb"])
