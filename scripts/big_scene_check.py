"""One-off check of a hierarchy with more than one apex node (more than 256 super nodes x 16 ... i.e. above 1.05 M triangles): default mode against brute force on a small image."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, api, synth
api.init(0)
d = tempfile.mkdtemp()
fi = api.FileInfo(synth.ensure_scene_file(directory=d, n_triangles=1200000, n_ellipsoids=300, n_spheres=100, n_lights=2, width=64, height=48, maxdepth=2, seed=5))
scene = api.Scene(fi.scene_desc)
cam = fi.camera()
a = scene.render(cam, fi.params(), want_ids=True)
b = scene.render(cam, fi.params(flags=abi.RT_FLAG_NO_FILTER), want_ids=True)
same = np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["ids"], b["ids"])
print("1.2 M triangles (two apex nodes): default mode == brute force:", same, "shadow rays", a["stats"].shadow_rays, b["stats"].shadow_rays, "blocked", a["stats"].shadow_blocked, b["stats"].shadow_blocked)
