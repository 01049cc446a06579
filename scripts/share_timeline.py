"""One rank's share of scene7 at 3840x2160 (rank 3 of 8; a second argument "synthetic" takes the synthetic scene), a few frames: run under rocprofv3 --kernel-trace and read the
timeline of the last frame with scripts/share_timeline_read.py (gaps between launches, launches side by side)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api  # noqa: E402

api.init(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == "synthetic":
    from raytracer_amd import synth
    fi = api.FileInfo(synth.ensure_scene_file())
else:
    fi = api.FileInfo(os.path.join(root, "tests", "golden", "scenes", "scene7.test"))
    fi.set_size(3840, 2160)
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
prm = fi.params(tile_rank=3 % world, tile_world=world)
buf = torch.zeros(api.rank_tile_count(prm) * 768, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for _ in range(6):
    scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False)
