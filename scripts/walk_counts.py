"""Work counters of one default-mode frame (synthetic scene at 3840x2160 unless a .test file is named): per-ray box tests,
queued leaf pairs and full triangle tests of the closest-hit and the shadow walks, per ray."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, synth  # noqa: E402

api.init(0)
if len(sys.argv) > 1:
    fi = api.FileInfo(sys.argv[1])
    fi.set_size(3840, 2160)
else:
    fi = api.FileInfo(synth.ensure_scene_file())
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
prm = fi.params()
buf = torch.zeros(api.rank_tile_count(prm) * 768, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
scene.render_tiles_device(cam, prm, buf.data_ptr())
s = scene.render_tiles_device(cam, prm, buf.data_ptr())
print("closest rays %d (hits %d), shadow rays %d (blocked %d), device %.2f ms" % (s.closest_rays, s.hits, s.shadow_rays, s.shadow_blocked, s.device_ms))
for name, i, rays in (("closest", 0, s.closest_rays), ("shadow", 1, s.shadow_rays)):
    print("%-8s box tests per ray %.1f, queued leaf pairs per ray %.2f, executed FLOP per ray %.0f" % (name, s.pair_pretests[i] / rays, s.queued_pairs[i] / rays, s.executed_flop[i] / rays))
for name, i, rays in (("closest", 0, s.closest_rays), ("shadow", 1, s.shadow_rays)):
    waves = max(s.walk_waves[i], 1)
    print("%-8s %d waves, %.1f rays per wave, first-level nodes per wave %.1f, walk steps per wave %.1f" % (name, waves, rays / waves, s.first_level_nodes[i] / waves, s.walk_steps[i] / waves))
print("full triangle tests per ray (both) %.3f" % (s.full_tests / (s.closest_rays + s.shadow_rays)))
print("closest ms by kind %s, shadow ms by kind %s" % ([round(x, 2) for x in s.closest_kind_ms], [round(x, 2) for x in s.shadow_kind_ms]))
