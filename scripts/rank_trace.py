"""Render one rank's share of the 4K scene7 frame a few times (for rocprofv3 --kernel-trace): python rank_trace.py WORLD RANK"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, distributed

world, rank = int(sys.argv[1]), int(sys.argv[2])
api.init(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fi = api.FileInfo(os.path.join(root, "tests", "golden", "scenes", "scene7.test"))
fi.set_size(3840, 2160)
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
prm = fi.params(tile_rank=rank, tile_world=world)
buf = torch.zeros(distributed.stride_tiles(prm, world) * 768, dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    st = scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True)
print("device_ms", st.device_ms)
