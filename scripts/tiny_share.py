import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import api, distributed
api.init(0)
fi = api.FileInfo('/root/repo/tests/golden/scenes/scene7.test'); fi.set_size(3840, 2160)
cam = fi.camera(); scene = api.Scene(fi.scene_desc)
for world in (8, 16, 64, 256, 4096):
    prm = fi.params(tile_rank=0, tile_world=world)
    buf = torch.zeros(distributed.stride_tiles(prm, world) * 768, dtype=torch.uint8, device="cuda:0")
    scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); st = scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False); ts.append((time.perf_counter() - t0) * 1e3)
    print("1/%d of the frame: %.3f ms (tiles %d)" % (world, min(ts), api.rank_tile_count(prm)))
