"""Fixed cost of a frame: wall time of rt_render_tiles_device for frames of growing size (scene7 at 3840x2160, a centred
rectangle of n tiles), with and without statistics (the statistics add two HIP events around every scan launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, synth  # noqa: E402

api.init(0)
fi = api.FileInfo(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "scene7.test"))
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
W, H = 3840, 2160


def best(fn, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
    return min(out)


for tiles_x, tiles_y in ((1, 1), (4, 4), (16, 16), (32, 32), (64, 64), (128, 64), (240, 135)):
    w, h = tiles_x * 16, tiles_y * 16
    prm = fi.params(width=W, height=H, rect_x0=(W - w) // 2 // 16 * 16, rect_y0=(H - h) // 2 // 16 * 16, rect_w=w, rect_h=h)
    buf = torch.zeros(api.rank_tile_count(prm) * 768, dtype=torch.uint8, device="cuda:0")
    scene.render_tiles_device(cam, prm, buf.data_ptr())
    a = best(lambda: scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True))
    b = best(lambda: scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False))
    s = scene.render_tiles_device(cam, prm, buf.data_ptr())
    print("%5d tiles: %.3f ms with statistics, %.3f ms without; device_ms %.3f, %d + %d scan launches" %
          (tiles_x * tiles_y, a, b, s.device_ms, s.closest_launches, s.shadow_launches))
for world in (8,):
    for r in (0, 3):
        prm = fi.params(width=W, height=H, tile_rank=r, tile_world=world)
        buf = torch.zeros(api.rank_tile_count(prm) * 768, dtype=torch.uint8, device="cuda:0")
        scene.render_tiles_device(cam, prm, buf.data_ptr())
        a = best(lambda: scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True))
        b = best(lambda: scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False))
        for chunk in (0, 1024, 2048, 8192):
            prm.chunk_tiles = chunk
            c = best(lambda: scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False))
            print("rank %d of %d, chunk_tiles %d: %.3f ms without statistics" % (r, world, chunk, c))
        print("rank %d of %d: %.3f ms with statistics, %.3f without" % (r, world, a, b))
