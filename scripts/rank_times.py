"""Time each rank's share of a frame on one GPU (tile_world = N, ranks in turn): what an N-GPU run can reach at best (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, distributed

api.init(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fi = api.FileInfo(os.path.join(root, "tests", "golden", "scenes", "scene7.test"))
fi.set_size(3840, 2160)
cam = fi.camera()
scene = api.Scene(fi.scene_desc)
for world in (1, 2, 4, 8):
    prm0 = fi.params(tile_world=world)
    stride = distributed.stride_tiles(prm0, world)
    buf = torch.zeros(stride * 768, dtype=torch.uint8, device="cuda:0")
    times = []
    for rank in range(world):
        prm = fi.params(tile_rank=rank, tile_world=world)
        scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True)   # warm-up (allocations)
        t0 = time.perf_counter()
        st = scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=True)
        times.append((time.perf_counter() - t0) * 1e3)
    print("world %d: per-rank ms min %.2f max %.2f  -> ideal speed-up %.2f of %d (device_ms last rank %.2f)" % (world, min(times), max(times), times_1 / max(times) if world > 1 else 1.0, world, st.device_ms) if world > 1 else "world 1: %.2f ms (device %.2f)" % (times[0], st.device_ms))
    if world == 1:
        times_1 = times[0]
