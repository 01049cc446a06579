"""Time each rank's share of a frame on one GPU (tile_world = N, ranks in turn): the bound an N-GPU run of the tile split
cannot beat, before any exchange.  Run on the GPU box:  python scripts/rank_times.py [synthetic|scene7_4k] > profiles/...txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, distributed, synth

api.init(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for workload in (sys.argv[1:] or ["synthetic", "scene7_4k"]):
    if workload == "synthetic":
        fi = api.FileInfo(synth.ensure_scene_file())
    else:
        fi = api.FileInfo(os.path.join(root, "tests", "golden", "scenes", "scene7.test"))
        fi.set_size(3840, 2160)
    cam = fi.camera()
    scene = api.Scene(fi.scene_desc)
    print("%s @ %dx%d, maxdepth %d: wall time of rt_render_tiles_device per rank share, best of 3" % (workload, fi.width, fi.height, fi.max_recursion_depth))
    whole = None
    for world in (1, 2, 4, 8):
        stride = distributed.stride_tiles(fi.params(tile_world=world), world)
        buf = torch.zeros(stride * 768, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        times = []
        for rank in range(world):
            prm = fi.params(tile_rank=rank, tile_world=world)
            best = 1e9
            for _ in range(4):      # the first call sizes the arenas
                t0 = time.perf_counter()
                scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False)
                best = min(best, (time.perf_counter() - t0) * 1e3)
            times.append(best)
        if world == 1:
            whole = times[0]
        # the slowest rank's share again, two frames in flight (rt_render_tiles_begin / _end): time per frame in the steady state
        slow = max(range(world), key=lambda r: times[r])
        prm = fi.params(tile_rank=slow, tile_world=world)
        bufs = [buf, torch.zeros_like(buf)]
        torch.cuda.synchronize()
        frames = 12
        ticket = api.render_tiles_begin(scene, cam, prm, bufs[0].data_ptr())
        t0 = time.perf_counter()
        for k in range(1, frames + 1):
            nxt = api.render_tiles_begin(scene, cam, prm, bufs[k % 2].data_ptr())
            api.render_tiles_end(ticket)
            ticket = nxt
        in_flight = (time.perf_counter() - t0) * 1e3 / frames
        api.render_tiles_end(ticket)
        print("  %d ranks: share min %.2f ms, max %.2f ms -> at best %.2fx of %d (whole frame %.2f ms); two frames in flight: %.2f ms per frame -> %.2fx"
              % (world, min(times), max(times), whole / max(times), world, whole, in_flight, whole / in_flight))
        if world > 1:
            print("      per rank: " + " ".join("%.2f" % t for t in times))
    scene.close()
