"""Tile-interleaved rendering on the GPU: the per-rank tile buffers of a 2- and 3-way split, laid out as the
all-gather delivers them, assemble (rt_assemble_tiles_device) into the image a single rank renders."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_split_frames_assemble_to_the_single_rank_frame(gpu, world):
    import torch
    from raytracer_amd import distributed
    fi = gpu.FileInfo(scene_path("scene6"))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    rect = dict(rect_x0=200, rect_y0=150, rect_w=250, rect_h=170)     # partial tiles on the right and bottom edges
    whole = scene.render(cam, fi.params(**rect))["rgb"]
    stride = distributed.stride_tiles(fi.params(**rect), world)
    gathered = torch.zeros(world * stride * 768, dtype=torch.uint8, device="cuda:0")
    for rank in range(world):
        prm = fi.params(tile_rank=rank, tile_world=world, **rect)
        part = gathered[rank * stride * 768:(rank + 1) * stride * 768]
        scene.render_tiles_device(cam, prm, part.data_ptr())
        host = scene.render(cam, prm)["rgb"]                          # rt_render with a split fills only this rank's tiles
        mask = np.zeros(whole.shape[:2], bool)
        tiles_x = distributed.tile_grid(prm)[0]
        for t in distributed.rank_tiles(prm, rank, world):
            ty, tx = distributed.tile_position(t, tiles_x)
            mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
        assert np.array_equal(host[mask], whole[mask]) and not host[~mask].any()
    frame = torch.zeros(whole.size, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    gpu.assemble_tiles_device(gathered.data_ptr(), stride, fi.params(tile_world=world, **rect), frame.data_ptr())
    assert np.array_equal(frame.cpu().numpy().reshape(whole.shape), whole)
    scene.close()


def test_library_communicator_gathers_tile_buffers_through_rccl(gpu):
    """One process per GPU, as the library sees it (rt_comm_*), with the one GPU a box has: a communicator of one rank, whose
    gather still goes through RCCL (a send to itself).  Frame assembled from the gathered buffer == frame rendered directly."""
    import torch
    from raytracer_amd import distributed
    fi = gpu.FileInfo(scene_path("scene6"))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    rect = dict(rect_x0=100, rect_y0=80, rect_w=330, rect_h=210)
    prm = fi.params(tile_rank=0, tile_world=1, **rect)
    whole = scene.render(cam, prm)["rgb"]
    uid = gpu.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    gpu.comm_init_rank(uid, 0, 1)
    try:
        stride = distributed.stride_tiles(prm, 1)
        tiles = torch.zeros(stride * 768, dtype=torch.uint8, device="cuda:0")
        gathered = torch.zeros(stride * 768, dtype=torch.uint8, device="cuda:0")
        frame = torch.zeros(whole.size, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        for _ in range(2):                      # twice: the exchange stream is reused
            scene.render_tiles_device(cam, prm, tiles.data_ptr())
            gpu.comm_gather_tiles(tiles.data_ptr(), stride, gathered.data_ptr())
            gpu.comm_wait()
            gpu.assemble_tiles_device(gathered.data_ptr(), stride, prm, frame.data_ptr())
            assert np.array_equal(frame.cpu().numpy().reshape(whole.shape), whole)
            gathered.zero_()
            torch.cuda.synchronize()
    finally:
        gpu.comm_destroy()
        scene.close()


MULTI_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import torch
from raytracer_amd import api
api.init_devices(1)
fi = api.FileInfo(%(scene)r)
scene = api.Scene(fi.scene_desc)
prm = fi.params(rect_x0=64, rect_y0=48, rect_w=300, rect_h=200)
multi = api.render_multi(scene, fi.camera(), prm, 1)
single = scene.render(fi.camera(), prm)
assert np.array_equal(multi["rgb"], single["rgb"]) and multi["rgb"].any()
assert multi["stats"][0].closest_rays == single["stats"].closest_rays and multi["stats"][0].hits == single["stats"].hits
# the same frame in two halves, the image left on the device; two frames under way, a third refused; then the single-device
# entry points from this thread (its HIP device must be the first context's again)
frames = [torch.zeros(multi["rgb"].size, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
torch.cuda.synchronize()
t0 = api.render_multi_begin(scene, fi.camera(), prm, frames[0].data_ptr())
t1 = api.render_multi_begin(scene, fi.camera(), prm, frames[1].data_ptr(), want_stats=False)
try:
    api.render_multi_begin(scene, fi.camera(), prm, frames[2].data_ptr())
    raise SystemExit("a third frame was accepted")
except api.RtError as e:
    assert "two frames are under way" in str(e)
stats0, ms0 = api.render_multi_end(t0, 1)
t2 = api.render_multi_begin(scene, fi.camera(), prm, frames[2].data_ptr())
api.render_multi_end(t1, 1)
stats2, ms2 = api.render_multi_end(t2, 1)
for f in frames:
    assert np.array_equal(f.cpu().numpy().reshape(multi["rgb"].shape), single["rgb"])
assert stats0[0].hits == single["stats"].hits and stats2[0].shadow_blocked == single["stats"].shadow_blocked and ms0 > 0.0
again = scene.render(fi.camera(), prm)
assert np.array_equal(again["rgb"], single["rgb"])
out = api.debug_eval(0, np.array([0.5], np.float32), np.array([1.0], np.float32))
assert out.view(np.uint32)[0] == 0x3EFFF992
print("MULTI_OK")
"""


SEVERAL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import torch
from raytracer_amd import api
n = %(n)d
api.init_devices(n)
fi = api.FileInfo(%(scene)r)
scene = api.Scene(fi.scene_desc)                     # one replica per context
cam = fi.camera()
for rect in (dict(rect_x0=64, rect_y0=48, rect_w=300, rect_h=200), dict(rect_x0=0, rect_y0=0, rect_w=33, rect_h=17), dict()):
    prm = fi.params(**rect)
    single = scene.render(cam, prm)                   # the first context alone
    multi = api.render_multi(scene, cam, prm, n)
    assert np.array_equal(multi["rgb"], single["rgb"]) and multi["rgb"].any(), rect
    for f in ("samples", "culled", "closest_rays", "hits", "shadow_rays", "shadow_blocked"):
        assert sum(getattr(st, f) for st in multi["stats"]) == getattr(single["stats"], f), f
    if rect.get("rect_w", 640) > 100:
        assert all(st.closest_rays > 0 for st in multi["stats"])       # every context had a share
# the image left on the device, two frames under way, different rectangles in turn
prm_a, prm_b = fi.params(rect_x0=64, rect_y0=48, rect_w=300, rect_h=200), fi.params(rect_x0=10, rect_y0=300, rect_w=500, rect_h=100)
want_a, want_b = scene.render(cam, prm_a)["rgb"], scene.render(cam, prm_b)["rgb"]
bufs = [torch.zeros(max(want_a.size, want_b.size), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
torch.cuda.synchronize()
ticket = api.render_multi_begin(scene, cam, prm_a, bufs[0].data_ptr())
for k in range(1, 7):
    nxt = api.render_multi_begin(scene, cam, prm_b if k %% 2 else prm_a, bufs[k %% 2].data_ptr())
    stats, ms = api.render_multi_end(ticket, n)
    want = want_b if (k - 1) %% 2 else want_a
    got = bufs[(k - 1) %% 2][:want.size].cpu().numpy().reshape(want.shape)
    assert np.array_equal(got, want), k
    assert sum(st.hits for st in stats) > 0
    ticket = nxt
api.render_multi_end(ticket, n)
scene.close()
print("SEVERAL_OK")
"""


@pytest.mark.parametrize("n", [2, 3, 8])
def test_several_device_contexts_on_the_one_gpu(gpu, n):
    """The n-device host logic of one process - a replica of the scene per context, a worker thread and a share of the tiles per
    context, the rank-major gather and the assembly, two frames under way - with every context on the one GPU of the box
    (RT_AMD_DEVICES_ON_ONE_GPU; the gather is then device-to-device copies, because RCCL refuses two ranks on one GPU).
    Images equal the single-context render; the contexts' ray counters add up to its counters."""
    import os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, RT_AMD_DEVICES_ON_ONE_GPU="1")
    done = subprocess.run([sys.executable, "-c", SEVERAL_SCRIPT % {"root": ROOT, "scene": scene_path("scene6"), "n": n}], env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "SEVERAL_OK" in done.stdout, done.stdout[-2000:] + done.stderr[-2000:]


@pytest.mark.parametrize("through_rccl", [False, True])
def test_single_process_multi_device_entry_point(gpu, through_rccl):
    """rt_init_devices + rt_render_multi in a process of their own (one device here: the partition, the per-device thread, the
    assembly and the copy back; with RT_AMD_COMM_ALWAYS also ncclCommInitAll and the grouped send / receive)."""
    import os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ)
    if through_rccl:
        env["RT_AMD_COMM_ALWAYS"] = "1"
    done = subprocess.run([sys.executable, "-c", MULTI_SCRIPT % {"root": ROOT, "scene": scene_path("scene6")}], env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "MULTI_OK" in done.stdout, done.stdout[-2000:] + done.stderr[-2000:]


def test_two_frames_in_flight_give_the_frames_of_one_at_a_time(gpu):
    """rt_render_tiles_begin / _end: two frames traced side by side (different shares of a split frame, then the same camera from two
    places, which rebuilds the box records for the farther one) == the same frames one after the other; a third begin is refused."""
    import torch
    fi = gpu.FileInfo(scene_path("scene7"))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    rect = dict(rect_x0=120, rect_y0=90, rect_w=400, rect_h=300)
    shares = [fi.params(tile_rank=r, tile_world=3, **rect) for r in range(3)]
    n = max(gpu.rank_tile_count(p) for p in shares) * 768
    want = [torch.zeros(n, dtype=torch.uint8, device="cuda:0") for _ in shares]
    got = [torch.full((n,), 5, dtype=torch.uint8, device="cuda:0") for _ in shares]
    torch.cuda.synchronize()
    stats = [scene.render_tiles_device(cam, p, w.data_ptr()) for p, w in zip(shares, want)]
    t0 = gpu.render_tiles_begin(scene, cam, shares[0], got[0].data_ptr())
    t1 = gpu.render_tiles_begin(scene, cam, shares[1], got[1].data_ptr())
    assert {t0, t1} == {0, 1}
    with pytest.raises(gpu.RtError, match="two frames are in flight"):
        gpu.render_tiles_begin(scene, cam, shares[2], got[2].data_ptr())
    s0 = gpu.render_tiles_end(t0)
    t2 = gpu.render_tiles_begin(scene, cam, shares[2], got[2].data_ptr())       # the freed slot
    s1 = gpu.render_tiles_end(t1)
    s2 = gpu.render_tiles_end(t2)
    with pytest.raises(gpu.RtError, match="no frame in flight"):
        gpu.render_tiles_end(t2)
    torch.cuda.synchronize()
    for w, g, a, b in zip(want, got, stats, (s0, s1, s2)):
        k = gpu.rank_tile_count(shares[0]) * 768
        assert w.any() and torch.equal(w[:k - 768], g[:k - 768])
        for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
            assert getattr(a, f) == getattr(b, f), f
    # a camera far outside the bound the box records were built for, begun while another frame is in flight
    far = fi.camera()
    for q in range(3):
        far.eye[q] = cam.eye[q] * 40.0
    prm = fi.params(**rect)
    m = gpu.rank_tile_count(prm) * 768
    near_want, far_want, near_got, far_got = (torch.zeros(m, dtype=torch.uint8, device="cuda:0") for _ in range(4))
    torch.cuda.synchronize()
    scene.render_tiles_device(cam, prm, near_want.data_ptr())
    scene.render_tiles_device(far, prm, far_want.data_ptr())
    scene2 = gpu.Scene(fi.scene_desc)                                            # fresh box records
    ta = gpu.render_tiles_begin(scene2, cam, prm, near_got.data_ptr())
    tb = gpu.render_tiles_begin(scene2, far, prm, far_got.data_ptr())
    gpu.render_tiles_end(ta)
    gpu.render_tiles_end(tb)
    torch.cuda.synchronize()
    assert torch.equal(near_want, near_got) and torch.equal(far_want, far_got)
    scene.close()
    scene2.close()
