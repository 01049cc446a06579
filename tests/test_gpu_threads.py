"""Calls from several host threads at once, which include/rt_amd.h declares safe (they take locks of the library and run
one after the other): whole renders of different rectangles, scenes created and destroyed meanwhile, frames begun and
ended from two threads, and a camera far outside the scene (which makes the library rebuild its bounding records) beside
frames that read those records.  Every result must equal the one the same call gives alone."""
import threading
import time

import numpy as np
import pytest

from conftest import scene_path
from raytracer_amd import abi
from test_gpu_edges import FLOOR, HEADER, SPHERES, write

pytestmark = pytest.mark.gpu


def _run(workers):
    errors = []

    def guarded(fn, name):
        def body():
            try:
                fn()
            except BaseException as e:          # an assertion in a thread must reach the test
                errors.append("%s: %r" % (name, e))
        return body

    threads = [threading.Thread(target=guarded(fn, name), name=name) for name, fn in workers]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a thread did not finish"
    assert not errors, errors


def test_renders_and_scene_churn_from_four_threads(gpu):
    fi = gpu.FileInfo(scene_path("scene6"))
    cam = fi.camera()
    scene = gpu.Scene(fi.scene_desc)
    fi4 = gpu.FileInfo(scene_path("scene4-specular"))
    rects = [dict(rect_x0=200, rect_y0=150, rect_w=96, rect_h=64), dict(rect_x0=0, rect_y0=0, rect_w=50, rect_h=37),
             dict(rect_x0=320, rect_y0=240, rect_w=128, rect_h=32)]
    modes = [0, abi.RT_FLAG_WALK_ALL, abi.RT_FLAG_NO_FILTER]
    alone = [scene.render(cam, fi.params(flags=m, **r), want_ids=True) for r, m in zip(rects, modes)]
    other = gpu.Scene(fi4.scene_desc)
    alone4 = other.render(fi4.camera(), fi4.params(rect_x0=280, rect_y0=200, rect_w=80, rect_h=48), want_ids=True)
    other.close()

    def renderer(i):
        def body():
            for _ in range(8):
                g = scene.render(cam, fi.params(flags=modes[i], **rects[i]), want_ids=True)
                assert np.array_equal(g["rgb"], alone[i]["rgb"]) and np.array_equal(g["ids"], alone[i]["ids"])
                for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"):
                    assert getattr(g["stats"], f) == getattr(alone[i]["stats"], f), f
        return body

    def churn():
        for _ in range(6):      # scenes come and go while the others render
            s = gpu.Scene(fi4.scene_desc)
            g = s.render(fi4.camera(), fi4.params(rect_x0=280, rect_y0=200, rect_w=80, rect_h=48), want_ids=True)
            s.close()
            assert np.array_equal(g["rgb"], alone4["rgb"]) and np.array_equal(g["ids"], alone4["ids"])

    _run([("render %d" % i, renderer(i)) for i in range(3)] + [("scene churn", churn)])
    scene.close()


def test_frames_in_flight_from_two_threads_beside_a_rebuild_of_the_bounding_records(gpu, tmp_path):
    import torch
    body = FLOOR + SPHERES
    near = write(tmp_path, "near", HEADER.format(w=96, h=64, depth=3, cam="0 2 9 0 0 0") + body)
    far = write(tmp_path, "far", HEADER.format(w=96, h=64, depth=3, cam="300 900 2500 0 0 0").replace("0 1 0 40\n", "0 1 0 0.5\n") + body)
    fi_near, fi_far = gpu.FileInfo(near), gpu.FileInfo(far)
    prm = fi_near.params(flags=abi.RT_FLAG_WALK_ALL)
    n = gpu.rank_tile_count(prm) * 768

    def tiles_alone(scene, cam):
        buf = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False)
        return buf.cpu().numpy()

    reference_scene = gpu.Scene(fi_near.scene_desc)
    want = {"near": tiles_alone(reference_scene, fi_near.camera()), "far": tiles_alone(reference_scene, fi_far.camera())}
    reference_scene.close()
    assert want["near"].any() and want["far"].any()

    scene = gpu.Scene(fi_near.scene_desc)     # its bounding records hold for the near eye only, until a far one comes

    def in_flight(which):
        cam = (fi_near if which == "near" else fi_far).camera()

        def body():
            buf = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
            for _ in range(12):
                buf.zero_()
                torch.cuda.synchronize()
                while True:
                    try:
                        ticket = gpu.render_tiles_begin(scene, cam, prm, buf.data_ptr())
                        break
                    except gpu.RtError as e:        # both frame slots are taken by the other threads: allowed, try again
                        assert "under way" in str(e) or "in flight" in str(e), e
                        time.sleep(0.0005)
                gpu.render_tiles_end(ticket)
                torch.cuda.synchronize()
                assert np.array_equal(buf.cpu().numpy(), want[which]), which
        return body

    def whole_renders():
        for k in range(10):
            cam = (fi_far if k % 2 else fi_near).camera()
            buf = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
            scene.render_tiles_device(cam, prm, buf.data_ptr(), want_stats=False)
            assert np.array_equal(buf.cpu().numpy(), want["far" if k % 2 else "near"]), k

    _run([("frames near", in_flight("near")), ("frames far", in_flight("far")), ("whole renders", whole_renders)])
    scene.close()
