"""The C-ABI library loads without a GPU and exports every symbol include/rt_amd.h declares."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from raytracer_amd import abi, api


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = api.lib()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(abi.PROTOTYPES) == names


def test_struct_sizes_match_the_header():
    # sizes the C compiler gives the structs of include/rt_amd.h on x86-64
    assert C.sizeof(abi.SceneDesc) == 8 * 8 + 7 * 8 + (6 + 3 + 3 + 6) * 4 + 8
    assert C.sizeof(abi.CameraDesc) == 14 * 4
    assert C.sizeof(abi.RenderParams) == 12 * 4
    assert C.sizeof(abi.Stats) == 6 * 8 + 6 * 8 + 3 * 8 + 3 * 8


def test_tile_partition_is_a_partition():
    for (w, h, world) in [(640, 480, 1), (640, 480, 8), (3840, 2160, 8), (100, 37, 3), (16, 16, 4)]:
        counts = []
        for rank in range(world):
            p = abi.RenderParams(width=w, height=h, max_bounce_count=1, tile_rank=rank, tile_world=world)
            counts.append(api.rank_tile_count(p))
        total = api.total_tile_count(abi.RenderParams(width=w, height=h))
        assert total == -(-w // 16) * -(-h // 16)
        assert sum(counts) == total and max(counts) - min(counts) <= 1


@pytest.mark.skipif(api.lib().rt_device_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_a_gpu():
    with pytest.raises(api.RtError):
        api.init(0)
    p = abi.RenderParams(width=16, height=16, max_bounce_count=1)
    cam = abi.CameraDesc()
    assert api.lib().rt_render(None, C.byref(cam), C.byref(p), None, None, None) != 0
    assert b"rt_init" in api.lib().rt_last_error()
    h = C.c_void_p()
    assert api.lib().rt_scene_create(C.byref(abi.SceneDesc()), C.byref(h)) != 0   # no CPU fallback
