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


def test_struct_sizes_match_the_header(tmp_path):
    """ctypes mirrors == what the C compiler makes of include/rt_amd.h (sizes and the offset of the last field)."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "rt_amd.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(rt_scene_desc), sizeof(rt_camera_desc), sizeof(rt_render_params),'
        ' sizeof(rt_stats), offsetof(rt_stats, executed_flop)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(abi.SceneDesc), C.sizeof(abi.CameraDesc), C.sizeof(abi.RenderParams), C.sizeof(abi.Stats),
                     abi.Stats.executed_flop.offset]


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
