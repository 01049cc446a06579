"""The maintainer's binding (the code a maintainer adds to the reference's main.cpp, as the integration guide at the repository root
shows it) compiled and run, and the whole-flow entry points (Source/main.cpp:25-98 on the GPU path).

* CPU: the two code blocks of the guide's first section are, word for word, the blocks compiled in oracle/integration_binding.cpp.
* GPU: that translation unit (the reference's own parse_input_file + the binding + librt_amd.so) renders scene1 and scene6 to
  the reference's full-frame bytes (SHA-256 of the golden fixtures, generated from the unmodified reference);
  rt_render_file_info and the rt_amd_render command line program do the same, the latter through the PNG it writes.
"""
import ctypes as C
import hashlib
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, scene_path

BINDING_SO = os.path.join(ROOT, "oracle", "_ref", "librt_binding.so")
CLI = os.path.join(ROOT, "raytracer_amd", "lib", "rt_amd_render")


def _golden_sha(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "render_%s.npz" % name))["full_sha256"].tobytes()


def _words(text):
    return re.sub(r"\s+", " ", text).strip()


def test_the_guide_s_blocks_are_the_compiled_ones():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 2
    src = open(os.path.join(ROOT, "oracle", "integration_binding.cpp")).read()
    for n in (1, 2):
        compiled = re.search(r"// \[binding block %d: begin\]\n(.*?)// \[binding block %d: end\]" % (n, n), src, flags=re.S).group(1)
        assert _words(blocks[n - 1]) == _words(compiled), "block %d of the integration guide differs from oracle/integration_binding.cpp" % n


def read_png_rgb8(path):
    """Minimal PNG reader (8-bit RGB, no interlace): enough for what rt_write_png and stb write."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, width, height = 8, b"", 0, 0
    while pos < len(data):
        length = int.from_bytes(data[pos:pos + 4], "big")
        kind = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        assert zlib.crc32(kind + body) == int.from_bytes(data[pos + 8 + length:pos + 12 + length], "big")
        if kind == b"IHDR":
            width, height = int.from_bytes(body[:4], "big"), int.from_bytes(body[4:8], "big")
            assert body[8:13] == bytes([8, 2, 0, 0, 0])      # 8 bits, colour type RGB, deflate, adaptive filters, no interlace
        elif kind == b"IDAT":
            idat += body
        pos += 12 + length
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(height, 1 + width * 3)
    out = np.zeros((height, width * 3), np.uint8)
    for y in range(height):
        f, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(width * 3, np.int32)
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + prev) & 255
        else:                                               # Sub, Average, Paeth: byte-serial
            cur = np.zeros(width * 3, np.int32)
            for i in range(width * 3):
                a = cur[i - 3] if i >= 3 else 0
                b = prev[i]
                c = prev[i - 3] if i >= 3 else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
            out[y] = cur
    return out.reshape(height, width, 3)


def test_png_reader_round_trip(tmp_path):
    """rt_write_png -> read_png_rgb8 (host code only: no GPU needed)."""
    from raytracer_amd import api
    rgb = np.random.default_rng(3).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    api.write_png(path, rgb)
    assert np.array_equal(read_png_rgb8(path), rgb)


@pytest.mark.parametrize("shape", [(1, 1), (1, 700), (33, 5), (480, 640), (301, 1023)])
def test_png_writer_pieces_concatenate_to_one_valid_stream(tmp_path, shape):
    """rt_write_png deflates the rows in pieces on several threads; an independent decoder (PIL) must read the pixels back.
    Also the background form (rt_write_png_begin / rt_write_png_end)."""
    from PIL import Image
    from raytracer_amd import api
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    rgb = np.stack([(xx // 3 + yy) % 256, (yy // 2) % 256, rng.integers(0, 256, shape)], axis=2).astype(np.uint8)
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    api.write_png(a, rgb)
    api.write_png_begin(b, rgb)
    rgb_after = rgb.copy()
    rgb[:] = 0                      # the writer works on its own copy
    api.write_png_end()
    for path in (a, b):
        assert np.array_equal(np.array(Image.open(path).convert("RGB")), rgb_after)
    assert np.array_equal(read_png_rgb8(a), rgb_after)
    with pytest.raises(api.RtError):
        api.write_png(str(tmp_path / "no_such_dir" / "x.png"), rgb_after)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["scene1", "scene6"])
def test_reference_front_end_plus_binding_renders_the_reference_image(gpu, name):
    if not os.path.exists(BINDING_SO):
        pytest.skip("oracle/_ref/librt_binding.so was not built (needs the reference sources)")
    lib = C.CDLL(BINDING_SO)
    lib.binding_main.restype = C.c_int
    lib.binding_main.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    w, h = C.c_uint32(0), C.c_uint32(0)
    pixels = np.zeros(640 * 480 * 3, np.uint8)
    assert lib.binding_main(os.fsencode(scene_path(name)), pixels.ctypes.data, pixels.size, C.byref(w), C.byref(h)) == 0
    assert (w.value, h.value) == (640, 480)
    if name == "scene6":      # one sample of scene6 hits a padding lane, whose material the reference reads past its array (see test_oracle_golden)
        fi = gpu.FileInfo(scene_path(name))
        scene = gpu.Scene(fi.scene_desc)
        ours = scene.render(fi.camera(), fi.params())["rgb"]
        scene.close()
        assert np.array_equal(pixels.reshape(480, 640, 3), ours)
    else:
        assert hashlib.sha256(pixels.tobytes()).digest() == _golden_sha(name)


@pytest.mark.gpu
def test_render_file_info_whole_flow(gpu):
    from raytracer_amd import abi, api
    fi = gpu.FileInfo(scene_path("scene1"))
    rgb = np.zeros((fi.height, fi.width, 3), np.uint8)
    stats = abi.Stats()
    assert api.lib().rt_render_file_info(fi._h, rgb.ctypes.data, C.byref(stats)) == 0
    assert hashlib.sha256(rgb.tobytes()).digest() == _golden_sha("scene1")
    assert stats.samples == 640 * 480 * 16 and stats.culled == 4278598 and stats.hits == 636595     # the reference's own counts for scene1


@pytest.mark.gpu
def test_command_line_program_writes_the_reference_image(gpu, tmp_path):
    """rt_amd_render scene1.test: parse, render on the GPU, write the PNG the scene's `output` command names."""
    if not os.path.exists(CLI):
        pytest.skip("rt_amd_render was not built")
    done = subprocess.run([CLI, scene_path("scene1")], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stdout + done.stderr
    fi = gpu.FileInfo(scene_path("scene1"))
    rgb = read_png_rgb8(str(tmp_path / fi.output_filename))
    assert rgb.shape == (480, 640, 3) and hashlib.sha256(rgb.tobytes()).digest() == _golden_sha("scene1")
    # a scene file that does not parse: the reference's message and a failing exit code
    bad = tmp_path / "bad.test"
    bad.write_text("camera 0 0 1 0 0 0 0 1 0 45\n")
    done = subprocess.run([CLI, str(bad)], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert done.returncode != 0 and "Failed to parse input file." in done.stdout
