"""The host front end (rt_parse_input_file, rt_camera_setup) against parse_input_file of the reference.

Golden: SHA-256 of every array the reference's parser produced for its nine scenes
(tests/golden/parsed.json, written by tests/golden/make_golden.py).  Error behaviour follows
Source/input_parsing.cpp: same messages for the same malformed inputs.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import checkers
from conftest import ALL_SCENES, ROOT, scene_path
from raytracer_amd import abi, api

PARSED = json.load(open(os.path.join(ROOT, "tests", "golden", "parsed.json")))


@pytest.mark.parametrize("name", ALL_SCENES)
def test_parsed_scene_matches_reference(name):
    fi = api.FileInfo(scene_path(name))
    want = PARSED[name]
    assert (fi.width, fi.height, fi.max_recursion_depth, fi.output_filename) == (want["width"], want["height"], want["max_recursion_depth"], want["output"])
    assert bytes(fi.camera11).hex() == want["camera11_hex"]
    assert bytes(fi.camera()).hex() == want["camera_desc_hex"]
    got = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in checkers.desc_arrays(fi.scene_desc).items()}
    assert got == want["arrays_sha256"]
    d = fi.scene_desc
    assert {"triangles": d.n_triangles, "spheres": d.n_spheres, "ellipsoids": d.n_ellipsoids,
            "point_lights": d.n_point_lights, "directional": d.has_directional_light} == want["counts"]


@pytest.mark.skipif(not checkers.have_ref(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("name", ALL_SCENES)
def test_parsed_scene_matches_live_reference(name):
    ref = checkers.load_ref()
    err = C.c_char_p()
    h = ref.ref_parse(scene_path(name).encode(), C.byref(err))
    assert h
    a = checkers.desc_arrays(ref.ref_parse_scene(h).contents)
    fi = api.FileInfo(scene_path(name))      # keep it alive: scene_desc points into it
    b = checkers.desc_arrays(fi.scene_desc)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
    ref.ref_parse_free(h)


def _parse_text(tmp_path, text):
    p = tmp_path / "s.test"
    p.write_text(text)
    return api.FileInfo(str(p))


ERRORS = [
    ("camera 0 0 1 0 0 0 0 1 0 45\n", "First command should be 'size'."),
    ("size 10\n", "'size' command should have 2 positive integer parameters."),
    ("size 10 -4\n", "'size' command should have 2 positive integer parameters."),
    ("size 8 8\nfoo 1\n", "Unknown command entered."),
    ("size 8 8\n   \n", "Unknown command entered."),
    ("size 8 8\nmaxdepth 2.5\n", "'maxdepth' command should have 1 positive integer parameter."),
    ("size 8 8\nvertex 0 0 0\nmaxverts 4\n", "'maxverts' should be specified before vertices are specified."),
    ("size 8 8\ncamera 0 0 1 0 0 0 0 1 0\n", "'camera' command should have 10 floating point parameters."),
    ("size 8 8\nvertex 0 0 1e3\n", "'vertex' command should have 3 floating point parameters."),
    ("size 8 8\nvertex 0 0 0\ntri 0 0 1\n", "Vertex index specified in 'tri' command is beyond the number of specified vertices."),
    ("size 8 8\ntri 0 1\n", "'tri' command should have 3 positive integer parameters."),
    ("size 8 8\nsphere 0 0 0\n", "'sphere' command should have 4 floating point parameters."),
    ("size 8 8\npushTransform 1\n", "'pushTransform' command does not take any parameters."),
    ("size 8 8\npopTransform\n", "Cannot perform 'popTransform' as there are no transforms on the stack."),
    ("size 8 8\ntranslate 1 2\n", "'translate' command should have 3 floating point parameters."),
    ("size 8 8\nscale 1 2 x\n", "'scale' command should have 3 floating point parameters."),
    ("size 8 8\nrotate 1 0 0\n", "'rotate' command should have 4 floating point parameters."),
    ("size 8 8\ndirectional 0 0 1 1 1\n", "'directional' command should have 6 floating point parameters."),
    ("size 8 8\npoint 0 0 1 1 1 1 1\n", "'point' command should have 6 floating point parameters."),
    ("size 8 8\nattenuation 1 0\n", "'attenuation' command should have 3 floating point parameters."),
    ("size 8 8\nambient 1\n", "'ambient' command should have 3 floating point parameters."),
    ("size 8 8\ndiffuse 1 1 1.0.0\n", "'diffuse' command should have 3 floating point parameters."),
    ("size 8 8\nspecular\n", "'specular' command should have 3 floating point parameters."),
    ("size 8 8\nemission a b c\n", "'emission' command should have 3 floating point parameters."),
    ("size 8 8\nshininess 1 2\n", "'shininess' command should have 1 floating point parameter."),
    ("size 8 8\noutput a b\n", "'output' command should have 1 parameter."),
    ("size 8 8\noutput " + "x" * 256 + "\n", "'output' name is too long."),
]


@pytest.mark.parametrize("text,message", ERRORS, ids=[e[1][:40] for e in ERRORS])
def test_error_messages(tmp_path, text, message):
    with pytest.raises(api.RtError) as e:
        _parse_text(tmp_path, text)
    assert str(e.value) == message


def test_missing_file():
    with pytest.raises(api.RtError) as e:
        api.FileInfo("/nonexistent/scene.test")
    assert str(e.value) == "Failed to open input file."


def test_defaults_and_comments(tmp_path):
    fi = _parse_text(tmp_path, "# comment\n\nsize 32 16\n#x\ncamera 0 0 5 0 0 0 0 1 0 40\n")
    assert (fi.width, fi.height, fi.max_recursion_depth, fi.output_filename) == (32, 16, 5, "raytrace.png")
    d = fi.scene_desc
    assert list(d.ambient) == pytest.approx([0.2, 0.2, 0.2]) and list(d.attenuation) == [1.0, 0.0, 0.0]
    assert list(d.bounding_box) == [0.0] * 6            # the box starts as the point (0, 0, 0)
    assert d.n_triangle8 == 0 and d.has_directional_light == 0
    cam = fi.camera11
    assert cam[9] == np.float32(32) * np.float32(40) / np.float32(16)   # fov.x = width * fov.y / height


def test_last_directional_wins_points_append_materials_snapshot(tmp_path):
    text = ("size 8 8\nmaxverts 3\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n"
            "directional 0 0 2 1 0 0\ndirectional 0 3 0 0 1 0\npoint 1 1 1 .5 .5 .5\npoint 2 2 2 .25 .25 .25\n"
            "diffuse .1 .2 .3\ntri 0 1 2\ndiffuse .9 .9 .9\nshininess 7\ntri 0 2 1\n")
    fi = _parse_text(tmp_path, text)
    d = fi.scene_desc
    assert list(d.directional_light) == [0.0, 1.0, 0.0, 0.0, 1.0, 0.0]
    assert d.n_point_lights == 2 and d.n_triangles == 2 and d.n_triangle8 == 1
    mats = np.ctypeslib.as_array(d.triangle_materials, shape=(20,))
    assert mats[0:3].tolist() == pytest.approx([0.1, 0.2, 0.3]) and mats[9] == 0.0
    assert mats[10:13].tolist() == pytest.approx([0.9, 0.9, 0.9]) and mats[19] == 7.0
    t8 = np.ctypeslib.as_array(d.triangle8s, shape=(9, 8))
    assert not t8[:, 2:].any()                          # lanes 2..7 of the only batch are zero padding


def test_sphere_versus_ellipsoid_classification(tmp_path):
    text = ("size 8 8\nsphere 1 2 3 2\npushTransform\nscale 2 2 2\nsphere 0 0 0 1\npopTransform\n"
            "pushTransform\ntranslate 0 1 0\nscale 1 2 3\nsphere 0 0 0 1\npopTransform\n")
    fi = _parse_text(tmp_path, text)
    d = fi.scene_desc
    assert d.n_spheres == 2 and d.n_ellipsoids == 1
    s8 = np.ctypeslib.as_array(d.sphere8s, shape=(4, 8))
    assert s8[:, 0].tolist() == [1.0, 2.0, 3.0, 2.0] and s8[:, 1].tolist() == [0.0, 0.0, 0.0, 2.0]
    inv = np.ctypeslib.as_array(d.ellipsoid8_inverse_transforms, shape=(3, 4, 8))[:, :, 0]
    assert np.allclose(np.diag(inv[:, :3]), [1.0, 0.5, 1.0 / 3.0]) and np.isclose(inv[1, 3], -0.5)


def test_set_size_recomputes_fov_x():
    fi = api.FileInfo(scene_path("scene7"))
    fi.set_size(3840, 2160)
    assert (fi.width, fi.height) == (3840, 2160)
    assert fi.camera11[9] == np.float32(3840) * np.float32(45) / np.float32(2160)   # 80 degrees, linear in the aspect ratio


def test_png_writer_roundtrip(tmp_path):
    from test_integration_binding import read_png_rgb8     # checks the signature, every chunk's CRC and the header, undoes the row filters
    rgb = (np.arange(7 * 5 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    path = str(tmp_path / "out.png")
    api.write_png(path, rgb)
    assert np.array_equal(read_png_rgb8(path), rgb)
