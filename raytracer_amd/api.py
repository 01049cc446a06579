"""Thin Python binding of librt_amd.so (include/rt_amd.h) for the tests and the benchmark.

It mirrors the reference's flow - parse_input_file -> camera set-up -> render -> image
(Source/main.cpp:33-93) - and adds nothing of its own: every function here is one call
through the C ABI.  There is no CPU fallback; a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "lib", "librt_amd.so")
_lib = None


class RtError(RuntimeError):
    pass


def lib():
    """The C-ABI library, loaded once.  Raises if it was not built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C raytracer_amd/csrc`" % LIB_PATH)
        _lib = abi.bind(C.CDLL(LIB_PATH))
    return _lib


def _check(rc):
    if rc != 0:
        raise RtError(lib().rt_last_error().decode("utf-8", "replace"))


def init(device=0):
    _check(lib().rt_init(int(device)))


class FileInfo:
    """A parsed `.test` file: the reference's FileInfo (Source/input_parsing.h:8-13)."""

    def __init__(self, filename):
        h = C.c_void_p()
        _check(lib().rt_parse_input_file(os.fsencode(filename), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().rt_file_info_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def scene_desc(self):
        return lib().rt_file_info_scene(self._h).contents

    @property
    def width(self):
        return lib().rt_file_info_width(self._h)

    @property
    def height(self):
        return lib().rt_file_info_height(self._h)

    @property
    def max_recursion_depth(self):
        return lib().rt_file_info_max_recursion_depth(self._h)

    @property
    def output_filename(self):
        return lib().rt_file_info_output_filename(self._h).decode()

    @property
    def camera11(self):
        cam = (C.c_float * 11)()
        lib().rt_file_info_camera(self._h, C.byref(cam))
        return cam

    def set_size(self, width, height):
        lib().rt_file_info_set_size(self._h, width, height)

    def camera(self):
        """Camera basis and half extents as main() computes them (Source/main.cpp:44-57)."""
        out = abi.CameraDesc()
        cam = self.camera11
        lib().rt_camera_setup(C.byref(cam), C.byref(out))
        return out

    def params(self, **overrides):
        p = abi.RenderParams(width=self.width, height=self.height, max_bounce_count=self.max_recursion_depth)
        for k, v in overrides.items():
            setattr(p, k, v)
        return p


class Scene:
    """Device-resident scene (rt_scene)."""

    def __init__(self, desc):
        h = C.c_void_p()
        _check(lib().rt_scene_create(C.byref(desc), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, params, want_ids=False):
        """rt_render: returns dict(rgb[h,w,3] uint8, ids[h,w,16] uint32 or None, stats)."""
        rw = params.rect_w or params.width
        rh = params.rect_h if params.rect_w else params.height
        rgb = np.zeros((rh, rw, 3), np.uint8)
        ids = np.zeros((rh, rw, 16), np.uint32) if want_ids else None
        stats = abi.Stats()
        _check(lib().rt_render(self._h, C.byref(camera), C.byref(params), rgb.ctypes.data,
                               ids.ctypes.data if want_ids else None, C.byref(stats)))
        return {"rgb": rgb, "ids": ids, "stats": stats}

    def render_tiles_device(self, camera, params, tiles_dev_ptr, want_stats=True):
        """rt_render_tiles_device into caller-owned device memory (an integer address)."""
        stats = abi.Stats()
        _check(lib().rt_render_tiles_device(self._h, C.byref(camera), C.byref(params), C.c_void_p(tiles_dev_ptr),
                                            C.byref(stats) if want_stats else None))
        return stats


def render_tiles_begin(scene, camera, params, tiles_dev_ptr):
    """rt_render_tiles_begin: starts a frame, returns its ticket."""
    ticket = C.c_int(-1)
    _check(lib().rt_render_tiles_begin(scene._h, C.byref(camera), C.byref(params), C.c_void_p(tiles_dev_ptr), C.byref(ticket)))
    return ticket.value


def render_tiles_end(ticket):
    """rt_render_tiles_end: waits for the frame, returns its statistics."""
    stats = abi.Stats()
    _check(lib().rt_render_tiles_end(ticket, C.byref(stats)))
    return stats


def render_tiles_noalloc(scene, camera, params, tiles_dev_ptr):
    """rt_render_tiles_noalloc into caller-owned device memory (an integer address)."""
    _check(lib().rt_render_tiles_noalloc(scene._h, C.byref(camera), C.byref(params), C.c_void_p(tiles_dev_ptr)))


def init_devices(n):
    _check(lib().rt_init_devices(int(n)))


def render_multi(scene, camera, params, n_devices):
    """rt_render_multi: the image of the rectangle traced by all devices of rt_init_devices; returns dict(rgb, stats per device)."""
    rw = params.rect_w or params.width
    rh = params.rect_h if params.rect_w else params.height
    rgb = np.zeros((rh, rw, 3), np.uint8)
    stats = (abi.Stats * n_devices)()
    _check(lib().rt_render_multi(scene._h, C.byref(camera), C.byref(params), rgb.ctypes.data, stats))
    return {"rgb": rgb, "stats": list(stats)}


def render_multi_begin(scene, camera, params, rgb_dev_ptr, want_stats=True):
    """rt_render_multi_begin: every device traces its share, the exchange into rgb_dev (memory of device 0) is enqueued; returns the ticket."""
    ticket = C.c_int(-1)
    _check(lib().rt_render_multi_begin(scene._h, C.byref(camera), C.byref(params), C.c_void_p(rgb_dev_ptr), 1 if want_stats else 0, C.byref(ticket)))
    return ticket.value


def render_multi_end(ticket, n_devices):
    """rt_render_multi_end: waits for the frame's exchange; returns (per-device statistics, device milliseconds of gather + assembly)."""
    stats = (abi.Stats * n_devices)()
    exchange_ms = C.c_double(0.0)
    _check(lib().rt_render_multi_end(int(ticket), stats, C.byref(exchange_ms)))
    return list(stats), exchange_ms.value


def comm_unique_id():
    buf = (C.c_uint8 * 128)()
    _check(lib().rt_comm_unique_id(buf))
    return bytes(buf)


def comm_init_rank(unique_id, rank, world):
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    _check(lib().rt_comm_init_rank(buf, int(rank), int(world)))


def comm_gather_tiles(tiles_ptr, tiles_per_rank_stride, gathered_ptr):
    _check(lib().rt_comm_gather_tiles(C.c_void_p(tiles_ptr), tiles_per_rank_stride, C.c_void_p(gathered_ptr) if gathered_ptr else None))


def comm_wait():
    _check(lib().rt_comm_wait())


def comm_destroy():
    lib().rt_comm_destroy()


def rank_tile_count(params):
    return lib().rt_rank_tile_count(C.byref(params))


def total_tile_count(params):
    return lib().rt_total_tile_count(C.byref(params))


def assemble_tiles_device(gathered_ptr, tiles_per_rank_stride, params, rgb_ptr):
    _check(lib().rt_assemble_tiles_device(C.c_void_p(gathered_ptr), tiles_per_rank_stride, C.byref(params), C.c_void_p(rgb_ptr)))


def write_png(filename, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(lib().rt_write_png(os.fsencode(filename), rgb.shape[1], rgb.shape[0], rgb.ctypes.data))


def write_png_begin(filename, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(lib().rt_write_png_begin(os.fsencode(filename), rgb.shape[1], rgb.shape[0], rgb.ctypes.data))


def write_png_end():
    _check(lib().rt_write_png_end())


# floats per item of `a`, fixed floats of `a`, floats per item of `b`, floats per item of the result (rt_debug_eval in include/rt_amd.h)
_DEBUG_SHAPES = {
    abi.RT_DEBUG_FP_POW: (1, 0, 1, 1), abi.RT_DEBUG_FP_MIN: (1, 0, 1, 1), abi.RT_DEBUG_FP_MAX: (1, 0, 1, 1), abi.RT_DEBUG_DIVIDE: (1, 0, 1, 1),
    abi.RT_DEBUG_DIVIDE_IN_RANGE: (1, 0, 1, 1), abi.RT_DEBUG_PRIMARY_RAY: (0, 16, 4, 3), abi.RT_DEBUG_SCENE_BOX: (0, 6, 6, 2),
    abi.RT_DEBUG_TRIANGLE: (9, 0, 6, 1), abi.RT_DEBUG_TRIANGLE_STAGED: (9, 0, 6, 1), abi.RT_DEBUG_SPHERE: (4, 0, 6, 1),
    abi.RT_DEBUG_ELLIPSOID: (24, 0, 6, 1), abi.RT_DEBUG_ELLIPSOID_STAGED: (24, 0, 6, 2), abi.RT_DEBUG_ELLIPSOID_SHADOW: (16, 0, 6, 1),
    abi.RT_DEBUG_ELLIPSOID_SHADOW_STAGED: (16, 0, 6, 2), abi.RT_DEBUG_ELLIPSOID_PARTS: (12, 0, 6, 8), abi.RT_DEBUG_TRIANGLE_PRETEST: (9, 0, 6, 3),
}


def debug_eval(op, a, b, out_per_item=None, n=None):
    """rt_debug_eval: one device routine on n items; a, b are float32 arrays laid out as include/rt_amd.h says.

    n defaults to the number of items `b` holds (its size over the operation's floats per item, whatever its shape); the sizes
    of both arrays are checked against the operation's layout before the library reads them."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if int(op) not in _DEBUG_SHAPES:
        raise ValueError("rt_debug_eval: unknown operation %r" % (op,))
    a_item, a_fixed, b_item, out_item = _DEBUG_SHAPES[int(op)]
    if out_per_item is not None and out_per_item != out_item:
        raise ValueError("operation %d returns %d floats per item, not %d" % (op, out_item, out_per_item))
    if n is None:
        if b.size % b_item:
            raise ValueError("b holds %d floats, not a multiple of %d per item" % (b.size, b_item))
        n = b.size // b_item
    if b.size < n * b_item or a.size < n * a_item + a_fixed:
        raise ValueError("rt_debug_eval: %d items need %d floats in a and %d in b (got %d, %d)" % (n, n * a_item + a_fixed, n * b_item, a.size, b.size))
    out = np.zeros(n * out_item, np.float32)
    _check(lib().rt_debug_eval(int(op), n, a.ctypes.data, b.ctypes.data, out.ctypes.data))
    return out
