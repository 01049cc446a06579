"""ctypes mirror of include/rt_amd.h (struct layouts and prototypes only, no logic)."""
import ctypes as C

c_float_p = C.POINTER(C.c_float)


class SceneDesc(C.Structure):
    """rt_scene_desc: flat views of the reference's `Scene` (Source/ray_tracing.h:88-106)."""
    _fields_ = [
        ("triangle8s", c_float_p),
        ("triangle_materials", c_float_p),
        ("sphere8s", c_float_p),
        ("sphere_materials", c_float_p),
        ("ellipsoid8_inverse_transforms", c_float_p),
        ("ellipsoid8_transforms", c_float_p),
        ("ellipsoid_materials", c_float_p),
        ("point_light_sources", c_float_p),
        ("n_triangle8", C.c_uint64),
        ("n_triangles", C.c_uint64),
        ("n_sphere8", C.c_uint64),
        ("n_spheres", C.c_uint64),
        ("n_ellipsoid8", C.c_uint64),
        ("n_ellipsoids", C.c_uint64),
        ("n_point_lights", C.c_uint64),
        ("bounding_box", C.c_float * 6),
        ("ambient", C.c_float * 3),
        ("attenuation", C.c_float * 3),
        ("directional_light", C.c_float * 6),
        ("has_directional_light", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class CameraDesc(C.Structure):
    """rt_camera_desc: eye, camera basis and tan(fov/2) (Source/main.cpp:44-57)."""
    _fields_ = [
        ("eye", C.c_float * 3),
        ("basis_i", C.c_float * 3),
        ("basis_j", C.c_float * 3),
        ("basis_k", C.c_float * 3),
        ("half_world", C.c_float * 2),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max_bounce_count", C.c_int32),
        ("rect_x0", C.c_uint32),
        ("rect_y0", C.c_uint32),
        ("rect_w", C.c_uint32),
        ("rect_h", C.c_uint32),
        ("tile_rank", C.c_uint32),
        ("tile_world", C.c_uint32),
        ("chunk_tiles", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("culled", C.c_uint64),
        ("closest_rays", C.c_uint64),
        ("hits", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shadow_blocked", C.c_uint64),
        ("closest_tests", C.c_uint64 * 3),
        ("shadow_tests", C.c_uint64 * 3),
        ("device_ms", C.c_double),
        ("closest_ms", C.c_double),
        ("shadow_ms", C.c_double),
        ("closest_launches", C.c_uint64),
        ("shadow_launches", C.c_uint64),
        ("full_tests", C.c_uint64),
        ("pair_pretests", C.c_uint64 * 2),
        ("queued_pairs", C.c_uint64 * 2),
        ("executed_flop", C.c_uint64 * 2),
        ("closest_kind_ms", C.c_double * 3),
        ("shadow_kind_ms", C.c_double * 3),
        ("closest_kind_launches", C.c_uint64 * 3),
        ("shadow_kind_launches", C.c_uint64 * 3),
        ("walk_waves", C.c_uint64 * 2),
        ("first_level_nodes", C.c_uint64 * 2),
        ("walk_steps", C.c_uint64 * 2),
        ("brute_wave_steps", (C.c_uint64 * 4) * 2),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = [list(x) if hasattr(x, "__len__") else x for x in v] if hasattr(v, "__len__") else v
        return out


RT_TILE = 16
RT_SAMPLES_PER_PIXEL = 16
RT_NOALLOC_MAX_TILES = 64
RT_FLAG_NO_FILTER = 1
RT_FLAG_WALK_ALL = 2
RT_FLAG_DEEP_WALK = 4
RT_FLAG_PLAIN_TEST = 8
RT_FLAG_EXACT_STAGES = 16
(RT_DEBUG_FP_POW, RT_DEBUG_FP_MIN, RT_DEBUG_FP_MAX, RT_DEBUG_DIVIDE, RT_DEBUG_DIVIDE_IN_RANGE, RT_DEBUG_PRIMARY_RAY,
 RT_DEBUG_SCENE_BOX, RT_DEBUG_TRIANGLE, RT_DEBUG_TRIANGLE_STAGED, RT_DEBUG_SPHERE, RT_DEBUG_ELLIPSOID, RT_DEBUG_ELLIPSOID_STAGED,
 RT_DEBUG_ELLIPSOID_SHADOW, RT_DEBUG_ELLIPSOID_SHADOW_STAGED, RT_DEBUG_ELLIPSOID_PARTS, RT_DEBUG_TRIANGLE_PRETEST) = range(16)
RT_ID_MISS = 0xFFFFFFFF
RT_ID_CULLED = 0xFFFFFFFE

# Every symbol include/rt_amd.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rt_render_tiles_device": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    "rt_render_tiles_begin": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(C.c_int)]),
    "rt_render_tiles_end": (C.c_int, [C.c_int, C.POINTER(Stats)]),
    "rt_render_tiles_noalloc": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p]),
    "rt_rank_tile_count": (C.c_uint32, [C.POINTER(RenderParams)]),
    "rt_total_tile_count": (C.c_uint32, [C.POINTER(RenderParams)]),
    "rt_assemble_tiles_device": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(RenderParams), C.c_void_p]),
    "rt_init_devices": (C.c_int, [C.c_int]),
    "rt_render_multi": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    "rt_render_multi_begin": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "rt_render_multi_end": (C.c_int, [C.c_int, C.POINTER(Stats), C.POINTER(C.c_double)]),
    "rt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rt_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_comm_gather_tiles": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rt_comm_wait": (C.c_int, []),
    "rt_comm_destroy": (None, []),
    "rt_parse_input_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rt_file_info_destroy": (None, [C.c_void_p]),
    "rt_file_info_scene": (C.POINTER(SceneDesc), [C.c_void_p]),
    "rt_file_info_width": (C.c_uint32, [C.c_void_p]),
    "rt_file_info_height": (C.c_uint32, [C.c_void_p]),
    "rt_file_info_max_recursion_depth": (C.c_int32, [C.c_void_p]),
    "rt_file_info_output_filename": (C.c_char_p, [C.c_void_p]),
    "rt_file_info_camera": (None, [C.c_void_p, C.POINTER(C.c_float * 11)]),
    "rt_file_info_set_size": (None, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rt_camera_setup": (None, [C.POINTER(C.c_float * 11), C.POINTER(CameraDesc)]),
    "rt_render_file_info": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rt_write_png": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_write_png_begin": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_write_png_end": (C.c_int, []),
    "rt_debug_eval": (C.c_int, [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def bind(lib, prototypes=PROTOTYPES):
    """Attach restype/argtypes to every prototype; raises AttributeError on a missing symbol."""
    for name, (restype, argtypes) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
