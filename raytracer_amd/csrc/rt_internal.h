// Internal declarations shared by the translation units of librt_amd.so.
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H

#include <cstdint>
#include <vector>

#include "../../include/rt_amd.h"

enum rt_error_code {
    RT_OK = 0,
    RT_ERR_ARGUMENT = 1,
    RT_ERR_PARSE = 2,
    RT_ERR_DEVICE = 3,
    RT_ERR_IO = 4,
    RT_ERR_LIMIT = 5,
};

// Records `message` as this thread's last error and returns `code`.
int rt_fail(int code, const char* message);

// A parsed `.test` file: the contents of the reference's FileInfo (Source/input_parsing.h:8-13).
struct rt_file_info {
    std::vector<float> triangle8s, triangle_materials;
    std::vector<float> sphere8s, sphere_materials;
    std::vector<float> ellipsoid8_inverse_transforms, ellipsoid8_transforms, ellipsoid_materials;
    std::vector<float> point_lights;
    uint64_t n_triangles = 0, n_spheres = 0, n_ellipsoids = 0;
    rt_scene_desc desc;      // pointers into the vectors above
    uint32_t width = 0, height = 0;
    int32_t max_recursion_depth = 5;
    float camera[11] = {0};  // eye, look_at, up, field_of_view x, y
    char filename[256];
};

#endif
