// TEST INFRASTRUCTURE ONLY - never linked into, imported by or called from the product path.
//
// The maintainer's binding, compiled: the reference's own front end
// (parse_input_file, compiled from the unmodified sources where they lie, as in ref_harness.cpp)
// feeding librt_amd.so through the two code blocks a maintainer adds to Source/main.cpp.
// tests/test_integration_binding.py checks that the blocks here are the blocks of the integration guide at the
// repository root word for word, and (on a GPU box) that the image this flow renders is the reference's.
// Build: oracle/Makefile, target `binding` -> oracle/_ref/librt_binding.so (needs the reference sources).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define TIME_BLOCK(block_name)

#include "maths.h"
#include "ray_tracing.h"
#include "input_parsing.h"
#include "render_work_queue.h"

#include "maths.cpp"
#include "ray_tracing.cpp"
#include "input_parsing.cpp"
#include "render_work_queue.cpp"

// [binding block 1: begin]
// after the other includes (main.cpp:10-13)
#include "rt_amd.h"        // link with -lrt_amd

// helper: view the parsed Scene as an rt_scene_desc (no copies; ray_tracing.h:88-106)
static rt_scene_desc describe(const Scene& s) {
    rt_scene_desc d = {};
    d.triangle8s                    = reinterpret_cast<const float*>(s.triangle8s.data());
    d.triangle_materials            = reinterpret_cast<const float*>(s.triangle_materials.data());
    d.sphere8s                      = reinterpret_cast<const float*>(s.sphere8s.data());
    d.sphere_materials              = reinterpret_cast<const float*>(s.sphere_materials.data());
    d.ellipsoid8_inverse_transforms = reinterpret_cast<const float*>(s.ellipsoid8_inverse_transforms.data());
    d.ellipsoid8_transforms         = reinterpret_cast<const float*>(s.ellipsoid8_transforms.data());
    d.ellipsoid_materials           = reinterpret_cast<const float*>(s.ellipsoid_materials.data());
    d.point_light_sources           = reinterpret_cast<const float*>(s.point_light_sources.data());
    d.n_triangle8 = s.triangle8s.size();              d.n_triangles  = s.triangle_materials.size();
    d.n_sphere8   = s.sphere8s.size();                d.n_spheres    = s.sphere_materials.size();
    d.n_ellipsoid8 = s.ellipsoid8_transforms.size();  d.n_ellipsoids = s.ellipsoid_materials.size();
    d.n_point_lights = s.point_light_sources.size();
    const AxisAlignedBoundingBox& b = s.bounding_box;
    const float bb[6] = {b.min_x, b.max_x, b.min_y, b.max_y, b.min_z, b.max_z};
    std::memcpy(d.bounding_box, bb, sizeof bb);
    std::memcpy(d.ambient, &s.ambient, 12);
    std::memcpy(d.attenuation, &s.attenuation_parameters, 12);
    std::memcpy(d.directional_light, &s.directional_light_source, 24);
    d.has_directional_light = s.has_directional_light_source;
    return d;
}
// [binding block 1: end]

// The flow of the reference's main() (Source/main.cpp:33-57: parse, camera basis, half image extents) with its render block
// (main.cpp:59-91) replaced by binding block 2.  Instead of writing a PNG (main.cpp:93) the pixels are handed to the caller.
// Returns 0, or -1 as main() does; *width and *height receive the image size; pixels_out (nullable) receives width*height*3 bytes
// when capacity allows.
extern "C" int binding_main(const char* scene_file, uint8_t* pixels_out, size_t capacity, uint32_t* width, uint32_t* height) {
    const ParseInputFileResult file_parsing_result = parse_input_file(scene_file);
    if (file_parsing_result.error != nullptr) {
        std::printf("Failed to parse input file. %s\n", file_parsing_result.error);
        return -1;
    }
    const FileInfo& file_info = file_parsing_result.file_info;
    const Scene& scene = file_info.scene;
    const Image& image = file_info.image;
    const Camera& camera = file_info.camera;
    const Vector camera_basis_k = normalise(camera.look_at - camera.eye);
    const Vector camera_basis_i = normalise(camera.up ^ camera_basis_k);
    const Vector camera_basis_j = camera_basis_k ^ camera_basis_i;
    const Dimensions half_image_dimensions_world{std::tan(0.5f * to_radians(camera.field_of_view.x)), std::tan(0.5f * to_radians(camera.field_of_view.y))};

    // [binding block 2: begin]
    rt_camera_desc cam;                       // eye, basis i/j/k, tan(fov/2): exactly main.cpp:44-57
    std::memcpy(cam.eye, &camera.eye, 12);
    std::memcpy(cam.basis_i, &camera_basis_i, 12);
    std::memcpy(cam.basis_j, &camera_basis_j, 12);
    std::memcpy(cam.basis_k, &camera_basis_k, 12);
    cam.half_world[0] = half_image_dimensions_world.width;
    cam.half_world[1] = half_image_dimensions_world.height;

    const rt_scene_desc desc = describe(scene);
    rt_render_params params = {};
    params.width = image.width; params.height = image.height;
    params.max_bounce_count = file_info.max_recursion_depth;

    rt_scene* gpu_scene = nullptr;
    if (rt_init(0) || rt_scene_create(&desc, &gpu_scene) ||
        rt_render(gpu_scene, &cam, &params, image.pixels, nullptr, nullptr)) {
        std::printf("GPU render failed. %s\n", rt_last_error());
        return -1;
    }
    rt_scene_destroy(gpu_scene);
    // [binding block 2: end]

    *width = image.width;
    *height = image.height;
    const size_t bytes = static_cast<size_t>(image.width) * image.height * 3;
    if (pixels_out && bytes <= capacity) std::memcpy(pixels_out, image.pixels, bytes);
    delete[] image.pixels;      // the parser allocates the pixels and the reference never frees them (input_parsing.cpp:162)
    return 0;
}
