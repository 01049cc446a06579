// TEST INFRASTRUCTURE ONLY - never linked into, imported by or called from the product path.
//
// Harness around the UNMODIFIED reference sources.  It is compiled from the
// reference's own files where they lie (include path given at build time, see
// oracle/Makefile); nothing of the reference is copied into this repository.
// The result, oracle/_ref/librt_ref.so, is used to pin the CPU restatement in
// oracle/rt_oracle.cpp and to generate the golden fixtures under tests/golden/.
//
// The reference's main.cpp cannot be built on Linux (profiling.cpp includes
// <Windows.h>), and no stand-in for that header is written here.  The render
// path does not need it: this translation unit includes the four source files
// of the path in the order main.cpp uses (Source/main.cpp:10-23), leaves the
// Win32 profiler out, and defines TIME_BLOCK as the empty macro the reference
// itself uses whenever PROFILING is off (Source/profiling.cpp:131-135).
// FLT_MAX, which the reference picks up through <Windows.h>, comes from <cfloat>.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
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

#include "../include/rt_amd.h"

static_assert(sizeof(Triangle8) == 288 && sizeof(Sphere8) == 128 && sizeof(Mat3x4x8) == 384 && sizeof(Material) == 40, "layout");

namespace {

Scene scene_from_desc(const rt_scene_desc& d) {
    Scene s;
    s.triangle8s.resize(d.n_triangle8);
    if (d.n_triangle8) std::memcpy(s.triangle8s.data(), d.triangle8s, d.n_triangle8 * sizeof(Triangle8));
    s.triangle_materials.resize(d.n_triangles);
    if (d.n_triangles) std::memcpy(s.triangle_materials.data(), d.triangle_materials, d.n_triangles * sizeof(Material));
    s.sphere8s.resize(d.n_sphere8);
    if (d.n_sphere8) std::memcpy(s.sphere8s.data(), d.sphere8s, d.n_sphere8 * sizeof(Sphere8));
    s.sphere_materials.resize(d.n_spheres);
    if (d.n_spheres) std::memcpy(s.sphere_materials.data(), d.sphere_materials, d.n_spheres * sizeof(Material));
    s.ellipsoid8_inverse_transforms.resize(d.n_ellipsoid8);
    s.ellipsoid8_transforms.resize(d.n_ellipsoid8);
    if (d.n_ellipsoid8) {
        std::memcpy(s.ellipsoid8_inverse_transforms.data(), d.ellipsoid8_inverse_transforms, d.n_ellipsoid8 * sizeof(Mat3x4x8));
        std::memcpy(s.ellipsoid8_transforms.data(), d.ellipsoid8_transforms, d.n_ellipsoid8 * sizeof(Mat3x4x8));
    }
    s.ellipsoid_materials.resize(d.n_ellipsoids);
    if (d.n_ellipsoids) std::memcpy(s.ellipsoid_materials.data(), d.ellipsoid_materials, d.n_ellipsoids * sizeof(Material));
    s.bounding_box = AxisAlignedBoundingBox{d.bounding_box[0], d.bounding_box[1], d.bounding_box[2], d.bounding_box[3], d.bounding_box[4], d.bounding_box[5]};
    s.ambient = Colour{d.ambient[0], d.ambient[1], d.ambient[2]};
    s.attenuation_parameters = AttenuationParameters{d.attenuation[0], d.attenuation[1], d.attenuation[2]};
    s.point_light_sources.resize(d.n_point_lights);
    if (d.n_point_lights) std::memcpy(s.point_light_sources.data(), d.point_light_sources, d.n_point_lights * sizeof(PointLightSource));
    s.directional_light_source = DirectionalLightSource{
        Vector{d.directional_light[0], d.directional_light[1], d.directional_light[2]},
        Colour{d.directional_light[3], d.directional_light[4], d.directional_light[5]}};
    s.has_directional_light_source = d.has_directional_light != 0;
    return s;
}

// Scene whose colour after one bounce encodes (kind, index + 1) of the hit:
// with no lights and zero ambient, intersect() returns exactly the emission
// of the hit primitive (Source/ray_tracing.cpp:682).
Scene id_scene_from(const Scene& src) {
    Scene s = src;
    s.ambient = Colour{0.0f, 0.0f, 0.0f};
    s.point_light_sources.clear();
    s.has_directional_light_source = false;
    for (std::size_t i = 0; i < s.triangle_materials.size(); ++i) s.triangle_materials[i].emission = Colour{static_cast<float>(i + 1), 0.0f, 0.0f};
    for (std::size_t i = 0; i < s.sphere_materials.size(); ++i) s.sphere_materials[i].emission = Colour{0.0f, static_cast<float>(i + 1), 0.0f};
    for (std::size_t i = 0; i < s.ellipsoid_materials.size(); ++i) s.ellipsoid_materials[i].emission = Colour{0.0f, 0.0f, static_cast<float>(i + 1)};
    return s;
}

struct Parsed {
    ParseInputFileResult result;
    rt_scene_desc desc;
};

void fill_desc(const Scene& s, rt_scene_desc& d) {
    std::memset(&d, 0, sizeof d);
    d.triangle8s = reinterpret_cast<const float*>(s.triangle8s.data());
    d.triangle_materials = reinterpret_cast<const float*>(s.triangle_materials.data());
    d.sphere8s = reinterpret_cast<const float*>(s.sphere8s.data());
    d.sphere_materials = reinterpret_cast<const float*>(s.sphere_materials.data());
    d.ellipsoid8_inverse_transforms = reinterpret_cast<const float*>(s.ellipsoid8_inverse_transforms.data());
    d.ellipsoid8_transforms = reinterpret_cast<const float*>(s.ellipsoid8_transforms.data());
    d.ellipsoid_materials = reinterpret_cast<const float*>(s.ellipsoid_materials.data());
    d.point_light_sources = reinterpret_cast<const float*>(s.point_light_sources.data());
    d.n_triangle8 = s.triangle8s.size();
    d.n_triangles = s.triangle_materials.size();
    d.n_sphere8 = s.sphere8s.size();
    d.n_spheres = s.sphere_materials.size();
    d.n_ellipsoid8 = s.ellipsoid8_inverse_transforms.size();
    d.n_ellipsoids = s.ellipsoid_materials.size();
    d.n_point_lights = s.point_light_sources.size();
    const AxisAlignedBoundingBox& b = s.bounding_box;
    const float bb[6] = {b.min_x, b.max_x, b.min_y, b.max_y, b.min_z, b.max_z};
    std::memcpy(d.bounding_box, bb, sizeof bb);
    d.ambient[0] = s.ambient.red; d.ambient[1] = s.ambient.green; d.ambient[2] = s.ambient.blue;
    d.attenuation[0] = s.attenuation_parameters.constant; d.attenuation[1] = s.attenuation_parameters.linear; d.attenuation[2] = s.attenuation_parameters.quadratic;
    const DirectionalLightSource& dl = s.directional_light_source;
    const float dd[6] = {dl.direction.x, dl.direction.y, dl.direction.z, dl.colour.red, dl.colour.green, dl.colour.blue};
    std::memcpy(d.directional_light, dd, sizeof dd);
    d.has_directional_light = s.has_directional_light_source ? 1 : 0;
}

}  // namespace

extern "C" {

// ---- parser -----------------------------------------------------------------
void* ref_parse(const char* filename, const char** error) {
    Parsed* p = new Parsed{parse_input_file(filename), {}};
    if (error) *error = p->result.error;
    if (p->result.error != nullptr) { delete p; return nullptr; }
    fill_desc(p->result.file_info.scene, p->desc);
    return p;
}
void ref_parse_free(void* h) {
    Parsed* p = static_cast<Parsed*>(h);
    if (p) { delete[] p->result.file_info.image.pixels; delete p; }
}
const rt_scene_desc* ref_parse_scene(void* h) { return &static_cast<Parsed*>(h)->desc; }
uint32_t ref_parse_width(void* h) { return static_cast<Parsed*>(h)->result.file_info.image.width; }
uint32_t ref_parse_height(void* h) { return static_cast<Parsed*>(h)->result.file_info.image.height; }
int32_t ref_parse_max_depth(void* h) { return static_cast<Parsed*>(h)->result.file_info.max_recursion_depth; }
const char* ref_parse_filename(void* h) { return static_cast<Parsed*>(h)->result.file_info.image.filename; }
void ref_parse_camera(void* h, float cam[11]) {
    const Camera& c = static_cast<Parsed*>(h)->result.file_info.camera;
    const float v[11] = {c.eye.x, c.eye.y, c.eye.z, c.look_at.x, c.look_at.y, c.look_at.z, c.up.x, c.up.y, c.up.z, c.field_of_view.x, c.field_of_view.y};
    std::memcpy(cam, v, sizeof v);
}

// ---- camera set-up, with the reference's own operators, as main.cpp:44-57 ----
void ref_camera_setup(const float cam[11], rt_camera_desc* out) {
    const Vector eye{cam[0], cam[1], cam[2]}, look_at{cam[3], cam[4], cam[5]}, up{cam[6], cam[7], cam[8]};
    const Vector k = normalise(look_at - eye);
    const Vector i = normalise(up ^ k);
    const Vector j = k ^ i;
    out->eye[0] = eye.x; out->eye[1] = eye.y; out->eye[2] = eye.z;
    out->basis_i[0] = i.x; out->basis_i[1] = i.y; out->basis_i[2] = i.z;
    out->basis_j[0] = j.x; out->basis_j[1] = j.y; out->basis_j[2] = j.z;
    out->basis_k[0] = k.x; out->basis_k[1] = k.y; out->basis_k[2] = k.z;
    out->half_world[0] = std::tan(0.5f * to_radians(cam[9]));
    out->half_world[1] = std::tan(0.5f * to_radians(cam[10]));
}

// ---- scalar known-answer hooks -------------------------------------------------
float ref_fp_pow(float x, float y) { return fp_pow(x, y); }
float ref_fp_min(float x, float y) { return fp_min(x, y); }
float ref_fp_max(float x, float y) { return fp_max(x, y); }
float ref_fp_sqrt(float x) { return fp_sqrt(x); }
void ref_ray_direction_through_pixel(unsigned px, unsigned py, float xo, float yo, const rt_camera_desc* c, float half_w_px, float half_h_px, float out[3]) {
    const BasisVectors b{Vector{c->basis_i[0], c->basis_i[1], c->basis_i[2]}, Vector{c->basis_j[0], c->basis_j[1], c->basis_j[2]}, Vector{c->basis_k[0], c->basis_k[1], c->basis_k[2]}};
    const Vector d = ray_direction_through_pixel(px, py, xo, yo, b, Dimensions{c->half_world[0], c->half_world[1]}, Dimensions{half_w_px, half_h_px});
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void ref_intersect_aabb(const float o[3], const float d[3], const float bb[6], float out[2]) {
    const AABBIntersectionResult r = intersect(Ray{Vector{o[0], o[1], o[2]}, Vector{d[0], d[1], d[2]}}, AxisAlignedBoundingBox{bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]});
    out[0] = r.min_distance; out[1] = r.max_distance;
}
// one batch of 8: tri8 = 72 floats, out = 8 distances
void ref_intersect_triangle8(const float o[3], const float d[3], const float* tri8, float out[8]) {
    Triangle8 t; std::memcpy(&t, tri8, sizeof t);
    _mm256_storeu_ps(out, intersect(load(Vector{o[0], o[1], o[2]}), load(Vector{d[0], d[1], d[2]}), t));
}
void ref_intersect_sphere8(const float o[3], const float d[3], const float* sph8, float out[8]) {
    Sphere8 s; std::memcpy(&s, sph8, sizeof s);
    _mm256_storeu_ps(out, intersect(load(Vector{o[0], o[1], o[2]}), load(Vector{d[0], d[1], d[2]}), s));
}
// The ellipsoid routines on one batch of 8 (inv8, fwd8: Mat3x4x8, 96 floats each): the reference's own operators and
// functions (Mat3x4AVX * Vec3AVX, transform_direction, ellipsoid_is_absent, magnitude, intersect_with_unit_sphere), called
// in the order of the closest-hit body, Source/ray_tracing.cpp:556-579 (that body is inline in intersect(Ray, Scene, int)
// and cannot be called on its own).  Outputs per lane: o2, d2 as x[8] y[8] z[8]; absent as 1 / 0; the unit-sphere
// distance; the final distance (FLT_MAX where absent).
void ref_ellipsoid8(const float o[3], const float d[3], const float* inv8, const float* fwd8, float* o2, float* d2, float* absent, float* t2, float* dist) {
    Mat3x4x8 inverse, forward;
    std::memcpy(&inverse, inv8, sizeof inverse);
    std::memcpy(&forward, fwd8, sizeof forward);
    const Vec3AVX ray_start = load(Vector{o[0], o[1], o[2]});
    const Vec3AVX ray_direction = load(Vector{d[0], d[1], d[2]});
    const Mat3x4AVX ellipsoid_inverse_transform = load(inverse);
    const Mat3x4AVX ellipsoid_transform = load(forward);
    const Vec3AVX transformed_ray_start = ellipsoid_inverse_transform * ray_start;
    Vec3AVX transformed_ray_direction = transform_direction(ellipsoid_inverse_transform, ray_direction);
    _mm256_storeu_ps(o2, transformed_ray_start.x); _mm256_storeu_ps(o2 + 8, transformed_ray_start.y); _mm256_storeu_ps(o2 + 16, transformed_ray_start.z);
    _mm256_storeu_ps(d2, transformed_ray_direction.x); _mm256_storeu_ps(d2 + 8, transformed_ray_direction.y); _mm256_storeu_ps(d2 + 16, transformed_ray_direction.z);
    const __m256 ellipsoid_absent = ellipsoid_is_absent(transformed_ray_direction);
    _mm256_storeu_ps(absent, _mm256_and_ps(ellipsoid_absent, _mm256_set1_ps(1.0f)));
    const __m256 transformed_ray_direction_magnitude = magnitude(transformed_ray_direction);
    const __m256 inverse_transformed_ray_direction_magnitude = _mm256_div_ps(_mm256_set1_ps(1.0f), transformed_ray_direction_magnitude);
    transformed_ray_direction = inverse_transformed_ray_direction_magnitude * transformed_ray_direction;
    const __m256 transformed_intersection_distances = intersect_with_unit_sphere(transformed_ray_start, transformed_ray_direction);
    _mm256_storeu_ps(t2, transformed_intersection_distances);
    const Vec3AVX transformed_intersection_point = transformed_ray_start + transformed_intersection_distances * transformed_ray_direction;
    const Vec3AVX intersection_point = ellipsoid_transform * transformed_intersection_point;
    const __m256 all_intersection_distances = magnitude(intersection_point - ray_start);
    const __m256 infinity = _mm256_set1_ps(FLT_MAX);
    _mm256_storeu_ps(dist, _mm256_or_ps(_mm256_andnot_ps(ellipsoid_absent, all_intersection_distances), _mm256_and_ps(ellipsoid_absent, infinity)));
}
// path_is_blocked itself (Source/ray_tracing.cpp:347-458) on a scene that holds this one batch of ellipsoids and nothing else.
// point != 0: light[3] is a position; the direction the function derives (:350-354) is returned in dir_out.
// point == 0: light[3] is the directional light's direction; dir_out = -1 * light (:411).
int ref_ellipsoid8_blocks(const float start[3], const float light[3], int point, const float* inv8, float dir_out[3]) {
    Scene s;
    s.ellipsoid8_inverse_transforms.resize(1);
    s.ellipsoid8_transforms.resize(1);
    std::memcpy(&s.ellipsoid8_inverse_transforms[0], inv8, sizeof(Mat3x4x8));
    std::memset(&s.ellipsoid8_transforms[0], 0, sizeof(Mat3x4x8));
    const Vector from{start[0], start[1], start[2]};
    if (point) {
        const PointLightSource l{Vector{light[0], light[1], light[2]}, Colour{1.0f, 1.0f, 1.0f}};
        const Vector start_to_light = l.position - from;
        float lanes[8];
        const __m256 inverse_distance_to_light = _mm256_div_ps(_mm256_set1_ps(1.0f), _mm256_set1_ps(magnitude(start_to_light)));
        const Vec3AVX ray_direction = inverse_distance_to_light * load(start_to_light);
        _mm256_storeu_ps(lanes, ray_direction.x); dir_out[0] = lanes[0];
        _mm256_storeu_ps(lanes, ray_direction.y); dir_out[1] = lanes[0];
        _mm256_storeu_ps(lanes, ray_direction.z); dir_out[2] = lanes[0];
        return path_is_blocked(from, l, s) ? 1 : 0;
    }
    const DirectionalLightSource l{Vector{light[0], light[1], light[2]}, Colour{1.0f, 1.0f, 1.0f}};
    const Vector direction = -1.0f * l.direction;
    dir_out[0] = direction.x; dir_out[1] = direction.y; dir_out[2] = direction.z;
    return path_is_blocked(from, l, s) ? 1 : 0;
}
// full trace of one ray
void ref_trace(const rt_scene_desc* desc, const float o[3], const float d[3], int depth, float out[3]) {
    const Scene s = scene_from_desc(*desc);
    const Colour c = intersect(Ray{Vector{o[0], o[1], o[2]}, Vector{d[0], d[1], d[2]}}, s, depth);
    out[0] = c.red; out[1] = c.green; out[2] = c.blue;
}

// ---- render --------------------------------------------------------------------
// rgb: rect_w*rect_h*3; ids (nullable): rect_w*rect_h*16; sample_colours (nullable): rect_w*rect_h*16*3.
// Work is handed to n_threads workers through an atomic counter, like the reference's queue
// (Source/render_work_queue.cpp:59-80): whole rows, each rendered by the reference's render_scanline into a
// full-width image, when the rectangle spans the image's width; otherwise segments of SEGMENT pixels of a row
// (a crop has few rows: whole rows would leave most workers idle in the last round).
// The seconds each worker spent in its loop are kept for ref_worker_seconds.
namespace { std::vector<double> g_worker_seconds; }
int ref_worker_seconds(double* out, int n) {
    const int have = static_cast<int>(g_worker_seconds.size());
    for (int i = 0; i < n && i < have; ++i) out[i] = g_worker_seconds[static_cast<std::size_t>(i)];
    return have;
}
int ref_render(const rt_scene_desc* desc, const rt_camera_desc* cam, const rt_render_params* prm,
               uint8_t* rgb, uint32_t* ids, float* sample_colours, int n_threads, double* render_seconds) {
    const Scene scene = scene_from_desc(*desc);
    const Scene id_scene = ids ? id_scene_from(scene) : Scene{};
    const unsigned W = prm->width, H = prm->height;
    const unsigned x0 = prm->rect_w ? prm->rect_x0 : 0, y0 = prm->rect_w ? prm->rect_y0 : 0;
    const unsigned rw = prm->rect_w ? prm->rect_w : W, rh = prm->rect_w ? prm->rect_h : H;
    const bool whole_rows = (x0 == 0 && rw == W);

    Camera camera{};
    camera.eye = Vector{cam->eye[0], cam->eye[1], cam->eye[2]};
    const BasisVectors basis{Vector{cam->basis_i[0], cam->basis_i[1], cam->basis_i[2]}, Vector{cam->basis_j[0], cam->basis_j[1], cam->basis_j[2]}, Vector{cam->basis_k[0], cam->basis_k[1], cam->basis_k[2]}};
    const Dimensions half_world{cam->half_world[0], cam->half_world[1]};
    const Dimensions half_px{0.5f * static_cast<float>(W), 0.5f * static_cast<float>(H)};

    // full-height, full-width scratch image so render_scanline's addressing is untouched
    std::vector<uint8_t> full;
    if (whole_rows) full.resize(static_cast<std::size_t>(W) * H * 3);
    Image image{W, H, full.data(), ""};

    constexpr unsigned SEGMENT = 4;
    const unsigned segments = whole_rows ? 1u : (rw + SEGMENT - 1) / SEGMENT;
    const unsigned units = rh * segments;
    std::atomic<unsigned> next{0};
    g_worker_seconds.assign(static_cast<std::size_t>(n_threads < 1 ? 1 : n_threads), 0.0);
    auto worker = [&](int tid) {
        const auto began = std::chrono::steady_clock::now();
        for (;;) {
            const unsigned u = next.fetch_add(1);
            if (u >= units) break;
            const unsigned r = u / segments;
            const unsigned xa = whole_rows ? x0 : x0 + (u % segments) * SEGMENT, xb = whole_rows ? x0 + rw : std::min(x0 + rw, xa + SEGMENT);
            const unsigned y = y0 + r;
            if (whole_rows) {
                render_scanline(scene, camera, basis, image, half_world, y, prm->max_bounce_count);
                std::memcpy(rgb + static_cast<std::size_t>(r) * rw * 3, full.data() + static_cast<std::size_t>(y) * W * 3, static_cast<std::size_t>(rw) * 3);
            }
            if (!whole_rows || ids || sample_colours) {
                // per-sample walk with the reference's own functions, in render_scanline's order (render_work_queue.cpp:37-56)
                for (unsigned x = xa; x < xb; ++x) {
                    Colour colour{0.0f, 0.0f, 0.0f};
                    for (int sample = 0; sample < 16; ++sample) {
                        const float xo = static_cast<float>(sample % 4) * 0.25f + 2.0f;
                        const float yo = static_cast<float>(sample / 4) * 0.25f + 2.0f;
                        const Vector dir = ray_direction_through_pixel(x, y, xo, yo, basis, half_world, half_px);
                        const Ray ray{camera.eye, dir};
                        const AABBIntersectionResult bb = intersect(ray, scene.bounding_box);
                        const std::size_t si = (static_cast<std::size_t>(r) * rw + (x - x0)) * 16 + static_cast<std::size_t>(sample);
                        Colour c{0.0f, 0.0f, 0.0f};
                        uint32_t id = RT_ID_CULLED;
                        if (bb.max_distance >= bb.min_distance) {
                            if (!whole_rows || sample_colours) c = intersect(ray, scene, prm->max_bounce_count);
                            if (ids) {
                                const Colour e = intersect(ray, id_scene, 1);
                                if (e.red > 0.0f) id = RT_ID_MAKE(RT_ID_KIND_TRIANGLE, static_cast<uint32_t>(e.red) - 1);
                                else if (e.green > 0.0f) id = RT_ID_MAKE(RT_ID_KIND_SPHERE, static_cast<uint32_t>(e.green) - 1);
                                else if (e.blue > 0.0f) id = RT_ID_MAKE(RT_ID_KIND_ELLIPSOID, static_cast<uint32_t>(e.blue) - 1);
                                else id = RT_ID_MISS;
                            }
                        }
                        if (ids) ids[si] = id;
                        if (sample_colours) { sample_colours[si * 3] = c.red; sample_colours[si * 3 + 1] = c.green; sample_colours[si * 3 + 2] = c.blue; }
                        colour += c;
                    }
                    if (!whole_rows) {
                        colour = 0.0625f * colour;
                        uint8_t* px = rgb + (static_cast<std::size_t>(r) * rw + (x - x0)) * 3;
                        px[0] = static_cast<unsigned char>(colour.red * 255.0f);
                        px[1] = static_cast<unsigned char>(colour.green * 255.0f);
                        px[2] = static_cast<unsigned char>(colour.blue * 255.0f);
                    }
                }
            }
        }
        g_worker_seconds[static_cast<std::size_t>(tid)] = std::chrono::duration<double>(std::chrono::steady_clock::now() - began).count();
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto& t : threads) t.join();
    if (render_seconds) *render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

}  // extern "C"
