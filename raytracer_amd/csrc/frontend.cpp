// Host front end: the `.test` scene format, the camera set-up and a PNG writer.
//
// Format-compatible replacement for parse_input_file (Source/input_parsing.cpp:96-603)
// producing the same FileInfo contents bit for bit: the same std::stof conversions,
// the same transform-stack products and the same fused/unfused rounding the reference
// gets from clang's -ffp-contract=on in Source/maths.cpp.  This file is compiled with
// -ffp-contract=off; every fused multiply-add is an explicit fmaf().
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include <zlib.h>

#include "rt_internal.h"

namespace {

constexpr float TOL = 1.0e-4f;                       // maths.h:10
constexpr float PI_F = 3.14159265358979323846264f;   // maths.h:15

struct Vec { float x, y, z; };
struct Mat { float r[3][4]; };

inline float min_ss(float x, float y) { return x < y ? x : y; }   // maths.cpp:147-152
inline float max_ss(float x, float y) { return x > y ? x : y; }   // maths.cpp:154-159
inline float to_radians(float deg) { return deg / 180.0f * PI_F; } // maths.cpp:181-183

// a*b + c*d + e*f as clang contracts it: fma(e, f, fma(a, b, c*d))
inline float dot3(float a, float b, float c, float d, float e, float f) { return fmaf(e, f, fmaf(a, b, c * d)); }

inline Vec normalise(Vec v) {                                      // maths.cpp:211-222
    const float inv = 1.0f / sqrtf(dot3(v.x, v.x, v.y, v.y, v.z, v.z));
    return {inv * v.x, inv * v.y, inv * v.z};
}
inline Vec cross(Vec l, Vec r) {                                   // maths.cpp:203-209
    return {fmaf(l.y, r.z, -(l.z * r.y)), fmaf(l.z, r.x, -(l.x * r.z)), fmaf(l.x, r.y, -(l.y * r.x))};
}

Mat identity() { return Mat{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}}; }
Mat translation(float x, float y, float z) { return Mat{{{1, 0, 0, x}, {0, 1, 0, y}, {0, 0, 1, z}}}; }
Mat scaling(float x, float y, float z) { return Mat{{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}}}; }

// maths.cpp:234-252
Mat mul(const Mat& l, const Mat& r) {
    Mat out;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out.r[i][j] = dot3(l.r[i][0], r.r[0][j], l.r[i][1], r.r[1][j], l.r[i][2], r.r[2][j]);
        out.r[i][3] = dot3(l.r[i][0], r.r[0][3], l.r[i][1], r.r[1][3], l.r[i][2], r.r[2][3]) + l.r[i][3];
    }
    return out;
}

// maths.cpp:226-232
Vec transform_point(const Mat& m, Vec v) {
    return {dot3(m.r[0][0], v.x, m.r[0][1], v.y, m.r[0][2], v.z) + m.r[0][3],
            dot3(m.r[1][0], v.x, m.r[1][1], v.y, m.r[1][2], v.z) + m.r[1][3],
            dot3(m.r[2][0], v.x, m.r[2][1], v.y, m.r[2][2], v.z) + m.r[2][3]};
}

// maths.cpp:278-300.  (1-c)*u*v + w contracts to fma((1-c)*u, v, w); with a product on
// both sides of the +/- the left product is the one fused.
Mat rotation(float angle, float ax, float ay, float az) {
    const Vec a = normalise(Vec{ax, ay, az});
    const float s = sinf(angle), c = cosf(angle);
    const float k = 1.0f - c;
    Mat m;
    m.r[0][0] = fmaf(k * a.x, a.x, c);
    m.r[0][1] = fmaf(k * a.x, a.y, -(s * a.z));
    m.r[0][2] = fmaf(k * a.x, a.z, s * a.y);
    m.r[0][3] = 0.0f;
    m.r[1][0] = fmaf(k * a.x, a.y, s * a.z);
    m.r[1][1] = fmaf(k * a.y, a.y, c);
    m.r[1][2] = fmaf(k * a.y, a.z, -(s * a.x));
    m.r[1][3] = 0.0f;
    m.r[2][0] = fmaf(k * a.x, a.z, -(s * a.y));
    m.r[2][1] = fmaf(k * a.y, a.z, s * a.x);
    m.r[2][2] = fmaf(k * a.z, a.z, c);
    m.r[2][3] = 0.0f;
    return m;
}

bool all_digits(const std::string& s) {                            // input_parsing.cpp:15-23
    for (char ch : s) if (ch < '0' || ch > '9') return false;
    return true;
}
bool looks_like_float(const std::string& s) {                      // input_parsing.cpp:25-52
    const char f = s[0];
    if (!(f == '-' || f == '+' || f == '.' || (f >= '0' && f <= '9'))) return false;
    bool dot = (f == '.');
    for (size_t i = 1; i < s.size(); ++i) {
        const char ch = s[i];
        if (ch == '.') { if (dot) return false; dot = true; }
        else if (ch < '0' || ch > '9') return false;
    }
    return true;
}

enum class Arg { None, Ints, Floats, Any };
struct CommandSpec { const char* name; size_t argc; Arg kind; const char* error; };
// every command of the format with its arity, argument kind and the reference's message for a bad argument list
const CommandSpec COMMANDS[] = {
    {"size", 2, Arg::Ints, "'size' command should have 2 positive integer parameters."},
    {"output", 1, Arg::Any, "'output' command should have 1 parameter."},
    {"maxdepth", 1, Arg::Ints, "'maxdepth' command should have 1 positive integer parameter."},
    {"maxverts", 1, Arg::Ints, "'maxverts' command should have 1 positive integer parameter."},
    {"camera", 10, Arg::Floats, "'camera' command should have 10 floating point parameters."},
    {"vertex", 3, Arg::Floats, "'vertex' command should have 3 floating point parameters."},
    {"tri", 3, Arg::Ints, "'tri' command should have 3 positive integer parameters."},
    {"sphere", 4, Arg::Floats, "'sphere' command should have 4 floating point parameters."},
    {"pushTransform", 0, Arg::None, "'pushTransform' command does not take any parameters."},
    {"popTransform", 0, Arg::None, "'popTransform' command does not take any parameters"},
    {"translate", 3, Arg::Floats, "'translate' command should have 3 floating point parameters."},
    {"scale", 3, Arg::Floats, "'scale' command should have 3 floating point parameters."},
    {"rotate", 4, Arg::Floats, "'rotate' command should have 4 floating point parameters."},
    {"directional", 6, Arg::Floats, "'directional' command should have 6 floating point parameters."},
    {"point", 6, Arg::Floats, "'point' command should have 6 floating point parameters."},
    {"attenuation", 3, Arg::Floats, "'attenuation' command should have 3 floating point parameters."},
    {"ambient", 3, Arg::Floats, "'ambient' command should have 3 floating point parameters."},
    {"diffuse", 3, Arg::Floats, "'diffuse' command should have 3 floating point parameters."},
    {"specular", 3, Arg::Floats, "'specular' command should have 3 floating point parameters."},
    {"emission", 3, Arg::Floats, "'emission' command should have 3 floating point parameters."},
    {"shininess", 1, Arg::Floats, "'shininess' command should have 1 floating point parameter."},
};

struct Builder {
    rt_file_info* info;
    std::vector<Vec> vertices;
    Mat current = identity(), inverse_current = identity();
    std::vector<Mat> stack, inverse_stack;
    float material[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float* bb() { return info->desc.bounding_box; }

    // product of the whole stack, bottom first, then the current transform (input_parsing.cpp:243-248)
    Mat world() const {
        Mat t = identity();
        for (const Mat& m : stack) t = mul(t, m);
        return mul(t, current);
    }
    Mat inverse_world() const {                                    // input_parsing.cpp:311-316
        Mat t = identity();
        for (const Mat& m : inverse_stack) t = mul(m, t);
        return mul(inverse_current, t);
    }
    void grow(int axis, float lo, float hi) {
        bb()[2 * axis] = min_ss(bb()[2 * axis], lo);
        bb()[2 * axis + 1] = max_ss(bb()[2 * axis + 1], hi);
    }
    void push_material(std::vector<float>& to) { to.insert(to.end(), material, material + 10); }

    void triangle(size_t ia, size_t ib, size_t ic) {               // input_parsing.cpp:243-292
        const Mat t = world();
        const Vec a = transform_point(t, vertices[ia]), b = transform_point(t, vertices[ib]), c = transform_point(t, vertices[ic]);
        push_material(info->triangle_materials);
        const size_t n = info->n_triangles++;
        if (n % 8 == 0) info->triangle8s.resize(info->triangle8s.size() + 72, 0.0f);
        float* t8 = info->triangle8s.data() + (n / 8) * 72;
        const size_t j = n % 8;
        const float v[9] = {a.x, a.y, a.z, b.x - a.x, b.y - a.y, b.z - a.z, c.x - a.x, c.y - a.y, c.z - a.z};
        for (int k = 0; k < 9; ++k) t8[k * 8 + j] = v[k];
        grow(0, min_ss(a.x, min_ss(b.x, c.x)), max_ss(a.x, max_ss(b.x, c.x)));
        grow(1, min_ss(a.y, min_ss(b.y, c.y)), max_ss(a.y, max_ss(b.y, c.y)));
        grow(2, min_ss(a.z, min_ss(b.z, c.z)), max_ss(a.z, max_ss(b.z, c.z)));
    }

    void sphere(float cx, float cy, float cz, float radius) {      // input_parsing.cpp:299-445
        Mat t = world();
        Mat ti = inverse_world();
        const float sx = dot3(t.r[0][0], t.r[0][0], t.r[1][0], t.r[1][0], t.r[2][0], t.r[2][0]);
        const float sy = dot3(t.r[0][1], t.r[0][1], t.r[1][1], t.r[1][1], t.r[2][1], t.r[2][1]);
        const float sz = dot3(t.r[0][2], t.r[0][2], t.r[1][2], t.r[1][2], t.r[2][2], t.r[2][2]);
        const bool uniform = (fabsf(sy - sx) < TOL) && (fabsf(sz - sy) < TOL);
        if (!uniform) {
            // an ellipsoid: unit sphere at the origin under (transform, inverse transform)
            if (cx != 0.0f || cy != 0.0f || cz != 0.0f) {
                t = mul(t, translation(cx, cy, cz));
                ti = mul(translation(-cx, -cy, -cz), ti);
            }
            if (radius != 1.0f) {
                t = mul(t, scaling(radius, radius, radius));
                const float ir = 1.0f / radius;
                ti = mul(scaling(ir, ir, ir), ti);
            }
            push_material(info->ellipsoid_materials);
            const size_t n = info->n_ellipsoids++;
            if (n % 8 == 0) {
                info->ellipsoid8_inverse_transforms.resize(info->ellipsoid8_inverse_transforms.size() + 96, 0.0f);
                info->ellipsoid8_transforms.resize(info->ellipsoid8_transforms.size() + 96, 0.0f);
            }
            float* mi = info->ellipsoid8_inverse_transforms.data() + (n / 8) * 96;
            float* mt = info->ellipsoid8_transforms.data() + (n / 8) * 96;
            for (int row = 0; row < 3; ++row)
                for (int col = 0; col < 4; ++col) {
                    mi[(row * 4 + col) * 8 + n % 8] = ti.r[row][col];
                    mt[(row * 4 + col) * 8 + n % 8] = t.r[row][col];
                }
            for (int axis = 0; axis < 3; ++axis) {
                const float extent = sqrtf(dot3(t.r[axis][0], t.r[axis][0], t.r[axis][1], t.r[axis][1], t.r[axis][2], t.r[axis][2]));
                grow(axis, t.r[axis][3] - extent, t.r[axis][3] + extent);
            }
        } else {
            const Vec c = transform_point(t, Vec{cx, cy, cz});
            const float r = sqrtf(sx) * radius;
            push_material(info->sphere_materials);
            const size_t n = info->n_spheres++;
            if (n % 8 == 0) info->sphere8s.resize(info->sphere8s.size() + 32, 0.0f);
            float* s8 = info->sphere8s.data() + (n / 8) * 32;
            s8[n % 8] = c.x; s8[8 + n % 8] = c.y; s8[16 + n % 8] = c.z; s8[24 + n % 8] = r;
            grow(0, c.x - r, c.x + r);
            grow(1, c.y - r, c.y + r);
            grow(2, c.z - r, c.z + r);
        }
    }
};

// Returns nullptr on success or the reference's message.
const char* parse_into(rt_file_info* info, std::istream& in) {
    Builder b;
    b.info = info;
    std::string line;
    bool first = true;
    std::vector<std::string> tok;
    std::vector<float> f;
    while (std::getline(in, line, '\n')) {
        if (line.empty() || line[0] == '#') continue;
        // split on single spaces, dropping empty tokens (input_parsing.cpp:69-87)
        tok.clear();
        for (size_t pos = 0; pos <= line.size();) {
            size_t end = line.find(' ', pos);
            if (end == std::string::npos) end = line.size();
            if (end > pos) tok.emplace_back(line, pos, end - pos);
            pos = end + 1;
        }
        const std::string name = tok.empty() ? std::string() : tok[0];
        if (first && name != "size") return "First command should be 'size'.";
        const CommandSpec* spec = nullptr;
        for (const CommandSpec& c : COMMANDS) if (name == c.name) { spec = &c; break; }
        if (!spec) return "Unknown command entered.";
        const size_t argc = tok.size() - 1;
        if (name == "maxverts" && !b.vertices.empty()) return "'maxverts' should be specified before vertices are specified.";
        bool ok = (argc == spec->argc);
        for (size_t i = 1; ok && i < tok.size(); ++i) {
            if (spec->kind == Arg::Ints) ok = all_digits(tok[i]);
            else if (spec->kind == Arg::Floats) ok = looks_like_float(tok[i]);
        }
        if (!ok) return spec->error;
        f.clear();
        if (spec->kind == Arg::Floats) for (size_t i = 1; i < tok.size(); ++i) f.push_back(std::stof(tok[i]));

        if (name == "size") {
            info->width = static_cast<uint32_t>(std::stoi(tok[1]));
            info->height = static_cast<uint32_t>(std::stoi(tok[2]));
        } else if (name == "output") {
            if (tok[1].size() + 1 > sizeof info->filename) return "'output' name is too long.";
            std::memset(info->filename, 0, sizeof info->filename);
            std::memcpy(info->filename, tok[1].data(), tok[1].size());
        } else if (name == "maxdepth") {
            info->max_recursion_depth = std::stoi(tok[1]);
        } else if (name == "maxverts") {
            const unsigned long n = std::stoul(tok[1]);
            b.vertices.reserve(n < (1ul << 28) ? n : (1ul << 28));
        } else if (name == "camera") {
            for (int i = 0; i < 9; ++i) info->camera[i] = f[static_cast<size_t>(i)];
            info->camera[10] = f[9];
            info->camera[9] = static_cast<float>(info->width) * info->camera[10] / static_cast<float>(info->height);  // :216
        } else if (name == "vertex") {
            b.vertices.push_back(Vec{f[0], f[1], f[2]});
        } else if (name == "tri") {
            const size_t ia = std::stoul(tok[1]), ib = std::stoul(tok[2]), ic = std::stoul(tok[3]);
            const size_t n = b.vertices.size();
            if (ia >= n || ib >= n || ic >= n) return "Vertex index specified in 'tri' command is beyond the number of specified vertices.";
            b.triangle(ia, ib, ic);
        } else if (name == "sphere") {
            b.sphere(f[0], f[1], f[2], f[3]);
        } else if (name == "pushTransform") {
            b.stack.push_back(b.current);
            b.inverse_stack.push_back(b.inverse_current);
            b.current = identity();
            b.inverse_current = identity();
        } else if (name == "popTransform") {
            if (b.stack.empty()) return "Cannot perform 'popTransform' as there are no transforms on the stack.";
            b.current = b.stack.back(); b.stack.pop_back();
            b.inverse_current = b.inverse_stack.back(); b.inverse_stack.pop_back();
        } else if (name == "translate") {
            b.current = mul(b.current, translation(f[0], f[1], f[2]));
            b.inverse_current = mul(translation(-f[0], -f[1], -f[2]), b.inverse_current);
        } else if (name == "scale") {
            b.current = mul(b.current, scaling(f[0], f[1], f[2]));
            b.inverse_current = mul(scaling(1.0f / f[0], 1.0f / f[1], 1.0f / f[2]), b.inverse_current);
        } else if (name == "rotate") {
            const float rad = to_radians(f[3]);
            b.current = mul(b.current, rotation(rad, f[0], f[1], f[2]));
            b.inverse_current = mul(rotation(-rad, f[0], f[1], f[2]), b.inverse_current);
        } else if (name == "directional") {
            const Vec d = normalise(Vec{f[0], f[1], f[2]});
            const float v[6] = {d.x, d.y, d.z, f[3], f[4], f[5]};
            std::memcpy(info->desc.directional_light, v, sizeof v);
            info->desc.has_directional_light = 1;
        } else if (name == "point") {
            info->point_lights.insert(info->point_lights.end(), f.begin(), f.end());
        } else if (name == "attenuation") {
            for (int i = 0; i < 3; ++i) info->desc.attenuation[i] = f[static_cast<size_t>(i)];
        } else if (name == "ambient") {
            for (int i = 0; i < 3; ++i) info->desc.ambient[i] = f[static_cast<size_t>(i)];
        } else if (name == "diffuse") {
            for (int i = 0; i < 3; ++i) b.material[i] = f[static_cast<size_t>(i)];
        } else if (name == "specular") {
            for (int i = 0; i < 3; ++i) b.material[3 + i] = f[static_cast<size_t>(i)];
        } else if (name == "emission") {
            for (int i = 0; i < 3; ++i) b.material[6 + i] = f[static_cast<size_t>(i)];
        } else if (name == "shininess") {
            b.material[9] = f[0];
        }
        first = false;
    }
    return nullptr;
}

void refresh_desc(rt_file_info* info) {
    rt_scene_desc& d = info->desc;
    d.triangle8s = info->triangle8s.data();
    d.triangle_materials = info->triangle_materials.data();
    d.sphere8s = info->sphere8s.data();
    d.sphere_materials = info->sphere_materials.data();
    d.ellipsoid8_inverse_transforms = info->ellipsoid8_inverse_transforms.data();
    d.ellipsoid8_transforms = info->ellipsoid8_transforms.data();
    d.ellipsoid_materials = info->ellipsoid_materials.data();
    d.point_light_sources = info->point_lights.data();
    d.n_triangle8 = info->triangle8s.size() / 72;
    d.n_triangles = info->n_triangles;
    d.n_sphere8 = info->sphere8s.size() / 32;
    d.n_spheres = info->n_spheres;
    d.n_ellipsoid8 = info->ellipsoid8_transforms.size() / 96;
    d.n_ellipsoids = info->n_ellipsoids;
    d.n_point_lights = info->point_lights.size() / 6;
}

// ---- PNG ---------------------------------------------------------------------------
uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back(static_cast<uint8_t>(x >> s)); }
void put_chunk(std::vector<uint8_t>& png, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(png, static_cast<uint32_t>(data.size()));
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_of(png.data() + start, png.size() - start, 0));
}

}  // namespace

extern "C" {

int rt_parse_input_file(const char* filename, rt_file_info** out) {
    if (!filename || !out) return rt_fail(RT_ERR_ARGUMENT, "rt_parse_input_file: null argument");
    *out = nullptr;
    std::ifstream in(filename);
    if (!in.is_open()) return rt_fail(RT_ERR_PARSE, "Failed to open input file.");
    rt_file_info* info = new rt_file_info();
    // defaults of input_parsing.cpp:104-136
    std::memset(&info->desc, 0, sizeof info->desc);
    info->max_recursion_depth = 5;
    std::memset(info->filename, 0, sizeof info->filename);
    std::strcpy(info->filename, "raytrace.png");
    info->desc.ambient[0] = info->desc.ambient[1] = info->desc.ambient[2] = 0.2f;
    info->desc.attenuation[0] = 1.0f;
    const char* error = nullptr;
    try {
        error = parse_into(info, in);
    } catch (const std::exception&) {
        // std::stof / stoi / stoul threw; the reference lets this terminate the process (input_parsing.cpp:234)
        error = "Numeric parameter out of range.";
    }
    if (error) { delete info; return rt_fail(RT_ERR_PARSE, error); }
    refresh_desc(info);
    *out = info;
    return 0;
}

void rt_file_info_destroy(rt_file_info* info) { delete info; }
const rt_scene_desc* rt_file_info_scene(const rt_file_info* info) { return &info->desc; }
uint32_t rt_file_info_width(const rt_file_info* info) { return info->width; }
uint32_t rt_file_info_height(const rt_file_info* info) { return info->height; }
int32_t rt_file_info_max_recursion_depth(const rt_file_info* info) { return info->max_recursion_depth; }
const char* rt_file_info_output_filename(const rt_file_info* info) { return info->filename; }
void rt_file_info_camera(const rt_file_info* info, float camera[11]) { std::memcpy(camera, info->camera, sizeof info->camera); }
void rt_file_info_set_size(rt_file_info* info, uint32_t width, uint32_t height) {
    info->width = width;
    info->height = height;
    info->camera[9] = static_cast<float>(width) * info->camera[10] / static_cast<float>(height);
}

// main.cpp:44-57
void rt_camera_setup(const float c[11], rt_camera_desc* out) {
    const Vec eye{c[0], c[1], c[2]}, look_at{c[3], c[4], c[5]}, up{c[6], c[7], c[8]};
    const Vec k = normalise(Vec{look_at.x - eye.x, look_at.y - eye.y, look_at.z - eye.z});
    const Vec i = normalise(cross(up, k));
    const Vec j = cross(k, i);
    const float v[14] = {eye.x, eye.y, eye.z, i.x, i.y, i.z, j.x, j.y, j.z, k.x, k.y, k.z,
                         tanf(0.5f * to_radians(c[9])), tanf(0.5f * to_radians(c[10]))};
    std::memcpy(out, v, sizeof v);
}

// PNG of an RGB8 image (stands where Source/main.cpp:93 calls stbi_write_png: same pixels, not the same bytes).
// Rows are filtered with the Sub filter and deflated by zlib in up to 16 pieces on threads of their own: every piece but
// the last ends on a full flush, which leaves it byte aligned and independent of the others, so the pieces concatenate
// into one valid deflate stream; the Adler-32 of the whole is combined from the pieces'.
static int write_png_impl(const char* filename, uint32_t width, uint32_t height, const uint8_t* rgb) {
    const size_t row = static_cast<size_t>(width) * 3;
    const uint32_t n_pieces = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), (height + 31) / 32}));
    const uint32_t rows_per_piece = (height + n_pieces - 1) / n_pieces;
    struct Piece { std::vector<uint8_t> out; uLong adler = 1; size_t raw_bytes = 0; int rc = Z_OK; };
    std::vector<Piece> pieces(n_pieces);
    auto work = [&](uint32_t p) {
        Piece& piece = pieces[p];
        const uint32_t y0 = p * rows_per_piece, y1 = std::min(height, y0 + rows_per_piece);
        if (y0 >= y1) return;
        std::vector<uint8_t> raw(static_cast<size_t>(y1 - y0) * (row + 1));
        for (uint32_t y = y0; y < y1; ++y) {
            uint8_t* dst = raw.data() + static_cast<size_t>(y - y0) * (row + 1);
            const uint8_t* src = rgb + static_cast<size_t>(y) * row;
            dst[0] = 1;                                         // filter type Sub: each byte minus the byte three to its left
            for (size_t i = 0; i < row; ++i) dst[1 + i] = static_cast<uint8_t>(src[i] - (i >= 3 ? src[i - 3] : 0));
        }
        piece.raw_bytes = raw.size();
        piece.adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), static_cast<uInt>(raw.size()));
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if ((piece.rc = deflateInit2(&zs, 3, Z_DEFLATED, -15, 8, Z_FILTERED)) != Z_OK) return;
        piece.out.resize(deflateBound(&zs, static_cast<uLong>(raw.size())) + 64);
        zs.next_in = raw.data(); zs.avail_in = static_cast<uInt>(raw.size());
        zs.next_out = piece.out.data(); zs.avail_out = static_cast<uInt>(piece.out.size());
        const bool last = y1 == height;
        const int rc = deflate(&zs, last ? Z_FINISH : Z_FULL_FLUSH);
        piece.rc = (last ? rc == Z_STREAM_END : rc == Z_OK) && zs.avail_in == 0 ? Z_OK : Z_BUF_ERROR;
        piece.out.resize(piece.out.size() - zs.avail_out);
        deflateEnd(&zs);
    };
    {
        std::vector<std::thread> threads;
        for (uint32_t p = 1; p < n_pieces; ++p) threads.emplace_back(work, p);
        work(0);
        for (std::thread& t : threads) t.join();
    }
    std::vector<uint8_t> z = {0x78, 0x9C};
    uLong adler = 1;
    bool first = true;
    for (const Piece& piece : pieces) {
        if (piece.rc != Z_OK) return rt_fail(RT_ERR_IO, "rt_write_png: deflate failed");
        if (piece.raw_bytes == 0) continue;
        z.insert(z.end(), piece.out.begin(), piece.out.end());
        adler = first ? piece.adler : adler32_combine(adler, piece.adler, static_cast<z_off_t>(piece.raw_bytes));
        first = false;
    }
    put_be32(z, static_cast<uint32_t>(adler));
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8-bit, RGB
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE* fp = std::fopen(filename, "wb");
    if (!fp) return rt_fail(RT_ERR_IO, "rt_write_png: cannot open output file");
    const size_t written = std::fwrite(png.data(), 1, png.size(), fp);
    std::fclose(fp);
    return written == png.size() ? 0 : rt_fail(RT_ERR_IO, "rt_write_png: short write");
}

int rt_write_png(const char* filename, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!filename || !rgb || !width || !height) return rt_fail(RT_ERR_ARGUMENT, "rt_write_png: bad argument");
    try {
        return write_png_impl(filename, width, height, rgb);
    } catch (const std::exception& e) {
        return rt_fail(RT_ERR_IO, e.what());
    }
}

// The same in the background: the pixels are copied, a thread encodes and writes them, and the caller goes on (to the next frame).
namespace {
struct PendingImage {
    std::thread worker;
    std::vector<uint8_t> pixels;
    std::string filename, error;
    int rc = 0;
    bool busy = false;
    std::mutex guard;          // begin and end may be called from different threads
    // an image begun and never ended is still written: the writer is joined when the library is unloaded
    ~PendingImage() { if (worker.joinable()) worker.join(); }
} g_pending_image;
}
int rt_write_png_begin(const char* filename, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!filename || !rgb || !width || !height) return rt_fail(RT_ERR_ARGUMENT, "rt_write_png_begin: bad argument");
    std::lock_guard<std::mutex> lock(g_pending_image.guard);
    if (g_pending_image.busy) return rt_fail(RT_ERR_ARGUMENT, "rt_write_png_begin: an image is still being written (rt_write_png_end)");
    try {
        g_pending_image.pixels.assign(rgb, rgb + static_cast<size_t>(width) * height * 3);
        g_pending_image.filename = filename;
        g_pending_image.worker = std::thread([width, height] {
            g_pending_image.rc = rt_write_png(g_pending_image.filename.c_str(), width, height, g_pending_image.pixels.data());
            if (g_pending_image.rc) g_pending_image.error = rt_last_error();
        });
        g_pending_image.busy = true;
    } catch (const std::exception& e) {
        return rt_fail(RT_ERR_IO, e.what());
    }
    return 0;
}
int rt_write_png_end(void) {
    std::lock_guard<std::mutex> lock(g_pending_image.guard);
    if (!g_pending_image.busy) return 0;
    if (g_pending_image.worker.joinable()) g_pending_image.worker.join();
    g_pending_image.busy = false;
    return g_pending_image.rc ? rt_fail(g_pending_image.rc, g_pending_image.error.c_str()) : 0;
}

}  // extern "C"
