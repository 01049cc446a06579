// TEST INFRASTRUCTURE ONLY - the checker, never the thing shipped or measured as the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// CPU restatement of the reference renderer's per-pixel loop, in scalar C++ with
// every rounding step spelled out: compiled with -ffp-contract=off, each fused
// multiply-add the reference executes is an explicit fmaf() here, everything else
// is a separately rounded IEEE binary32 operation.  The reference is built with
// clang's default -ffp-contract=on, which fuses a*b+c inside one expression of its
// scalar code (Source/maths.cpp) but never across the AVX intrinsics of
// Source/ray_tracing.cpp, which fuse only where they say _mm256_fmadd_ps.
//
// PARITY: pinned.  This file is checked bit-for-bit (RGB8, per-sample colours and
// per-sample primary hit ids) against the unmodified reference compiled from its own
// sources (oracle/ref_harness.cpp -> oracle/_ref/librt_ref.so) on all nine scenes the
// reference ships, and against the fixtures generated from that build under
// tests/golden/ (tests/test_oracle_vs_reference.py, tests/test_oracle_golden.py).
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"

namespace {

constexpr float TOL = 1.0e-4f;  // maths.h:10

struct V3 { float x, y, z; };

inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t float_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// minss / maxss: the second operand comes back when either is NaN (maths.cpp:147-159)
inline float fp_min(float x, float y) { return x < y ? x : y; }
inline float fp_max(float x, float y) { return x > y ? x : y; }
inline float fp_sqrt(float x) { return sqrtf(x); }  // sqrtps, correctly rounded (maths.cpp:141-145)

// ---- scalar Vector operators as clang contracts them (maths.cpp:187-222) ------
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
// x*x' + y*y' + z*z'  ->  fma(z, z', fma(x, x', y*y'))   (maths.cpp:195-197)
inline float sdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.x, b.x, a.y * b.y)); }
// l.y*r.z - l.z*r.y  ->  fma(l.y, r.z, -(l.z*r.y))        (maths.cpp:203-209)
inline V3 scross(V3 l, V3 r) {
    return {fmaf(l.y, r.z, -(l.z * r.y)), fmaf(l.z, r.x, -(l.x * r.z)), fmaf(l.x, r.y, -(l.y * r.x))};
}
inline V3 divide(V3 v, float s) { const float inv = 1.0f / s; return {inv * v.x, inv * v.y, inv * v.z}; }  // maths.cpp:211-214
inline float smagnitude(V3 v) { return fp_sqrt(sdot(v, v)); }
inline V3 normalise(V3 v) { return divide(v, smagnitude(v)); }

// ---- the AVX lane arithmetic (ray_tracing.cpp:41-126) ----------------------------
// dot: mul, fmadd, fmadd in x, y, z order (ray_tracing.cpp:68-74)
inline float vdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// cross: two products and a subtraction, none fused (ray_tracing.cpp:76-83)
inline V3 vcross(V3 l, V3 r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }

struct M34 { float r[3][4]; };
inline M34 load_mat(const float* m8, int lane) {  // Mat3x4x8::rows[row][col][lane]
    M34 m;
    for (int row = 0; row < 3; ++row) for (int col = 0; col < 4; ++col) m.r[row][col] = m8[(row * 4 + col) * 8 + lane];
    return m;
}
// ray_tracing.cpp:105-126
inline V3 vmat_point(const M34& m, V3 v) {
    return {m.r[0][3] + fmaf(m.r[0][2], v.z, fmaf(m.r[0][1], v.y, m.r[0][0] * v.x)),
            m.r[1][3] + fmaf(m.r[1][2], v.z, fmaf(m.r[1][1], v.y, m.r[1][0] * v.x)),
            m.r[2][3] + fmaf(m.r[2][2], v.z, fmaf(m.r[2][1], v.y, m.r[2][0] * v.x))};
}
// ray_tracing.cpp:208-227
inline V3 vmat_dir(const M34& m, V3 v) {
    return {fmaf(m.r[0][2], v.z, fmaf(m.r[0][1], v.y, m.r[0][0] * v.x)),
            fmaf(m.r[1][2], v.z, fmaf(m.r[1][1], v.y, m.r[1][0] * v.x)),
            fmaf(m.r[2][2], v.z, fmaf(m.r[2][1], v.y, m.r[2][0] * v.x))};
}
// scalar Matrix*Vector: m0*x + m1*y + m2*z + m3 -> fma(m2, z, fma(m0, x, m1*y)) + m3 (maths.cpp:226-232)
inline V3 smat_point(const M34& m, V3 v) {
    return {fmaf(m.r[0][2], v.z, fmaf(m.r[0][0], v.x, m.r[0][1] * v.y)) + m.r[0][3],
            fmaf(m.r[1][2], v.z, fmaf(m.r[1][0], v.x, m.r[1][1] * v.y)) + m.r[1][3],
            fmaf(m.r[2][2], v.z, fmaf(m.r[2][0], v.x, m.r[2][1] * v.y)) + m.r[2][3]};
}
// maths.cpp:312-318
inline V3 smat_transpose_dir(const M34& m, V3 v) {
    return {fmaf(m.r[2][0], v.z, fmaf(m.r[0][0], v.x, m.r[1][0] * v.y)),
            fmaf(m.r[2][1], v.z, fmaf(m.r[0][1], v.x, m.r[1][1] * v.y)),
            fmaf(m.r[2][2], v.z, fmaf(m.r[0][2], v.x, m.r[1][2] * v.y))};
}

// ---- fp_pow and friends (maths.cpp:9-139) ------------------------------------------
inline int exponent_127(float x) { return static_cast<int>((float_to_bits(x) >> 23) & 0xFFu); }

float fp_log2(float x) {
    const int exponent = exponent_127(x) - 127;
    const float mantissa = bits_to_float((float_to_bits(x) & 0x007FFFFFu) | 0x3F800000u);  // maths.cpp:35-38
    static const float C[6] = {-2.80036404536401f, 5.0917108473873602f, -3.5507929732452235f,
                               1.6311487949127739f, -0.41656369634969576f, 0.044873611393688138f};
    float acc = 0.0f, mult = 1.0f;
    for (int k = 0; k < 6; ++k) { acc = fmaf(C[k], mult, acc); mult *= mantissa; }  // `+= c * m` contracts
    float result = acc + static_cast<float>(exponent);
    return fp_min(result, 0.0f);
}

float fp_exp2(float x) {
    const float clamped = fp_max(x, -127.0f);
    const float ax = bits_to_float(float_to_bits(clamped) & 0x7FFFFFFFu);
    const int ip = static_cast<int>(ax);
    const float frac = ax - static_cast<float>(ip);
    const uint32_t e127 = static_cast<unsigned char>(-ip + 127);
    const float inv_int = bits_to_float(e127 << 23);  // maths.cpp:81-85
    static const float C[4] = {0.99994648282537313f, -0.69137342261578116f, 0.23097554273105286f, -0.039602120115271706f};
    float acc = 0.0f, mult = 1.0f;
    for (int k = 0; k < 4; ++k) { acc = fmaf(C[k], mult, acc); mult *= frac; }
    return inv_int * acc;
}

float fp_pow(float x, float y) {
    if (exponent_127(y) == 0) return 1.0f;  // y zero or subnormal
    if (exponent_127(x) == 0) return 0.0f;  // x zero or subnormal
    return fp_exp2(y * fp_log2(x));
}

// ---- ray / primitive tests -------------------------------------------------------------
// ray_tracing.cpp:129-165, one lane
inline float hit_triangle(V3 o, V3 d, V3 a, V3 ab, V3 ac) {
    const V3 n = vcross(ab, ac);
    const float dn = vdot(d, n);
    const bool parallel = fabsf(dn) < TOL;
    const float t = vdot(sub(a, o), n) / dn;
    const bool behind = t < TOL;
    const V3 p = add(o, scale(t, d));
    const V3 ap = sub(p, a);
    const float nn = vdot(n, n);
    const float alpha = vdot(vcross(ab, ap), n) / nn;
    const float beta = vdot(vcross(ap, ac), n) / nn;
    const bool outside = (alpha < 0.0f) | (beta < 0.0f) | (alpha + beta > 1.0f);
    return (parallel | behind | outside) ? FLT_MAX : t;
}

// ray_tracing.cpp:169-206, one lane; line 202 selects distance_0 for the far root and is kept
inline float hit_sphere(V3 o, V3 d, V3 c, float radius) {
    const V3 oc = sub(c, o);
    const float mid = vdot(oc, d);
    const float oc2 = vdot(oc, oc);
    const float mid2 = mid * mid;
    const float perp2 = oc2 - mid2;
    const float r2 = radius * radius;
    const float disc = r2 - perp2;
    const bool none = disc < 0.0f;
    const float h = sqrtf(disc);
    const float d0 = mid - h;
    const float d1 = mid + h;
    const bool inv0 = none | (d0 < TOL);
    const bool inv1 = none | (d1 < TOL);
    const float p0 = inv0 ? FLT_MAX : d0;
    const float p1 = inv1 ? FLT_MAX : d0;
    return p0 < p1 ? p0 : p1;  // _mm256_min_ps
}

// ray_tracing.cpp:229-263, one lane
inline float hit_unit_sphere(V3 o, V3 d) {
    const float mid = vdot(scale(-1.0f, o), d);
    const float oo = vdot(o, o);
    const float mid2 = mid * mid;
    const float perp2 = oo - mid2;
    const float disc = 1.0f - perp2;
    const bool none = disc < 0.0f;
    const float h = sqrtf(disc);
    const float d0 = mid - h;
    const float d1 = mid + h;
    const bool inv0 = none | (d0 < TOL);
    const bool inv1 = none | (d1 < TOL);
    const float p0 = inv0 ? FLT_MAX : d0;
    const float p1 = inv1 ? FLT_MAX : d1;
    return p0 < p1 ? p0 : p1;
}

// ray_tracing.cpp:336-344
inline bool ellipsoid_absent(V3 d) { return (d.x < TOL) & (d.y < TOL) & (d.z < TOL); }

struct Counters {
    uint64_t samples = 0, culled = 0, closest_rays = 0, hits = 0, shadow_rays = 0, shadow_blocked = 0;
    uint64_t shadow_tests[3] = {0, 0, 0};
    void merge(const Counters& o) {
        samples += o.samples; culled += o.culled; closest_rays += o.closest_rays; hits += o.hits;
        shadow_rays += o.shadow_rays; shadow_blocked += o.shadow_blocked;
        for (int k = 0; k < 3; ++k) shadow_tests[k] += o.shadow_tests[k];
    }
};

struct SceneView {
    const rt_scene_desc* d;
    V3 tri_a(size_t b, int j) const { const float* p = d->triangle8s + b * 72; return {p[j], p[8 + j], p[16 + j]}; }
    V3 tri_ab(size_t b, int j) const { const float* p = d->triangle8s + b * 72 + 24; return {p[j], p[8 + j], p[16 + j]}; }
    V3 tri_ac(size_t b, int j) const { const float* p = d->triangle8s + b * 72 + 48; return {p[j], p[8 + j], p[16 + j]}; }
    V3 sph_c(size_t b, int j) const { const float* p = d->sphere8s + b * 32; return {p[j], p[8 + j], p[16 + j]}; }
    float sph_r(size_t b, int j) const { return d->sphere8s[b * 32 + 24 + j]; }
};

struct Material { float diffuse[3], specular[3], emission[3], shininess; };

// ray_tracing.cpp:347-405 (light != nullptr) and :407-458 (directional)
bool path_is_blocked(const SceneView& sv, V3 start, const float* point_light, Counters& cn) {
    const rt_scene_desc& s = *sv.d;
    V3 dir;
    float limit;
    V3 light_pos{0, 0, 0};
    if (point_light) {
        light_pos = {point_light[0], point_light[1], point_light[2]};
        const V3 to_light = sub(light_pos, start);
        limit = smagnitude(to_light);
        const float inv = 1.0f / limit;
        dir = scale(inv, to_light);
    } else {
        dir = scale(-1.0f, V3{s.directional_light[0], s.directional_light[1], s.directional_light[2]});
        limit = FLT_MAX;
    }
    cn.shadow_rays++;
    for (size_t b = 0; b < s.n_triangle8; ++b) {
        bool any = false;
        cn.shadow_tests[0] += 8;
        for (int j = 0; j < 8; ++j) any |= hit_triangle(start, dir, sv.tri_a(b, j), sv.tri_ab(b, j), sv.tri_ac(b, j)) < limit;
        if (any) { cn.shadow_blocked++; return true; }
    }
    for (size_t b = 0; b < s.n_sphere8; ++b) {
        bool any = false;
        cn.shadow_tests[1] += 8;
        for (int j = 0; j < 8; ++j) any |= hit_sphere(start, dir, sv.sph_c(b, j), sv.sph_r(b, j)) < limit;
        if (any) { cn.shadow_blocked++; return true; }
    }
    for (size_t b = 0; b < s.n_ellipsoid8; ++b) {
        bool any = false;
        cn.shadow_tests[2] += 8;
        for (int j = 0; j < 8; ++j) {
            const M34 minv = load_mat(s.ellipsoid8_inverse_transforms + b * 96, j);
            const V3 o2 = vmat_point(minv, start);
            V3 d2 = vmat_dir(minv, dir);
            const bool absent = ellipsoid_absent(d2);
            const float inv_mag = 1.0f / sqrtf(vdot(d2, d2));
            d2 = scale(inv_mag, d2);
            float lim2 = FLT_MAX;
            if (point_light) {
                const V3 l2 = vmat_point(minv, light_pos);
                const V3 dl = sub(l2, o2);
                lim2 = sqrtf(vdot(dl, dl));
            }
            float t2 = hit_unit_sphere(o2, d2);
            if (absent) t2 = FLT_MAX;
            any |= t2 < lim2;
        }
        if (any) { cn.shadow_blocked++; return true; }
    }
    return false;
}

struct Hit { int kind; size_t index; float t; };  // kind -1: nothing

// closest-hit search of one ray: ray_tracing.cpp:484-608 and the kind choice of :616-655
Hit closest_hit(const SceneView& sv, V3 o, V3 d) {
    const rt_scene_desc& s = *sv.d;
    constexpr size_t INVALID = static_cast<size_t>(-1);
    // per kind: eight running lane minima (strict <, so the first batch keeps a lane tie),
    // then the lanes are scanned in order with strict < (lowest lane keeps a tie)
    float lane_t[8];
    uint32_t lane_b[8];
    auto reduce = [&](size_t& index, float& dist) {
        index = INVALID; dist = FLT_MAX;
        for (int j = 0; j < 8; ++j) if (lane_t[j] < dist) { index = 8 * static_cast<size_t>(lane_b[j]) + j; dist = lane_t[j]; }
    };
    auto reset = [&]() { for (int j = 0; j < 8; ++j) { lane_t[j] = FLT_MAX; lane_b[j] = 0xFFFFFFFFu; } };

    size_t tri_i, sph_i, ell_i;
    float tri_t, sph_t, ell_t;

    reset();
    for (size_t b = 0; b < s.n_triangle8; ++b)
        for (int j = 0; j < 8; ++j) {
            const float t = hit_triangle(o, d, sv.tri_a(b, j), sv.tri_ab(b, j), sv.tri_ac(b, j));
            if (t < lane_t[j]) { lane_t[j] = t; lane_b[j] = static_cast<uint32_t>(b); }
        }
    reduce(tri_i, tri_t);

    reset();
    for (size_t b = 0; b < s.n_sphere8; ++b)
        for (int j = 0; j < 8; ++j) {
            const float t = hit_sphere(o, d, sv.sph_c(b, j), sv.sph_r(b, j));
            if (t < lane_t[j]) { lane_t[j] = t; lane_b[j] = static_cast<uint32_t>(b); }
        }
    reduce(sph_i, sph_t);

    reset();
    for (size_t b = 0; b < s.n_ellipsoid8; ++b)
        for (int j = 0; j < 8; ++j) {
            const M34 minv = load_mat(s.ellipsoid8_inverse_transforms + b * 96, j);
            const M34 m = load_mat(s.ellipsoid8_transforms + b * 96, j);
            const V3 o2 = vmat_point(minv, o);
            V3 d2 = vmat_dir(minv, d);
            const bool absent = ellipsoid_absent(d2);
            const float inv_mag = 1.0f / sqrtf(vdot(d2, d2));
            d2 = scale(inv_mag, d2);
            const float t2 = hit_unit_sphere(o2, d2);
            const V3 p2 = add(o2, scale(t2, d2));
            const V3 p = vmat_point(m, p2);
            const V3 op = sub(p, o);
            const float dist = sqrtf(vdot(op, op));
            const float t = absent ? FLT_MAX : dist;
            if (t < lane_t[j]) { lane_t[j] = t; lane_b[j] = static_cast<uint32_t>(b); }
        }
    reduce(ell_i, ell_t);

    if (tri_i == INVALID && sph_i == INVALID && ell_i == INVALID) return {-1, 0, FLT_MAX};
    if (tri_t - sph_t <= TOL && tri_t - ell_t <= TOL) return {0, tri_i, tri_t};
    if (sph_t - tri_t <= TOL && sph_t - ell_t <= TOL) return {1, sph_i, sph_t};
    if (ell_t - tri_t <= TOL && ell_t - sph_t <= TOL) return {2, ell_i, ell_t};
    return {-1, 0, FLT_MAX};  // the reference asserts this cannot happen
}

inline Material load_material(const float* base, uint64_t count, size_t index) {
    Material m;
    if (index < count) std::memcpy(&m, base + index * 10, sizeof m);
    else std::memset(&m, 0, sizeof m);  // a hit on a padding lane reads past the vector in the reference
    return m;
}

// intersect(Ray, Scene, max_bounce_count): ray_tracing.cpp:475-735
void trace(const SceneView& sv, V3 o, V3 d, int max_bounce_count, float colour[3], uint32_t* primary_id, Counters& cn) {
    const rt_scene_desc& s = *sv.d;
    float c[3] = {0.0f, 0.0f, 0.0f};
    float w[3] = {1.0f, 1.0f, 1.0f};
    if (primary_id) *primary_id = RT_ID_MISS;
    for (int bounce = 0; bounce < max_bounce_count; ++bounce) {
        cn.closest_rays++;
        const Hit hit = closest_hit(sv, o, d);
        if (hit.kind < 0) break;
        cn.hits++;
        if (bounce == 0 && primary_id) *primary_id = RT_ID_MAKE(hit.kind, hit.index);

        const V3 p = add(o, scale(hit.t, d));
        Material mat;
        V3 n;
        const size_t b = hit.index / 8;
        const int j = static_cast<int>(hit.index % 8);
        if (hit.kind == 0) {
            const V3 a = sv.tri_a(b, j), ab = sv.tri_ab(b, j), ac = sv.tri_ac(b, j);
            const V3 vb = add(ab, a), vc = add(ac, a);                  // ray_tracing.cpp:631-632
            n = normalise(scross(sub(vb, a), sub(vc, a)));              // maths.cpp:304-306
            mat = load_material(s.triangle_materials, s.n_triangles, hit.index);
        } else if (hit.kind == 1) {
            n = normalise(sub(p, sv.sph_c(b, j)));                      // maths.cpp:308-310
            mat = load_material(s.sphere_materials, s.n_spheres, hit.index);
        } else {
            const M34 minv = load_mat(s.ellipsoid8_inverse_transforms + b * 96, j);
            const V3 pe = smat_point(minv, p);                           // maths.cpp:320-326
            const V3 ne = normalise(pe);
            n = normalise(smat_transpose_dir(minv, ne));
            mat = load_material(s.ellipsoid_materials, s.n_ellipsoids, hit.index);
        }

        for (int k = 0; k < 3; ++k) c[k] += w[k] * (s.ambient[k] + mat.emission[k]);  // :682

        const V3 to_eye = scale(-1.0f, d);
        const V3 shift = scale(2.0f * TOL, n);
        const V3 p_above = add(p, shift);
        if (s.has_directional_light) {
            if (!path_is_blocked(sv, p_above, nullptr, cn)) {
                const V3 to_light = scale(-1.0f, V3{s.directional_light[0], s.directional_light[1], s.directional_light[2]});
                const float di = fp_max(sdot(n, to_light), 0.0f);
                const V3 half = normalise(add(to_eye, to_light));
                const float si = fp_pow(fp_max(sdot(n, half), 0.0f), mat.shininess);
                for (int k = 0; k < 3; ++k) {
                    const float contribution = s.directional_light[3 + k] * (di * mat.diffuse[k] + si * mat.specular[k]);
                    c[k] += w[k] * contribution;
                }
            }
        }
        for (uint64_t l = 0; l < s.n_point_lights; ++l) {
            const float* light = s.point_light_sources + l * 6;
            if (path_is_blocked(sv, p_above, light, cn)) continue;
            const V3 to_light_full = sub(V3{light[0], light[1], light[2]}, p);
            const float dist = smagnitude(to_light_full);
            const V3 to_light = divide(to_light_full, dist);
            const float di = fp_max(sdot(n, to_light), 0.0f);
            const V3 half = normalise(add(to_eye, to_light));
            const float si = fp_pow(fp_max(sdot(n, half), 0.0f), mat.shininess);
            const float linear_term = s.attenuation[1] * dist;                  // :328-334
            const float quadratic_term = s.attenuation[2] * dist * dist;
            const float att = 1.0f / (s.attenuation[0] + linear_term + quadratic_term);
            for (int k = 0; k < 3; ++k) {
                const float contribution = (att * light[3 + k]) * (di * mat.diffuse[k] + si * mat.specular[k]);  // :722
                c[k] += w[k] * contribution;
            }
        }
        for (int k = 0; k < 3; ++k) w[k] *= mat.specular[k];
        o = p_above;
        d = sub(d, scale(2.0f * sdot(d, n), n));  // :728, not renormalised
    }
    colour[0] = c[0]; colour[1] = c[1]; colour[2] = c[2];
}

// ray_tracing.cpp:460-473
inline V3 ray_direction_through_pixel(unsigned px, unsigned py, float xo, float yo, const rt_camera_desc& cam, float half_w_px, float half_h_px) {
    const float alpha = cam.half_world[0] * (half_w_px - (static_cast<float>(px) + xo)) / half_w_px;
    const float beta = cam.half_world[1] * (half_h_px - (static_cast<float>(py) + yo)) / half_h_px;
    const V3 i{cam.basis_i[0], cam.basis_i[1], cam.basis_i[2]}, j{cam.basis_j[0], cam.basis_j[1], cam.basis_j[2]}, k{cam.basis_k[0], cam.basis_k[1], cam.basis_k[2]};
    return normalise(add(add(scale(alpha, i), scale(beta, j)), k));
}

// ray_tracing.cpp:265-297
inline void intersect_aabb(V3 o, V3 d, const float bb[6], float& min_distance, float& max_distance) {
    const float ix = 1.0f / d.x;
    const float x0 = (bb[0] - o.x) * ix, x1 = (bb[1] - o.x) * ix;
    const float min_x = fp_min(x0, x1), max_x = fp_max(x0, x1);
    const float iy = 1.0f / d.y;
    const float y0 = (bb[2] - o.y) * iy, y1 = (bb[3] - o.y) * iy;
    const float min_y = fp_min(y0, y1), max_y = fp_max(y0, y1);
    const float iz = 1.0f / d.z;
    const float z0 = (bb[4] - o.z) * iz, z1 = (bb[5] - o.z) * iz;
    const float min_z = fp_min(z0, z1), max_z = fp_max(z0, z1);
    min_distance = fp_max(min_z, fp_max(min_y, min_x));
    max_distance = fp_min(max_z, fp_min(max_y, max_x));
}

}  // namespace

extern "C" {

float rto_fp_pow(float x, float y) { return fp_pow(x, y); }
float rto_fp_min(float x, float y) { return fp_min(x, y); }
float rto_fp_max(float x, float y) { return fp_max(x, y); }
float rto_fp_sqrt(float x) { return fp_sqrt(x); }
void rto_ray_direction_through_pixel(unsigned px, unsigned py, float xo, float yo, const rt_camera_desc* c, float half_w_px, float half_h_px, float out[3]) {
    const V3 d = ray_direction_through_pixel(px, py, xo, yo, *c, half_w_px, half_h_px);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void rto_intersect_aabb(const float o[3], const float d[3], const float bb[6], float out[2]) {
    intersect_aabb(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, bb, out[0], out[1]);
}
void rto_intersect_triangle8(const float o[3], const float d[3], const float* t8, float out[8]) {
    for (int j = 0; j < 8; ++j)
        out[j] = hit_triangle(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, V3{t8[j], t8[8 + j], t8[16 + j]},
                              V3{t8[24 + j], t8[32 + j], t8[40 + j]}, V3{t8[48 + j], t8[56 + j], t8[64 + j]});
}
void rto_intersect_sphere8(const float o[3], const float d[3], const float* s8, float out[8]) {
    for (int j = 0; j < 8; ++j) out[j] = hit_sphere(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, V3{s8[j], s8[8 + j], s8[16 + j]}, s8[24 + j]);
}
// The ellipsoid routines on one batch of 8, same contract as ref_ellipsoid8 in ref_harness.cpp (ray_tracing.cpp:556-579).
void rto_ellipsoid8(const float o[3], const float d[3], const float* inv8, const float* fwd8, float* o2_out, float* d2_out, float* absent_out, float* t2_out, float* dist_out) {
    const V3 ro{o[0], o[1], o[2]}, rd{d[0], d[1], d[2]};
    for (int j = 0; j < 8; ++j) {
        const M34 minv = load_mat(inv8, j);
        const M34 m = load_mat(fwd8, j);
        const V3 o2 = vmat_point(minv, ro);
        V3 d2 = vmat_dir(minv, rd);
        o2_out[j] = o2.x; o2_out[8 + j] = o2.y; o2_out[16 + j] = o2.z;
        d2_out[j] = d2.x; d2_out[8 + j] = d2.y; d2_out[16 + j] = d2.z;
        const bool absent = ellipsoid_absent(d2);
        absent_out[j] = absent ? 1.0f : 0.0f;
        const float inv_mag = 1.0f / sqrtf(vdot(d2, d2));
        d2 = scale(inv_mag, d2);
        const float t2 = hit_unit_sphere(o2, d2);
        t2_out[j] = t2;
        const V3 p2 = add(o2, scale(t2, d2));
        const V3 p = vmat_point(m, p2);
        const V3 op = sub(p, ro);
        const float dist = sqrtf(vdot(op, op));
        dist_out[j] = absent ? FLT_MAX : dist;
    }
}
// path_is_blocked on a scene that holds one batch of ellipsoids and nothing else, same contract as ref_ellipsoid8_blocks.
int rto_ellipsoid8_blocks(const float start[3], const float light[3], int point, const float* inv8, float dir_out[3]) {
    rt_scene_desc desc;
    std::memset(&desc, 0, sizeof desc);
    desc.ellipsoid8_inverse_transforms = inv8;
    desc.n_ellipsoid8 = 1;
    const V3 from{start[0], start[1], start[2]};
    V3 dir;
    if (point) {
        const V3 to_light = sub(V3{light[0], light[1], light[2]}, from);
        dir = scale(1.0f / smagnitude(to_light), to_light);
    } else {
        desc.directional_light[0] = light[0]; desc.directional_light[1] = light[1]; desc.directional_light[2] = light[2];
        desc.has_directional_light = 1;
        dir = scale(-1.0f, V3{light[0], light[1], light[2]});
    }
    dir_out[0] = dir.x; dir_out[1] = dir.y; dir_out[2] = dir.z;
    Counters cn;
    return path_is_blocked(SceneView{&desc}, from, point ? light : nullptr, cn) ? 1 : 0;
}
void rto_trace(const rt_scene_desc* desc, const float o[3], const float d[3], int depth, float out[3]) {
    Counters cn;
    trace(SceneView{desc}, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, depth, out, nullptr, cn);
}

// main.cpp:44-57 with the scalar operators above; std::tan on float is tanf
void rto_camera_setup(const float cam[11], rt_camera_desc* out) {
    const V3 eye{cam[0], cam[1], cam[2]}, look_at{cam[3], cam[4], cam[5]}, up{cam[6], cam[7], cam[8]};
    const V3 k = normalise(sub(look_at, eye));
    const V3 i = normalise(scross(up, k));
    const V3 j = scross(k, i);
    out->eye[0] = eye.x; out->eye[1] = eye.y; out->eye[2] = eye.z;
    out->basis_i[0] = i.x; out->basis_i[1] = i.y; out->basis_i[2] = i.z;
    out->basis_j[0] = j.x; out->basis_j[1] = j.y; out->basis_j[2] = j.z;
    out->basis_k[0] = k.x; out->basis_k[1] = k.y; out->basis_k[2] = k.z;
    const float PI = 3.14159265358979323846264f;
    out->half_world[0] = tanf(0.5f * (cam[9] / 180.0f * PI));   // to_radians, maths.cpp:181-183
    out->half_world[1] = tanf(0.5f * (cam[10] / 180.0f * PI));
}

namespace { std::vector<double> g_worker_seconds; }
int rto_worker_seconds(double* out, int n) {
    const int have = static_cast<int>(g_worker_seconds.size());
    for (int i = 0; i < n && i < have; ++i) out[i] = g_worker_seconds[static_cast<size_t>(i)];
    return have;
}
// Same contract as ref_render in ref_harness.cpp (work units: segments of 4 pixels of a row), plus exact counters.
int rto_render(const rt_scene_desc* desc, const rt_camera_desc* cam, const rt_render_params* prm,
               uint8_t* rgb, uint32_t* ids, float* sample_colours, int n_threads, rt_stats* stats, double* render_seconds) {
    const SceneView sv{desc};
    const unsigned W = prm->width, H = prm->height;
    const unsigned x0 = prm->rect_w ? prm->rect_x0 : 0, y0 = prm->rect_w ? prm->rect_y0 : 0;
    const unsigned rw = prm->rect_w ? prm->rect_w : W, rh = prm->rect_w ? prm->rect_h : H;
    const float half_w_px = 0.5f * static_cast<float>(W), half_h_px = 0.5f * static_cast<float>(H);
    const V3 eye{cam->eye[0], cam->eye[1], cam->eye[2]};
    const uint32_t tile_world = prm->tile_world ? prm->tile_world : 1;
    const uint32_t tiles_x = (rw + RT_TILE - 1) / RT_TILE;

    constexpr unsigned SEGMENT = 4;
    const unsigned segments = (rw + SEGMENT - 1) / SEGMENT, units = rh * segments;
    std::atomic<unsigned> next{0};
    std::vector<Counters> per_thread(static_cast<size_t>(n_threads < 1 ? 1 : n_threads));
    g_worker_seconds.assign(per_thread.size(), 0.0);
    auto worker = [&](int tid) {
        Counters& cn = per_thread[static_cast<size_t>(tid)];
        const auto began = std::chrono::steady_clock::now();
        for (;;) {
            const unsigned u = next.fetch_add(1);
            if (u >= units) break;
            const unsigned r = u / segments;
            const unsigned y = y0 + r;
            for (unsigned xr = (u % segments) * SEGMENT; xr < rw && xr < (u % segments + 1) * SEGMENT; ++xr) {
                // the tile numbering of the library (include/rt_amd.h): rows of tiles rotated against each other
                const uint32_t tile_row = r / RT_TILE, place = (xr / RT_TILE + tile_row % tiles_x) % tiles_x;
                const uint32_t tile = tile_row * tiles_x + place;
                if (tile % tile_world != prm->tile_rank % tile_world) continue;
                const unsigned x = x0 + xr;
                float colour[3] = {0.0f, 0.0f, 0.0f};
                for (int sample = 0; sample < 16; ++sample) {
                    // render_work_queue.cpp:40-41: offsets 2.0, 2.25, 2.5, 2.75
                    const float xo = static_cast<float>(sample % 4) * 0.25f + 2.0f;
                    const float yo = static_cast<float>(sample / 4) * 0.25f + 2.0f;
                    const V3 dir = ray_direction_through_pixel(x, y, xo, yo, *cam, half_w_px, half_h_px);
                    float mn, mx;
                    intersect_aabb(eye, dir, desc->bounding_box, mn, mx);
                    const size_t si = (static_cast<size_t>(r) * rw + xr) * 16 + static_cast<size_t>(sample);
                    float c[3] = {0.0f, 0.0f, 0.0f};
                    uint32_t id = RT_ID_CULLED;
                    cn.samples++;
                    if (mx >= mn) trace(sv, eye, dir, prm->max_bounce_count, c, &id, cn);
                    else cn.culled++;
                    if (ids) ids[si] = id;
                    if (sample_colours) { sample_colours[si * 3] = c[0]; sample_colours[si * 3 + 1] = c[1]; sample_colours[si * 3 + 2] = c[2]; }
                    colour[0] += c[0]; colour[1] += c[1]; colour[2] += c[2];
                }
                uint8_t* px = rgb + (static_cast<size_t>(r) * rw + xr) * 3;
                for (int k = 0; k < 3; ++k) {
                    const float v = 0.0625f * colour[k];
                    // x86 truncating conversion to int32, low byte kept (render_work_queue.cpp:53-55)
                    px[k] = static_cast<uint8_t>(static_cast<int32_t>(v * 255.0f) & 0xFF);
                }
            }
        }
        g_worker_seconds[static_cast<size_t>(tid)] = std::chrono::duration<double>(std::chrono::steady_clock::now() - began).count();
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto& t : threads) t.join();
    if (render_seconds) *render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        Counters total;
        for (const Counters& c : per_thread) total.merge(c);
        std::memset(stats, 0, sizeof *stats);
        stats->samples = total.samples; stats->culled = total.culled; stats->closest_rays = total.closest_rays;
        stats->hits = total.hits; stats->shadow_rays = total.shadow_rays; stats->shadow_blocked = total.shadow_blocked;
        stats->closest_tests[0] = total.closest_rays * desc->n_triangles;
        stats->closest_tests[1] = total.closest_rays * desc->n_spheres;
        stats->closest_tests[2] = total.closest_rays * desc->n_ellipsoids;
        for (int k = 0; k < 3; ++k) stats->shadow_tests[k] = total.shadow_tests[k];
    }
    return 0;
}

}  // extern "C"
