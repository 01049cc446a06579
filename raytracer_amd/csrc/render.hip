// gfx950 (MI355X) implementation of the reference renderer's per-pixel loop, and the
// C ABI around it (include/rt_amd.h).
//
// One image is traced as a stream of ray batches instead of one recursive call per sample:
//   k_primary        : camera rays + scene-box cull       (ray_tracing.cpp:460-473, :265-297, render_work_queue.cpp:37-49)
//   k_closest_scan, k_closest_packet (box hierarchy), k_closest_tri_brute,
//   k_closest_sph*, k_closest_ell* (every pair): closest hit per primitive kind (ray_tracing.cpp:484-608)
//   k_hit            : kind choice, hit point, normal, ambient + emission (ray_tracing.cpp:610-686)
//   k_shadow_scan, k_shadow_dfs (box hierarchy), k_shadow_tri_pretest, k_shadow_tri_brute,
//   k_shadow_brute, k_shadow_sph/ell (every pair): any-hit toward one light (ray_tracing.cpp:347-458)
//   k_shade          : Blinn-Phong per unblocked light, attenuation, reflection (ray_tracing.cpp:687-728)
//   k_resolve        : 16-sample average and 8-bit conversion (render_work_queue.cpp:51-55)
// Rays that are still alive are kept in index lists compacted in order (k_scan_flags,
// k_scan_blocks, k_fill_list), so every launch runs with full wavefronts of rays that are
// neighbours in the image.
// Two ways through the primitive lists give the same bytes:
//   brute force (RT_FLAG_NO_FILTER): every ray against every primitive, records staged in LDS tiles and
//     read as broadcasts; a pair is decided by a sign test or a conservative pre-test where that is
//     certain and by the literal test of the reference otherwise (RT_FLAG_PLAIN_TEST: the literal test
//     on every pair; RT_FLAG_EXACT_STAGES: the literal test cut into stages that are skipped per wave);
//   default: conservative bounds (axis-aligned boxes, slab test) over a hierarchy of the primitives
//     decide almost every pair; what is left goes through the same pre-tests and the same literal test.
//     The closest hit is an atomicMin over (distance, scan index) keys, so the order of
//     the visits does not matter.
//
// Arithmetic contract: IEEE binary32, round to nearest, denormals kept, no contraction
// (-ffp-contract=off); the fused multiply-adds the reference executes are explicit
// __builtin_fmaf calls, in the reference's operand order.  Division and square root are
// the correctly rounded ones (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded on first use (load_rccl)
#include <dlfcn.h>

#include <algorithm>
#include <atomic>

#include <cfloat>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_internal.h"

// ------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error = "";

int rt_fail(int code, const char* message) {
    g_last_error = message ? message : "";
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char buf_[512];                                                                   \
            std::snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return rt_fail(RT_ERR_DEVICE, buf_);                                              \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side data
// ------------------------------------------------------------------------------------------
namespace {

constexpr float TOL = 1.0e-4f;            // maths.h:10
constexpr int BLOCK = 256;                // threads per workgroup in every kernel
constexpr int TILE_PRIMS = 256;           // primitives staged in LDS per step (triangles: 64 B each)
#ifndef RT_CLUSTER
#define RT_CLUSTER 16
#endif
constexpr uint32_t CLUSTER = RT_CLUSTER;          // children per node of the box hierarchy
constexpr uint32_t SAMPLES_PER_TILE = RT_TILE * RT_TILE * RT_SAMPLES_PER_PIXEL;  // 4096

// Primitive lists are stored in "scan order": entry k holds the primitive the reference
// keeps in lane j = k / nb of batch i = k % nb (original index 8*i + j).  The reference's winner
// among equal distances is the smallest (distance, lane, batch), not the smallest index
// (ray_tracing.cpp:489-516): that is the smallest (distance, k), which the brute-force scans get
// by walking k upward with a strict `<` and the hierarchy scans by an atomicMin over
// (distance, k) keys (pack_hit).

// Bounding records of one primitive kind (triangles, ellipsoids or spheres) for the conservative
// pre-test of the scans: axis-aligned boxes, 2 x float4 per node, (lo, 0) (hi, 0), see box_keeps.
// Four levels: leaves (one per primitive), clusters (CLUSTER consecutive leaves), top nodes
// (CLUSTER consecutive clusters) and super nodes (CLUSTER consecutive top nodes); the box of a node
// contains the boxes of everything below it.  A scan starts at the top nodes, or at the super
// nodes when the top level is long (LEVELS of the scan kernels).  Leaves are stored in CLUSTER ORDER: position s holds the primitive with scan index
// k_of[s], and consecutive positions are neighbours in space (leaf order of a kd tree, built by
// the host at scene creation).  One hierarchy serves every ray: from the eye, reflected, shadow.
struct Hierarchy {
    const uint32_t* k_of;
    const float4 *leaf_box, *cl_box, *top_box, *super_box, *root_box;
    const uint4 *cl_nodes, *top_nodes, *super_nodes, *root_nodes, *apex_nodes;   // the children of every cluster / top node / super node / root node / apex node, packed (PackedChildren)
    uint32_t n, n_clusters, n_tops, n_supers, n_roots, n_apex;
};

struct DevScene {
    const float4* tri;   // 4 x float4 per triangle: (a, n.x) (a_to_b, n.y) (a_to_c, n.z) (n.n, RN(1/n.n) or NaN, 0, 0)
    const float4* trif;  // pre-test records, 4 x float4 per triangle, see pretest_stage1
    const float4* trif_pos;  // the same in cluster order (th.k_of), with the scan index in the spare word [2].w
    Hierarchy th;        // bounding records of the triangles, see Hierarchy
    Hierarchy eh;        // bounding records of the ellipsoids
    Hierarchy sh;        // bounding records of the spheres
    const float4* sph;   // (centre, radius)
    const float4* ell;   // 6 x float4: rows of the inverse transform, rows of the transform
    const float4* ell_light;  // [point light][ellipsoid]: inverse transform applied to the light position
    const float* ell_ok;      // per ellipsoid: 1 if the staged form of the test is proven for the record (ell_record_is_regular), else 0
    const float* tri_mat;
    const float* sph_mat;
    const float* ell_mat;
    const float* lights;  // 8 floats per light: kind (0 directional, 1 point), xyz, rgb, 0; directional first
    uint32_t n_tri, tri_nb;   // scan entries (8 * batches), batches
    uint32_t n_sph, sph_nb;
    uint32_t n_ell, ell_nb;
    uint32_t n_tri_mat, n_sph_mat, n_ell_mat;
    uint32_t n_lights, has_directional;
    float bb[6];
    float ambient[3];
    float atten[3];
    float origin_bound;   // every bounce / shadow ray starts within this max-norm distance of 0 (scene box + slack)
    float box_bound;      // the box records hold for origins within this max-norm distance of 0 (>= origin_bound; covers the eye)
};

struct Frame {
    float eye[3], bi[3], bj[3], bk[3], hw[2];
    float half_w_px, half_h_px;
    uint32_t rect_x0, rect_y0, rect_w, rect_h;
    uint32_t tiles_x;
    uint32_t tile_rank, tile_world;
};

// Per-sample state of one chunk, structure of arrays, indexed by the sample's slot.
struct State {
    float *ox, *oy, *oz, *dx, *dy, *dz;   // ray of the current bounce
    float *nox, *noy, *noz, *ndx, *ndy, *ndz;   // ray of the next bounce (the host swaps the two sets per bounce)
    float *wr, *wg, *wb;                  // colour weighting
    float *cr, *cg, *cb;                  // accumulated colour
    float *px, *py, *pz, *nx, *ny, *nz;   // hit point and unit normal of the current bounce
    unsigned long long* key;              // closest triangle of the current bounce, see pack_hit
    unsigned long long* key_ell;          // closest ellipsoid of the current bounce, same format
    unsigned long long* key_sph;          // closest sphere of the current bounce, see pack_hit_signed
    uint32_t* hit_id;                     // RT_ID_MAKE(kind, original index) of the current bounce
    uint32_t* blocked;                    // bit l: light l of the current group is blocked
    uint32_t* list[2];                    // compacted slots of live rays, ping-pong
    uint32_t* counts;                     // counts[b]: live rays entering bounce b
    unsigned long long* flag_bits;        // per producing workgroup: 4 wave ballots of "this entry goes on" (ordered compaction)
    uint32_t* wg_offset;                  // per producing workgroup: position of its first entry in the next list
    uint32_t* scan_block;                 // per 1024 producing workgroups: their total, then the position of their first entry
    uint32_t* primary_id;                 // nullable
    unsigned long long* stats;            // ST_* counters
};

// Work counters live in STAT_COPIES copies, picked by workgroup number and summed by the host:
// one hot address per counter serialises the atomics of a whole launch (10 ns each).
constexpr int STAT_COPIES = 64;
enum { ST_SAMPLES, ST_CULLED, ST_CLOSEST_RAYS, ST_HITS, ST_SHADOW_RAYS, ST_SHADOW_BLOCKED,
       ST_SHADOW_TESTS_TRI, ST_SHADOW_TESTS_SPH, ST_SHADOW_TESTS_ELL, ST_FULL_TESTS, ST_PAIR_PRETESTS, ST_PAIR_PRETESTS_SHADOW, ST_LIVE, ST_QUEUED, ST_QUEUED_SHADOW, ST_FLOP, ST_FLOP_SHADOW,
       ST_FIRST_NODES, ST_FIRST_NODES_SHADOW, ST_WALK_BATCHES, ST_WALK_BATCHES_SHADOW, ST_WALK_WAVES, ST_WALK_WAVES_SHADOW,
       ST_BRUTE_STEPS, ST_BRUTE_STAGE1, ST_BRUTE_STAGE2, ST_BRUTE_EXACT, ST_BRUTE_STEPS_SHADOW, ST_BRUTE_STAGE1_SHADOW, ST_BRUTE_STAGE2_SHADOW, ST_BRUTE_EXACT_SHADOW, ST_COUNT };

struct V3 { float x, y, z; };

// ------------------------------------------------------------------------------------------
// arithmetic building blocks
// ------------------------------------------------------------------------------------------
#define DEV __device__ __forceinline__
#define RT_STRINGIFY_(x) #x
#define RT_STRINGIFY(x) RT_STRINGIFY_(x)

DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV float fp_min(float x, float y) { return x < y ? x : y; }   // minss: second operand on NaN (maths.cpp:147-152)
DEV float fp_max(float x, float y) { return x > y ? x : y; }   // maxss (maths.cpp:154-159)
DEV float sqrt_(float x) { return __builtin_sqrtf(x); }

DEV V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 scale(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
// scalar Vector dot as clang contracts it: fma(z, z', fma(x, x', y*y'))  (maths.cpp:195-197)
DEV float sdot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.x, b.x, a.y * b.y)); }
// scalar cross: fma(l.y, r.z, -(l.z*r.y))  (maths.cpp:203-209)
DEV V3 scross(V3 l, V3 r) { return {fma_(l.y, r.z, -(l.z * r.y)), fma_(l.z, r.x, -(l.x * r.z)), fma_(l.x, r.y, -(l.y * r.x))}; }
DEV V3 divide(V3 v, float s) { const float inv = 1.0f / s; return {inv * v.x, inv * v.y, inv * v.z}; }
DEV V3 normalise(V3 v) { return divide(v, sqrt_(sdot(v, v))); }
// AVX dot: mul, fmadd, fmadd in x, y, z order (ray_tracing.cpp:68-74)
DEV float vdot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// AVX cross: unfused (ray_tracing.cpp:76-83)
DEV V3 vcross(V3 l, V3 r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
DEV V3 xyz(float4 v) { return {v.x, v.y, v.z}; }

// rows as float4 (m0, m1, m2, m3); AVX forms of ray_tracing.cpp:105-126 and :208-227
DEV V3 vmat_point(float4 r0, float4 r1, float4 r2, V3 v) {
    return {r0.w + fma_(r0.z, v.z, fma_(r0.y, v.y, r0.x * v.x)),
            r1.w + fma_(r1.z, v.z, fma_(r1.y, v.y, r1.x * v.x)),
            r2.w + fma_(r2.z, v.z, fma_(r2.y, v.y, r2.x * v.x))};
}
DEV V3 vmat_dir(float4 r0, float4 r1, float4 r2, V3 v) {
    return {fma_(r0.z, v.z, fma_(r0.y, v.y, r0.x * v.x)),
            fma_(r1.z, v.z, fma_(r1.y, v.y, r1.x * v.x)),
            fma_(r2.z, v.z, fma_(r2.y, v.y, r2.x * v.x))};
}
// scalar Matrix*Vector: fma(m2, z, fma(m0, x, m1*y)) + m3 (maths.cpp:226-232)
DEV V3 smat_point(float4 r0, float4 r1, float4 r2, V3 v) {
    return {fma_(r0.z, v.z, fma_(r0.x, v.x, r0.y * v.y)) + r0.w,
            fma_(r1.z, v.z, fma_(r1.x, v.x, r1.y * v.y)) + r1.w,
            fma_(r2.z, v.z, fma_(r2.x, v.x, r2.y * v.y)) + r2.w};
}
// maths.cpp:312-318
DEV V3 smat_transpose_dir(float4 r0, float4 r1, float4 r2, V3 v) {
    return {fma_(r2.x, v.z, fma_(r0.x, v.x, r1.x * v.y)),
            fma_(r2.y, v.z, fma_(r0.y, v.x, r1.y * v.y)),
            fma_(r2.z, v.z, fma_(r0.z, v.x, r1.z * v.y))};
}

// Closest triangle of a ray as one 64-bit key: distance bits in the high word, scan index in the
// low word.  Distances the full test accepts are >= the tolerance, hence positive, so unsigned
// order of the keys is the order of (distance, scan index): atomicMin keeps the reference's winner.
DEV unsigned long long pack_hit(float t, uint32_t k) { return (static_cast<unsigned long long>(__float_as_uint(t)) << 32) | k; }
constexpr unsigned long long NO_HIT_KEY = ~0ull;
// The sphere test can return a distance below the tolerance, negative ones included (hit_sphere:
// an origin inside a sphere gets the near root).  The high word is therefore the distance mapped
// to an unsigned number of the same order (sign bit flipped for positive values, all bits for
// negative ones).  -0 sorts before +0 here while the reference treats them as equal; two spheres
// that return zeros of different sign for one ray are not expected to exist.
DEV unsigned long long pack_hit_signed(float t, uint32_t k) {
    const uint32_t b = __float_as_uint(t);
    return (static_cast<unsigned long long>(b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u)) << 32) | k;
}
DEV float unpack_distance_signed(unsigned long long key) {
    const uint32_t b = static_cast<uint32_t>(key >> 32);
    return __uint_as_float(b ^ ((b >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// fp_pow and its helpers (maths.cpp:9-139)
DEV int exponent_127(float x) { return static_cast<int>((__float_as_uint(x) >> 23) & 0xFFu); }
DEV float fp_log2(float x) {
    const int exponent = exponent_127(x) - 127;
    const float m = __uint_as_float((__float_as_uint(x) & 0x007FFFFFu) | 0x3F800000u);
    float acc = 0.0f, mult = 1.0f;
    acc = fma_(-2.80036404536401f, mult, acc); mult *= m;
    acc = fma_(5.0917108473873602f, mult, acc); mult *= m;
    acc = fma_(-3.5507929732452235f, mult, acc); mult *= m;
    acc = fma_(1.6311487949127739f, mult, acc); mult *= m;
    acc = fma_(-0.41656369634969576f, mult, acc); mult *= m;
    acc = fma_(0.044873611393688138f, mult, acc);
    const float result = acc + static_cast<float>(exponent);
    return fp_min(result, 0.0f);
}
DEV float fp_exp2(float x) {
    const float clamped = fp_max(x, -127.0f);
    const float ax = __uint_as_float(__float_as_uint(clamped) & 0x7FFFFFFFu);
    const int ip = static_cast<int>(ax);
    const float frac = ax - static_cast<float>(ip);
    const float inv_int = __uint_as_float((static_cast<uint32_t>(127 - ip) & 0xFFu) << 23);
    float acc = 0.0f, mult = 1.0f;
    acc = fma_(0.99994648282537313f, mult, acc); mult *= frac;
    acc = fma_(-0.69137342261578116f, mult, acc); mult *= frac;
    acc = fma_(0.23097554273105286f, mult, acc); mult *= frac;
    acc = fma_(-0.039602120115271706f, mult, acc);
    return inv_int * acc;
}
DEV float fp_pow(float x, float y) {
    if (exponent_127(y) == 0) return 1.0f;
    if (exponent_127(x) == 0) return 0.0f;
    return fp_exp2(y * fp_log2(x));
}

// ------------------------------------------------------------------------------------------
// ray / primitive tests (one ray, one primitive)
// ------------------------------------------------------------------------------------------
// ray_tracing.cpp:129-165 with the ray-independent n = a_to_b x a_to_c and n.n taken from the record
DEV float hit_triangle(V3 o, V3 d, float4 t0, float4 t1, float4 t2, float nn) {
    const V3 a = xyz(t0), ab = xyz(t1), ac = xyz(t2);
    const V3 n{t0.w, t1.w, t2.w};
    const float dn = vdot(d, n);
    const bool parallel = __builtin_fabsf(dn) < TOL;
    const float t = vdot(sub(a, o), n) / dn;
    const bool behind = t < TOL;
    const V3 p = add(o, scale(t, d));
    const V3 ap = sub(p, a);
    const float alpha = vdot(vcross(ab, ap), n) / nn;
    const float beta = vdot(vcross(ap, ac), n) / nn;
    const bool outside = (alpha < 0.0f) | (beta < 0.0f) | (alpha + beta > 1.0f);
    return (parallel | behind | outside) ? FLT_MAX : t;
}

// The same test evaluated in stages, for the brute-force scans, where almost every pair misses and
// the three IEEE divisions are the largest single cost of the test as written above (13 issue slots
// each on gfx950).  The result is that of hit_triangle bit for bit; the stages only avoid computing
// what the outcome does not depend on.
//   tri_front: d.n and (a-o).n exactly as hit_triangle computes them.  A lane whose |d.n| is below
//              the tolerance is "parallel"; a lane whose two dot products differ in sign has a
//              quotient t <= -0, which is "behind".  Either way the result is FLT_MAX.
//   tri_head:  t by the IEEE division, the hit point, and alpha's numerator x1 = (ab x ap).n, all
//              exactly as in hit_triangle.  alpha itself, RN(x1 / nn), is only ever compared, and
//              rounding is monotonic: with nn in [2^-60, 2^60], x1 < -2^-60 gives a quotient below
//              -2^-120, which rounds to a negative normal number (alpha < 0), and x1 > x1_max, the
//              record's RN(nn * (1 + 2^-20)) > nn * (1 + 2^-21), gives alpha > 1, which rejects the
//              pair whatever beta is (beta < 0 rejects; beta >= 0 makes alpha + beta > 1).  Two
//              comparisons of x1 itself: no multiplication, and t < tolerance is left to the tail
//              (with the sign test of the front, or an eye in front of the scene, it is rare).
//   tri_tail:  t >= tolerance; a1 = x1 * y with y = RN(1/nn) from the record is within 2^-22
//              (relative) of alpha, so a1 < -TINY implies alpha < 0 and a1 > 1 + BAND alpha > 1;
//              beta's numerator and b1 = x2 * y likewise: b1 < -TINY implies beta < 0,
//              a1 + b1 > 1 + BAND implies alpha + beta > 1, and a1, b1 > TINY with
//              a1 + b1 < 1 - BAND implies the pair is accepted with distance t.  Anything else
//              (a quotient within TINY of zero, a sum within BAND of one) is "unsure": the caller
//              then runs hit_triangle itself.
// The argument needs every intermediate to be finite, so that no comparison meets a NaN that
// hit_triangle would meet differently: the record holds y = x1_max = NaN unless |a|, |a_to_b|,
// |a_to_c| <= TAME_COORD per component and nn is within [2^-60, 2^60] (a tile of the list that
// holds such a record goes through hit_triangle: stage_tile), and the scans take this path only
// for rays with |o| <= TAME_COORD and |d| <= 2 per component (ray_is_tame).  Then for a lane that is not parallel |t| < 2^54,
// |ap| < 2^56 and both numerators stay below 2^96.
constexpr float BARY_TINY = 1.0e-30f;
constexpr float BARY_BAND = 9.5367431640625e-7f;   // 2^-20
constexpr float X1_TINY = 8.673617379884035e-19f;  // 2^-60
constexpr float TAME_COORD = 4096.0f;
DEV bool ray_is_tame(V3 o, V3 d) {
    return fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) <= TAME_COORD && fmaxf(fabsf(d.x), fmaxf(fabsf(d.y), fabsf(d.z))) <= 2.0f;
}
// Decisions that concern the whole wave are taken on lane masks (one bit per lane, in scalar registers).
using Mask = unsigned long long;
DEV Mask ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
struct TriFront { float dn, num; };
DEV TriFront tri_front(V3 o, V3 d, float4 t0, float4 t1, float4 t2, Mask& out) {
    const V3 n{t0.w, t1.w, t2.w};
    TriFront f;
    f.dn = vdot(d, n);
    f.num = vdot(sub(xyz(t0), o), n);
    out = ballot(__builtin_fabsf(f.dn) < TOL) | ballot(static_cast<int>(__float_as_uint(f.dn) ^ __float_as_uint(f.num)) < 0);
    return f;
}
// The same for rays that all start at one point (rays from the eye): (a-o).n is then a property of
// the triangle, computed once per tile by stage_triangle_tile with the very expression above.
// Triangles behind such rays are rare, so the sign test is left to tri_head's t < tolerance.
DEV TriFront tri_front_shared(V3 d, float4 t0, float4 t1, float4 t2, float num, Mask& out) {
    const V3 n{t0.w, t1.w, t2.w};
    TriFront f;
    f.dn = vdot(d, n);
    f.num = num;
    out = ballot(__builtin_fabsf(f.dn) < TOL);
    return f;
}
// num / den, correctly rounded, for the operands tri_head can meet on a lane that is still in play:
// |den| in [1e-4, 2^28], |num| < 2^41, signs equal.  This is the instruction sequence the compiler
// emits for an IEEE division (reciprocal, one Newton step on it, quotient, two residual corrections)
// without its range scaling (v_div_scale) and special-case fix-up (v_div_fixup), which do nothing
// for |num| >= 2^-100 here.  Below that the quotient is below 2^-86 either way, i.e. far under the
// tolerance it is compared with next, and its exact bits reach nothing else.
DEV float divide_in_range(float num, float den) {
    const float y0 = __builtin_amdgcn_rcpf(den);
    const float e = fma_(-den, y0, 1.0f);
    const float y1 = fma_(e, y0, y0);
    const float q0 = num * y1;
    const float r0 = fma_(-den, q0, num);
    const float q1 = fma_(r0, y1, q0);
    const float r1 = fma_(-den, q1, num);
    return fma_(r1, y1, q1);
}
struct TriHead { float t, x1; V3 ap; };
DEV TriHead tri_head(V3 o, V3 d, const TriFront& f, float4 t0, float4 t1, float4 t2, float x1_max, Mask& out) {
    const V3 n{t0.w, t1.w, t2.w};
    TriHead h;
    h.t = divide_in_range(f.num, f.dn);
    const V3 p = add(o, scale(h.t, d));
    h.ap = sub(p, xyz(t0));
    h.x1 = vdot(vcross(xyz(t1), h.ap), n);
    out |= ballot(h.x1 < -X1_TINY) | ballot(h.x1 > x1_max);
    return h;
}
// `out`: this lane is among those the earlier stages rejected
DEV float tri_tail(const TriHead& h, bool out, float4 t0, float4 t1, float4 t2, float y, bool& unsure) {
    const V3 n{t0.w, t1.w, t2.w};
    const float a1 = h.x1 * y;
    const float b1 = vdot(vcross(h.ap, xyz(t2)), n) * y;
    const float sum = a1 + b1;
    out = out | !(h.t >= TOL) | (a1 < -BARY_TINY) | (a1 > 1.0f + BARY_BAND) | (b1 < -BARY_TINY) | (sum > 1.0f + BARY_BAND);
    const bool in = (a1 > BARY_TINY) & (b1 > BARY_TINY) & (sum < 1.0f - BARY_BAND);
    unsure = !(out | in);
    return out ? FLT_MAX : h.t;
}

// Conservative pre-test of one ray against one triangle, in two stages.  A stage returns
// false only when the full test above is certain to return FLT_MAX.  It is not part of the
// reference; it exists because the full test spends most of its time in three IEEE divisions
// (about 17 issue slots each on gfx950) while almost every ray misses almost every triangle.
//
// With n, a and nn as in the full test, w1 = (n x a_to_b)/nn and w2 = (a_to_c x n)/nn, the
// full test's barycentrics are alpha = w1.(p - a), beta = w2.(p - a) in real arithmetic.
// Here t1 = ((a-o).n) * rcp(d.n) (the same two dot products, bit for bit, as the full test,
// and a reciprocal instead of a division), ap = t1*d - (a-o), alpha1 = w1.ap, beta1 = w2.ap.
// Rounding: the full test's hit point is off by 3 eps S per component (S = |o|_inf + |t| + |a|_inf),
// its cross, dot and division add 30 eps Q |ap|_inf + eps, propagated through w1 (|w1|_1 <= 6Q)
// 18 eps Q S: 48 eps Q S + eps in all.  Here rcp is within 1 ulp (5 eps |t| on t1, times 6Q),
// the stored w1 is off by 8 eps Q per component and the dot adds its own roundings: 54 + 30
// eps Q S.  Together |alpha1 - alpha| and |beta1 - beta| are bounded by
//     m = 132 eps Q (|o|_inf + |t| + |a|_inf) + eps,  Q = |n|_inf max(|a_to_b|_inf, |a_to_c|_inf)/nn,
// eps = 2^-24.  With G = 2^-15 Q (about four times 132 eps Q) and H = G |a|_inf + 2^-22,
// m <= G (|t1| + |o|_inf) + H.  Stage 1 keeps a ray when alpha1 is within [-2m, 1 + 2m], stage 2
// when min(alpha1, beta1, 1 - alpha1 - beta1) >= -2m.  The record stores Ga = 2G and
// Ha = 2H + 0.5, so that thr = Ga (|t1| + |o|_inf) + Ha = 0.5 + 2m.
// The distance itself is not tested: the few triangles a ray's line crosses behind its origin or
// beyond the current best go to the full test, which rejects them.  Any NaN or infinity makes
// the comparisons false, which keeps the ray.
// Record: (n, Ga) (a, Ha) (w1, 0) (w2, 0).
struct Pre { V3 ap; float alpha, thr; };
DEV bool pretest_stage1(V3 o, V3 d, float o_max, float4 f0, float4 f1, float4 f2, Pre& s) {
    const V3 n = xyz(f0), a = xyz(f1), w1 = xyz(f2);
    const float dn = vdot(d, n);
    const V3 ao = sub(a, o);
    const float num = vdot(ao, n);
    const float t1 = num * __builtin_amdgcn_rcpf(dn);
    s.ap = V3{fma_(t1, d.x, -ao.x), fma_(t1, d.y, -ao.y), fma_(t1, d.z, -ao.z)};
    s.alpha = vdot(s.ap, w1);
    s.thr = fma_(f0.w, __builtin_fabsf(t1) + o_max, f1.w);
    return !(__builtin_fabsf(s.alpha - 0.5f) > s.thr);
}
// The same for rays that all start at one point: a - o and (a - o).n are then properties of the triangle, computed once per
// tile (stage_pretest_tile) with the very expressions above.  p0 = (n, (a-o).n), p1 = (a - o, Ga), p2 = (w1, Ha).
DEV bool pretest_stage1_shared(V3 d, float o_max, float4 p0, float4 p1, float4 p2, Pre& s) {
    const float dn = vdot(d, xyz(p0));
    const float t1 = p0.w * __builtin_amdgcn_rcpf(dn);
    s.ap = V3{fma_(t1, d.x, -p1.x), fma_(t1, d.y, -p1.y), fma_(t1, d.z, -p1.z)};
    s.alpha = vdot(s.ap, xyz(p2));
    s.thr = fma_(p1.w, __builtin_fabsf(t1) + o_max, p2.w);
    return !(__builtin_fabsf(s.alpha - 0.5f) > s.thr);
}
// pretest_stage1 in two parts for the brute-force scans of rays with origins of their own, whose triangles lie behind them as
// often as ahead: the two dot products first - if they differ in sign the quotient t of the full test is <= -0, below the
// tolerance, and the pair is a certain miss (a NaN has no meaningful sign, but then the full test returns NaN or FLT_MAX, never a
// hit); a wave all of whose rays have the triangle behind them goes to the next one after 11 issue slots instead of 24.
struct PreFront { float dn, num; V3 ao; };
DEV PreFront pretest_front(V3 o, V3 d, float4 f0, float4 f1, bool& behind) {
    PreFront p;
    p.dn = vdot(d, xyz(f0));
    p.ao = sub(xyz(f1), o);
    p.num = vdot(p.ao, xyz(f0));
    behind = static_cast<int>(__float_as_uint(p.dn) ^ __float_as_uint(p.num)) < 0;
    return p;
}
DEV bool pretest_stage1_rest(const PreFront& p, V3 d, float o_max, float4 f0, float4 f1, float4 f2, Pre& s) {
    const float t1 = p.num * __builtin_amdgcn_rcpf(p.dn);
    s.ap = V3{fma_(t1, d.x, -p.ao.x), fma_(t1, d.y, -p.ao.y), fma_(t1, d.z, -p.ao.z)};
    s.alpha = vdot(s.ap, xyz(f2));
    s.thr = fma_(f0.w, __builtin_fabsf(t1) + o_max, f1.w);
    return !(__builtin_fabsf(s.alpha - 0.5f) > s.thr);
}
DEV bool pretest_stage2(const Pre& s, float4 f3) {
    const float beta = vdot(s.ap, xyz(f3));
    const float rest = 1.0f - (s.alpha + beta);
    const float u = __builtin_fminf(__builtin_fminf(s.alpha, beta), rest);
    return !(u < 0.5f - s.thr);
}

// Box pre-test (slab test) for any ray: the record is an axis-aligned box that contains every hit
// point the full tests can accept on the primitives below the node; a ray whose segment
// [0, limit] misses the box cannot hit them.  inv = 1/d per component (approximate reciprocal,
// 1 ulp; a zero component gives an infinity: the products are then infinities of the right sign,
// or NaN when the origin lies on the slab's plane, which min/max drop).  Each t carries a relative
// error below 6 * 2^-24 (difference, reciprocal, product), so an interval that is non-empty in exact
// arithmetic has tmin <= tmax * (1 + 2e-6) as evaluated; the sign of tmax is exact.  A box of
// infinities keeps every ray; NaN keeps the ray.
struct BoxRay { V3 o, inv; float limit; };
DEV BoxRay make_box_ray(V3 o, V3 d, float limit) {
    return BoxRay{o, V3{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)}, limit};
}
DEV bool box_keeps(const BoxRay& r, float4 lo, float4 hi) {
    const float ax = (lo.x - r.o.x) * r.inv.x, bx = (hi.x - r.o.x) * r.inv.x;
    const float ay = (lo.y - r.o.y) * r.inv.y, by = (hi.y - r.o.y) * r.inv.y;
    const float az = (lo.z - r.o.z) * r.inv.z, bz = (hi.z - r.o.z) * r.inv.z;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    const float far = __builtin_fminf(tmax, r.limit) * 1.000002f;
    return !(tmax < 0.0f) && !(tmin > far);
}
// The same with the box enlarged by `grow` (>= 0) on every side.
DEV bool box_keeps_grown(const BoxRay& r, float4 lo, float4 hi, float grow) {
    lo.x -= grow; lo.y -= grow; lo.z -= grow;
    hi.x += grow; hi.y += grow; hi.z += grow;
    return box_keeps(r, lo, hi);
}
// The CLUSTER = 16 children of a node in one 128-byte record (8 x uint4), which is what a lane that
// walks the node fetches: w[0] = (base.xyz, scale.x), w[1] = (scale.y, scale.z, -, -) as float bits,
// w[2..7] = byte u of each: lo.x, hi.x, lo.y, hi.y, lo.z, hi.z of child u in units of `scale` above
// `base`.  A child's box, fma(q, scale, base) per coordinate, contains its box record: the builder
// (k_prep_packed_children) evaluates this expression and moves q until it does.  A node with a
// child that is not a finite box has base = NaN: every child decodes to NaN and keeps every ray.
// Against 16 separate 32-byte records this is a quarter of the bytes and 8 loads instead of 32.
DEV uint32_t dword_of(uint4 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
// The slab test of a ray against the 16 children of a packed node without decoding their boxes: with
// A = scale * inv and B = (base - o) * inv per axis (once per node), the ray parameter of the plane at byte q is
// t = q * A + B, one fused multiply-add per plane instead of a decode, a subtraction and a product.
// Against box_keeps on the decoded box this evaluates t with an ABSOLUTE error: A and B are rounded (eps * 255 scale
// |inv| and 2 eps |base - o| |inv|), and the box the bytes stand for in real arithmetic may lie inside the decoded one
// by the rounding of the builder's check (eps (|base| + 255 scale)).  Their sum is below
//   S = 2 eps |inv| (|base| + 2 |base - o| + 510 scale)     per axis;
// the slack used is 2 eps |inv| (|base| + 3 |base - o| + 765 scale), more than twice what is needed.  Each plane is
// moved outward by it, so that the slab interval only grows.
// The sign of inv says which plane of an axis the ray meets first: load_oriented_node fetches the low and high bytes of
// the axes where inv < 0 the other way round (instead of a minimum and a maximum per child and axis), after which the "near" planes are
// t = q_near * A + (B - slack) and the "far" ones t = q_far * A + (B + slack).  What is left is relative (the final
// rounding, the approximate reciprocal: below 4 eps) and is covered by the factor 1.000002 on the far side, as in
// box_keeps; it is folded into the far planes' B (as an addition that bounds it from above) and into the limit.
// `grow` enlarges every box (sphere_growth).  A zero direction component gives infinities and NaNs on that axis:
// min3 / max3 drop the NaNs, and an infinity left over is the verdict box_keeps would reach (the origin outside the
// slab) or keeps the ray.  base = NaN (a node the builder gave up on) keeps every ray.
struct NodeRay { V3 a, b_near, b_far; float limit_far; };
DEV NodeRay make_node_ray(uint4 w0, uint4 w1, V3 o, V3 inv, float limit, float grow) {
    const V3 base{__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
    const V3 sc{__uint_as_float(w0.w), __uint_as_float(w1.x), __uint_as_float(w1.y)};
    constexpr float TWO_EPS = 2.0f * 5.9604644775390625e-8f, FAR = 1.000002f;
    NodeRay n;
    n.limit_far = limit * FAR;
    auto axis = [&](float base_q, float scale_q, float o_q, float inv_q, float& a, float& b_near, float& b_far) __attribute__((always_inline)) {
        const float bo = base_q - o_q;
        a = scale_q * inv_q;
        const float b = bo * inv_q;
        const float reach = fma_(3.0f, __builtin_fabsf(bo), fma_(765.0f, scale_q, __builtin_fabsf(base_q)));
        const float slack = fma_(reach, TWO_EPS, grow) * __builtin_fabsf(inv_q);
        b_near = b - slack;
        const float late = b + slack;
        // (FAR - 1) times the largest |t| a far plane of this node can have, so that q * A + b_far >= FAR * (q * A + late) where that is positive
        b_far = fma_(FAR - 1.0f, fma_(255.0f, __builtin_fabsf(a), __builtin_fabsf(late)), late);
    };
    axis(base.x, sc.x, o.x, inv.x, n.a.x, n.b_near.x, n.b_far.x);
    axis(base.y, sc.y, o.y, inv.y, n.a.y, n.b_near.y, n.b_far.y);
    axis(base.z, sc.z, o.z, inv.z, n.a.z, n.b_near.z, n.b_far.z);
    return n;
}
// Loads a node for a ray with reciprocal direction inv: w[2], w[4], w[6] are the bytes of the planes the ray meets first on
// x, y, z and w[3], w[5], w[7] those it meets last.  The record holds low then high per axis, and the lane picks with the
// address it loads from: no instruction and no register beyond the node's (exchanging the words after the load costs 24
// conditional moves and, at 96 VGPRs, spills).
DEV void load_oriented_node(const uint4* __restrict__ src, V3 inv, uint4 (&w)[8]) {
    const bool fx = inv.x < 0.0f, fy = inv.y < 0.0f, fz = inv.z < 0.0f;
    w[0] = src[0]; w[1] = src[1];
    w[2] = src[2u + fx]; w[3] = src[3u - fx];
    w[4] = src[4u + fy]; w[5] = src[5u - fy];
    w[6] = src[6u + fz]; w[7] = src[7u - fz];
}
DEV float packed_byte(uint4 bytes, int child) { return static_cast<float>((dword_of(bytes, child >> 2) >> (8 * (child & 3))) & 255u); }
// on a node loaded by load_oriented_node
DEV bool child_keeps(const NodeRay& n, const uint4 (&w)[8], int child) {
    const float nx = fma_(packed_byte(w[2], child), n.a.x, n.b_near.x), fx = fma_(packed_byte(w[3], child), n.a.x, n.b_far.x);
    const float ny = fma_(packed_byte(w[4], child), n.a.y, n.b_near.y), fy = fma_(packed_byte(w[5], child), n.a.y, n.b_far.y);
    const float nz = fma_(packed_byte(w[6], child), n.a.z, n.b_near.z), fz = fma_(packed_byte(w[7], child), n.a.z, n.b_far.z);
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    const float tmax = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return !(tmax < 0.0f) && !(tmin > __builtin_fminf(tmax, n.limit_far));
}
// bit u set: child first + u of the node exists (the last node of a level may be short)
DEV uint32_t children_present(uint32_t first, uint32_t n_child) {
    const uint32_t left = n_child - first;     // first < n_child for every queued node
    return left >= CLUSTER ? 0xFFFFu : (1u << left) - 1u;
}

// Sphere (centre, radius^2) around a box record, for the per-wave test (packet_sphere_keeps)
DEV float4 box_sphere(float4 lo, float4 hi) {
    const V3 c{0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    const V3 h{hi.x - c.x, hi.y - c.y, hi.z - c.z};
    const V3 l{c.x - lo.x, c.y - lo.y, c.z - lo.z};
    const V3 e{fmaxf(h.x, l.x), fmaxf(h.y, l.y), fmaxf(h.z, l.z)};
    return make_float4(c.x, c.y, c.z, vdot(e, e) * 1.00001f);
}

// Per-wave ("packet") pre-test, ahead of box_keeps.  A wave's rays are neighbours in the image, so
// they fit in a narrow bound: a cone of directions (unit axis A, cosine cw of the half angle) and a
// ball (O, rho) around the origins.  One lane tests one top node against the bound of the whole
// wave, so a node the wave cannot reach costs 1/64 of a lane-instruction sequence instead of one
// per ray.  The node's box is replaced by the sphere around it (box_sphere: centre c, radius r).
// Moving every origin to O enlarges the sphere to r + rho; seen from O it spans the half angle
// psi = asin((r + rho)/|c - O|), and the wave reaches it only if angle(A, c - O) <= theta_w + psi.
// sin(psi) is inflated by 2e-5 relative, which covers the approximate sqrt/rsq (1 ulp each) and the
// rounding of its cosine; cw is lowered by 1e-6 when it is formed and the threshold by another 2e-6
// (the dot products, |A| off 1 by a few 2^-24, the products).  The test needs theta_w < 90 degrees
// (cw > 0), so that the angle sum stays below 180; a wider wave, NaN or infinity anywhere keeps the node.
struct Packet { V3 A; float cw, sw; V3 O; float rho, o1; };
DEV float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
DEV float wave_min(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
DEV float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
// Bound of the wave's valid rays; d must be of unit length up to rounding.  o may be null (common point).
template <int R>
DEV Packet make_packet(const V3 (&d)[R], const V3* o, const bool (&valid)[R]) {
    Packet p;
    V3 sd{0.0f, 0.0f, 0.0f}, so{0.0f, 0.0f, 0.0f};
    float n = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (valid[r]) {
            sd = add(sd, d[r]);
            n += 1.0f;
            if (o) so = add(so, o[r]);
        }
    }
    sd = {wave_sum(sd.x), wave_sum(sd.y), wave_sum(sd.z)};
    n = wave_sum(n);
    p.A = normalise(sd);
    float cw = 2.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) cw = fminf(cw, vdot(d[r], p.A));
    cw = wave_min(cw) - 1.0e-6f;
    // NaN (no valid ray, or directions that cancel) must end as "keep": fminf drops NaN operands, so test A itself
    if (!(fabsf(p.A.x) <= 2.0f) || !(fabsf(p.A.y) <= 2.0f) || !(fabsf(p.A.z) <= 2.0f) || !(n > 0.0f)) cw = -1.0f;
    p.cw = cw;
    p.sw = __builtin_amdgcn_sqrtf(fmaxf(fma_(-cw, cw, 1.0f), 0.0f)) * 1.000001f;
    p.O = {0.0f, 0.0f, 0.0f};
    p.rho = 0.0f;
    p.o1 = 0.0f;
    if (o) {
        const float inv_n = 1.0f / fmaxf(n, 1.0f);
        so = {wave_sum(so.x) * inv_n, wave_sum(so.y) * inv_n, wave_sum(so.z) * inv_n};
        float far2 = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (valid[r]) { const V3 e = sub(o[r], so); far2 = fmaxf(far2, vdot(e, e)); }
        }
        far2 = wave_max(far2);
        p.O = so;
        p.o1 = fabsf(so.x) + fabsf(so.y) + fabsf(so.z);
        // o - O is rounded per component by 2^-24 (|o| + |O|): four times that, on top of the relative slack
        p.rho = fma_(__builtin_amdgcn_sqrtf(far2), 1.00001f, 4.8e-7f * p.o1);
        if (!(far2 < 1.0e30f)) p.cw = -1.0f;
    }
    return p;
}
// The bound is the same in every lane; say so to the compiler (scalar registers).
DEV Packet uniform_packet(Packet p) {
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    p.A = {uni(p.A.x), uni(p.A.y), uni(p.A.z)};
    p.cw = uni(p.cw); p.sw = uni(p.sw);
    p.O = {uni(p.O.x), uni(p.O.y), uni(p.O.z)};
    p.rho = uni(p.rho); p.o1 = uni(p.o1);
    return p;
}
DEV bool packet_sphere_keeps(const Packet& p, float4 c) {
    const V3 v = sub(xyz(c), p.O);
    const float len2 = vdot(v, v);
    const float inv_len = __builtin_amdgcn_rsqf(len2);
    // c - O is rounded per component by 2^-24 (|c| + |O|): the centre may be off by that much (four times it is added)
    const float slack = 2.4e-7f * ((fabsf(c.x) + fabsf(c.y)) + (fabsf(c.z) + p.o1));
    const float reach = (__builtin_amdgcn_sqrtf(c.w) + p.rho) + slack;
    const float sin_psi = reach * inv_len * 1.00002f;
    const float cos_psi = __builtin_amdgcn_sqrtf(fmaxf(fma_(-sin_psi, sin_psi, 1.0f), 0.0f));
    const float thr = fma_(p.cw, cos_psi, -(p.sw * sin_psi)) - 2.0e-6f;
    const float along = vdot(v, p.A) * inv_len;
    return !(along < thr) || !(sin_psi < 1.0f) || !(p.cw > 0.0f);
}

// A lower bound of the distance from any origin of the wave to any point of the sphere c (centre, radius^2) around a
// node's box; the roundings are covered as in packet_sphere_keeps.  NaN when anything is not finite (a comparison
// `> far` is then false: the node is kept).
DEV float packet_sphere_distance(const Packet& p, float4 c) {
    const V3 v = sub(xyz(c), p.O);
    const float len = __builtin_amdgcn_sqrtf(vdot(v, v));
    const float slack = 4.8e-7f * ((fabsf(c.x) + fabsf(c.y)) + (fabsf(c.z) + p.o1));
    const float reach = (__builtin_amdgcn_sqrtf(c.w) + p.rho) + slack;
    return fma_(len, 0.99998f, -(reach * 1.00002f));
}

// ray_tracing.cpp:169-206; the far root is replaced by the near one at :202 and that is kept.
// In two parts for the brute-force scans: a wave all of whose lanes have a negative discriminant skips the square root
// (IEEE, about 15 issue slots) and the rest; sphere_roots on a negative discriminant gives FLT_MAX, as the whole does.
struct SphereFront { float mid, disc; };
DEV SphereFront sphere_front(V3 o, V3 d, float4 s) {
    const V3 oc = sub(xyz(s), o);
    SphereFront f;
    f.mid = vdot(oc, d);
    const float oc2 = vdot(oc, oc);
    const float mid2 = f.mid * f.mid;
    const float perp2 = oc2 - mid2;
    const float r2 = s.w * s.w;
    f.disc = r2 - perp2;
    return f;
}
DEV float sphere_roots(const SphereFront& f) {
    const bool none = f.disc < 0.0f;
    const float h = sqrt_(f.disc);
    const float d0 = f.mid - h;
    const float d1 = f.mid + h;
    const bool inv0 = none | (d0 < TOL);
    const bool inv1 = none | (d1 < TOL);
    const float p0 = inv0 ? FLT_MAX : d0;
    const float p1 = inv1 ? FLT_MAX : d0;
    return p0 < p1 ? p0 : p1;
}
DEV float hit_sphere(V3 o, V3 d, float4 s) { return sphere_roots(sphere_front(o, d, s)); }

// ray_tracing.cpp:229-263; oo = o.o as :235 computes it
DEV float unit_sphere_distance(V3 o, float oo, V3 d) {
    const float mid = vdot(scale(-1.0f, o), d);
    const float mid2 = mid * mid;
    const float perp2 = oo - mid2;
    const float disc = 1.0f - perp2;
    const bool none = disc < 0.0f;
    const float h = sqrt_(disc);
    const float d0 = mid - h;
    const float d1 = mid + h;
    const bool inv0 = none | (d0 < TOL);
    const bool inv1 = none | (d1 < TOL);
    const float p0 = inv0 ? FLT_MAX : d0;
    const float p1 = inv1 ? FLT_MAX : d1;
    return p0 < p1 ? p0 : p1;
}
DEV float hit_unit_sphere(V3 o, V3 d) { return unit_sphere_distance(o, vdot(o, o), d); }

// closest-hit body for one ellipsoid, ray_tracing.cpp:555-579
DEV float hit_ellipsoid(V3 o, V3 d, float4 i0, float4 i1, float4 i2, float4 m0, float4 m1, float4 m2) {
    const V3 o2 = vmat_point(i0, i1, i2, o);
    V3 d2 = vmat_dir(i0, i1, i2, d);
    const bool absent = (d2.x < TOL) & (d2.y < TOL) & (d2.z < TOL);   // ray_tracing.cpp:336-344
    const float inv_mag = 1.0f / sqrt_(vdot(d2, d2));
    d2 = scale(inv_mag, d2);
    const float t2 = hit_unit_sphere(o2, d2);
    const V3 p2 = add(o2, scale(t2, d2));
    const V3 p = vmat_point(m0, m1, m2, p2);
    const V3 op = sub(p, o);
    const float dist = sqrt_(vdot(op, op));
    return absent ? FLT_MAX : dist;
}

// shadow body for one ellipsoid: true when it blocks (ray_tracing.cpp:376-401, :432-454).
// l2 is the light position in ellipsoid space (point lights only).
DEV bool ellipsoid_blocks(V3 o, V3 d, float4 i0, float4 i1, float4 i2, bool point, V3 l2) {
    const V3 o2 = vmat_point(i0, i1, i2, o);
    V3 d2 = vmat_dir(i0, i1, i2, d);
    const bool absent = (d2.x < TOL) & (d2.y < TOL) & (d2.z < TOL);
    const float inv_mag = 1.0f / sqrt_(vdot(d2, d2));
    d2 = scale(inv_mag, d2);
    float limit = FLT_MAX;
    if (point) {
        const V3 dl = sub(l2, o2);
        limit = sqrt_(vdot(dl, dl));
    }
    float t2 = hit_unit_sphere(o2, d2);
    t2 = absent ? FLT_MAX : t2;
    return t2 < limit;
}


// The ellipsoid tests in stages, for the brute-force scans.  As written (hit_ellipsoid, ellipsoid_blocks) a pair costs two
// matrix products, two IEEE square roots and an IEEE division before anything is decided, and almost every ray misses
// almost every ellipsoid.  The stages return what the test as written returns, bit for bit where it matters (see below),
// and skip what the outcome does not depend on; a stage is skipped when no ray of the wave needs it.
//   stage 0 (ell_parts, ell_certain_miss): O = M^-1 o, D = M^-1 d (not normalised), q = D.D and o'.o' = O.O exactly as the
//            test computes them.  With u = D/|D| in real arithmetic the test's discriminant is, up to rounding,
//            disc* = 1 - (O.O - (O.u)^2).  Its evaluated value differs from disc* by less than 23 eps (|O|^2 + 1)
//            (eps = 2^-24): the normalised direction carries 4.5 eps (q 3 eps, square root, reciprocal and product one
//            each), the mid point 8 eps |O|, its square 17.1 eps |O|^2, O.O 3 eps |O|^2, the two differences one eps of
//            their results each.  So disc* < -K (|O|^2 + 1) with K = 32 eps makes the evaluated discriminant negative:
//            the unit sphere is missed and the distance is FLT_MAX.  In terms of quantities that need no division,
//                 q (|O|^2 (1 - K) - (1 + K)) > (O.D)^2,
//            evaluated as  fma(O.O, 1 - K', -(1 + K')) * q  >  m * m,  m = O.D,  with K' = 2^-17 = 128 eps: the evaluation's
//            own roundings (m^2: 7.1 eps |O|^2 q, the left side 8.5 eps |O|^2 q) are covered by K' - K = 96 eps.  A NaN
//            anywhere makes the comparison false, which leaves the pair to the next stage.
//   stage 1 (ell_root): the absent mask, the normalisation and the unit-sphere distance t' exactly as written.  A lane that
//            is absent, or whose t' is not below FLT_MAX (a miss, or NaN), is out: as written its result is FLT_MAX
//            (absent), or the world distance of the "point" o' + FLT_MAX d', which is never below FLT_MAX for a regular
//            record (below), or NaN; none of them is below any closest distance so far nor below a light's distance.
//   tail    (ell_world_distance): p' = o' + t' d', p = M p', |p - o| exactly as written, for every lane of a wave that has
//            a lane with a root.
// Range.  The argument needs finite intermediates, no underflow that matters, and a transform that takes the point at
// FLT_MAX to an overflow: a record is REGULAR (ell_record_is_regular, in double, once per scene) if it is all zeros (a
// padding lane: D = 0, absent either way), or if every entry is finite, |M^-1| entries <= 2^20, translations <= 2^30,
// the singular values of both 3x3 parts lie within [2^-30, 2^20] resp. [2^-30, ...] and their condition numbers are at
// most 2^14.  Rays must be tame (ray_is_tame: |o| <= 4096, |d| <= 2 per component) with |d|^2 >= 1/4.  Then |O| < 2^35,
// q in [2^-63, 2^48], every product above is finite, and the left side of the stage-0 comparison can only be positive when
// |O|^2 > 1, i.e. when q |O|^2 is far above the underflow range.  For a miss, |p'| >= 0.99 FLT_MAX, so some component of
// M p' exceeds 0.98 sigma_min(M) FLT_MAX / sqrt(3) - |translation| > 1e29 in magnitude (rounding: 12 eps times the condition
// number, below 2 %), its square overflows, and the distance is +inf or NaN.  A tile of the list that holds a record that
// is not regular, and a wave that holds a ray that is not tame, go through the test as written.
constexpr float ELL_MARGIN = 7.62939453125e-6f;        // K' = 2^-17
DEV bool ray_is_tame_for_ellipsoids(V3 o, V3 d) { return ray_is_tame(o, d) && vdot(d, d) >= 0.25f; }
struct EllParts { V3 O, D; float oo, q; };
DEV float ell_margin_term(float oo) { return fma_(oo, 1.0f - ELL_MARGIN, -(1.0f + ELL_MARGIN)); }
DEV EllParts ell_parts(float4 i0, float4 i1, float4 i2, V3 o, V3 d) {
    EllParts e;
    e.O = vmat_point(i0, i1, i2, o);
    e.D = vmat_dir(i0, i1, i2, d);
    e.oo = vdot(e.O, e.O);
    e.q = vdot(e.D, e.D);
    return e;
}
// the same for rays that all start at one point: O and O.O are then properties of the ellipsoid (stage_ellipsoid_tile)
DEV EllParts ell_parts_shared(float4 i0, float4 i1, float4 i2, float oo, V3 d) {
    EllParts e;
    e.O = V3{i0.w, i1.w, i2.w};
    e.D = vmat_dir(i0, i1, i2, d);
    e.oo = oo;
    e.q = vdot(e.D, e.D);
    return e;
}
// `margin_term` = ell_margin_term(e.oo)
DEV bool ell_certain_miss(const EllParts& e, float margin_term) {
    const float m = vdot(e.O, e.D);
    return margin_term * e.q > m * m;
}
struct EllRoot { V3 d2; float t2; bool absent; };
DEV EllRoot ell_root(const EllParts& e) {
    EllRoot r;
    r.absent = (e.D.x < TOL) & (e.D.y < TOL) & (e.D.z < TOL);   // ray_tracing.cpp:336-344
    const float inv_mag = 1.0f / sqrt_(e.q);
    r.d2 = scale(inv_mag, e.D);
    r.t2 = unit_sphere_distance(e.O, e.oo, r.d2);
    return r;
}
DEV bool ell_root_is_out(const EllRoot& r) { return r.absent | !(r.t2 < FLT_MAX); }
DEV float ell_world_distance(V3 o, const EllParts& e, const EllRoot& r, float4 m0, float4 m1, float4 m2) {
    const V3 p2 = add(e.O, scale(r.t2, r.d2));
    const V3 p = vmat_point(m0, m1, m2, p2);
    const V3 op = sub(p, o);
    const float dist = sqrt_(vdot(op, op));
    return r.absent ? FLT_MAX : dist;
}
// shadow rays: the limit in ellipsoid space (ray_tracing.cpp:388-389) and the comparison (:392-397, :445-450)
DEV bool ell_root_blocks(const EllParts& e, const EllRoot& r, bool point, V3 l2) {
    float limit = FLT_MAX;
    if (point) {
        const V3 dl = sub(l2, e.O);
        limit = sqrt_(vdot(dl, dl));
    }
    const float t2 = r.absent ? FLT_MAX : r.t2;
    return t2 < limit;
}
constexpr float ELL_LIGHT_COORD = 1.152921504606847e18f;    // 2^60: a light farther than this in ellipsoid space makes its tile go as written

// smallest and largest eigenvalue of a symmetric 3x3 matrix (trigonometric form; absolute error a few 1e-16 of the largest)
__device__ void sym3_eig_range(double a00, double a01, double a02, double a11, double a12, double a22, double& lo, double& hi) {
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    const double p2 = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * p1;
    const double pp = sqrt(p2 / 6.0);
    if (!(pp > 1.0e-300)) { lo = hi = q; return; }
    const double b00 = (a00 - q) / pp, b11 = (a11 - q) / pp, b22 = (a22 - q) / pp, b01 = a01 / pp, b02 = a02 / pp, b12 = a12 / pp;
    const double det = b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02);
    const double phi = acos(fmax(-1.0, fmin(1.0, det / 2.0))) / 3.0;
    hi = q + 2.0 * pp * cos(phi);
    lo = q + 2.0 * pp * cos(phi + 2.0943951023931953);
}
// `rec`: the six rows of an ellipsoid record (inverse transform, transform)
__device__ bool ell_record_is_regular(const float4* rec) {
    bool finite = true, zero = true;
    double a_max = 0.0, t_max = 0.0, f_max = 0.0, w_max = 0.0;
    for (int row = 0; row < 3; ++row) {
        const float4 i = rec[row], m = rec[3 + row];
        finite = finite && isfinite(i.x) && isfinite(i.y) && isfinite(i.z) && isfinite(i.w) && isfinite(m.x) && isfinite(m.y) && isfinite(m.z) && isfinite(m.w);
        zero = zero && i.x == 0.0f && i.y == 0.0f && i.z == 0.0f && i.w == 0.0f;
        a_max = fmax(a_max, fmax(fabs(static_cast<double>(i.x)), fmax(fabs(static_cast<double>(i.y)), fabs(static_cast<double>(i.z)))));
        f_max = fmax(f_max, fmax(fabs(static_cast<double>(m.x)), fmax(fabs(static_cast<double>(m.y)), fabs(static_cast<double>(m.z)))));
        t_max = fmax(t_max, fabs(static_cast<double>(i.w)));
        w_max = fmax(w_max, fabs(static_cast<double>(m.w)));
    }
    if (!finite) return false;
    if (zero) return true;
    constexpr double P20 = 1048576.0, P30 = 1073741824.0, PM60 = 8.673617379884035e-19, PM28 = 3.725290298461914e-9;
    if (!(a_max <= P20 && t_max <= P30 && f_max <= P30 && w_max <= P30)) return false;
    for (int part = 0; part < 2; ++part) {
        const double M[3][3] = {{rec[3 * part].x, rec[3 * part].y, rec[3 * part].z}, {rec[3 * part + 1].x, rec[3 * part + 1].y, rec[3 * part + 1].z}, {rec[3 * part + 2].x, rec[3 * part + 2].y, rec[3 * part + 2].z}};
        double g[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g[r][c] = M[0][r] * M[0][c] + M[1][r] * M[1][c] + M[2][r] * M[2][c];
        double lo, hi;
        sym3_eig_range(g[0][0], g[0][1], g[0][2], g[1][1], g[1][2], g[2][2], lo, hi);
        if (!(lo >= PM60 && lo >= PM28 * hi && hi <= P20 * P20 * P20)) return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// workgroup helpers
// ------------------------------------------------------------------------------------------
DEV unsigned long long* stat_slot(const State& st, int which) { return st.stats + (blockIdx.x & (STAT_COPIES - 1)) * ST_COUNT + which; }
DEV void block_add_stat(unsigned long long* stat, unsigned long long value) {
    // wave reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) value += __shfl_down(value, off);
    if ((threadIdx.x & 63u) == 0 && value) atomicAdd(stat, value);
}

// ------------------------------------------------------------------------------------------
// scene preparation: reference AoSoA-8 -> scan-ordered records
// ------------------------------------------------------------------------------------------
// Full-test record of one triangle: (a, n.x) (a_to_b, n.y) (a_to_c, n.z) (n.n, y, 0, x1_max) with n = a_to_b x a_to_c
// and n.n as ray_tracing.cpp:131,148 compute them, and y = RN(1/n.n), x1_max = RN(n.n * (1 + 2^-20)) for the staged
// form of the test (tri_tail, tri_head), both NaN where that form is not proven.
DEV void make_tri_record(V3 a, V3 ab, V3 ac, float4* o) {
    const V3 n = vcross(ab, ac);
    const float nn = vdot(n, n);
    o[0] = make_float4(a.x, a.y, a.z, n.x);
    o[1] = make_float4(ab.x, ab.y, ab.z, n.y);
    o[2] = make_float4(ac.x, ac.y, ac.z, n.z);
    const float coord_max = fmaxf(fmaxf(fabsf(a.x), fmaxf(fabsf(a.y), fabsf(a.z))),
                                  fmaxf(fmaxf(fabsf(ab.x), fmaxf(fabsf(ab.y), fabsf(ab.z))), fmaxf(fabsf(ac.x), fmaxf(fabsf(ac.y), fabsf(ac.z)))));
    const bool regular = coord_max <= TAME_COORD && nn >= 8.673617379884035e-19f && nn <= 1.152921504606847e18f;   // 2^-60, 2^60
    o[3] = make_float4(nn, regular ? 1.0f / nn : __builtin_nanf(""), 0.0f, regular ? nn * (1.0f + BARY_BAND) : __builtin_nanf(""));
}
// Pre-test record of one triangle (pretest_stage1 / pretest_stage2): (n, Ga) (a, Ha) (w1, 0) (w2, 0); `full` is its full-test record.
// Nothing here has to match the reference's rounding.
DEV void make_pretest_record(V3 a, V3 ab, V3 ac, const float4* full, float4* q) {
    const V3 n{full[0].w, full[1].w, full[2].w};
    const float nn = full[3].x;
    const float inv_nn = 1.0f / nn;
    const V3 w1 = scale(inv_nn, vcross(n, ab));
    const V3 w2 = scale(inv_nn, vcross(ac, n));
    const float n_max = fmaxf(fabsf(n.x), fmaxf(fabsf(n.y), fabsf(n.z)));
    const float e_max = fmaxf(fmaxf(fabsf(ab.x), fmaxf(fabsf(ab.y), fabsf(ab.z))), fmaxf(fabsf(ac.x), fmaxf(fabsf(ac.y), fabsf(ac.z))));
    const float a_max = fmaxf(fabsf(a.x), fmaxf(fabsf(a.y), fabsf(a.z)));
    const float G = 1.0001f * (1.0f / 32768.0f) * (n_max * e_max * inv_nn);
    const float H = 1.0001f * (G * a_max) + (1.0f / 4194304.0f);
    q[0] = make_float4(n.x, n.y, n.z, 2.0f * G);
    q[1] = make_float4(a.x, a.y, a.z, 2.0f * H + 0.5f);
    q[2] = make_float4(w1.x, w1.y, w1.z, 0.0f);
    q[3] = make_float4(w2.x, w2.y, w2.z, 0.0f);
}
__global__ void __launch_bounds__(BLOCK) k_prep_triangles(const float* __restrict__ tri8, uint32_t nb, float4* __restrict__ out, float4* __restrict__ outf) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= 8u * nb) return;
    const uint32_t j = k / nb, i = k % nb;
    const float* p = tri8 + static_cast<size_t>(i) * 72 + j;
    const V3 a{p[0], p[8], p[16]}, ab{p[24], p[32], p[40]}, ac{p[48], p[56], p[64]};
    float4* o = out + static_cast<size_t>(k) * 4;
    make_tri_record(a, ab, ac, o);
    make_pretest_record(a, ab, ac, o, outf + static_cast<size_t>(k) * 4);
}

__global__ void __launch_bounds__(BLOCK) k_reorder_pretest_records(const float4* __restrict__ trif, const uint32_t* __restrict__ k_of, uint32_t n, float4* __restrict__ out) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= n) return;
    const uint32_t k = k_of[pos];
    const float4* f = trif + static_cast<size_t>(k) * 4;
    float4* o = out + static_cast<size_t>(pos) * 4;
    o[0] = f[0]; o[1] = f[1];
    o[2] = make_float4(f[2].x, f[2].y, f[2].z, __uint_as_float(k));
    o[3] = f[3];
}

__global__ void __launch_bounds__(BLOCK) k_prep_spheres(const float* __restrict__ sph8, uint32_t nb, float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= 8u * nb) return;
    const uint32_t j = k / nb, i = k % nb;
    const float* p = sph8 + static_cast<size_t>(i) * 32 + j;
    out[k] = make_float4(p[0], p[8], p[16], p[24]);
}

__global__ void __launch_bounds__(BLOCK) k_prep_ellipsoids(const float* __restrict__ inv8, const float* __restrict__ fwd8, uint32_t nb, float4* __restrict__ out, float* __restrict__ ok) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= 8u * nb) return;
    const uint32_t j = k / nb, i = k % nb;
    const float* pi = inv8 + static_cast<size_t>(i) * 96 + j;
    const float* pf = fwd8 + static_cast<size_t>(i) * 96 + j;
    float4* o = out + static_cast<size_t>(k) * 6;
    for (int row = 0; row < 3; ++row) {
        o[row] = make_float4(pi[(row * 4 + 0) * 8], pi[(row * 4 + 1) * 8], pi[(row * 4 + 2) * 8], pi[(row * 4 + 3) * 8]);
        o[3 + row] = make_float4(pf[(row * 4 + 0) * 8], pf[(row * 4 + 1) * 8], pf[(row * 4 + 2) * 8], pf[(row * 4 + 3) * 8]);
    }
    ok[k] = ell_record_is_regular(o) ? 1.0f : 0.0f;
}

// Box records (box_keeps), built in double.  Leaves: a box that contains every hit point o + t d the
// full test of the kind can accept on the primitive, for origins within origin_bound (max norm) of
// 0, rounded outward to binary32; clusters and top nodes: the union of their children, which is exact.
DEV void store_box(float4* __restrict__ out, uint32_t pos, const double* lo, const double* hi, bool ok) {
    const float inf = __builtin_inff();
    float4 l = make_float4(-inf, -inf, -inf, 0.0f), h = make_float4(inf, inf, inf, 0.0f);   // keeps every ray
    if (ok && isfinite(lo[0]) && isfinite(lo[1]) && isfinite(lo[2]) && isfinite(hi[0]) && isfinite(hi[1]) && isfinite(hi[2])) {
        l = make_float4(nextafterf(static_cast<float>(lo[0]), -inf), nextafterf(static_cast<float>(lo[1]), -inf), nextafterf(static_cast<float>(lo[2]), -inf), 0.0f);
        h = make_float4(nextafterf(static_cast<float>(hi[0]), inf), nextafterf(static_cast<float>(hi[1]), inf), nextafterf(static_cast<float>(hi[2]), inf), 0.0f);
    }
    out[static_cast<size_t>(pos) * 2] = l;
    out[static_cast<size_t>(pos) * 2 + 1] = h;
}
// Triangles.  Let m bound the barycentric error of the full test (as in the two-stage pre-test,
// with |t| bounded by T, the distance from an origin to the farthest point of the triangle, with
// slack): whenever the full test accepts, its hit point lies within delta of the triangle scaled
// by (1 + 3m) about its centroid, where delta = 26 eps T covers the hit point being off the plane.
__global__ void __launch_bounds__(BLOCK) k_prep_tri_box(const float4* __restrict__ tri, const uint32_t* __restrict__ k_of, uint32_t n_tri, float origin_bound, float4* __restrict__ out) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= n_tri) return;
    const uint32_t k = k_of[pos];
    const float4 t0 = tri[static_cast<size_t>(k) * 4], t1 = tri[static_cast<size_t>(k) * 4 + 1], t2 = tri[static_cast<size_t>(k) * 4 + 2], t3 = tri[static_cast<size_t>(k) * 4 + 3];
    const double a[3] = {t0.x, t0.y, t0.z}, n[3] = {t0.w, t1.w, t2.w}, ab[3] = {t1.x, t1.y, t1.z}, ac[3] = {t2.x, t2.y, t2.z};
    const double nn = static_cast<double>(t3.x);
    auto max3 = [](const double* v) { return fmax(fabs(v[0]), fmax(fabs(v[1]), fabs(v[2]))); };
    auto len3 = [](const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {0.0, 0.0, 0.0};
    bool ok = false;
    if (nn > 0.0) {
        const double eps = 5.9604644775390625e-8;   // 2^-24
        const double B = origin_bound;
        const double Q = max3(n) * fmax(max3(ab), max3(ac)) / nn;
        const double edges = len3(ab) + len3(ac);
        const double T = 2.0 * (1.7320508075688772 * (max3(a) + B) + edges);   // origin to any accepted hit, with slack
        const double m = 2.0 * (48.0 * eps * Q * (B + T + max3(a)) + eps);
        const double delta = 2.0 * 13.0 * eps * T;
        ok = m < 1.0e6;
        for (int i = 0; i < 3; ++i) {
            const double c = a[i] + (ab[i] + ac[i]) / 3.0;
            const double v[3] = {a[i], a[i] + ab[i], a[i] + ac[i]};
            lo[i] = 1.0e300; hi[i] = -1.0e300;
            for (int j = 0; j < 3; ++j) {
                const double w = c + (1.0 + 3.0 * m) * (v[j] - c);
                lo[i] = fmin(lo[i], w); hi[i] = fmax(hi[i], w);
            }
            lo[i] -= delta; hi[i] += delta;
        }
    }
    store_box(out, pos, lo, hi, ok);
}
__global__ void __launch_bounds__(BLOCK) k_prep_cluster_box(const float4* __restrict__ member, uint32_t n_member, uint32_t n_nodes, float4* __restrict__ out) {
    const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n_nodes) return;
    const uint32_t first = c * CLUSTER, count = min(CLUSTER, n_member - first);
    const float inf = __builtin_inff();
    float4 lo = make_float4(inf, inf, inf, 0.0f), hi = make_float4(-inf, -inf, -inf, 0.0f);
    bool all = false;
    for (uint32_t t = 0; t < count; ++t) {
        const float4 l = member[static_cast<size_t>(first + t) * 2], h = member[static_cast<size_t>(first + t) * 2 + 1];
        // NaN anywhere (not produced by the leaf kernels) would be dropped by min/max: keep every ray instead
        if (!(l.x <= h.x) || !(l.y <= h.y) || !(l.z <= h.z)) all = true;
        lo.x = fminf(lo.x, l.x); lo.y = fminf(lo.y, l.y); lo.z = fminf(lo.z, l.z);
        hi.x = fmaxf(hi.x, h.x); hi.y = fmaxf(hi.y, h.y); hi.z = fmaxf(hi.z, h.z);
    }
    if (all) { lo = make_float4(-inf, -inf, -inf, 0.0f); hi = make_float4(inf, inf, inf, 0.0f); }
    out[static_cast<size_t>(c) * 2] = lo;
    out[static_cast<size_t>(c) * 2 + 1] = hi;
}

// PackedChildren of every node of one level from the box records of the level below
__global__ void __launch_bounds__(BLOCK) k_prep_packed_children(const float4* __restrict__ member, uint32_t n_member, uint32_t n_nodes, uint4* __restrict__ out) {
    static_assert(CLUSTER == 16, "a packed record holds 16 children");
    const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n_nodes) return;
    const uint32_t first = c * CLUSTER, count = min(CLUSTER, n_member - first);
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    bool bad = false;
    for (uint32_t t = 0; t < count; ++t) {
        const float4 l = member[static_cast<size_t>(first + t) * 2], h = member[static_cast<size_t>(first + t) * 2 + 1];
        const float cl[3] = {l.x, l.y, l.z}, ch[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            if (!(fabsf(cl[a]) < inf) || !(fabsf(ch[a]) < inf) || !(cl[a] <= ch[a])) bad = true;
            lo[a] = fminf(lo[a], cl[a]); hi[a] = fmaxf(hi[a], ch[a]);
        }
    }
    float scale[3] = {0.0f, 0.0f, 0.0f};
    uint32_t bytes[6][4] = {};
    if (!bad) {
        for (int a = 0; a < 3; ++a) {
            const float ext = hi[a] - lo[a];
            scale[a] = ext > 0.0f ? nextafterf(ext / 255.0f, inf) * 1.000001f : 0.0f;
            if (!(scale[a] < inf) || fma_(255.0f, scale[a], lo[a]) < hi[a]) bad = true;
        }
    }
    if (!bad) {
        for (uint32_t t = 0; t < count; ++t) {
            const float4 l = member[static_cast<size_t>(first + t) * 2], h = member[static_cast<size_t>(first + t) * 2 + 1];
            const float cl[3] = {l.x, l.y, l.z}, ch[3] = {h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                float ql = 0.0f, qh = 0.0f;
                if (scale[a] > 0.0f) {
                    ql = fminf(fmaxf(floorf((cl[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
                    qh = fminf(fmaxf(ceilf((ch[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
                    while (ql > 0.0f && fma_(ql, scale[a], lo[a]) > cl[a]) ql -= 1.0f;    // the decoded box must contain the record
                    while (qh < 255.0f && fma_(qh, scale[a], lo[a]) < ch[a]) qh += 1.0f;
                }
                if (fma_(ql, scale[a], lo[a]) > cl[a] || fma_(qh, scale[a], lo[a]) < ch[a]) bad = true;
                bytes[2 * a][t >> 2] |= static_cast<uint32_t>(ql) << (8 * (t & 3));
                bytes[2 * a + 1][t >> 2] |= static_cast<uint32_t>(qh) << (8 * (t & 3));
            }
        }
    }
    uint4* o = out + static_cast<size_t>(c) * 8;
    if (bad) {
        const uint32_t nan = 0x7FC00000u;
        o[0] = make_uint4(nan, nan, nan, 0u);
        o[1] = make_uint4(0u, 0u, 0u, 0u);
        for (int w = 2; w < 8; ++w) o[w] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    o[0] = make_uint4(__float_as_uint(lo[0]), __float_as_uint(lo[1]), __float_as_uint(lo[2]), __float_as_uint(scale[0]));
    o[1] = make_uint4(__float_as_uint(scale[1]), __float_as_uint(scale[2]), 0u, 0u);
    for (int w = 0; w < 6; ++w) o[2 + w] = make_uint4(bytes[w][0], bytes[w][1], bytes[w][2], bytes[w][3]);
}

// ---- bounding records of the ellipsoids (all in double) ----
// The full test (hit_ellipsoid / ellipsoid_blocks) maps the ray with the inverse transform
// (rows i0..i2: linear part Ainv, translation t) and tests the unit sphere, so the solid it
// tests is { x : |Ainv x + t| <= 1 }: centre c = -B t with B = Ainv^-1, and every point of it
// within sigma_max(B) of c.  In binary32 the test also accepts rays that miss by a little:
// with o2 the mapped origin (|o2| <= D2 = sigma_max(Ainv) * D, D = largest distance of an origin
// from c), perp2 = o2.o2 - mid^2 carries up to 8 eps D2^2 of rounding, the mapped origin is off
// by e_o <= 4 eps (3 sigma_max(Ainv) |o|_inf + |t|_inf) per component and the mapped direction by
// an angle <= 4 sqrt(3) eps kappa (kappa = condition number).  A ray the full test accepts therefore
// passes, in ellipsoid space, within rho = sqrt(1 + 16 eps D2^2) + 2 (sqrt(3) e_o + 4 sqrt(3) eps kappa D2)
// of the origin (every noise term doubled), and in world space within sigma_max(B) * rho of c.
struct EllBound { double c[3]; double r_geom, s_inv, kappa, t_max; bool ok; };
__device__ double sym3_max_eig(double a00, double a01, double a02, double a11, double a12, double a22) {
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    const double p2 = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * p1;
    const double pp = sqrt(p2 / 6.0);
    if (!(pp > 1.0e-300)) return q;
    const double b00 = (a00 - q) / pp, b11 = (a11 - q) / pp, b22 = (a22 - q) / pp, b01 = a01 / pp, b02 = a02 / pp, b12 = a12 / pp;
    const double det = b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02);
    const double r = fmax(-1.0, fmin(1.0, det / 2.0));
    return (q + 2.0 * pp * cos(acos(r) / 3.0)) * (1.0 + 1.0e-9) + 1.0e-300;
}
__device__ EllBound ell_bound(float4 i0, float4 i1, float4 i2) {
    EllBound b;
    const double A[3][3] = {{i0.x, i0.y, i0.z}, {i1.x, i1.y, i1.z}, {i2.x, i2.y, i2.z}};
    const double t[3] = {i0.w, i1.w, i2.w};
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    b.ok = isfinite(det) && det != 0.0 && isfinite(t[0]) && isfinite(t[1]) && isfinite(t[2]);
    b.c[0] = b.c[1] = b.c[2] = 0.0; b.r_geom = b.s_inv = b.kappa = b.t_max = 0.0;
    if (!b.ok) return b;
    double B[3][3];   // inverse of A
    B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    for (int r = 0; r < 3; ++r) b.c[r] = -(B[r][0] * t[0] + B[r][1] * t[1] + B[r][2] * t[2]);
    auto gram_max = [](const double M[3][3]) {   // largest eigenvalue of M^T M
        double g[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g[r][c] = M[0][r] * M[0][c] + M[1][r] * M[1][c] + M[2][r] * M[2][c];
        return sym3_max_eig(g[0][0], g[0][1], g[0][2], g[1][1], g[1][2], g[2][2]);
    };
    b.r_geom = sqrt(gram_max(B));
    b.s_inv = sqrt(gram_max(A));
    b.kappa = b.r_geom * b.s_inv;
    b.t_max = fmax(fabs(t[0]), fmax(fabs(t[1]), fabs(t[2])));
    b.ok = isfinite(b.r_geom) && isfinite(b.s_inv) && isfinite(b.c[0]) && isfinite(b.c[1]) && isfinite(b.c[2]) && b.r_geom > 0.0;
    return b;
}
// world-space radius around b.c that a ray the full test accepts stays within; D, o_max as above.  Infinite = no bound.
__device__ double ell_reach(const EllBound& b, double D, double o_max) {
    const double eps = 5.9604644775390625e-8;   // 2^-24
    const double D2 = b.s_inv * D;
    const double e_o = 4.0 * eps * (3.0 * b.s_inv * o_max + b.t_max);
    const double rho = sqrt(1.0 + 16.0 * eps * D2 * D2) + 2.0 * (1.7320508075688772 * e_o + 4.0 * 1.7320508075688772 * eps * b.kappa * D2);
    if (!(rho < 8.0)) return INFINITY;
    return b.r_geom * rho * (1.0 + 1.0e-6);
}
// box records of the ellipsoids: the hit point lies within ell_reach of the centre
__global__ void __launch_bounds__(BLOCK) k_prep_ell_box(const float4* __restrict__ ell, const uint32_t* __restrict__ k_of, uint32_t n, float origin_bound, float4* __restrict__ out) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= n) return;
    const uint32_t k = k_of[pos];
    const EllBound b = ell_bound(ell[static_cast<size_t>(k) * 6], ell[static_cast<size_t>(k) * 6 + 1], ell[static_cast<size_t>(k) * 6 + 2]);
    double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {0.0, 0.0, 0.0};
    bool ok = false;
    if (b.ok) {
        const double B = origin_bound;
        const double c_len = sqrt(b.c[0] * b.c[0] + b.c[1] * b.c[1] + b.c[2] * b.c[2]);
        const double D = 1.7320508075688772 * B + c_len;
        const double r = ell_reach(b, D, B) * (1.0 + 1.0e-6);
        ok = r < 1.0e30;
        for (int i = 0; i < 3; ++i) { lo[i] = b.c[i] - r; hi[i] = b.c[i] + r; }
    }
    store_box(out, pos, lo, hi, ok);
}

// ---- bounding records of the spheres (all in double) ----
// hit_sphere accepts when r*r - (oc.oc - (oc.d)^2) >= 0 in binary32, oc = c - o.  With |oc| <= D the
// roundings of oc (relative 2^-24 per component), of the two dot products, of the square and of
// the difference move perp2 by less than 11 eps D^2, and a direction longer than 1 lowers it by
// (|d|^2 - 1) (oc.d)^2: reflected directions are not renormalised.  The records allow for
// |d|^2 <= 1 + SPH_DIR_SLACK: a line the test accepts then passes within
// reach = sqrt(r^2 + (16 eps + SPH_DIR_SLACK) D^2) of the centre (sph_reach), and its hit point
// o + t d, at sqrt(perp^2 + h^2) of the centre with h^2 = r^2 - perp2 as evaluated, within
// sph_reach(reach): the noise enters twice (k_prep_sph_box).  A ray whose direction is longer than
// that (as evaluated: SPH_DIR_CHECK) has every box enlarged by sphere_growth() in the closest-hit
// scan (box_keeps_grown); the shadow scans, whose directions were normalised a moment ago, keep
// every sphere for a wave that holds such a ray (a directional light of the caller's making).
// The distance the test returns can be negative (hit_sphere: origin inside the sphere); the origin
// then lies in the box, which the slab test keeps.
constexpr float SPH_DIR_SLACK = 32.0f * 5.9604644775390625e-8f;
constexpr float SPH_DIR_CHECK = 1.0f + 28.0f * 5.9604644775390625e-8f;   // |d|^2 as evaluated carries up to 3 eps
// Enlargement of every sphere's box for a ray with direction d (not normalised), origins and centres
// within max-norm B of 0: the excess e of |d|^2 over what the records allow adds e |oc|^2 to the
// squared distance perp^2 the test accepts and e (r^2 + D^2) to the hit point's offset along the
// ray, together below 36 B^2 e (|oc|, D <= 2 sqrt(3) B, r <= B).  With s the other additive terms of
// the box (s < 400 eps B): (sqrt(reach^2 + x) + s)^2 <= (reach + s)^2 + 1.5 x + 2 s^2; the growth
// below is sqrt(72 B^2 e + 2 s^2), which covers 1.5 x.
DEV float sphere_growth(V3 d, float B) {
    const float e = vdot(d, d) - SPH_DIR_CHECK;
    if (!(e > 0.0f)) return 0.0f;
    return __builtin_sqrtf(fma_(e * 72.0072f, B * B, 1.2e-9f * (B * B))) * 1.0001f;
}
__device__ double sph_reach(double r, double D) {
    const double eps = 5.9604644775390625e-8;   // 2^-24
    return sqrt(r * r * (1.0 + 4.0 * eps) + (16.0 * eps + SPH_DIR_SLACK) * D * D) * (1.0 + 1.0e-6) + 4.0 * eps * D;
}
__device__ bool sph_usable(float4 s) { return isfinite(s.x) && isfinite(s.y) && isfinite(s.z) && isfinite(s.w); }
// box records of the spheres
__global__ void __launch_bounds__(BLOCK) k_prep_sph_box(const float4* __restrict__ sph, const uint32_t* __restrict__ k_of, uint32_t n, float origin_bound, float4* __restrict__ out) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= n) return;
    const float4 s = sph[k_of[pos]];
    double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {0.0, 0.0, 0.0};
    const bool ok = sph_usable(s);
    if (ok) {
        const double c[3] = {s.x, s.y, s.z};
        const double c_len = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const double D = 1.7320508075688772 * origin_bound + c_len;
        const double r1 = sph_reach(fabs(static_cast<double>(s.w)), D);
        const double r = sph_reach(r1 * (1.0 + 1.0e-5), D);   // and the cross term 2 h (oc.d) (|d|^2 - 1) of the hit point's offset
        for (int i = 0; i < 3; ++i) { lo[i] = c[i] - r; hi[i] = c[i] + r; }
    }
    store_box(out, pos, lo, hi, ok);
}

// inverse transform applied to each point light's position (ray_tracing.cpp:388), once per scene
__global__ void __launch_bounds__(BLOCK) k_prep_ellipsoid_lights(const float4* __restrict__ ell, uint32_t n_ell, const float* __restrict__ point_lights,
                                                                 uint32_t n_point, float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_ell) return;
    const float4 i0 = ell[static_cast<size_t>(k) * 6], i1 = ell[static_cast<size_t>(k) * 6 + 1], i2 = ell[static_cast<size_t>(k) * 6 + 2];
    for (uint32_t l = 0; l < n_point; ++l) {
        const V3 lp{point_lights[l * 6], point_lights[l * 6 + 1], point_lights[l * 6 + 2]};
        const V3 l2 = vmat_point(i0, i1, i2, lp);
        out[static_cast<size_t>(l) * n_ell + k] = make_float4(l2.x, l2.y, l2.z, 0.0f);
    }
}

// Pixels of a 16x16 tile are numbered along a Z curve, so that any 4, 16 or 64 consecutive pixels
// form a 2x2, 4x4 or 8x8 block: rays that are neighbours in a list are neighbours in the image in
// both directions, which keeps the packet bounds of the scans narrow.
// Tile numbers.  Tile u of a rectangle tiles_x tiles wide lies in row u / tiles_x, and the rows are rotated against each
// other: its column is (u % tiles_x - row) mod tiles_x.  Rank r of n renders the tiles u = r, r + n, ...: rotated, they lie
// on diagonals instead of in columns when n divides tiles_x (3840 pixels, 8 ranks: per-rank times within 2 to 4 % of each
// other where columns gave 7 to 10 %).
__host__ __device__ inline uint32_t tile_row(uint32_t u, uint32_t tiles_x) { return u / tiles_x; }
__host__ __device__ inline uint32_t tile_column(uint32_t u, uint32_t tiles_x) {
    const uint32_t place = u % tiles_x, shift = (u / tiles_x) % tiles_x;
    return place >= shift ? place - shift : place + tiles_x - shift;
}
__host__ __device__ inline uint32_t tile_number(uint32_t column, uint32_t row, uint32_t tiles_x) {
    const uint32_t place = column + row % tiles_x;
    return row * tiles_x + (place >= tiles_x ? place - tiles_x : place);
}
DEV uint32_t tile_pixel_x(uint32_t q) { return (q & 1u) | ((q >> 1) & 2u) | ((q >> 2) & 4u) | ((q >> 3) & 8u); }
DEV uint32_t tile_pixel_y(uint32_t q) { return ((q >> 1) & 1u) | ((q >> 2) & 2u) | ((q >> 3) & 4u) | ((q >> 4) & 8u); }

// ray_tracing.cpp:460-473
DEV V3 primary_direction(const Frame& f, uint32_t x, uint32_t y, float xo, float yo) {
    const float alpha = f.hw[0] * (f.half_w_px - (static_cast<float>(x) + xo)) / f.half_w_px;
    const float beta = f.hw[1] * (f.half_h_px - (static_cast<float>(y) + yo)) / f.half_h_px;
    const V3 bi{f.bi[0], f.bi[1], f.bi[2]}, bj{f.bj[0], f.bj[1], f.bj[2]}, bk{f.bk[0], f.bk[1], f.bk[2]};
    return normalise(add(add(scale(alpha, bi), scale(beta, bj)), bk));
}
// ray_tracing.cpp:271-294; bb = min_x max_x min_y max_y min_z max_z
DEV void scene_box_distances(V3 o, V3 d, const float* bb, float& min_distance, float& max_distance) {
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

// ------------------------------------------------------------------------------------------
// k_primary: one thread per sample of the chunk
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_primary(Frame f, DevScene sc, State st, uint32_t first_tile_slot) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;   // the grid covers the chunk's samples exactly
    const uint32_t tile_slot = first_tile_slot + s / SAMPLES_PER_TILE;
    const uint32_t tile = tile_slot * f.tile_world + f.tile_rank;
    const uint32_t in_tile = s % SAMPLES_PER_TILE;
    const uint32_t pixel = in_tile / RT_SAMPLES_PER_PIXEL, sample = in_tile % RT_SAMPLES_PER_PIXEL;
    const uint32_t rx = tile_column(tile, f.tiles_x) * RT_TILE + tile_pixel_x(pixel);
    const uint32_t ry = tile_row(tile, f.tiles_x) * RT_TILE + tile_pixel_y(pixel);
    const bool inside = rx < f.rect_w && ry < f.rect_h;
    const uint32_t x = f.rect_x0 + rx, y = f.rect_y0 + ry;

    // render_work_queue.cpp:40-41: offsets 2.0, 2.25, 2.5, 2.75 (exact in binary32)
    const float xo = static_cast<float>(sample % 4u) * 0.25f + 2.0f;
    const float yo = static_cast<float>(sample / 4u) * 0.25f + 2.0f;
    const V3 d = primary_direction(f, x, y, xo, yo);
    const V3 o{f.eye[0], f.eye[1], f.eye[2]};
    float min_distance, max_distance;
    scene_box_distances(o, d, sc.bb, min_distance, max_distance);
    const bool live = inside && (max_distance >= min_distance);   // render_work_queue.cpp:46

    st.cr[s] = 0.0f; st.cg[s] = 0.0f; st.cb[s] = 0.0f;
    if (live) {
        st.ox[s] = o.x; st.oy[s] = o.y; st.oz[s] = o.z;
        st.dx[s] = d.x; st.dy[s] = d.y; st.dz[s] = d.z;
        st.wr[s] = 1.0f; st.wg[s] = 1.0f; st.wb[s] = 1.0f;
    }
    if (live) { st.key[s] = NO_HIT_KEY; st.key_ell[s] = NO_HIT_KEY; st.key_sph[s] = NO_HIT_KEY; }
    if (st.primary_id) st.primary_id[s] = live ? RT_ID_MISS : RT_ID_CULLED;
    const unsigned long long live_mask = __ballot(live);    // list 0 is built from these by k_scan_flags + k_fill_list
    if ((threadIdx.x & 63u) == 0) st.flag_bits[blockIdx.x * 4u + (threadIdx.x >> 6)] = live_mask;
}

// ------------------------------------------------------------------------------------------
// ordered compaction: the entries a kernel flagged (one ballot per wave) become the next list,
// in the order of the entries themselves
// ------------------------------------------------------------------------------------------
// The order matters for speed, not for results: with it, consecutive list entries are neighbours
// in the image, which is what makes the per-wave pre-test of the scans narrow.  (Appending with one
// atomic per workgroup leaves the workgroups in their completion order, scattered over the part
// of the image that is in flight.)
// k_scan_flags: per block of SCAN_THREADS producing workgroups, the exclusive prefix of their flag
// counts (wg_offset) and the block's total (scan_block[b]).
// k_scan_blocks: one workgroup; exclusive prefix over the block totals in place, the list length,
// and the work counters that are list lengths (one atomic per launch instead of one per wave).
constexpr int SCAN_THREADS = 1024;
DEV uint32_t block_exclusive_scan(uint32_t value, uint32_t* partial /* SCAN_THREADS words */, uint32_t& total) {
    partial[threadIdx.x] = value;
    __syncthreads();
    for (uint32_t step = 1; step < SCAN_THREADS; step <<= 1) {      // inclusive Hillis-Steele scan
        const uint32_t add = threadIdx.x >= step ? partial[threadIdx.x - step] : 0u;
        __syncthreads();
        partial[threadIdx.x] += add;
        __syncthreads();
    }
    total = partial[SCAN_THREADS - 1];
    return partial[threadIdx.x] - value;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_flags(State st, const uint32_t* __restrict__ n_in_ptr, uint32_t n_in_fixed) {
    __shared__ uint32_t partial[SCAN_THREADS];
    const uint32_t n_in = n_in_ptr ? *n_in_ptr : n_in_fixed;
    const uint32_t n_wg = (n_in + BLOCK - 1) / BLOCK;
    if (blockIdx.x * SCAN_THREADS >= n_wg) return;
    const uint32_t g = blockIdx.x * SCAN_THREADS + threadIdx.x;
    uint32_t count = 0;
    if (g < n_wg) {
        const ulonglong2 a = reinterpret_cast<const ulonglong2*>(st.flag_bits)[g * 2u], b = reinterpret_cast<const ulonglong2*>(st.flag_bits)[g * 2u + 1u];
        count = __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y);
    }
    uint32_t total;
    const uint32_t before = block_exclusive_scan(count, partial, total);
    if (g < n_wg) st.wg_offset[g] = before;
    if (threadIdx.x == 0) st.scan_block[blockIdx.x] = total;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_blocks(State st, const uint32_t* __restrict__ n_in_ptr, uint32_t n_in_fixed, uint32_t* __restrict__ n_out, int stat_in, int stat_out) {
    __shared__ uint32_t partial[SCAN_THREADS];
    const uint32_t n_in = n_in_ptr ? *n_in_ptr : n_in_fixed;
    const uint32_t n_wg = (n_in + BLOCK - 1) / BLOCK;
    const uint32_t n_blocks = (n_wg + SCAN_THREADS - 1) / SCAN_THREADS;   // <= SCAN_THREADS: a chunk has at most 2^20 producing workgroups
    const uint32_t mine = threadIdx.x < n_blocks ? st.scan_block[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t before = block_exclusive_scan(mine, partial, total);
    if (threadIdx.x < n_blocks) st.scan_block[threadIdx.x] = before;
    if (threadIdx.x == 0) {
        *n_out = total;
        if (stat_in >= 0) atomicAdd(stat_slot(st, stat_in), static_cast<unsigned long long>(n_in));
        if (stat_out >= 0) atomicAdd(stat_slot(st, stat_out), static_cast<unsigned long long>(total));
    }
}
// k_fill_list: workgroup g places the flagged ones of entries g*256.. (the entry itself from list_in, or its index).
__global__ void __launch_bounds__(BLOCK) k_fill_list(State st, const uint32_t* __restrict__ n_in_ptr, uint32_t n_in_fixed,
                                                     const uint32_t* __restrict__ list_in, uint32_t* __restrict__ list_out) {
    const uint32_t n_in = n_in_ptr ? *n_in_ptr : n_in_fixed;
    const uint32_t base = blockIdx.x * BLOCK;
    if (base >= n_in) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t pos = st.scan_block[blockIdx.x / SCAN_THREADS] + st.wg_offset[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) pos += __popcll(st.flag_bits[blockIdx.x * 4u + w]);
    const unsigned long long mask = st.flag_bits[blockIdx.x * 4u + wave];
    if ((mask >> lane) & 1ull) {
        pos += __popcll(mask & ((1ull << lane) - 1ull));
        list_out[pos] = list_in ? list_in[base + threadIdx.x] : base + threadIdx.x;
    }
}

// ------------------------------------------------------------------------------------------
// closest-hit scans: one kernel per primitive kind, then k_hit
// ------------------------------------------------------------------------------------------
// Each kind has its own launch and leaves its closest hit as a 64-bit key (pack_hit) in
// st.key / st.key_sph / st.key_ell; k_hit combines the three.

// LDS staging: a tile of `count` records of WORDS float4 each is copied by the whole
// workgroup; `lds` must hold TILE_PRIMS * WORDS float4.
template <int WORDS>
DEV void stage_tile(float4* lds, const float4* __restrict__ src, uint32_t first, uint32_t count) {
    const uint32_t n4 = count * WORDS;
    const float4* g = src + static_cast<size_t>(first) * WORDS;
    for (uint32_t i = threadIdx.x; i < n4; i += BLOCK) lds[i] = g[i];
}
// A tile of full-test records of triangles, one triangle per thread, laid out for the order in which the staged
// test wants the words: four arrays of TILE_PRIMS float4,
//   [0] (n, (a-o).n)   [1] (a, x1_max)   [2] (a_to_b, y)   [3] (a_to_c, n.n)
// so that a step that ends at the front reads 16 bytes of the record (rays with one origin; 32 otherwise), one that
// ends at the head 48, and only the tail and the test as written all 64.  (a-o).n is computed here, with the very
// expression of tri_front, when all rays of the workgroup start at `o` (`shared`: rays from the eye); else the word
// is unused.  Returns true if this thread's record is outside the range the staged form of the test is proven for
// (make_tri_record marks it with NaN): the workgroup then takes the whole tile through the test as written.
static_assert(TILE_PRIMS == BLOCK, "stage_triangle_tile: one triangle per thread");
DEV bool stage_triangle_tile(float4* lds, const float4* __restrict__ src, uint32_t first, uint32_t count, bool shared, V3 o) {
    if (threadIdx.x >= count) return false;
    const float4* g = src + static_cast<size_t>(first + threadIdx.x) * 4;
    const float4 t0 = g[0], t1 = g[1], t2 = g[2], t3 = g[3];
    const V3 n{t0.w, t1.w, t2.w};
    lds[threadIdx.x] = make_float4(n.x, n.y, n.z, shared ? vdot(sub(xyz(t0), o), n) : 0.0f);
    lds[TILE_PRIMS + threadIdx.x] = make_float4(t0.x, t0.y, t0.z, t3.w);
    lds[2 * TILE_PRIMS + threadIdx.x] = make_float4(t1.x, t1.y, t1.z, t3.y);
    lds[3 * TILE_PRIMS + threadIdx.x] = make_float4(t2.x, t2.y, t2.z, t3.x);
    return !(t3.w == t3.w);
}
// record k of such a tile in the words of make_tri_record: (a, n.x) (a_to_b, n.y) (a_to_c, n.z) (n.n, y, (a-o).n, x1_max)
struct TileTriangle { float4 t0, t1, t2, t3; };
DEV TileTriangle tile_triangle(const float4* lds, uint32_t k) {
    const float4 r0 = lds[k], r1 = lds[TILE_PRIMS + k], r2 = lds[2 * TILE_PRIMS + k], r3 = lds[3 * TILE_PRIMS + k];
    return TileTriangle{make_float4(r1.x, r1.y, r1.z, r0.x), make_float4(r2.x, r2.y, r2.z, r0.y), make_float4(r3.x, r3.y, r3.z, r0.z), make_float4(r3.w, r2.w, r0.w, r1.w)};
}

// rays base + r*BLOCK + thread of the live list; lanes past the end repeat the first ray and are not stored
template <int R>
DEV void load_rays(const State& st, const uint32_t* __restrict__ list_in, uint32_t base, uint32_t n_rays,
                   uint32_t (&slot)[R], bool (&valid)[R], V3 (&o)[R], V3 (&d)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = base + r * BLOCK + threadIdx.x;
        valid[r] = i < n_rays;
        slot[r] = list_in[valid[r] ? i : base];
        o[r] = {st.ox[slot[r]], st.oy[slot[r]], st.oz[slot[r]]};
        d[r] = {st.dx[slot[r]], st.dy[slot[r]], st.dz[slot[r]]};
    }
}

// Work split of a hierarchy scan launch.  A workgroup owns BLOCK*R rays and normally walks the
// whole list of top nodes.  When a launch has few rays (deep bounces, one rank's share of a frame)
// the time of the launch is the time of its busiest wave, so the top-node list is cut into S
// segments and (ray block, segment) pairs are spread over the grid; results combine through
// atomicMin / atomicOr.  S is derived from the ray count alone, so every workgroup agrees on it.
#ifndef RT_SEG_TARGET
#define RT_SEG_TARGET 2048
#endif
constexpr uint32_t SEG_TARGET = RT_SEG_TARGET;   // workgroups a small launch is spread over
struct ScanSplit { uint32_t ray_base, first_prim, end_prim; bool active; };
// STEP: granule of a segment, in nodes of the level the scan starts at (it walks 64 at a time, one per lane; a short segment leaves lanes idle there, which is cheap)
DEV ScanSplit scan_split(uint32_t n_rays, uint32_t rays_per_block, uint32_t n_prims, uint32_t STEP) {
    ScanSplit sp{0, 0, 0, false};
    const uint32_t n_blocks = (n_rays + rays_per_block - 1) / rays_per_block;
    if (n_blocks == 0) return sp;
    const uint32_t n_tiles = (n_prims + STEP - 1) / STEP;
    uint32_t segments = SEG_TARGET / n_blocks;
    segments = segments < 1 ? 1 : (segments > n_tiles ? n_tiles : segments);
    if (segments == 0 || blockIdx.x >= n_blocks * segments) return sp;
    const uint32_t tiles_per_segment = (n_tiles + segments - 1) / segments;
    const uint32_t seg = blockIdx.x % segments;
    sp.ray_base = (blockIdx.x / segments) * rays_per_block;
    sp.first_prim = seg * tiles_per_segment * STEP;
    sp.end_prim = min(n_prims, (seg + 1) * tiles_per_segment * STEP);
    sp.active = sp.first_prim < sp.end_prim;
    return sp;
}

// The hierarchy scans read the 16-byte top-node records straight from memory, one record per lane
// (coalesced; the record arrays stay in L2).  A record the packet pre-test keeps is broadcast from
// the lane that holds it with v_readlane.  Waves are independent: no LDS staging, no barriers.
#ifndef RT_SHADOW_CHECK_STEPS
#define RT_SHADOW_CHECK_STEPS 1
#endif
constexpr uint32_t SHADOW_CHECK_STEPS = RT_SHADOW_CHECK_STEPS;          // shadow scans drain and re-check "all blocked" every so many steps of 64 top nodes
DEV float readlane_f(float v, uint32_t l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), static_cast<int>(l))); }
DEV float4 readlane4(float4 v, uint32_t l) { return make_float4(readlane_f(v.x, l), readlane_f(v.y, l), readlane_f(v.z, l), readlane_f(v.w, l)); }
// The lanes of a wave hand data to each other through LDS (queue entries, blocked bits).  The
// hardware executes a wave's LDS operations in order, but the compiler reasons per thread: without
// a fence it may keep a value one lane loaded earlier although another lane has changed it since.
// wave_sync() orders the wave's memory operations as a barrier would, at no run-time cost.
DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Work queues of a wave, in LDS.  An entry is one word: (ray of the wave) << 24 | node, where
// the ray is r*64 + lane of the lane that holds it (its slot is looked up in a per-wave table) and
// the node is a top node, a cluster or a triangle position (hence at most 2^24 triangles).
// A push adds up to PUSH_CHILDREN entries per lane.  256 entries per queue and two children per push keep the LDS of a
// workgroup that walks from the super nodes (four queues, staged rays) at 31 KB: five workgroups per CU, like the walks
// from the top nodes (384 entries, four children per push and four workgroups: 291 ms on the 1 M-triangle scene; this: 268 ms).
#ifndef RT_QCAP
#define RT_QCAP 256
#endif
#ifndef RT_PUSH_CHILDREN
#define RT_PUSH_CHILDREN 2
#endif
constexpr uint32_t PUSH_CHILDREN = RT_PUSH_CHILDREN;                  // children of a node handed down per push
constexpr uint32_t QCAP = RT_QCAP, PUSH_MAX = 64 * PUSH_CHILDREN;   // the most entries a push can add
// A walk step hands its kept children to the next queue, a quarter of the children at a time; that queue is worked off
// first when it holds more than this (at most QCAP - PUSH_MAX, the room a push may need).  Early is better than full:
// what the deeper levels find (a closer hit, a blocked ray) spares the pairs still queued above them, against lanes
// left idle by short queues.  1 M-triangle scene, box tests per ray and frame: 32: 385, 297 ms; 64: 400, 303 ms;
// 96: 416, 292 ms; 128: 429, 302 ms (and 523 tests when the queues are left to fill up).
#ifndef RT_DRAIN_ABOVE
#define RT_DRAIN_ABOVE 96
#endif
constexpr uint32_t DRAIN_ABOVE = RT_DRAIN_ABOVE;
#ifndef RT_FIRST_DRAIN_ABOVE
#define RT_FIRST_DRAIN_ABOVE (RT_QCAP < 384 ? RT_QCAP - 128 : 128)
#endif
constexpr uint32_t FIRST_DRAIN_ABOVE = RT_FIRST_DRAIN_ABOVE;   // the same for the queue of the level a walk starts at (a push adds at most 64 entries per ray of a lane)
static_assert(FIRST_DRAIN_ABOVE + 128 <= QCAP, "a push must fit the queue");
static_assert(DRAIN_ABOVE + PUSH_MAX <= QCAP, "a push must fit the queue");
#ifndef RT_STAGE_RAYS
#define RT_STAGE_RAYS(R, LEVELS) 1
#endif
// The walks ask for at least 5 waves per SIMD, i.e. at most 96 VGPRs: left alone the shadow scans take 97 to 109,
// which the allocation granule of 8 turns into 4 waves; the LDS of a workgroup (26 to 31 KB) allows 5 workgroups per CU
// as well.  The rays of a wave are staged in LDS once; walks from the super nodes used to re-derive a ray from the ray
// state for every (ray, node) pair, two rays per lane, to save that LDS (308 -> 298 ms when it went, at 4 workgroups per
// CU then).  Brute force is left alone.
#ifndef RT_SCAN_ATTR
#define RT_SCAN_ATTR __attribute__((amdgpu_waves_per_eu(5)))
#endif
#ifndef RT_DEEP_RAYS
#define RT_DEEP_RAYS 1
#endif
constexpr int DEEP_RAYS = RT_DEEP_RAYS;   // rays per lane of the walks that start at the super nodes
#ifndef RT_SHORTEN_RAYS
#define RT_SHORTEN_RAYS 1
#endif
#ifndef RT_SHADOW_DFS
#define RT_SHADOW_DFS 1      // shadow rays: each lane walks its own ray depth first (k_shadow_dfs); 0: the queued walk (k_shadow_scan)
#endif
#ifndef RT_ORDER_PASSES
#define RT_ORDER_PASSES 3
#endif
#ifndef RT_ORDER_NEAR
#define RT_ORDER_NEAR 0.125f
#endif
#ifndef RT_ORDER_MID
#define RT_ORDER_MID 0.5f
#endif
constexpr uint32_t ORDER_PASSES = RT_ORDER_PASSES;
constexpr float ORDER_NEAR = RT_ORDER_NEAR, ORDER_MID = RT_ORDER_MID;   // class limits, as fractions of the range of distances of the nodes a wave can reach
constexpr uint32_t NODE_MASK = 0xFFFFFFu;
// For every set bit b < n_bits of a lane's `bits`, append base + b*step.  Entries of one bit are contiguous, in lane order.
DEV void push_bits(uint32_t* q, uint32_t& count, uint32_t bits, uint32_t n_bits, uint32_t base, uint32_t step) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = 0; b < n_bits; ++b) {
        const bool flag = ((bits >> b) & 1u) != 0u;
        const unsigned long long mask = __ballot(flag);
        if (flag) q[count + __popcll(mask & ((1ull << lane) - 1ull))] = base + b * step;
        count += __popcll(mask);
    }
}

// The pre-test record of the triangle at position `pos` of the cluster order; it carries the scan
// index k in a spare word, so that the full-test record is only looked up for the pairs that need it.
struct TriPair { float4 f0, f1, f2, f3; };
DEV TriPair load_tri_pair(const DevScene& sc, uint32_t pos) {
    const float4* __restrict__ f = sc.trif_pos + static_cast<size_t>(pos) * 4;
    return TriPair{f[0], f[1], f[2], f[3]};
}
// two-stage pre-test and, if it keeps the ray, the full test; FLT_MAX otherwise
DEV float pretests_then_full(const DevScene& sc, const TriPair& f, V3 o, V3 d, uint32_t& n_full, uint32_t& k) {
    k = __float_as_uint(f.f2.w);
    Pre pre;
    const float o_max = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
    if (!(pretest_stage1(o, d, o_max, f.f0, f.f1, f.f2, pre) && pretest_stage2(pre, f.f3))) return FLT_MAX;
    const float4* __restrict__ rec = sc.tri + static_cast<size_t>(k) * 4;
    n_full++;
    return hit_triangle(o, d, rec[0], rec[1], rec[2], rec[3].x);
}

// Brute force over the triangle list (RT_FLAG_NO_FILTER, and lists too short for a hierarchy): every
// ray of the workgroup against every triangle.  Tiles of TILE_PRIMS records are staged in LDS and read
// back as broadcasts (every lane of a wave reads the same record), R rays per lane share each read.
// STAGED: the test is evaluated in stages (tri_front / tri_head / tri_tail) and a stage is skipped
// when no lane of the wave needs it; otherwise every pair goes through hit_triangle as written
// (RT_FLAG_PLAIN_TEST, and waves that hold a ray outside the range the staged form is proven for).
// `limit[r]`: a result counts only below it (FLT_MAX for closest-hit rays, the light's distance for
// shadow rays toward a point light).  visit(r, k, t) is called, for all lanes of the wave at once,
// when some lane may have a result for ray r; lanes without one see FLT_MAX.  `everyone`: the lanes
// whose results the caller uses; what the others get is unspecified (a stage is skipped when none
// of these lanes needs it).
#ifndef RT_PIN_READS
#define RT_PIN_READS 1
#endif
#if RT_PIN_READS
#define PIN(v) asm volatile("" : "+v"(v))
#else
#define PIN(v) (void)(v)
#endif
template <int R, bool STAGED, bool SHARED, typename Visit>
DEV void brute_force_tile(const float4* lds, uint32_t first, uint32_t count, const V3 (&o)[R], const V3 (&d)[R], bool staged_wave, Mask everyone, Visit&& visit) {
    if (STAGED && staged_wave) {
        const uint32_t lane = threadIdx.x & 63u;
        // Each of the four parts of a record is read as one 16-byte read (left alone the compiler reads the words it needs:
        // 12-byte and strided reads, which take the LDS twice as long), where its stage begins: the position in the tile
        // is made opaque before the head and the tail, so that their reads cannot be moved ahead of the branch before
        // them, and a step that ends at the front costs the LDS one read (two for rays of any origin).  (Reading the
        // front's parts a step ahead, two triangles per turn of the loop with registers of their own, was measured, with
        // the read issued at the end of the step before and at its start: -1 % either way.  The front's wait for the LDS
        // is covered by the other waves of the SIMD.)
        struct FrontParts { float4 r0, r1; };
        const auto front_parts = [&](uint32_t k) __attribute__((always_inline)) {
            FrontParts f;
            f.r0 = lds[k];
            f.r1 = SHARED ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : lds[TILE_PRIMS + k];
            return f;
        };
        const auto step = [&](uint32_t k, const FrontParts& fp) __attribute__((always_inline)) {
            uint32_t kv = k * static_cast<uint32_t>(sizeof(float4));
            const auto part = [&](uint32_t j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds + j * TILE_PRIMS) + kv); };
            const float4 r0 = fp.r0;
            float4 r1 = fp.r1;
            asm volatile("" : : "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r0.w));
            if (!SHARED) asm volatile("" : : "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w));
            float4 t0, t1, t2, t3;
            t0.w = r0.x; t1.w = r0.y; t2.w = r0.z; t3.z = r0.w;
            if (!SHARED) { t0.x = r1.x; t0.y = r1.y; t0.z = r1.z; t3.w = r1.w; }
            t1.x = t1.y = t1.z = t2.x = t2.y = t2.z = t3.x = t3.y = 0.0f;   // (not read before the head)
            TriFront front[R];
            Mask out[R], in_play = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                front[r] = SHARED ? tri_front_shared(d[r], t0, t1, t2, t3.z, out[r]) : tri_front(o[r], d[r], t0, t1, t2, out[r]);
                in_play |= everyone & ~out[r];
            }
            if (in_play == 0) return;
            PIN(kv);
            if (SHARED) { r1 = part(1); asm volatile("" : : "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w)); t0.x = r1.x; t0.y = r1.y; t0.z = r1.z; t3.w = r1.w; }
            const float4 r2 = part(2);
            asm volatile("" : : "v"(r2.x), "v"(r2.y), "v"(r2.z), "v"(r2.w));
            t1.x = r2.x; t1.y = r2.y; t1.z = r2.z; t3.y = r2.w;
            TriHead head[R];
            in_play = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) { head[r] = tri_head(o[r], d[r], front[r], t0, t1, t2, t3.w, out[r]); in_play |= everyone & ~out[r]; }
            if (in_play == 0) return;
            PIN(kv);
            const float4 r3 = part(3);
            asm volatile("" : : "v"(r3.x), "v"(r3.y), "v"(r3.z), "v"(r3.w));
            t2.x = r3.x; t2.y = r3.y; t2.z = r3.z; t3.x = r3.w;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bool unsure;
                float t = tri_tail(head[r], ((out[r] >> lane) & 1ull) != 0ull, t0, t1, t2, t3.y, unsure);
                if ((ballot(unsure) & everyone) != 0) t = hit_triangle(o[r], d[r], t0, t1, t2, t3.x);
                visit(r, first + k, t);
            }
        };
        for (uint32_t k = 0; k < count; ++k) step(k, front_parts(k));
    } else if constexpr (STAGED) {   // a wave with a ray outside the staged form's range: rare, kept small
#pragma nounroll
        for (uint32_t k = 0; k < count; ++k) {
            const TileTriangle tt = tile_triangle(lds, k);
#pragma unroll
            for (int r = 0; r < R; ++r) visit(r, first + k, hit_triangle(o[r], d[r], tt.t0, tt.t1, tt.t2, tt.t3.x));
        }
    } else {
#pragma unroll 2
        for (uint32_t k = 0; k < count; ++k) {
            const TileTriangle tt = tile_triangle(lds, k);
#pragma unroll
            for (int r = 0; r < R; ++r) visit(r, first + k, hit_triangle(o[r], d[r], tt.t0, tt.t1, tt.t2, tt.t3.x));
        }
    }
}
// The default form of the brute-force triangle scans: every (ray, triangle) pair is visited and decided either by the two-stage
// conservative pre-test (pretest_stage1 / pretest_stage2: the reference's own d.n and (a-o).n, a reciprocal instead of the
// division, and the barycentrics from precomputed w1, w2 with proven margins - the test the hierarchy's leaves use) or, for
// the pairs it keeps, by the test as written (hit_triangle).  A wave goes on to the next stage only when one of its lanes
// kept the pair, which in a scene of small triangles happens once in a few hundred steps: the common path is stage 1, 18 issue
// slots for rays with one origin and 24 for any rays, against 40 / 48 of the exact stages (tri_front / tri_head) and 96.5 of the
// test as written.  The tile in LDS holds the pre-test records (stage_pretest_tile); the full-test record of a pair that
// reaches the test as written is read from memory (one 64-byte line for the whole wave).
// Counters (per wave, uniform): steps that reached the second stage and the test as written.
// (Fetching the records of the next tile into registers while the current one is scanned was measured: no gain in either scan.)
DEV void stage_pretest_tile(float4* lds, const float4* __restrict__ trif, uint32_t first, uint32_t count, bool shared, V3 o) {
    if (threadIdx.x >= count) return;
    const float4* g = trif + static_cast<size_t>(first + threadIdx.x) * 4;
    const float4 f0 = g[0], f1 = g[1], f2 = g[2], f3 = g[3];      // (n, Ga) (a, Ha) (w1, -) (w2, -)
    if (shared) {
        const V3 ao = sub(xyz(f1), o);
        lds[threadIdx.x] = make_float4(f0.x, f0.y, f0.z, vdot(ao, xyz(f0)));
        lds[TILE_PRIMS + threadIdx.x] = make_float4(ao.x, ao.y, ao.z, f0.w);
        lds[2 * TILE_PRIMS + threadIdx.x] = make_float4(f2.x, f2.y, f2.z, f1.w);
    } else {
        lds[threadIdx.x] = f0;
        lds[TILE_PRIMS + threadIdx.x] = f1;
        lds[2 * TILE_PRIMS + threadIdx.x] = f2;
    }
    lds[3 * TILE_PRIMS + threadIdx.x] = f3;
}
struct BruteCounts { uint32_t steps = 0, stage1 = 0, stage2 = 0, exact = 0; };
template <int R, bool SHARED, typename Visit>
DEV void pretest_tile(const float4* lds, const float4* __restrict__ records, uint32_t first, uint32_t count, const V3 (&o)[R], const V3 (&d)[R],
                      const float (&o_max)[R], Mask everyone, BruteCounts& counts, Visit&& visit) {
    counts.steps += count * R;      // (the counters are in units of 64 pairs: a step of R rays per lane counts R times)
    // Every part of a record is read as one 16-byte read: left alone the compiler reads the words a stage needs with 12-byte and
    // strided reads, which occupy the LDS twice as long (the four SIMDs of a CU share it), so all four words are declared used.
#define RT_WHOLE_WORD(v) asm volatile("" : : "v"((v).x), "v"((v).y), "v"((v).z), "v"((v).w))
    for (uint32_t k = 0; k < count; ++k) {
        const float4 p0 = lds[k], p1 = lds[TILE_PRIMS + k];
        RT_WHOLE_WORD(p0); RT_WHOLE_WORD(p1);
        float4 p2;
        if (SHARED) { p2 = lds[2 * TILE_PRIMS + k]; RT_WHOLE_WORD(p2); }
        Pre pre[R];
        bool kept[R];
        Mask keep = 0;
        if (SHARED) {      // (triangles behind the eye are rare: no sign test)
            counts.stage1 += R;
#pragma unroll
            for (int r = 0; r < R; ++r) { kept[r] = pretest_stage1_shared(d[r], o_max[r], p0, p1, p2, pre[r]); keep |= ballot(kept[r]); }
        } else {
            PreFront front[R];
            bool behind[R];
            Mask ahead = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) { front[r] = pretest_front(o[r], d[r], p0, p1, behind[r]); ahead |= ballot(!behind[r]); }
            if ((ahead & everyone) == 0) continue;
            counts.stage1 += R;
            p2 = lds[2 * TILE_PRIMS + k];
            RT_WHOLE_WORD(p2);
#pragma unroll
            for (int r = 0; r < R; ++r) { kept[r] = pretest_stage1_rest(front[r], d[r], o_max[r], p0, p1, p2, pre[r]) && !behind[r]; keep |= ballot(kept[r]); }
        }
        if ((keep & everyone) == 0) continue;
        counts.stage2 += R;
        const float4 p3 = lds[3 * TILE_PRIMS + k];
        RT_WHOLE_WORD(p3);
        keep = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) { kept[r] = kept[r] && pretest_stage2(pre[r], p3); keep |= ballot(kept[r]); }
        if ((keep & everyone) == 0) continue;
        counts.exact += R;
        const float4* rec = records + static_cast<size_t>(first + k) * 4;
        const float4 t0 = rec[0], t1 = rec[1], t2 = rec[2];
        const float nn = rec[3].x;
#pragma unroll
        for (int r = 0; r < R; ++r) visit(r, first + k, hit_triangle(o[r], d[r], t0, t1, t2, nn));
    }
#undef RT_WHOLE_WORD
}
DEV void add_brute_counts(const State& st, const BruteCounts& c, int first_stat) {
    if ((threadIdx.x & 63u) != 0) return;
    if (c.steps) atomicAdd(stat_slot(st, first_stat), static_cast<unsigned long long>(c.steps));
    if (c.stage1) atomicAdd(stat_slot(st, first_stat + 1), static_cast<unsigned long long>(c.stage1));
    if (c.stage2) atomicAdd(stat_slot(st, first_stat + 2), static_cast<unsigned long long>(c.stage2));
    if (c.exact) atomicAdd(stat_slot(st, first_stat + 3), static_cast<unsigned long long>(c.exact));
}

// Reflected rays for the brute-force closest-hit scan, ordered by direction.  The staged test skips a stage when no ray of
// the wave needs it, and the rays a wave takes from the list in image order were reflected by different small triangles:
// they point anywhere.  A copy of the list in the order of 512 direction bins (3 bits per component; within a bin in the
// order the workgroups arrive, i.e. roughly image order, so origins stay close too) gives the scan waves whose rays
// agree on which triangles lie behind them.  Only the scan reads the copy: results go to the rays' slots, and every
// other kernel keeps the list in image order.  Three launches: count per bin, offsets, scatter (k_bin_*).
#ifndef RT_DIR_LEVELS
#define RT_DIR_LEVELS 8
#endif
constexpr uint32_t DIR_BINS = 512;
static_assert(RT_DIR_LEVELS * RT_DIR_LEVELS * RT_DIR_LEVELS <= DIR_BINS, "levels per component of a direction");
static_assert(2 * DIR_BINS <= SCAN_THREADS, "counts and cursors live in State::scan_block between two compactions");
DEV uint32_t direction_bin(V3 d) {
    constexpr float LEVELS = RT_DIR_LEVELS;
    const auto level = [](float v) { return static_cast<uint32_t>(fminf(fmaxf((v + 1.0f) * (0.5f * LEVELS), 0.0f), LEVELS - 1.0f)); };   // (a NaN lands in level 0)
    return level(d.x) + RT_DIR_LEVELS * (level(d.y) + RT_DIR_LEVELS * level(d.z));
}
// The same for the hit points the shadow rays start at, by the cell of an 8 x 8 x 8 grid over the scene box: neighbours in
// image order lie at any depth of a scene of small triangles, and what is behind a shadow ray depends on where it starts.
DEV uint32_t origin_bin(V3 p, const float* bb) {
    const auto level = [](float v, float lo, float hi) { return static_cast<uint32_t>(fminf(fmaxf((v - lo) / (hi - lo) * 8.0f, 0.0f), 7.0f)); };   // (a NaN lands in level 0)
    return level(p.x, bb[0], bb[3]) | (level(p.y, bb[1], bb[4]) << 3) | (level(p.z, bb[2], bb[5]) << 6);
}
// KEY 0: the rays of `bounce` by direction; KEY 1: the hit points of `bounce` (the list of bounce + 1) by cell
template <int KEY>
DEV uint32_t ray_bin(const DevScene& sc, const State& st, uint32_t slot) {
    return KEY == 0 ? direction_bin(V3{st.dx[slot], st.dy[slot], st.dz[slot]}) : origin_bin(V3{st.px[slot], st.py[slot], st.pz[slot]}, sc.bb);
}
template <int KEY>
__global__ void __launch_bounds__(BLOCK) k_bin_count(DevScene sc, State st, uint32_t bounce, uint32_t* __restrict__ counts) {
    __shared__ uint32_t lds_count[DIR_BINS];
    for (uint32_t b = threadIdx.x; b < DIR_BINS; b += BLOCK) lds_count[b] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < st.counts[bounce + KEY]) atomicAdd(&lds_count[ray_bin<KEY>(sc, st, st.list[(bounce + KEY) & 1u][i])], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < DIR_BINS; b += BLOCK) if (lds_count[b]) atomicAdd(&counts[b], lds_count[b]);
}
__global__ void __launch_bounds__(DIR_BINS) k_bin_offsets(const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors) {
    __shared__ uint32_t sums[DIR_BINS];
    const uint32_t mine = counts[threadIdx.x];
    sums[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t step = 1; step < DIR_BINS; step <<= 1) {
        const uint32_t add = threadIdx.x >= step ? sums[threadIdx.x - step] : 0u;
        __syncthreads();
        sums[threadIdx.x] += add;
        __syncthreads();
    }
    cursors[threadIdx.x] = sums[threadIdx.x] - mine;
}
template <int KEY>
__global__ void __launch_bounds__(BLOCK) k_bin_scatter(DevScene sc, State st, uint32_t bounce, uint32_t* __restrict__ cursors, uint32_t* __restrict__ out) {
    __shared__ uint32_t lds_count[DIR_BINS], lds_first[DIR_BINS];
    for (uint32_t b = threadIdx.x; b < DIR_BINS; b += BLOCK) lds_count[b] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = i < st.counts[bounce + KEY];
    uint32_t slot = 0, bin = 0, place = 0;
    if (valid) {
        slot = st.list[(bounce + KEY) & 1u][i];
        bin = ray_bin<KEY>(sc, st, slot);
        place = atomicAdd(&lds_count[bin], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < DIR_BINS; b += BLOCK) if (lds_count[b]) lds_first[b] = atomicAdd(&cursors[b], lds_count[b]);
    __syncthreads();
    if (valid) out[lds_first[bin] + place] = slot;
}

// `list`: the rays in the order the scan takes them (the list of the bounce, or its copy in the order of k_bin_scatter)
// FORM: 0 the test as written on every pair (RT_FLAG_PLAIN_TEST); 1 the test in exact stages (RT_FLAG_EXACT_STAGES);
// 2 conservative pre-tests, then the test as written for the pairs they keep (pretest_tile; the default).
template <int R, int FORM>
__global__ void __launch_bounds__(BLOCK) k_closest_tri_brute(DevScene sc, State st, uint32_t bounce, const uint32_t* __restrict__ list) {
    constexpr bool STAGED = FORM == 1;
    __shared__ float4 lds[TILE_PRIMS * 4];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    uint32_t slot[R];
    bool valid[R];
    V3 o[R], d[R];
    load_rays<R>(st, list, base, n_rays, slot, valid, o, d);
    float best_t[R]; uint32_t best_k[R];
    bool tame = true;
#pragma unroll
    for (int r = 0; r < R; ++r) { best_t[r] = FLT_MAX; best_k[r] = 0xFFFFFFFFu; tame = tame && ray_is_tame(o[r], d[r]); }
    const bool staged_wave = __all(tame);
    // rays from the eye share their origin: then (a-o).n is computed once per triangle and tile (stage_triangle_tile), not once per pair
    bool same = true;
#pragma unroll
    for (int r = 0; r < R; ++r) same = same && o[r].x == o[0].x && o[r].y == o[0].y && o[r].z == o[0].z;
    const V3 o_first{__shfl(o[0].x, 0), __shfl(o[0].y, 0), __shfl(o[0].z, 0)};
    same = same && o[0].x == o_first.x && o[0].y == o_first.y && o[0].z == o_first.z;
    __shared__ float lds_origin[3];
    if (threadIdx.x == 0) { lds_origin[0] = o[0].x; lds_origin[1] = o[0].y; lds_origin[2] = o[0].z; }
    __syncthreads();
    const V3 o_block{lds_origin[0], lds_origin[1], lds_origin[2]};
    same = same && o[0].x == o_block.x && o[0].y == o_block.y && o[0].z == o_block.z;
    const bool shared_origin = FORM != 0 && __syncthreads_and(same && (tame || FORM == 2));
    auto visit = [&](int r, uint32_t k, float t) __attribute__((always_inline)) {
        if (t < best_t[r]) { best_t[r] = t; best_k[r] = k; }
    };
    const Mask everyone = ballot(true);
    if (FORM == 2) {
        float o_max[R];
#pragma unroll
        for (int r = 0; r < R; ++r) o_max[r] = fmaxf(fabsf(o[r].x), fmaxf(fabsf(o[r].y), fabsf(o[r].z)));
        BruteCounts counts;
        for (uint32_t first = 0; first < sc.n_tri; first += TILE_PRIMS) {
            const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_tri - first);
            __syncthreads();
            stage_pretest_tile(lds, sc.trif, first, count, shared_origin, o_block);
            __syncthreads();
            if (shared_origin) pretest_tile<R, true>(lds, sc.tri, first, count, o, d, o_max, everyone, counts, visit);
            else pretest_tile<R, false>(lds, sc.tri, first, count, o, d, o_max, everyone, counts, visit);
        }
        add_brute_counts(st, counts, ST_BRUTE_STEPS);
#pragma unroll
        for (int r = 0; r < R; ++r) if (valid[r]) st.key[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit(best_t[r], best_k[r]);
        return;
    }
    for (uint32_t first = 0; first < sc.n_tri; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_tri - first);
        __syncthreads();
        const bool as_written = __syncthreads_or(stage_triangle_tile(lds, sc.tri, first, count, shared_origin, o_block));
        if (shared_origin && !as_written) {
            brute_force_tile<R, STAGED, true>(lds, first, count, o, d, staged_wave, everyone, visit);
        } else {
            brute_force_tile<R, STAGED, false>(lds, first, count, o, d, staged_wave && !as_written, everyone, visit);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) st.key[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit(best_t[r], best_k[r]);
}

// Closest hit of every live ray over the primitives of kind PRIM (0 triangles, 1 ellipsoids, 2 spheres):
// the wave walks the box hierarchy of the kind.  Top nodes are tested one per lane
// against the bound of the whole wave (packet test), then per ray; (ray, node) pairs go through
// three per-wave queues in LDS (top node -> cluster -> leaf), each walked one pair per lane, and
// the (ray, leaf) pairs that remain through the full test of the kind (for triangles after the
// two-stage pre-test).  Result: st.key[slot] / st.key_ell[slot] / st.key_sph[slot], by atomicMin.
template <int R, int LEVELS, int PRIM>
__global__ void __launch_bounds__(BLOCK) RT_SCAN_ATTR k_closest_scan(DevScene sc, State st, uint32_t bounce) {
    static_assert(LEVELS == 3 || LEVELS == 4, "a walk from the top nodes or from the super nodes");
    constexpr int N_QUEUES = LEVELS;
    static_assert(64 * R <= 256, "a queue entry has 8 bits for the ray");
    const Hierarchy& h = PRIM == 0 ? sc.th : (PRIM == 1 ? sc.eh : sc.sh);
    // A hit found shortens the ray for the rest of the walk: a node that begins beyond the closest hit so far holds nothing
    // closer (and nothing as close: the box test's `limit` is exclusive of ties by its safety factor).  The distance found
    // so far is kept per ray of the wave, in LDS, as the bits of a positive float (their order as integers is the order of the
    // floats).  Not for spheres, whose accepted distances may be negative (pack_hit_signed).
    constexpr bool SHORTENS = RT_SHORTEN_RAYS && PRIM != 2;
    constexpr bool STAGED = RT_STAGE_RAYS(R, LEVELS);
    // per wave: one queue per level ((ray, top node), (ray, cluster), (ray, leaf), with LEVELS = 4 also (ray, super node)), slot table,
    // closest distances (with staged rays they live in a spare word of the ray instead: the workgroup's LDS decides how many fit a CU)
    constexpr uint32_t WAVE_LDS = N_QUEUES * QCAP + 64 * R + (SHORTENS && !STAGED ? 64 * R : 0);
    __shared__ uint32_t lds_queue[(BLOCK / 64) * WAVE_LDS];
    // the wave's rays, staged once for the lanes that walk their pairs: (o, growth) (d, closest distance so far) (1/d, -) per ray
    __shared__ float4 lds_rays[STAGED ? (BLOCK / 64) * 64 * R * 3 : 1];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t* __restrict__ list_in = st.list[bounce & 1u];
    const ScanSplit split = scan_split(n_rays, BLOCK * R, LEVELS == 4 ? h.n_supers : h.n_tops, LEVELS == 4 ? 1u : 16u);
    if (!split.active) return;
    const uint32_t base = split.ray_base;

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* const q0 = lds_queue + wave * WAVE_LDS;      // (ray, top node)
    uint32_t* const q1 = q0 + QCAP;                        // (ray, cluster)
    uint32_t* const q2 = q1 + QCAP;                        // (ray, leaf)
    uint32_t* const qs = q2 + QCAP;                        // (ray, super node), LEVELS = 4 only
    uint32_t* const wave_slots = q0 + N_QUEUES * QCAP;     // slot of ray r*64 + lane
    uint32_t q0_count = 0, q1_count = 0, q2_count = 0, qs_count = 0, n_full = 0, n_queued = 0;
    uint32_t n_first = 0, n_batches = 0;    // per wave: nodes of the starting level tested per ray, steps of 64 (ray, node) pairs walked
    unsigned long long n_pairs = 0;
    float4* const wave_rays_lds = lds_rays + (STAGED ? wave * (64 * R * 3) : 0);
    // closest distance found so far for ray r*64 + lane (SHORTENS)
    auto best_word = [&](uint32_t ray) __attribute__((always_inline)) {
        return STAGED ? reinterpret_cast<uint32_t*>(wave_rays_lds + ray * 3 + 1) + 3 : wave_slots + 64 * R + ray;
    };
    auto best_of = [&](uint32_t ray) __attribute__((always_inline)) { return SHORTENS ? __uint_as_float(*best_word(ray)) : FLT_MAX; };
    // ray number `ray` of the wave (r*64 + lane of its owner) for the lane that handles one of its pairs
    struct PairRay { V3 o, d, inv; float grow, best; };
    auto pair_ray = [&](uint32_t ray) __attribute__((always_inline)) {
        PairRay p;
        if (STAGED) {
            const float4 a = wave_rays_lds[ray * 3], b = wave_rays_lds[ray * 3 + 1], c = wave_rays_lds[ray * 3 + 2];
            p.o = xyz(a); p.d = xyz(b); p.inv = xyz(c); p.grow = a.w; p.best = SHORTENS ? b.w : FLT_MAX;
        } else {
            const uint32_t s = wave_slots[ray];
            p.o = {st.ox[s], st.oy[s], st.oz[s]};
            p.d = {st.dx[s], st.dy[s], st.dz[s]};
            p.inv = make_box_ray(p.o, p.d, FLT_MAX).inv;
            p.grow = PRIM == 2 ? sphere_growth(p.d, sc.box_bound) : 0.0f;
            p.best = best_of(ray);
        }
        return p;
    };

    // a wave owns 64*R consecutive rays of the list, so that its packet bound is as narrow as the list allows
    uint32_t slot[R];
    bool valid[R];
    float grow[R];
    float d_len[R];         // length of the direction, rounded up (0 for a lane without a ray): distance found so far = closest t * d_len
    BoxRay box_ray[R];
    Packet packet;
    bool unsafe = false;    // an origin outside the bound the box records were built for: keep everything
    constexpr bool GROWN = PRIM == 2;   // spheres: the boxes grow with the length of the direction (sphere_growth)
    {
        V3 o[R], d[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + wave * (64 * R) + r * 64 + lane;
            valid[r] = i < n_rays;
            slot[r] = list_in[valid[r] ? i : base];
            o[r] = {st.ox[slot[r]], st.oy[slot[r]], st.oz[slot[r]]};
            d[r] = {st.dx[slot[r]], st.dy[slot[r]], st.dz[slot[r]]};
            box_ray[r] = make_box_ray(o[r], d[r], FLT_MAX);
            grow[r] = GROWN ? sphere_growth(d[r], sc.box_bound) : 0.0f;
            unsafe |= !(fmaxf(fabsf(o[r].x), fmaxf(fabsf(o[r].y), fabsf(o[r].z))) <= sc.box_bound);
            if (STAGED) {
                float4* const mine = wave_rays_lds + (r * 64 + lane) * 3;
                mine[0] = make_float4(o[r].x, o[r].y, o[r].z, grow[r]);
                mine[1] = make_float4(d[r].x, d[r].y, d[r].z, FLT_MAX);
                mine[2] = make_float4(box_ray[r].inv.x, box_ray[r].inv.y, box_ray[r].inv.z, 0.0f);
            } else if (SHORTENS) {
                *best_word(r * 64 + lane) = __float_as_uint(FLT_MAX);
            }
            d_len[r] = valid[r] ? __builtin_amdgcn_sqrtf(vdot(d[r], d[r])) * 1.00001f : 0.0f;
            d[r] = normalise(d[r]);   // for the bound of the wave: reflected directions drift off unit length
            wave_slots[r * 64 + lane] = slot[r];
        }
        packet = make_packet<R>(d, o, valid);
        if (GROWN) {   // a larger ball of origins is the same as larger boxes
            float g = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) g = fmaxf(g, valid[r] ? grow[r] : 0.0f);
            packet.rho += wave_max(g);
        }
        packet = uniform_packet(packet);
    }
    const bool keep_all = __any(unsafe);
    uint32_t wave_rays = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) wave_rays += __popcll(__ballot(valid[r]));

    // (ray, leaf) pairs: the full test of the kind (triangles: after the two-stage pre-test), one pair per lane
    auto drain2 = [&]() __attribute__((always_inline)) {
        wave_sync();
        if (PRIM == 0) {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t s = wave_slots[e >> 24];
                const PairRay pr = pair_ray(e >> 24);
                uint32_t k;
                const float t = pretests_then_full(sc, load_tri_pair(sc, e & NODE_MASK), pr.o, pr.d, n_full, k);
                if (t < FLT_MAX) {
                    atomicMin(&st.key[s], pack_hit(t, k));
                    if (SHORTENS) atomicMin(best_word(e >> 24), __float_as_uint(t));
                }
            }
        } else if (PRIM == 1) {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t s = wave_slots[e >> 24];
                const PairRay pr = pair_ray(e >> 24);
                const uint32_t k = h.k_of[e & NODE_MASK];
                const float4* __restrict__ rec = sc.ell + static_cast<size_t>(k) * 6;
                const float t = hit_ellipsoid(pr.o, pr.d, rec[0], rec[1], rec[2], rec[3], rec[4], rec[5]);
                n_full++;
                if (t < FLT_MAX) {
                    atomicMin(&st.key_ell[s], pack_hit(t, k));
                    if (SHORTENS) atomicMin(best_word(e >> 24), __float_as_uint(t));
                }
            }
        } else {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t s = wave_slots[e >> 24];
                const PairRay pr = pair_ray(e >> 24);
                const uint32_t k = h.k_of[e & NODE_MASK];
                const float t = hit_sphere(pr.o, pr.d, sc.sph[k]);
                n_full++;
                if (t < FLT_MAX) atomicMin(&st.key_sph[s], pack_hit_signed(t, k));
            }
        }
        wave_sync();
        n_queued += q2_count;
        q2_count = 0;
    };
    // (ray, node) pairs: one pair per lane, the lane walks the CLUSTER children of its node and queues
    // the ones its ray may reach one level down
    auto walk = [&](uint32_t* q_in, uint32_t& in_count, const uint4* __restrict__ nodes, uint32_t n_child,
                    uint32_t* q_out, uint32_t& out_count, auto&& drain_out) __attribute__((always_inline)) {
        wave_sync();
        for (uint32_t i0 = 0; i0 < in_count; i0 += 64) {
            const bool active = i0 + lane < in_count;
            const uint32_t e = q_in[active ? i0 + lane : 0u];
            const uint32_t first = (e & NODE_MASK) * CLUSTER;
            const PairRay pr = pair_ray(e >> 24);
            uint4 w[8];
            load_oriented_node(nodes + static_cast<size_t>(e & NODE_MASK) * 8, pr.inv, w);
            const NodeRay node_ray = make_node_ray(w[0], w[1], pr.o, pr.inv, pr.best, GROWN ? pr.grow : 0.0f);
            uint32_t keep_bits = 0;
#pragma unroll
            for (int u = 0; u < static_cast<int>(CLUSTER); ++u) {
                // no short circuit: the test is cheaper than a branch around it
                keep_bits |= (child_keeps(node_ray, w, u) | keep_all) ? 1u << u : 0u;
            }
            keep_bits = active ? keep_bits & children_present(first, n_child) : 0u;
#pragma nounroll
            for (uint32_t quarter = 0; quarter < CLUSTER / PUSH_CHILDREN; ++quarter) {     // not unrolled: one copy of the code behind drain_out
                const uint32_t bits = (keep_bits >> (PUSH_CHILDREN * quarter)) & ((1u << PUSH_CHILDREN) - 1u);
                if (__any(bits != 0u)) {
                    if (out_count > DRAIN_ABOVE) drain_out();
                    push_bits(q_out, out_count, bits, PUSH_CHILDREN, (e & ~NODE_MASK) | (first + PUSH_CHILDREN * quarter), 1u);
                }
            }
        }
        n_pairs += static_cast<unsigned long long>(in_count) * CLUSTER;
        n_batches += (in_count + 63u) / 64u;
        wave_sync();
        in_count = 0;
    };
    auto drain1 = [&]() __attribute__((always_inline)) { walk(q1, q1_count, h.cl_nodes, h.n, q2, q2_count, drain2); };
    auto drain0 = [&]() __attribute__((always_inline)) { walk(q0, q0_count, h.top_nodes, h.n_clusters, q1, q1_count, drain1); };
    auto drain_supers = [&]() __attribute__((always_inline)) { walk(qs, qs_count, h.super_nodes, h.n_tops, q0, q0_count, drain0); };
    // the level the scan starts at: its boxes are tested one per lane against the wave's bound, then per ray
    const float4* __restrict__ first_box = LEVELS == 4 ? h.super_box : h.top_box;
    uint32_t* const q_first = LEVELS == 4 ? qs : q0;
    uint32_t& first_count = LEVELS == 4 ? qs_count : q0_count;
    auto drain_first = [&]() __attribute__((always_inline)) { if constexpr (LEVELS == 4) drain_supers(); else drain0(); };

    // The nodes of the starting level are visited nearest first: in three classes of their distance from the wave's origins
    // (relative to the range of distances of what the wave can reach), each class in index order, and the queues are drained
    // after every class, so that what a class finds shortens the rays for the classes behind it.  Per wave, a node farther
    // than every ray's closest hit so far is dropped by the lane that holds it, before any per-ray test.  The classes of up
    // to 32 steps of 64 nodes are kept as 2-bit codes per lane (3 = the wave cannot reach the node); longer lists are walked
    // in one pass.  A walk from the top nodes first tests the super nodes above them, one per lane: a step none of whose
    // four super nodes the wave can reach is skipped without looking at its top nodes.
    const uint32_t n_steps = (split.end_prim - split.first_prim + 63u) / 64u;
    const bool ordered = SHORTENS && ORDER_PASSES > 1 && n_steps <= 32u;
    unsigned long long codes = ~0ull;
    if (ordered) {
        unsigned long long group_reach[2] = {~0ull, ~0ull};   // LEVELS = 3: bit s = the wave may reach super node first_prim / 16 + s
        float near_key = FLT_MAX, far_key = -FLT_MAX;          // range of distances of the nodes the wave can reach
        auto widen = [&](bool reach, float key) __attribute__((always_inline)) {
            near_key = reach ? fminf(near_key, key) : near_key;   // fminf / fmaxf drop a NaN key
            far_key = reach ? fmaxf(far_key, key) : far_key;
        };
        if constexpr (LEVELS == 3) {
            const uint32_t first_super = split.first_prim / CLUSTER, end_super = (split.end_prim + CLUSTER - 1u) / CLUSTER;
#pragma unroll
            for (uint32_t half = 0; half < 2; ++half) {
                const uint32_t super = first_super + half * 64u + lane;
                const size_t at = min(super, h.n_supers - 1u);
                const float4 sphere = box_sphere(h.super_box[at * 2], h.super_box[at * 2 + 1]);
                const bool reach = super < end_super && (packet_sphere_keeps(packet, sphere) || keep_all);
                group_reach[half] = __ballot(reach);
                widen(reach, packet_sphere_distance(packet, sphere));
            }
        } else {
            for (uint32_t c0 = split.first_prim; c0 < split.end_prim; c0 += 64) {
                const size_t mine_at = min(c0 + lane, split.end_prim - 1u);
                const float4 sphere = box_sphere(first_box[mine_at * 2], first_box[mine_at * 2 + 1]);
                widen(c0 + lane < split.end_prim && (packet_sphere_keeps(packet, sphere) || keep_all), packet_sphere_distance(packet, sphere));
            }
        }
        near_key = wave_min(near_key);
        const float per_unit = __builtin_amdgcn_rcpf(wave_max(far_key) - near_key);
        uint32_t step = 0;
        for (uint32_t c0 = split.first_prim; c0 < split.end_prim; c0 += 64, ++step) {
            if (LEVELS == 3 && ((group_reach[step >> 4] >> (4u * (step & 15u))) & 15ull) == 0ull) continue;
            const size_t mine_at = min(c0 + lane, split.end_prim - 1u);
            const float4 sphere = box_sphere(first_box[mine_at * 2], first_box[mine_at * 2 + 1]);
            const bool reach = (packet_sphere_keeps(packet, sphere) || keep_all) && c0 + lane < split.end_prim;
            const float rel = (packet_sphere_distance(packet, sphere) - near_key) * per_unit;
            const uint32_t cls = rel <= ORDER_NEAR ? 0u : (rel <= ORDER_MID ? 1u : 2u);    // NaN: the last class
            codes &= ~(3ull << (2u * step));
            codes |= static_cast<unsigned long long>(reach ? cls : 3u) << (2u * step);
        }
    }
    wave_sync();
#pragma nounroll
    for (uint32_t pass = 0; pass < (ordered ? 3u : 1u); ++pass) {
        uint32_t step = 0;
        for (uint32_t c0 = split.first_prim; c0 < split.end_prim; c0 += 64, ++step) {
            const bool in_pass = (static_cast<uint32_t>(codes >> (2u * step)) & 3u) == pass;
            if (ordered && !__any(in_pass)) continue;
            const size_t mine_at = min(c0 + lane, split.end_prim - 1u);
            const float4 mine_lo = first_box[mine_at * 2], mine_hi = first_box[mine_at * 2 + 1];
            const float4 sphere = box_sphere(mine_lo, mine_hi);
            bool reach = ordered ? in_pass : (packet_sphere_keeps(packet, sphere) || keep_all) && c0 + lane < split.end_prim;
            if (SHORTENS) {   // what the drains so far have found
                float far = 0.0f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    box_ray[r].limit = best_of(r * 64 + lane);
                    far = fmaxf(far, box_ray[r].limit * d_len[r]);
                }
                far = wave_max(far);
                reach = reach && (!(packet_sphere_distance(packet, sphere) > far) || keep_all);
            }
            unsigned long long mask = __ballot(reach);
            while (mask) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_ctzll(mask));
                mask &= mask - 1ull;
                const float4 lo = readlane4(mine_lo, bit), hi = readlane4(mine_hi, bit);
                uint32_t keep_bits = 0;
                n_pairs += wave_rays;   // the per-ray tests of this top node
                ++n_first;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    keep_bits |= ((GROWN ? box_keeps_grown(box_ray[r], lo, hi, grow[r]) : box_keeps(box_ray[r], lo, hi)) || keep_all) && valid[r] ? 1u << r : 0u;
                if (__any(keep_bits != 0u)) {
                    if (first_count > FIRST_DRAIN_ABOVE) drain_first();
                    push_bits(q_first, first_count, keep_bits, R, (lane << 24) | (c0 + bit), 64u << 24);
                }
            }
        }
        drain_first();
        if constexpr (LEVELS == 4) drain0();
        drain1();
        drain2();
    }
    if (PRIM == 0) block_add_stat(stat_slot(st, ST_FULL_TESTS), n_full);
    if (lane == 0 && n_pairs) atomicAdd(stat_slot(st, ST_PAIR_PRETESTS), n_pairs);
    if (lane == 0 && n_queued) atomicAdd(stat_slot(st, ST_QUEUED), static_cast<unsigned long long>(n_queued));
    if (lane == 0) {
        atomicAdd(stat_slot(st, ST_FIRST_NODES), static_cast<unsigned long long>(n_first));
        atomicAdd(stat_slot(st, ST_WALK_BATCHES), static_cast<unsigned long long>(n_batches));
        atomicAdd(stat_slot(st, ST_WALK_WAVES), 1ull);
    }
    {   // FLOP executed in per-ray tests (n_full is per lane; the others are wave totals)
        constexpr unsigned long long CHILD = 24, LEAF = PRIM == 0 ? 40 : (PRIM == 1 ? 95 : 19), FULL = PRIM == 0 ? 54 : 0;
        block_add_stat(stat_slot(st, ST_FLOP), FULL * n_full + (lane == 0 ? CHILD * n_pairs + LEAF * n_queued : 0ull));
    }
}

// Closest hit for rays that leave the eye (bounce 0).  The 64 rays of a wave are the 16 samples
// of four neighbouring pixels and share their origin, so the bound of the wave (make_packet) is
// almost as narrow as a single ray: the whole hierarchy is walked ONCE PER WAVE against that bound
// - top nodes one per lane, then the 16 clusters of a kept top node and the 16 leaves of a kept
// cluster on 16 lanes each - and only the leaves the wave can reach are tested per ray (box test,
// then the full test of the kind; for triangles through the two-stage pre-test).  No queues: the
// closest hit is a per-lane running minimum, ties going to the smaller scan index as in the
// brute-force scan; it is merged into st.key* by atomicMin because a small launch is split over
// segments of the top level (scan_split).  A node the wave's bound cannot reach holds no hit for
// any of its rays, so the result is that of the per-ray walk.
template <int PRIM>
__global__ void __launch_bounds__(BLOCK) k_closest_packet(DevScene sc, State st, uint32_t bounce) {
    const Hierarchy& h = PRIM == 0 ? sc.th : (PRIM == 1 ? sc.eh : sc.sh);
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t* __restrict__ list_in = st.list[bounce & 1u];
    const ScanSplit split = scan_split(n_rays, BLOCK, h.n_tops, 16u);
    if (!split.active) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t i = split.ray_base + wave * 64 + lane;
    const bool valid = i < n_rays;
    const uint32_t slot = list_in[valid ? i : split.ray_base];
    const V3 o{st.ox[slot], st.oy[slot], st.oz[slot]};
    const V3 d{st.dx[slot], st.dy[slot], st.dz[slot]};
    BoxRay ray = make_box_ray(o, d, FLT_MAX);     // its limit follows the closest hit found (not for spheres: their accepted distances may be negative)
    const float grow = PRIM == 2 ? sphere_growth(d, sc.box_bound) : 0.0f;
    const bool unsafe = !(fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) <= sc.box_bound);
    Packet packet;
    {
        const V3 dn[1] = {normalise(d)};
        const V3 on[1] = {o};
        const bool vn[1] = {valid};
        packet = make_packet<1>(dn, on, vn);
        if (PRIM == 2) packet.rho += wave_max(valid ? grow : 0.0f);
        packet = uniform_packet(packet);
    }
    const bool keep_all = __any(unsafe);
    const uint32_t wave_rays = __popcll(__ballot(valid));
    float best_t = FLT_MAX;
    uint32_t best_k = 0xFFFFFFFFu, n_full = 0;
    unsigned long long n_pairs = 0, n_queued = 0;
    const uint32_t sub = lane & 15u;   // the child a lane looks at, one level down

    for (uint32_t c0 = split.first_prim; c0 < split.end_prim; c0 += 64) {
        const size_t mine_at = min(c0 + lane, split.end_prim - 1u);
        const bool reach = packet_sphere_keeps(packet, box_sphere(h.top_box[mine_at * 2], h.top_box[mine_at * 2 + 1]));
        unsigned long long top_mask = __ballot((reach || keep_all) && c0 + lane < split.end_prim);
        while (top_mask) {
            const uint32_t top = c0 + static_cast<uint32_t>(__builtin_ctzll(top_mask));
            top_mask &= top_mask - 1ull;
            const uint32_t cluster_at = top * CLUSTER + sub;
            const size_t cl = min(cluster_at, h.n_clusters - 1u);
            const bool cl_reach = packet_sphere_keeps(packet, box_sphere(h.cl_box[cl * 2], h.cl_box[cl * 2 + 1]));
            unsigned long long cl_mask = __ballot((cl_reach || keep_all) && lane < CLUSTER && cluster_at < h.n_clusters);
            while (cl_mask) {
                const uint32_t cluster = top * CLUSTER + static_cast<uint32_t>(__builtin_ctzll(cl_mask));
                cl_mask &= cl_mask - 1ull;
                const uint32_t leaf_at = cluster * CLUSTER + sub;
                const size_t lf = min(leaf_at, h.n - 1u);
                const float4 leaf_lo = h.leaf_box[lf * 2], leaf_hi = h.leaf_box[lf * 2 + 1];
                const bool leaf_reach = packet_sphere_keeps(packet, box_sphere(leaf_lo, leaf_hi));
                unsigned long long leaf_mask = __ballot((leaf_reach || keep_all) && lane < CLUSTER && leaf_at < h.n);
                while (leaf_mask) {
                    const uint32_t bit = static_cast<uint32_t>(__builtin_ctzll(leaf_mask));
                    leaf_mask &= leaf_mask - 1ull;
                    const uint32_t pos = cluster * CLUSTER + bit;
                    // every lane: its own ray against the leaf at position pos
                    const float4 lo = readlane4(leaf_lo, bit), hi = readlane4(leaf_hi, bit);
                    n_pairs += wave_rays;
                    const bool in_box = valid && ((PRIM == 2 ? box_keeps_grown(ray, lo, hi, grow) : box_keeps(ray, lo, hi)) || keep_all);
                    n_queued += __popcll(__ballot(in_box));
                    if (in_box) {
                        float t;
                        uint32_t k;
                        if (PRIM == 0) {
                            t = pretests_then_full(sc, load_tri_pair(sc, pos), o, d, n_full, k);
                        } else if (PRIM == 1) {
                            k = h.k_of[pos];
                            const float4* __restrict__ rec = sc.ell + static_cast<size_t>(k) * 6;
                            t = hit_ellipsoid(o, d, rec[0], rec[1], rec[2], rec[3], rec[4], rec[5]);
                            n_full++;
                        } else {
                            k = h.k_of[pos];
                            t = hit_sphere(o, d, sc.sph[k]);
                            n_full++;
                        }
                        if (t < FLT_MAX && (t < best_t || (t == best_t && k < best_k))) {
                            best_t = t; best_k = k;
                            if (RT_SHORTEN_RAYS && PRIM != 2) ray.limit = t;
                        }
                    }
                }
            }
        }
    }
    if (valid && best_k != 0xFFFFFFFFu) {
        if (PRIM == 0) atomicMin(&st.key[slot], pack_hit(best_t, best_k));
        else if (PRIM == 1) atomicMin(&st.key_ell[slot], pack_hit(best_t, best_k));
        else atomicMin(&st.key_sph[slot], pack_hit_signed(best_t, best_k));
    }
    if (PRIM == 0) block_add_stat(stat_slot(st, ST_FULL_TESTS), n_full);
    if (lane == 0 && n_pairs) atomicAdd(stat_slot(st, ST_PAIR_PRETESTS), n_pairs);
    if (lane == 0 && n_queued) atomicAdd(stat_slot(st, ST_QUEUED), n_queued);
    {   // FLOP executed in per-ray tests (n_full is per lane; the others are wave totals)
        constexpr unsigned long long CHILD = 24, LEAF = PRIM == 0 ? 40 : (PRIM == 1 ? 95 : 19), FULL = PRIM == 0 ? 54 : 0;
        block_add_stat(stat_slot(st, ST_FLOP), FULL * n_full + (lane == 0 ? CHILD * n_pairs + LEAF * n_queued : 0ull));
    }
}

template <int R>
__global__ void __launch_bounds__(BLOCK) k_closest_sph(DevScene sc, State st, uint32_t bounce) {
    __shared__ float4 lds[TILE_PRIMS];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    uint32_t slot[R];
    bool valid[R];
    V3 o[R], d[R];
    load_rays<R>(st, st.list[bounce & 1u], base, n_rays, slot, valid, o, d);
    float best_t[R]; uint32_t best_k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best_t[r] = FLT_MAX; best_k[r] = 0xFFFFFFFFu; }
    for (uint32_t first = 0; first < sc.n_sph; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_sph - first);
        __syncthreads();
        stage_tile<1>(lds, sc.sph, first, count);
        __syncthreads();
        for (uint32_t k = 0; k < count; ++k) {
            const float4 s = lds[k];
            SphereFront front[R];
            Mask missed = ~0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) { front[r] = sphere_front(o[r], d[r], s); missed &= ballot(front[r].disc < 0.0f); }
            if (missed == ballot(true)) continue;      // no lane of the wave has a real root
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = sphere_roots(front[r]);
                if (t < best_t[r]) { best_t[r] = t; best_k[r] = first + k; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) st.key_sph[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit_signed(best_t[r], best_k[r]);
}

template <int R>
__global__ void __launch_bounds__(BLOCK) k_closest_ell(DevScene sc, State st, uint32_t bounce) {
    __shared__ float4 lds[TILE_PRIMS * 6];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    uint32_t slot[R];
    bool valid[R];
    V3 o[R], d[R];
    load_rays<R>(st, st.list[bounce & 1u], base, n_rays, slot, valid, o, d);
    float best_t[R]; uint32_t best_k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best_t[r] = FLT_MAX; best_k[r] = 0xFFFFFFFFu; }
    for (uint32_t first = 0; first < sc.n_ell; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_ell - first);
        __syncthreads();
        stage_tile<6>(lds, sc.ell, first, count);
        __syncthreads();
        for (uint32_t k = 0; k < count; ++k) {
            const float4 i0 = lds[k * 6], i1 = lds[k * 6 + 1], i2 = lds[k * 6 + 2];
            const float4 m0 = lds[k * 6 + 3], m1 = lds[k * 6 + 4], m2 = lds[k * 6 + 5];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = hit_ellipsoid(o[r], d[r], i0, i1, i2, m0, m1, m2);
                if (t < best_t[r]) { best_t[r] = t; best_k[r] = first + k; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) st.key_ell[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit(best_t[r], best_k[r]);
}


// True for every thread of the workgroup when all its rays start at one point (rays from the eye) and `fit` holds for all
// of them; o_block is then that point.  Ends with a barrier.
template <int R>
DEV bool workgroup_shares_origin(const V3 (&o)[R], bool fit, V3& o_block) {
    __shared__ float lds_origin[3];
    bool same = true;
#pragma unroll
    for (int r = 0; r < R; ++r) same = same && o[r].x == o[0].x && o[r].y == o[0].y && o[r].z == o[0].z;
    if (threadIdx.x == 0) { lds_origin[0] = o[0].x; lds_origin[1] = o[0].y; lds_origin[2] = o[0].z; }
    __syncthreads();
    o_block = V3{lds_origin[0], lds_origin[1], lds_origin[2]};
    same = same && o[0].x == o_block.x && o[0].y == o_block.y && o[0].z == o_block.z;
    return __syncthreads_and(same && fit) != 0;
}

// A tile of ellipsoid records for the staged scans, one ellipsoid per thread: four arrays of TILE_PRIMS float4,
//   [row] (row of the inverse transform . xyz, X[row]),  row = 0, 1, 2
// with X the translation column, or - when all rays of the workgroup start at `o` (`shared`: rays from the eye) - that
// origin in ellipsoid space, M^-1 o, computed here with the expression of the test; and
//   [3]   (O.O, ell_margin_term(O.O), -, -) for a shared origin; the light's position in ellipsoid space for shadow rays
//         toward a point light (`light_row`: that light's row of DevScene::ell_light, else null).
// A step that ends at stage 0 (ell_certain_miss) reads three (shadow rays, reflected rays) or four (rays from the eye)
// 16-byte words; the rows of the transform itself, which only the tail needs, stay in memory.
// Returns true if this thread's record, or the light's position in its space, is outside the range the staged form is proven for.
DEV bool stage_ellipsoid_tile(float4* lds, const DevScene& sc, uint32_t first, uint32_t count, bool shared, V3 o, const float4* __restrict__ light_row) {
    if (threadIdx.x >= count) return false;
    const uint32_t k = first + threadIdx.x;
    const float4* g = sc.ell + static_cast<size_t>(k) * 6;
    float4 i0 = g[0], i1 = g[1], i2 = g[2];
    bool irregular = sc.ell_ok[k] == 0.0f;
    float4 extra = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (shared) {
        const V3 O = vmat_point(i0, i1, i2, o);
        const float oo = vdot(O, O);
        i0.w = O.x; i1.w = O.y; i2.w = O.z;
        extra = make_float4(oo, ell_margin_term(oo), 0.0f, 0.0f);
    } else if (light_row) {
        extra = light_row[k];
        irregular = irregular || !(fmaxf(fabsf(extra.x), fmaxf(fabsf(extra.y), fabsf(extra.z))) <= ELL_LIGHT_COORD);
    }
    lds[threadIdx.x] = i0;
    lds[TILE_PRIMS + threadIdx.x] = i1;
    lds[2 * TILE_PRIMS + threadIdx.x] = i2;
    lds[3 * TILE_PRIMS + threadIdx.x] = extra;
    return irregular;
}

// Closest hit over one staged tile, R rays per lane.  Lanes past the end of the list repeat a ray of the list (load_rays),
// so every lane's vote counts.
template <int R, bool SHARED>
DEV void ellipsoid_tile_closest(const float4* lds, const float4* __restrict__ records, uint32_t first, uint32_t count,
                                const V3 (&o)[R], const V3 (&d)[R], float (&best_t)[R], uint32_t (&best_k)[R]) {
    for (uint32_t k = 0; k < count; ++k) {
        const float4 i0 = lds[k], i1 = lds[TILE_PRIMS + k], i2 = lds[2 * TILE_PRIMS + k];
        EllParts e[R];
        Mask open = 0;
        if (SHARED) {
            const float2 extra = *reinterpret_cast<const float2*>(lds + 3 * TILE_PRIMS + k);    // (O.O, margin term)
#pragma unroll
            for (int r = 0; r < R; ++r) { e[r] = ell_parts_shared(i0, i1, i2, extra.x, d[r]); open |= ballot(!ell_certain_miss(e[r], extra.y)); }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) { e[r] = ell_parts(i0, i1, i2, o[r], d[r]); open |= ballot(!ell_certain_miss(e[r], ell_margin_term(e[r].oo))); }
        }
        if (open == 0) continue;
        EllRoot root[R];
        Mask in_play = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) { root[r] = ell_root(e[r]); in_play |= ballot(!ell_root_is_out(root[r])); }
        if (in_play == 0) continue;
        const float4* m = records + static_cast<size_t>(first + k) * 6 + 3;
        const float4 m0 = m[0], m1 = m[1], m2 = m[2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t = ell_world_distance(o[r], e[r], root[r], m0, m1, m2);
            if (t < best_t[r]) { best_t[r] = t; best_k[r] = first + k; }
        }
    }
}

// Brute force over the ellipsoid list (RT_FLAG_NO_FILTER, and lists too short for a hierarchy) with the test in stages
// (ell_parts ...); k_closest_ell above is the same scan with the test as written (RT_FLAG_PLAIN_TEST).
template <int R>
__global__ void __launch_bounds__(BLOCK) k_closest_ell_staged(DevScene sc, State st, uint32_t bounce) {
    __shared__ float4 lds[TILE_PRIMS * 4];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    uint32_t slot[R];
    bool valid[R];
    V3 o[R], d[R];
    load_rays<R>(st, st.list[bounce & 1u], base, n_rays, slot, valid, o, d);
    float best_t[R]; uint32_t best_k[R];
    bool tame = true;
#pragma unroll
    for (int r = 0; r < R; ++r) { best_t[r] = FLT_MAX; best_k[r] = 0xFFFFFFFFu; tame = tame && ray_is_tame_for_ellipsoids(o[r], d[r]); }
    const bool staged_wave = __all(tame);
    V3 o_block;
    const bool shared_origin = workgroup_shares_origin<R>(o, tame, o_block);
    for (uint32_t first = 0; first < sc.n_ell; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_ell - first);
        __syncthreads();
        const bool as_written = __syncthreads_or(stage_ellipsoid_tile(lds, sc, first, count, shared_origin, o_block, nullptr));
        if (as_written || !staged_wave) {      // rare: the records straight from memory
            for (uint32_t k = 0; k < count; ++k) {
                const float4* rec = sc.ell + static_cast<size_t>(first + k) * 6;
                const float4 i0 = rec[0], i1 = rec[1], i2 = rec[2], m0 = rec[3], m1 = rec[4], m2 = rec[5];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float t = hit_ellipsoid(o[r], d[r], i0, i1, i2, m0, m1, m2);
                    if (t < best_t[r]) { best_t[r] = t; best_k[r] = first + k; }
                }
            }
        } else if (shared_origin) {
            ellipsoid_tile_closest<R, true>(lds, sc.ell, first, count, o, d, best_t, best_k);
        } else {
            ellipsoid_tile_closest<R, false>(lds, sc.ell, first, count, o, d, best_t, best_k);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) st.key_ell[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit(best_t[r], best_k[r]);
}

// Brute force over the sphere list, like k_closest_sph, for workgroups whose rays start at one point (rays from the eye):
// centre - origin and its square are then properties of the sphere, computed once per tile with the expressions of
// sphere_front, and a pair costs one dot product and three more operations.
template <int R>
__global__ void __launch_bounds__(BLOCK) k_closest_sph_eye(DevScene sc, State st, uint32_t bounce) {
    __shared__ float4 lds[TILE_PRIMS];
    __shared__ float lds_r2[TILE_PRIMS];
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    uint32_t slot[R];
    bool valid[R];
    V3 o[R], d[R];
    load_rays<R>(st, st.list[bounce & 1u], base, n_rays, slot, valid, o, d);
    float best_t[R]; uint32_t best_k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best_t[r] = FLT_MAX; best_k[r] = 0xFFFFFFFFu; }
    V3 o_block;
    const bool shared_origin = workgroup_shares_origin<R>(o, true, o_block);
    for (uint32_t first = 0; first < sc.n_sph; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_sph - first);
        __syncthreads();
        if (threadIdx.x < count) {
            const float4 s = sc.sph[first + threadIdx.x];
            if (shared_origin) {
                const V3 oc = sub(xyz(s), o_block);
                lds[threadIdx.x] = make_float4(oc.x, oc.y, oc.z, vdot(oc, oc));
                lds_r2[threadIdx.x] = s.w * s.w;
            } else {
                lds[threadIdx.x] = s;
            }
        }
        __syncthreads();
        for (uint32_t k = 0; k < count; ++k) {
            const float4 s = lds[k];
            SphereFront front[R];
            Mask missed = ~0ull;
            if (shared_origin) {
                const float r2 = lds_r2[k];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    front[r].mid = vdot(xyz(s), d[r]);
                    const float mid2 = front[r].mid * front[r].mid;
                    const float perp2 = s.w - mid2;
                    front[r].disc = r2 - perp2;
                    missed &= ballot(front[r].disc < 0.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) { front[r] = sphere_front(o[r], d[r], s); missed &= ballot(front[r].disc < 0.0f); }
            }
            if (missed == ballot(true)) continue;      // no lane of the wave has a real root
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = sphere_roots(front[r]);
                if (t < best_t[r]) { best_t[r] = t; best_k[r] = first + k; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (valid[r]) st.key_sph[slot[r]] = best_k[r] == 0xFFFFFFFFu ? NO_HIT_KEY : pack_hit_signed(best_t[r], best_k[r]);
}

// k_hit: one ray per thread.  Kind choice (ray_tracing.cpp:610-680), hit point, normal,
// ambient + emission (:682), the reflected ray, and the flag "this ray goes on" for the list compaction.
__global__ void __launch_bounds__(BLOCK) k_hit(DevScene sc, State st, uint32_t bounce) {
    const uint32_t n_rays = st.counts[bounce];
    const uint32_t base = blockIdx.x * BLOCK;
    if (base >= n_rays) return;
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < n_rays;
    const uint32_t s = st.list[bounce & 1u][valid ? i : base];

    float t_kind[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    uint32_t k_kind[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (sc.n_tri) {
        const unsigned long long key = st.key[s];
        k_kind[0] = static_cast<uint32_t>(key & 0xFFFFFFFFull);
        if (k_kind[0] != 0xFFFFFFFFu) t_kind[0] = __uint_as_float(static_cast<uint32_t>(key >> 32));
    }
    if (sc.n_sph) {
        const unsigned long long key = st.key_sph[s];
        k_kind[1] = static_cast<uint32_t>(key & 0xFFFFFFFFull);
        if (k_kind[1] != 0xFFFFFFFFu) t_kind[1] = unpack_distance_signed(key);
    }
    if (sc.n_ell) {
        const unsigned long long key = st.key_ell[s];
        k_kind[2] = static_cast<uint32_t>(key & 0xFFFFFFFFull);
        if (k_kind[2] != 0xFFFFFFFFu) t_kind[2] = __uint_as_float(static_cast<uint32_t>(key >> 32));
    }
    const float tri_t = t_kind[0], sph_t = t_kind[1], ell_t = t_kind[2];
    const bool any = (k_kind[0] != 0xFFFFFFFFu) | (k_kind[1] != 0xFFFFFFFFu) | (k_kind[2] != 0xFFFFFFFFu);
    // ray_tracing.cpp:616, :637, :655
    int kind = -1;
    if (any) {
        if (tri_t - sph_t <= TOL && tri_t - ell_t <= TOL) kind = 0;
        else if (sph_t - tri_t <= TOL && sph_t - ell_t <= TOL) kind = 1;
        else if (ell_t - tri_t <= TOL && ell_t - sph_t <= TOL) kind = 2;
    }
    const bool hit = valid && kind >= 0;
    if (hit) {
        const V3 o{st.ox[s], st.oy[s], st.oz[s]};
        const V3 d{st.dx[s], st.dy[s], st.dz[s]};
        V3 p, n;
        uint32_t index;
        const float* mat;
        uint32_t mat_count;
        if (kind == 0) {
            const uint32_t k = k_kind[0];
            index = 8u * (k % sc.tri_nb) + k / sc.tri_nb;
            p = add(o, scale(tri_t, d));
            const float4 t0 = sc.tri[static_cast<size_t>(k) * 4], t1 = sc.tri[static_cast<size_t>(k) * 4 + 1], t2 = sc.tri[static_cast<size_t>(k) * 4 + 2];
            const V3 a = xyz(t0);
            const V3 vb = add(xyz(t1), a), vc = add(xyz(t2), a);      // ray_tracing.cpp:631-632
            n = normalise(scross(sub(vb, a), sub(vc, a)));            // maths.cpp:304-306
            mat = sc.tri_mat; mat_count = sc.n_tri_mat;
        } else if (kind == 1) {
            const uint32_t k = k_kind[1];
            index = 8u * (k % sc.sph_nb) + k / sc.sph_nb;
            p = add(o, scale(sph_t, d));
            n = normalise(sub(p, xyz(sc.sph[k])));                    // maths.cpp:308-310
            mat = sc.sph_mat; mat_count = sc.n_sph_mat;
        } else {
            const uint32_t k = k_kind[2];
            index = 8u * (k % sc.ell_nb) + k / sc.ell_nb;
            p = add(o, scale(ell_t, d));
            const float4 i0 = sc.ell[static_cast<size_t>(k) * 6], i1 = sc.ell[static_cast<size_t>(k) * 6 + 1], i2 = sc.ell[static_cast<size_t>(k) * 6 + 2];
            const V3 pe = smat_point(i0, i1, i2, p);                  // maths.cpp:320-326
            const V3 ne = normalise(pe);
            n = normalise(smat_transpose_dir(i0, i1, i2, ne));
            mat = sc.ell_mat; mat_count = sc.n_ell_mat;
        }
        // A hit on a zero-padding lane has no material; the reference reads past its vector there.
        float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
        if (index < mat_count) { e0 = mat[index * 10 + 6]; e1 = mat[index * 10 + 7]; e2 = mat[index * 10 + 8]; }
        st.cr[s] += st.wr[s] * (sc.ambient[0] + e0);                   // ray_tracing.cpp:682
        st.cg[s] += st.wg[s] * (sc.ambient[1] + e1);
        st.cb[s] += st.wb[s] * (sc.ambient[2] + e2);
        st.px[s] = p.x; st.py[s] = p.y; st.pz[s] = p.z;
        st.nx[s] = n.x; st.ny[s] = n.y; st.nz[s] = n.z;
        // the reflected ray (ray_tracing.cpp:685, :728; the direction is not renormalised).  It does not depend on the
        // shadow rays, so the next bounce's closest-hit scan can run beside this bounce's shadow scans.
        const V3 above = add(p, scale(2.0f * TOL, n));
        st.nox[s] = above.x; st.noy[s] = above.y; st.noz[s] = above.z;
        const V3 refl = sub(d, scale(2.0f * sdot(d, n), n));
        st.ndx[s] = refl.x; st.ndy[s] = refl.y; st.ndz[s] = refl.z;
        const uint32_t id = RT_ID_MAKE(static_cast<uint32_t>(kind), index);
        st.hit_id[s] = id;
        st.blocked[s] = 0u;
        st.key[s] = NO_HIT_KEY;      // for the next bounce's triangle scan
        st.key_ell[s] = NO_HIT_KEY;
        st.key_sph[s] = NO_HIT_KEY;
        if (bounce == 0 && st.primary_id) st.primary_id[s] = id;
    }
    const unsigned long long hit_mask = __ballot(hit);      // the next list is built from these by k_scan_flags + k_fill_list
    if ((threadIdx.x & 63u) == 0) st.flag_bits[blockIdx.x * 4u + (threadIdx.x >> 6)] = hit_mask;
}

// ------------------------------------------------------------------------------------------
// shadow scans: any-hit of every hit point toward light (first_light + blockIdx.y), per kind
// ------------------------------------------------------------------------------------------
// Shadow rays of one launch: origin above the surface (ray_tracing.cpp:685-686), direction and
// limit per light kind (:350-354, :410-412).  A ray some earlier kind already blocked starts blocked.
template <int R>
struct ShadowRays {
    uint32_t slot[R];
    bool valid[R], blocked[R];
    V3 o[R], d[R];
    float limit[R];
};
template <int R>
DEV void load_shadow_rays(const State& st, const uint32_t* __restrict__ list_in, uint32_t base, uint32_t n_rays,
                          bool point, V3 lv, uint32_t light_bit, ShadowRays<R>& s) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = base + r * BLOCK + threadIdx.x;
        s.valid[r] = i < n_rays;
        s.slot[r] = list_in[s.valid[r] ? i : base];
        const V3 p{st.px[s.slot[r]], st.py[s.slot[r]], st.pz[s.slot[r]]};
        const V3 n{st.nx[s.slot[r]], st.ny[s.slot[r]], st.nz[s.slot[r]]};
        s.o[r] = add(p, scale(2.0f * TOL, n));
        if (point) {
            const V3 to_light = sub(lv, s.o[r]);
            s.limit[r] = sqrt_(sdot(to_light, to_light));
            const float inv = 1.0f / s.limit[r];
            s.d[r] = scale(inv, to_light);
        } else {
            s.d[r] = scale(-1.0f, lv);
            s.limit[r] = FLT_MAX;
        }
        s.blocked[r] = (st.blocked[s.slot[r]] & light_bit) != 0u;
    }
}
template <int R>
DEV bool shadow_wave_done(const ShadowRays<R>& s) {
    bool all = true;
#pragma unroll
    for (int r = 0; r < R; ++r) all = all && (s.blocked[r] || !s.valid[r]);
    return __all(all) != 0;
}
// tests spent on still-undecided rays of this wave, for `count` primitives (wave-uniform)
template <int R>
DEV unsigned long long shadow_live_tests(const ShadowRays<R>& s, uint32_t count) {
    unsigned long long n = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) n += static_cast<unsigned long long>(count) * __popcll(__ballot(s.valid[r] && !s.blocked[r]));
    return n;
}
template <int R>
DEV void store_shadow_result(const State& st, const ShadowRays<R>& s, uint32_t light_bit, unsigned long long* stat, unsigned long long wave_tests) {
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (s.valid[r] && s.blocked[r]) atomicOr(&st.blocked[s.slot[r]], light_bit);
    if ((threadIdx.x & 63u) == 0 && wave_tests) atomicAdd(stat, wave_tests);
}

// One shadow ray: origin above the surface (ray_tracing.cpp:685-686), direction and limit per
// light kind (:350-354 point, :410-412 directional).  Deterministic, so it can be rebuilt on demand.
struct ShadowRay { V3 o, d; float limit; };
DEV ShadowRay make_shadow_ray(const State& st, uint32_t slot, bool point, V3 lv) {
    ShadowRay s;
    const V3 p{st.px[slot], st.py[slot], st.pz[slot]};
    const V3 n{st.nx[slot], st.ny[slot], st.nz[slot]};
    s.o = add(p, scale(2.0f * TOL, n));
    if (point) {
        const V3 to_light = sub(lv, s.o);
        s.limit = sqrt_(sdot(to_light, to_light));
        const float inv = 1.0f / s.limit;
        s.d = scale(inv, to_light);
    } else {
        s.d = scale(-1.0f, lv);
        s.limit = FLT_MAX;
    }
    return s;
}

// Brute-force any-hit over the triangle list toward light first_light + blockIdx.y (see k_closest_tri_brute), one ray per lane.
// A blocked ray needs no further tests, and in a dense scene most rays are blocked somewhere along the list: a lane
// whose ray is blocked, or has seen the whole list, takes another ray.  "Blocked" is an OR over the triangles, so a
// ray may start anywhere in the list: the workgroup stages the tiles of the list round and round, a ray that joins at
// tile s is tested against tiles s, s+1, ... up to s-1, and every lane counts the tiles its ray has still to see.
// A workgroup owns a contiguous range of the ray list, SHADOW_RANGE_BLOCKS x BLOCK rays (fewer when the launch would
// otherwise have fewer than SHADOW_MIN_RANGES ranges over all lights): every SHADOW_COMPACT_TILES tiles it retires the
// rays that are done, moves the open ones to the first lanes (ray data passed through LDS, order kept) and fills the
// lanes behind them with the next rays of its range.  Waves left without rays at the end of the range only take part
// in the staging and the barriers, beside the other workgroups of their CU.  Neighbours in the list are neighbours in
// the image, so the rays of a wave agree on most triangles and whole stages of the test are skipped; handing out rays
// from one cursor per launch instead keeps every lane busy to the end but mixes rays from all over the image in a wave
// (measured: 0.51 of the FP32 peak against 0.57 with ranges; ranges of 2, 8 or 16 capacities handed out by a cursor, so that a workgroup
// goes on to another range instead of trailing off with the long-lived rays of its own: 4426 / 4269 / 4209 ms per launch against 4177).
#ifndef RT_SHADOW_COMPACT_TILES
#define RT_SHADOW_COMPACT_TILES 4
#endif
#ifndef RT_SHADOW_RANGE_BLOCKS
#define RT_SHADOW_RANGE_BLOCKS 16
#endif
#ifndef RT_SHADOW_MIN_RANGES
#define RT_SHADOW_MIN_RANGES 6144
#endif
// Eight waves per SIMD (64 VGPRs; left alone the kernel takes 80, i.e. six): 58.1 -> 59.8 % on the synthetic crop, seven: 59.5 %.
#ifndef RT_SHADOW_BRUTE_ATTR
#define RT_SHADOW_BRUTE_ATTR __attribute__((amdgpu_waves_per_eu(8)))
#endif
constexpr uint32_t SHADOW_COMPACT_TILES = RT_SHADOW_COMPACT_TILES, SHADOW_RANGE_BLOCKS = RT_SHADOW_RANGE_BLOCKS, SHADOW_MIN_RANGES = RT_SHADOW_MIN_RANGES;
template <int FORM>
__global__ void __launch_bounds__(BLOCK) RT_SHADOW_BRUTE_ATTR k_shadow_tri_brute(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first, const uint32_t* __restrict__ list_in) {
    __shared__ float4 lds[TILE_PRIMS * 4];
    // Rays change lanes through the tile's memory, between two tiles: (o, limit) (d, tiles left) and the slot of the
    // rays the workgroup holds, by lane.  (Memory of their own would leave six workgroups per CU instead of nine.)
    constexpr bool STAGED = FORM == 1;
    float4 (*parked)[BLOCK] = reinterpret_cast<float4 (*)[BLOCK]>(lds);
    uint32_t* parked_slot = reinterpret_cast<uint32_t*>(lds + 2 * BLOCK);
    static_assert(TILE_PRIMS * 4 >= 2 * BLOCK + BLOCK / 4, "the parked rays fit the tile");
    __shared__ uint32_t wave_open[BLOCK / 64];
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t n_blocks = (n_rays + BLOCK - 1) / BLOCK;
    const uint32_t blocks_per_wg = min(SHADOW_RANGE_BLOCKS, max(1u, n_blocks * gridDim.y / SHADOW_MIN_RANGES));
    if (static_cast<unsigned long long>(blockIdx.x) * blocks_per_wg >= n_blocks) return;      // (also: no rays)
    uint32_t range_next = blockIdx.x * blocks_per_wg * BLOCK;                                 // the next ray of the range
    const uint32_t range_end = static_cast<uint32_t>(min(static_cast<unsigned long long>(n_rays), static_cast<unsigned long long>(range_next) + static_cast<unsigned long long>(blocks_per_wg) * BLOCK));
    const float* __restrict__ L = sc.lights + (first_light + blockIdx.y) * 8;
    const bool point = L[0] != 0.0f;
    const V3 lv{L[1], L[2], L[3]};
    const uint32_t light_bit = 1u << (first_light + blockIdx.y - group_first);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_tiles = (sc.n_tri + TILE_PRIMS - 1) / TILE_PRIMS;
    const uint32_t interval = min(SHADOW_COMPACT_TILES, n_tiles);
    // the ray this lane is testing: open while `active`, not blocked and with tiles left to see
    uint32_t slot = 0, left = 0;
    bool active = false, blocked = false;
    V3 o[1] = {V3{0.0f, 0.0f, 0.0f}}, d[1] = {V3{0.0f, 0.0f, 0.0f}};
    float limit = 0.0f;
    bool staged_wave = false;
    uint32_t held = 0;                         // lanes [0, held) hold a ray
    uint32_t tile = 0;                         // the tile of the list that is staged next
    unsigned long long wave_tests = 0;
    for (;;) {
        // retire what is done, pack the open rays in order, take new ones
        if (active && blocked) atomicOr(&st.blocked[slot], light_bit);
        const bool open = active && !blocked && left != 0u;
        const Mask open_mask = ballot(open);
        if (lane == 0) wave_open[wave] = static_cast<uint32_t>(__popcll(open_mask));
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64; ++w) { const uint32_t c = wave_open[w]; before += w < wave ? c : 0u; total += c; }
        const uint32_t room = static_cast<uint32_t>(BLOCK) - total;
        const uint32_t next = range_next;
        const uint32_t take = min(room, range_end - range_next);
        range_next += take;
        if (total + take == 0) break;                                  // uniform over the workgroup: every ray of the range is done
        if (total != held || take != 0) {                              // uniform
            if (open) {
                const uint32_t to = before + static_cast<uint32_t>(__popcll(open_mask & ((1ull << lane) - 1ull)));
                parked[0][to] = make_float4(o[0].x, o[0].y, o[0].z, limit);
                parked[1][to] = make_float4(d[0].x, d[0].y, d[0].z, __uint_as_float(left));
                parked_slot[to] = slot;
            }
            if (threadIdx.x >= total && threadIdx.x < total + take) {
                const uint32_t s = list_in[next + (threadIdx.x - total)];
                const ShadowRay ray = make_shadow_ray(st, s, point, lv);
                const bool blocked_before = (st.blocked[s] & light_bit) != 0u;         // (by a scan of another kind of primitive)
                parked[0][threadIdx.x] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.limit);
                parked[1][threadIdx.x] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(blocked_before ? 0u : n_tiles));
                parked_slot[threadIdx.x] = s;
            }
            __syncthreads();
            held = total + take;
            // lanes past the rays of a partly filled wave repeat that wave's first ray (their results are dropped), so
            // that they meet no operands of their own; waves without any ray skip the tests
            const uint32_t wave_first = wave * 64u;
            active = threadIdx.x < held;
            const uint32_t from = active ? threadIdx.x : (wave_first < held ? wave_first : 0u);
            const float4 p0 = parked[0][from], p1 = parked[1][from];
            o[0] = xyz(p0); d[0] = xyz(p1); limit = p0.w; left = __float_as_uint(p1.w); slot = parked_slot[from];
            blocked = false;
            staged_wave = __all(ray_is_tame(o[0], d[0]));
        }
        for (uint32_t step = 0; step < interval; ++step) {
            const uint32_t first = tile * TILE_PRIMS;
            const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_tri - first);
            __syncthreads();
            const bool as_written = __syncthreads_or(stage_triangle_tile(lds, sc.tri, first, count, false, V3{0.0f, 0.0f, 0.0f}));
            const Mask testing = ballot(active && !blocked && left != 0u);
            if (testing != 0) {
                wave_tests += static_cast<unsigned long long>(count) * static_cast<uint32_t>(__popcll(testing));
                // Only the lanes in `testing` have a specified result: a lane whose ray has been once round the list (left == 0) waits
                // for the next refill with an open ray, and a pair the staged form is unsure of comes back as a hit unless a lane
                // that counts is unsure too (brute_force_tile) - it must not block that ray.
                const bool counted = active && left != 0u;
                brute_force_tile<1, STAGED, false>(lds, first, count, o, d, staged_wave && !as_written, testing, [&](int, uint32_t, float t) __attribute__((always_inline)) {
                    blocked |= (t < limit) & counted;
                });
            }
            left -= left != 0u ? 1u : 0u;
            tile = tile + 1 == n_tiles ? 0u : tile + 1;
        }
    }
    if (lane == 0 && wave_tests) atomicAdd(stat_slot(st, ST_SHADOW_TESTS_TRI), wave_tests);
}

// The same scan in the default form of the brute-force triangle scans (pretest_tile), R rays per lane: the pre-test reads
// three 16-byte words of LDS per step for 24 issue slots of arithmetic, so with one ray per lane the LDS - shared by the four
// SIMDs of a CU - limits the scan as much as the arithmetic does; two rays per lane share each read.  The workgroup holds
// BLOCK x R rays, ray i in lane i % BLOCK as its ray number i / BLOCK; every SHADOW_COMPACT_TILES tiles the open ones move to
// the front (in order) and the room behind them is filled from the workgroup's range of the list.  Only a ray's slot and the
// number of tiles it has still to see change lanes (8 bytes per ray, through the tile's memory between two tiles); the ray
// itself is rebuilt from its hit point (make_shadow_ray: the same expressions give the same bits).
template <int R>
__global__ void __launch_bounds__(BLOCK) RT_SHADOW_BRUTE_ATTR k_shadow_tri_pretest(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first, const uint32_t* __restrict__ list_in) {
    __shared__ float4 lds[TILE_PRIMS * 4];
    constexpr uint32_t CAPACITY = BLOCK * R, WAVES = BLOCK / 64;
    uint2* parked = reinterpret_cast<uint2*>(lds);        // (slot, tiles left) of the rays the workgroup holds, between two tiles
    static_assert(sizeof(uint2) * CAPACITY <= sizeof(float4) * TILE_PRIMS * 4, "the parked rays fit the tile");
    __shared__ uint32_t wave_open[R * WAVES];
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t n_blocks = (n_rays + BLOCK - 1) / BLOCK;
    const uint32_t blocks_per_wg = min(SHADOW_RANGE_BLOCKS * R, max(1u, n_blocks * gridDim.y / SHADOW_MIN_RANGES));
    if (static_cast<unsigned long long>(blockIdx.x) * blocks_per_wg >= n_blocks) return;      // (also: no rays)
    uint32_t range_next = blockIdx.x * blocks_per_wg * BLOCK;                                 // the next ray of the range
    const uint32_t range_end = static_cast<uint32_t>(min(static_cast<unsigned long long>(n_rays), static_cast<unsigned long long>(range_next) + static_cast<unsigned long long>(blocks_per_wg) * BLOCK));
    const float* __restrict__ L = sc.lights + (first_light + blockIdx.y) * 8;
    const bool point = L[0] != 0.0f;
    const V3 lv{L[1], L[2], L[3]};
    const uint32_t light_bit = 1u << (first_light + blockIdx.y - group_first);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_tiles = (sc.n_tri + TILE_PRIMS - 1) / TILE_PRIMS;
    const uint32_t interval = min(SHADOW_COMPACT_TILES, n_tiles);
    uint32_t slot[R], left[R];
    bool active[R], blocked[R];
    V3 o[R], d[R];
    float limit[R], o_max[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { slot[r] = 0; left[r] = 0; active[r] = false; blocked[r] = false; o[r] = d[r] = V3{0.0f, 0.0f, 0.0f}; limit[r] = 0.0f; o_max[r] = 0.0f; }
    uint32_t held = 0;                         // rays 0 .. held - 1 are held
    uint32_t tile = 0;                         // the tile of the list that is staged next
    unsigned long long wave_tests = 0;
    BruteCounts counts;
    for (;;) {
        // retire what is done, pack the open rays in order, take new ones
        bool open[R];
        Mask open_mask[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (active[r] && blocked[r]) atomicOr(&st.blocked[slot[r]], light_bit);
            open[r] = active[r] && !blocked[r] && left[r] != 0u;
            open_mask[r] = ballot(open[r]);
            if (lane == 0) wave_open[r * WAVES + wave] = static_cast<uint32_t>(__popcll(open_mask[r]));
        }
        __syncthreads();
        uint32_t before[R], total = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (uint32_t w = 0; w < WAVES; ++w) {
                if (w == wave) before[r] = total;
                total += wave_open[r * WAVES + w];
            }
        }
        const uint32_t room = CAPACITY - total;
        const uint32_t next = range_next;
        const uint32_t take = min(room, range_end - range_next);
        range_next += take;
        if (total + take == 0) break;                                  // uniform over the workgroup: every ray of the range is done
        if (total != held || take != 0) {                              // uniform
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (open[r]) parked[before[r] + static_cast<uint32_t>(__popcll(open_mask[r] & ((1ull << lane) - 1ull)))] = make_uint2(slot[r], left[r]);
            for (uint32_t j = threadIdx.x; j < take; j += BLOCK) {
                const uint32_t s = list_in[next + j];
                const bool blocked_before = (st.blocked[s] & light_bit) != 0u;         // (by a scan of another kind of primitive)
                parked[total + j] = make_uint2(s, blocked_before ? 0u : n_tiles);
            }
            __syncthreads();
            held = total + take;
            // a lane without a ray of some number repeats a ray the workgroup holds (its results are dropped), so that it meets no
            // operands of its own
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t i = static_cast<uint32_t>(r) * BLOCK + threadIdx.x, wave_first = static_cast<uint32_t>(r) * BLOCK + wave * 64u;
                active[r] = i < held;
                const uint2 mine = parked[active[r] ? i : (wave_first < held ? wave_first : 0u)];
                slot[r] = mine.x; left[r] = mine.y;
                const ShadowRay ray = make_shadow_ray(st, slot[r], point, lv);
                o[r] = ray.o; d[r] = ray.d; limit[r] = ray.limit;
                o_max[r] = fmaxf(fabsf(ray.o.x), fmaxf(fabsf(ray.o.y), fabsf(ray.o.z)));
                blocked[r] = false;
            }
        }
        for (uint32_t step = 0; step < interval; ++step) {
            const uint32_t first = tile * TILE_PRIMS;
            const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_tri - first);
            __syncthreads();
            stage_pretest_tile(lds, sc.trif, first, count, false, V3{0.0f, 0.0f, 0.0f});
            __syncthreads();
            Mask testing = 0;
            bool counted[R];
            uint32_t n_testing = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                counted[r] = active[r] && !blocked[r] && left[r] != 0u;
                const Mask m = ballot(counted[r]);
                testing |= m;
                n_testing += static_cast<uint32_t>(__popcll(m));
            }
            if (testing != 0) {
                wave_tests += static_cast<unsigned long long>(count) * n_testing;
                // (a lane of `testing` may hold a ray that does not count beside one that does: its result is computed like any other and dropped)
                pretest_tile<R, false>(lds, sc.tri, first, count, o, d, o_max, testing, counts, [&](int r, uint32_t, float t) __attribute__((always_inline)) {
                    blocked[r] |= (t < limit[r]) & counted[r];
                });
            }
#pragma unroll
            for (int r = 0; r < R; ++r) left[r] -= left[r] != 0u ? 1u : 0u;
            tile = tile + 1 == n_tiles ? 0u : tile + 1;
        }
    }
    if (lane == 0 && wave_tests) atomicAdd(stat_slot(st, ST_SHADOW_TESTS_TRI), wave_tests);
    add_brute_counts(st, counts, ST_BRUTE_STEPS_SHADOW);
}

// Any-hit of every hit point toward light first_light + blockIdx.y over the primitives of kind PRIM:
// the box hierarchy of the kind, queues
// as in k_closest_scan; a (ray, leaf) pair that blocks sets the ray's bit in LDS, which the lane
// that holds the ray picks up after the drain; the wave stops when all its rays are blocked.
template <int R, int LEVELS, int PRIM>
__global__ void __launch_bounds__(BLOCK) RT_SCAN_ATTR k_shadow_scan(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first) {
    static_assert(LEVELS == 3 || LEVELS == 4, "a walk from the top nodes or from the super nodes");
    constexpr int N_QUEUES = LEVELS;
    static_assert(64 * R <= 256, "a queue entry has 8 bits for the ray");
    const Hierarchy& h = PRIM == 0 ? sc.th : (PRIM == 1 ? sc.eh : sc.sh);
    constexpr uint32_t WAVE_LDS = N_QUEUES * QCAP + 64 * R;   // per wave: one queue per level ((ray, top node), (ray, cluster), (ray, leaf), with LEVELS = 4 also (ray, super node)), slot table
    __shared__ uint32_t lds_queue[(BLOCK / 64) * WAVE_LDS];
    __shared__ uint32_t lds_blocked[BLOCK];   // bit r: ray r of that thread is blocked (set by drains)
    // the wave's shadow rays, staged once for the lanes that walk their pairs: (o, limit) (d, -) (1/d, limit of the box test) per ray
    constexpr bool STAGED = RT_STAGE_RAYS(R, LEVELS);
    __shared__ float4 lds_rays[STAGED ? (BLOCK / 64) * 64 * R * 3 : 1];
    const uint32_t n_rays = st.counts[bounce + 1];
    const ScanSplit split = scan_split(n_rays, BLOCK * R, LEVELS == 4 ? h.n_supers : h.n_tops, LEVELS == 4 ? 1u : 16u);
    if (!split.active) return;
    const uint32_t base = split.ray_base;
    const uint32_t* __restrict__ list_in = st.list[(bounce + 1) & 1u];
    const uint32_t light = first_light + blockIdx.y;
    const float* __restrict__ L = sc.lights + light * 8;
    const uint32_t light_bit = 1u << (light - group_first);
    const bool point = L[0] != 0.0f;
    const V3 lv{L[1], L[2], L[3]};

    // a wave owns 64*R consecutive rays of the list (see k_closest_scan)
    uint32_t slot[R];
    bool valid[R], blocked[R];
    V3 d[R];
    BoxRay box_ray[R];
    // the segment toward a point light ends at the light for triangles and spheres (their tests compare the distance itself);
    // the ellipsoid test compares in its own space, so its boxes are tested against the whole half line
    constexpr bool LIMITED = PRIM != 1;
    Packet packet;
    bool unsafe = false;
    {
        V3 o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + (threadIdx.x >> 6) * (64 * R) + r * 64 + (threadIdx.x & 63u);
            valid[r] = i < n_rays;
            slot[r] = list_in[valid[r] ? i : base];
            const ShadowRay s = make_shadow_ray(st, slot[r], point, lv);
            o[r] = s.o;
            d[r] = s.d;
            box_ray[r] = make_box_ray(s.o, s.d, LIMITED ? s.limit : FLT_MAX);
            if (STAGED) {
                float4* const mine = lds_rays + ((threadIdx.x >> 6) * (64 * R) + r * 64 + (threadIdx.x & 63u)) * 3;
                mine[0] = make_float4(s.o.x, s.o.y, s.o.z, s.limit);
                mine[1] = make_float4(s.d.x, s.d.y, s.d.z, 0.0f);
                mine[2] = make_float4(box_ray[r].inv.x, box_ray[r].inv.y, box_ray[r].inv.z, box_ray[r].limit);
            }
            unsafe |= !(fmaxf(fabsf(s.o.x), fmaxf(fabsf(s.o.y), fabsf(s.o.z))) <= sc.box_bound);
            if (!point) unsafe |= !(vdot(s.d, s.d) <= SPH_DIR_CHECK);   // a directional light's direction is the caller's: the pre-tests and the spheres' boxes need unit length
            blocked[r] = (st.blocked[slot[r]] & light_bit) != 0u;
            lds_queue[(threadIdx.x >> 6) * WAVE_LDS + N_QUEUES * QCAP + r * 64 + (threadIdx.x & 63u)] = slot[r];
        }
        packet = uniform_packet(make_packet<R>(d, o, valid));
    }
    lds_blocked[threadIdx.x] = 0u;
    wave_sync();
    const bool keep_all = __any(unsafe);

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* const q0 = lds_queue + wave * WAVE_LDS;      // (ray, top node)
    uint32_t* const q1 = q0 + QCAP;                        // (ray, cluster)
    uint32_t* const q2 = q1 + QCAP;                        // (ray, leaf)
    uint32_t* const qs = q2 + QCAP;                        // (ray, super node), LEVELS = 4 only
    uint32_t* const wave_slots = q0 + N_QUEUES * QCAP;     // slot of ray r*64 + lane
    uint32_t q0_count = 0, q1_count = 0, q2_count = 0, qs_count = 0, n_full = 0, n_queued = 0;
    uint32_t n_first = 0, n_batches = 0;    // per wave: nodes of the starting level tested per ray, steps of 64 (ray, node) pairs walked
    unsigned long long n_pairs = 0;
    // shadow ray number `ray` of the wave (r*64 + lane of its owner) for the lane that handles one of its pairs
    struct PairRay { ShadowRay s; BoxRay box; };
    auto pair_ray = [&](uint32_t ray) __attribute__((always_inline)) {
        PairRay p;
        if (STAGED) {
            const float4* __restrict__ rec = lds_rays + (wave * (64 * R) + ray) * 3;
            const float4 a = rec[0], b = rec[1], c = rec[2];
            p.s = ShadowRay{xyz(a), xyz(b), a.w};
            p.box = BoxRay{xyz(a), xyz(c), c.w};
        } else {
            p.s = make_shadow_ray(st, wave_slots[ray], point, lv);
            p.box = make_box_ray(p.s.o, p.s.d, LIMITED ? p.s.limit : FLT_MAX);
        }
        return p;
    };
    // blocked bits of the wave's rays live in lds_blocked[thread that holds the ray], bit r; ray = r*64 + lane
    auto ray_blocked = [&](uint32_t ray) {
        return ((__hip_atomic_load(&lds_blocked[wave * 64u + (ray & 63u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> (ray >> 6)) & 1u) != 0u;
    };
    // (ray, leaf) pairs: the full test of the kind (triangles: after the two-stage pre-test), one pair per lane; a pair that blocks sets its ray's bit
    auto drain2 = [&]() __attribute__((always_inline)) {
        wave_sync();
        if (PRIM == 0) {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t ray = e >> 24;
                const ShadowRay s = pair_ray(ray).s;
                uint32_t k;
                if (pretests_then_full(sc, load_tri_pair(sc, e & NODE_MASK), s.o, s.d, n_full, k) < s.limit) atomicOr(&lds_blocked[wave * 64u + (ray & 63u)], 1u << (ray >> 6));
            }
        } else if (PRIM == 2) {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t ray = e >> 24;
                const ShadowRay s = pair_ray(ray).s;
                n_full++;
                if (hit_sphere(s.o, s.d, sc.sph[h.k_of[e & NODE_MASK]]) < s.limit) atomicOr(&lds_blocked[wave * 64u + (ray & 63u)], 1u << (ray >> 6));
            }
        } else {
            for (uint32_t i = lane; i < q2_count; i += 64) {
                const uint32_t e = q2[i];
                const uint32_t ray = e >> 24;
                const ShadowRay s = pair_ray(ray).s;
                const uint32_t k = h.k_of[e & NODE_MASK];
                const float4* __restrict__ rec = sc.ell + static_cast<size_t>(k) * 6;
                const V3 l2 = point ? xyz(sc.ell_light[static_cast<size_t>(light - sc.has_directional) * sc.n_ell + k]) : V3{0.0f, 0.0f, 0.0f};
                n_full++;
                if (ellipsoid_blocks(s.o, s.d, rec[0], rec[1], rec[2], point, l2)) atomicOr(&lds_blocked[wave * 64u + (ray & 63u)], 1u << (ray >> 6));
            }
        }
        wave_sync();
        n_queued += q2_count;
        q2_count = 0;
        const uint32_t mine = __hip_atomic_load(&lds_blocked[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
        for (int r = 0; r < R; ++r) blocked[r] |= ((mine >> r) & 1u) != 0u;
    };
    // (ray, node) pairs: one pair per lane, the lane walks the CLUSTER children of its node and queues
    // the ones its ray may reach one level down; pairs of rays that are blocked by now are dropped
    auto walk = [&](uint32_t* q_in, uint32_t& in_count, const uint4* __restrict__ nodes, uint32_t n_child,
                    uint32_t* q_out, uint32_t& out_count, auto&& drain_out) __attribute__((always_inline)) {
        wave_sync();
        for (uint32_t i0 = 0; i0 < in_count; i0 += 64) {
            bool active = i0 + lane < in_count;
            const uint32_t e = q_in[active ? i0 + lane : 0u];
            const uint32_t first = (e & NODE_MASK) * CLUSTER;
            active = active && !ray_blocked(e >> 24);
            const BoxRay ray = pair_ray(e >> 24).box;
            uint4 w[8];
            load_oriented_node(nodes + static_cast<size_t>(e & NODE_MASK) * 8, ray.inv, w);
            const NodeRay node_ray = make_node_ray(w[0], w[1], ray.o, ray.inv, ray.limit, 0.0f);
            uint32_t keep_bits = 0;
#pragma unroll
            for (int u = 0; u < static_cast<int>(CLUSTER); ++u) {
                // no short circuit: the test is cheaper than a branch around it
                keep_bits |= (child_keeps(node_ray, w, u) | keep_all) ? 1u << u : 0u;
            }
            keep_bits = active ? keep_bits & children_present(first, n_child) : 0u;
#pragma nounroll
            for (uint32_t quarter = 0; quarter < CLUSTER / PUSH_CHILDREN; ++quarter) {     // not unrolled: one copy of the code behind drain_out
                const uint32_t bits = (keep_bits >> (PUSH_CHILDREN * quarter)) & ((1u << PUSH_CHILDREN) - 1u);
                if (__any(bits != 0u)) {
                    if (out_count > DRAIN_ABOVE) drain_out();
                    push_bits(q_out, out_count, bits, PUSH_CHILDREN, (e & ~NODE_MASK) | (first + PUSH_CHILDREN * quarter), 1u);
                }
            }
        }
        n_pairs += static_cast<unsigned long long>(in_count) * CLUSTER;
        n_batches += (in_count + 63u) / 64u;
        wave_sync();
        in_count = 0;
    };
    auto drain1 = [&]() __attribute__((always_inline)) { walk(q1, q1_count, h.cl_nodes, h.n, q2, q2_count, drain2); };
    auto drain0 = [&]() __attribute__((always_inline)) { walk(q0, q0_count, h.top_nodes, h.n_clusters, q1, q1_count, drain1); };
    auto drain_supers = [&]() __attribute__((always_inline)) { walk(qs, qs_count, h.super_nodes, h.n_tops, q0, q0_count, drain0); };
    // the level the scan starts at: its boxes are tested one per lane against the wave's bound, then per ray
    const float4* __restrict__ first_box = LEVELS == 4 ? h.super_box : h.top_box;
    uint32_t* const q_first = LEVELS == 4 ? qs : q0;
    uint32_t& first_count = LEVELS == 4 ? qs_count : q0_count;
    auto drain_first = [&]() __attribute__((always_inline)) { if constexpr (LEVELS == 4) drain_supers(); else drain0(); };

    unsigned long long wave_tests = 0;   // tests spent on still-undecided rays (wave-uniform)
    {
        auto live_rays = [&]() {
            uint32_t n = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) n += __popcll(__ballot(valid[r] && !blocked[r]));
            return n;
        };
        uint32_t live = live_rays(), steps = 0;
        for (uint32_t c0 = split.first_prim; c0 < split.end_prim && live != 0; c0 += 64) {
            const size_t mine_at = min(c0 + lane, split.end_prim - 1u);
            const float4 mine_lo = first_box[mine_at * 2], mine_hi = first_box[mine_at * 2 + 1];
            constexpr uint32_t NODE_PRIMS = LEVELS == 4 ? CLUSTER * CLUSTER * CLUSTER : CLUSTER * CLUSTER;   // primitives below a node of the first level
            wave_tests += static_cast<unsigned long long>(min(h.n, min(c0 + 64u, split.end_prim) * NODE_PRIMS) - c0 * NODE_PRIMS) * live;
            const bool reach = packet_sphere_keeps(packet, box_sphere(mine_lo, mine_hi));
            unsigned long long mask = __ballot((reach || keep_all) && c0 + lane < split.end_prim);
            while (mask) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_ctzll(mask));
                mask &= mask - 1ull;
                const float4 lo = readlane4(mine_lo, bit), hi = readlane4(mine_hi, bit);
                uint32_t keep_bits = 0;
                n_pairs += live;   // the per-ray tests of this top node
                ++n_first;
#pragma unroll
                for (int r = 0; r < R; ++r) keep_bits |= (box_keeps(box_ray[r], lo, hi) || keep_all) && valid[r] && !blocked[r] ? 1u << r : 0u;
                if (__any(keep_bits != 0u)) {
                    if (first_count > FIRST_DRAIN_ABOVE) drain_first();
                    push_bits(q_first, first_count, keep_bits, R, (lane << 24) | (c0 + bit), 64u << 24);
                }
            }
            // work the queues off before looking at "all blocked"; the last step does it for the whole scan
            if (++steps % SHADOW_CHECK_STEPS == 0 || c0 + 64 >= split.end_prim) { drain_first(); if constexpr (LEVELS == 4) drain0(); drain1(); drain2(); live = live_rays(); }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (valid[r] && blocked[r]) atomicOr(&st.blocked[slot[r]], light_bit);
    if ((threadIdx.x & 63u) == 0 && wave_tests) atomicAdd(stat_slot(st, PRIM == 0 ? ST_SHADOW_TESTS_TRI : (PRIM == 1 ? ST_SHADOW_TESTS_ELL : ST_SHADOW_TESTS_SPH)), wave_tests);
    if (PRIM == 0) block_add_stat(stat_slot(st, ST_FULL_TESTS), n_full);
    if ((threadIdx.x & 63u) == 0 && n_pairs) atomicAdd(stat_slot(st, ST_PAIR_PRETESTS_SHADOW), n_pairs);
    if ((threadIdx.x & 63u) == 0 && n_queued) atomicAdd(stat_slot(st, ST_QUEUED_SHADOW), static_cast<unsigned long long>(n_queued));
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(stat_slot(st, ST_FIRST_NODES_SHADOW), static_cast<unsigned long long>(n_first));
        atomicAdd(stat_slot(st, ST_WALK_BATCHES_SHADOW), static_cast<unsigned long long>(n_batches));
        atomicAdd(stat_slot(st, ST_WALK_WAVES_SHADOW), 1ull);
    }
    {   // FLOP executed in per-ray tests (n_full is per lane; the others are wave totals)
        constexpr unsigned long long CHILD = 24, LEAF = PRIM == 0 ? 40 : (PRIM == 1 ? 71 : 19), FULL = PRIM == 0 ? 54 : 0;
        block_add_stat(stat_slot(st, ST_FLOP_SHADOW), FULL * n_full + ((threadIdx.x & 63u) == 0 ? CHILD * n_pairs + LEAF * n_queued : 0ull));
    }
}

// Shadow rays, each lane walking its own ray depth first.  A shadow ray only asks whether anything blocks it, and in a
// scene where most rays are blocked the answer is one leaf away: the queued walk (k_shadow_scan) expands every kept child
// of every level side by side and visits 18 nodes per blocked ray on the 1 M-triangle scene, where a walk that follows one
// child at a time needs a third of that.  Above the super nodes the hierarchy continues here with root nodes (16 super
// nodes each) and apex nodes (16 root nodes each), all PackedChildren records, so that every level is walked by the
// same code: a lane keeps, per level, the node it is in and the children of it still to visit (two words per level in LDS),
// takes the deepest pending child and holds it until the wave runs the part that serves it - an inner node is loaded and
// tested as in the walks above (load_oriented_node / child_keeps), a leaf goes through the full test of its kind.  Lanes
// are at different levels of different nodes in the same instruction.  A step runs either the node part or the leaf
// part, whichever has more lanes waiting (leaves weighted by DFS_LEAF_WEIGHT: their part is cheaper).  Rays end at very
// different times, so a wave owns 64 * rounds consecutive rays of the list and hands a new one to every idle lane
// whenever DFS_REFILL_MIN lanes are idle: the wave stays full until its share runs out.
// (The same walk for closest-hit rays loses: without an order among the children nothing shortens the ray early, 912 box
// tests per ray against the 480 of the queued walk with its nearest-class-first order, 479 ms against 199.)
#ifndef RT_DFS_LEAF_WEIGHT
#define RT_DFS_LEAF_WEIGHT 3
#endif
#ifndef RT_DFS_REFILL_MIN
#define RT_DFS_REFILL_MIN 16
#endif
#ifndef RT_DFS_BLOCKER_CACHE
#define RT_DFS_BLOCKER_CACHE 2
#endif
#ifndef RT_DFS_ATTR
#define RT_DFS_ATTR __attribute__((amdgpu_waves_per_eu(6)))
#endif
constexpr uint32_t DFS_LEAF_WEIGHT = RT_DFS_LEAF_WEIGHT, DFS_REFILL_MIN = RT_DFS_REFILL_MIN, DFS_LEVELS = 5;
template <int PRIM>
__global__ void __launch_bounds__(BLOCK) RT_DFS_ATTR k_shadow_dfs(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first, uint32_t rounds) {
    const Hierarchy& h = PRIM == 0 ? sc.th : (PRIM == 1 ? sc.eh : sc.sh);
    __shared__ uint32_t lds_node[DFS_LEVELS][BLOCK];   // per lane and level: the node whose children are being visited (0: an apex node, 1: a root node, 2: a super node, 3: a top node, 4: a cluster)
    __shared__ uint32_t lds_left[DFS_LEVELS][BLOCK];   // and its children still to visit (bit u = child 16 * node + u)
#if RT_DFS_BLOCKER_CACHE
    // The leaf that blocked the lane's last blocked ray is the first thing its next ray is tested against: the rays a lane
    // takes one after the other are a few pixels apart, and "blocked" is an OR over the leaves in any order.
    __shared__ uint32_t lds_blocker[BLOCK];
    lds_blocker[threadIdx.x] = 0xFFFFFFFFu;
#if RT_DFS_BLOCKER_CACHE > 1
    // and the leaf that blocked a ray of the wave most recently is the second.  (A ring of the last 2, 4 or 8 such leaves
    // per wave, newest first, finds a little more - 220.5, 217.3, 211.8 executed tests per ray against 221.8 - and costs
    // more steps than that saves: 187, 194, 205 ms per frame of the 1 M-triangle scene against 176.)
    __shared__ uint32_t lds_wave_blocker[BLOCK / 64];
    if ((threadIdx.x & 63u) == 0) lds_wave_blocker[threadIdx.x >> 6] = 0xFFFFFFFFu;
    constexpr uint32_t TRY_WAVE_BLOCKER = 1u << 31;        // (a bit of `pending`: the wave's leaf is still to be tried)
#endif
#endif
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t tid = threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t wave_first = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 64u * rounds;     // this wave's share of the list
    if (wave_first >= n_rays) return;
    const uint32_t wave_end = min(wave_first + 64u * rounds, n_rays);
    const uint32_t* __restrict__ list_in = st.list[(bounce + 1) & 1u];
    const uint32_t light = first_light + blockIdx.y;
    const float* __restrict__ L = sc.lights + light * 8;
    const uint32_t light_bit = 1u << (light - group_first);
    const bool point = L[0] != 0.0f;
    const V3 lv{L[1], L[2], L[3]};
    constexpr bool LIMITED = PRIM != 1;      // see k_shadow_scan

    uint32_t next = wave_first;              // the first ray of the share not handed out yet (the same in every lane)
    uint32_t slot = 0, pending = 0, apex_next = 0, n_taken = 0;
    ShadowRay s{};
    BoxRay ray{};
    bool open = false, unsafe = false, node_work = false, leaf_work = false;
    uint32_t item = 0, level = 0;            // the node to test, and which record array it is in (0 apex .. 4 cluster); or the leaf position
    uint32_t n_nodes = 0, n_leaves = 0, n_full = 0, n_steps = 0;   // per lane: inner nodes tested, leaves tested, full triangle tests; per wave: steps
    for (;;) {
        const Mask idle = ballot(!open);
        const uint32_t n_idle = __popcll(idle);
        if (next < wave_end && n_idle >= DFS_REFILL_MIN) {
            const uint32_t mine = next + __popcll(idle & ((1ull << lane) - 1ull));
            if (!open && mine < wave_end) {
                slot = list_in[mine];
                s = make_shadow_ray(st, slot, point, lv);
                ray = make_box_ray(s.o, s.d, LIMITED ? s.limit : FLT_MAX);
                unsafe = !(fmaxf(fabsf(s.o.x), fmaxf(fabsf(s.o.y), fabsf(s.o.z))) <= sc.box_bound);   // outside the bound the boxes were built for: keep everything
                if (!point) unsafe |= !(vdot(s.d, s.d) <= SPH_DIR_CHECK);
                open = (st.blocked[slot] & light_bit) == 0u;
                pending = 0u;
                apex_next = 0u;
                ++n_taken;
#if RT_DFS_BLOCKER_CACHE
                const uint32_t before = lds_blocker[tid];
                if (open && before != 0xFFFFFFFFu) { item = before; leaf_work = true; }
#if RT_DFS_BLOCKER_CACHE > 1
                pending = TRY_WAVE_BLOCKER;
#endif
#endif
            }
            next = min(next + n_idle, wave_end);
        }
        if (!__any(open)) {
            if (next >= wave_end) break;
            continue;      // every lane is idle: the next turn hands out rays
        }
        ++n_steps;
#if RT_DFS_BLOCKER_CACHE > 1
        wave_sync();      // (what other lanes have written is read from memory, not from a register of this lane)
        if (open && !node_work && !leaf_work && (pending & TRY_WAVE_BLOCKER) != 0u) {
            pending = 0u;
            const uint32_t recent = lds_wave_blocker[tid >> 6];
            if (recent != 0xFFFFFFFFu && recent != lds_blocker[tid]) { item = recent; leaf_work = true; }
        }
#endif
        if (open && !node_work && !leaf_work) {
            if (pending == 0u) {
                if (apex_next >= h.n_apex) {
                    open = false;      // nothing blocks it
                } else {
                    item = apex_next++;
                    level = 0u;
                    node_work = true;
                }
            } else {
                const uint32_t l = 31u - static_cast<uint32_t>(__builtin_clz(pending));     // the deepest level with children left
                uint32_t left = lds_left[l][tid];
                const uint32_t c = static_cast<uint32_t>(__builtin_ctz(left));
                left &= left - 1u;
                lds_left[l][tid] = left;
                if (left == 0u) pending &= ~(1u << l);
                item = lds_node[l][tid] * CLUSTER + c;
                if (l == DFS_LEVELS - 1u) leaf_work = true;
                else { node_work = true; level = l + 1u; }
            }
        }
        const uint32_t n_node = __popcll(__ballot(node_work)), n_leaf = __popcll(__ballot(leaf_work));
        if (n_leaf != 0u && (n_node == 0u || n_leaf * DFS_LEAF_WEIGHT >= n_node)) {
            if (leaf_work) {
                leaf_work = false;
                ++n_leaves;
                bool blocks;
                if (PRIM == 0) {
                    uint32_t k;
                    blocks = pretests_then_full(sc, load_tri_pair(sc, item), s.o, s.d, n_full, k) < s.limit;
                } else if (PRIM == 2) {
                    blocks = hit_sphere(s.o, s.d, sc.sph[h.k_of[item]]) < s.limit;
                } else {
                    const uint32_t k = h.k_of[item];
                    const float4* __restrict__ rec = sc.ell + static_cast<size_t>(k) * 6;
                    const V3 l2 = point ? xyz(sc.ell_light[static_cast<size_t>(light - sc.has_directional) * sc.n_ell + k]) : V3{0.0f, 0.0f, 0.0f};
                    blocks = ellipsoid_blocks(s.o, s.d, rec[0], rec[1], rec[2], point, l2);
                }
                if (blocks) {
                    atomicOr(&st.blocked[slot], light_bit);
                    open = false;
#if RT_DFS_BLOCKER_CACHE
                    lds_blocker[tid] = item;
#if RT_DFS_BLOCKER_CACHE > 1
                    lds_wave_blocker[tid >> 6] = item;       // (any one of the lanes that write in this step)
#endif
#endif
                }
            }
        } else if (n_node != 0u) {
            if (node_work) {
                node_work = false;
                const uint4* __restrict__ nodes = level == 0u ? h.apex_nodes : (level == 1u ? h.root_nodes : (level == 2u ? h.super_nodes : (level == 3u ? h.top_nodes : h.cl_nodes)));
                const uint32_t n_child = level == 0u ? h.n_roots : (level == 1u ? h.n_supers : (level == 2u ? h.n_tops : (level == 3u ? h.n_clusters : h.n)));
                uint4 w[8];
                load_oriented_node(nodes + static_cast<size_t>(item) * 8, ray.inv, w);
                const NodeRay node_ray = make_node_ray(w[0], w[1], ray.o, ray.inv, ray.limit, 0.0f);
                uint32_t keep_bits = 0;
#pragma unroll
                for (int u = 0; u < static_cast<int>(CLUSTER); ++u) keep_bits |= (child_keeps(node_ray, w, u) | unsafe) ? 1u << u : 0u;
                keep_bits &= children_present(item * CLUSTER, n_child);
                ++n_nodes;
                if (keep_bits != 0u) {
                    lds_node[level][tid] = item;
                    lds_left[level][tid] = keep_bits;
                    pending |= 1u << level;
                }
            }
        }
    }
    // work counters, as k_shadow_scan keeps them
    block_add_stat(stat_slot(st, PRIM == 0 ? ST_SHADOW_TESTS_TRI : (PRIM == 1 ? ST_SHADOW_TESTS_ELL : ST_SHADOW_TESTS_SPH)), static_cast<unsigned long long>(n_taken) * h.n);
    if (PRIM == 0) block_add_stat(stat_slot(st, ST_FULL_TESTS), n_full);
    block_add_stat(stat_slot(st, ST_PAIR_PRETESTS_SHADOW), static_cast<unsigned long long>(n_nodes) * CLUSTER);
    block_add_stat(stat_slot(st, ST_QUEUED_SHADOW), n_leaves);
    {
        constexpr unsigned long long CHILD = 24, LEAF = PRIM == 0 ? 40 : (PRIM == 1 ? 71 : 19), FULL = PRIM == 0 ? 54 : 0;
        block_add_stat(stat_slot(st, ST_FLOP_SHADOW), FULL * n_full + CHILD * static_cast<unsigned long long>(n_nodes) * CLUSTER + LEAF * n_leaves);
    }
    if (lane == 0) {
        atomicAdd(stat_slot(st, ST_WALK_BATCHES_SHADOW), static_cast<unsigned long long>(n_steps));
        atomicAdd(stat_slot(st, ST_WALK_WAVES_SHADOW), 1ull);
    }
}

template <int R>
__global__ void __launch_bounds__(BLOCK) k_shadow_sph(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first) {
    __shared__ float4 lds[TILE_PRIMS];
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    const float* __restrict__ L = sc.lights + (first_light + blockIdx.y) * 8;
    const uint32_t light_bit = 1u << (first_light + blockIdx.y - group_first);
    ShadowRays<R> s;
    load_shadow_rays<R>(st, st.list[(bounce + 1) & 1u], base, n_rays, L[0] != 0.0f, V3{L[1], L[2], L[3]}, light_bit, s);
    unsigned long long wave_tests = 0;
    bool done = shadow_wave_done<R>(s);
    for (uint32_t first = 0; first < sc.n_sph; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_sph - first);
        if (__syncthreads_and(done)) break;
        stage_tile<1>(lds, sc.sph, first, count);
        __syncthreads();
        if (!done) {
            wave_tests += shadow_live_tests<R>(s, count);
            for (uint32_t k = 0; k < count; ++k) {
                const float4 sp = lds[k];
                SphereFront front[R];
                Mask missed = ~0ull;
#pragma unroll
                for (int r = 0; r < R; ++r) { front[r] = sphere_front(s.o[r], s.d[r], sp); missed &= ballot(front[r].disc < 0.0f); }
                if (missed == ballot(true)) continue;      // no lane of the wave has a real root
#pragma unroll
                for (int r = 0; r < R; ++r) s.blocked[r] |= sphere_roots(front[r]) < s.limit[r];
            }
            done = shadow_wave_done<R>(s);
        }
    }
    store_shadow_result<R>(st, s, light_bit, stat_slot(st, ST_SHADOW_TESTS_SPH), wave_tests);
}

template <int R>
__global__ void __launch_bounds__(BLOCK) k_shadow_ell(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first) {
    __shared__ float4 lds[TILE_PRIMS * 4];
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t base = blockIdx.x * (BLOCK * R);
    if (base >= n_rays) return;
    const uint32_t light = first_light + blockIdx.y;
    const float* __restrict__ L = sc.lights + light * 8;
    const bool point = L[0] != 0.0f;
    const uint32_t light_bit = 1u << (light - group_first);
    ShadowRays<R> s;
    load_shadow_rays<R>(st, st.list[(bounce + 1) & 1u], base, n_rays, point, V3{L[1], L[2], L[3]}, light_bit, s);
    unsigned long long wave_tests = 0;
    bool done = shadow_wave_done<R>(s);
    const uint32_t point_index = light - sc.has_directional;   // index among the point lights
    for (uint32_t first = 0; first < sc.n_ell; first += TILE_PRIMS) {
        const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), sc.n_ell - first);
        if (__syncthreads_and(done)) break;
        // three inverse-transform rows, then the light position in ellipsoid space
        for (uint32_t i = threadIdx.x; i < count * 4; i += BLOCK) {
            const uint32_t k = i / 4, w = i % 4;
            lds[i] = (w < 3) ? sc.ell[static_cast<size_t>(first + k) * 6 + w]
                             : (point ? sc.ell_light[static_cast<size_t>(point_index) * sc.n_ell + first + k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
        __syncthreads();
        if (!done) {
            wave_tests += shadow_live_tests<R>(s, count);
            for (uint32_t k = 0; k < count; ++k) {
                const float4 i0 = lds[k * 4], i1 = lds[k * 4 + 1], i2 = lds[k * 4 + 2], l2 = lds[k * 4 + 3];
#pragma unroll
                for (int r = 0; r < R; ++r) s.blocked[r] |= ellipsoid_blocks(s.o[r], s.d[r], i0, i1, i2, point, xyz(l2));
            }
            done = shadow_wave_done<R>(s);
        }
    }
    store_shadow_result<R>(st, s, light_bit, stat_slot(st, ST_SHADOW_TESTS_ELL), wave_tests);
}


// Brute-force any-hit over the ellipsoids (PRIM 1, the test in stages) or the spheres (PRIM 2) toward light
// first_light + blockIdx.y.  These scans follow the triangles' scan, which in a dense scene has blocked most rays already:
// a workgroup therefore owns a range of the ray list (blocks_per_wg x BLOCK entries), collects the rays of the range that
// are still open, in order, and scans the primitive list once per BLOCK x R of them, so that its waves hold open rays only
// (the scans as written below, k_shadow_sph / k_shadow_ell, load every ray of the list and run a wave to the end of the
// primitive list for its one open ray).  Tests are counted like there: primitives of the tiles staged while the ray was open.
#ifndef RT_SMALL_SHADOW_RANGE_BLOCKS
#define RT_SMALL_SHADOW_RANGE_BLOCKS 16
#endif
constexpr uint32_t SMALL_SHADOW_RANGE_BLOCKS = RT_SMALL_SHADOW_RANGE_BLOCKS;
template <int R>
DEV void ellipsoid_tile_shadow(const float4* lds, uint32_t count, ShadowRays<R>& s, bool point) {
    for (uint32_t k = 0; k < count; ++k) {
        const float4 i0 = lds[k], i1 = lds[TILE_PRIMS + k], i2 = lds[2 * TILE_PRIMS + k];
        EllParts e[R];
        Mask open = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            e[r] = ell_parts(i0, i1, i2, s.o[r], s.d[r]);
            open |= ballot(!ell_certain_miss(e[r], ell_margin_term(e[r].oo)) && s.valid[r] && !s.blocked[r]);
        }
        if (open == 0) continue;
        const float4 l2 = lds[3 * TILE_PRIMS + k];
#pragma unroll
        for (int r = 0; r < R; ++r) s.blocked[r] |= ell_root_blocks(e[r], ell_root(e[r]), point, xyz(l2));
    }
}
template <int R, int PRIM>
__global__ void __launch_bounds__(BLOCK) k_shadow_brute(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t group_first) {
    static_assert(PRIM == 1 || PRIM == 2, "ellipsoids or spheres");
    constexpr uint32_t PARTS = PRIM == 1 ? 4 : 1, CAPACITY = BLOCK * R;
    __shared__ float4 lds[TILE_PRIMS * PARTS];
    __shared__ uint32_t open_slot[CAPACITY + BLOCK];      // the open rays collected so far, in list order
    __shared__ uint32_t wave_open[BLOCK / 64];
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t n_blocks = (n_rays + BLOCK - 1) / BLOCK;
    const uint32_t blocks_per_wg = min(SMALL_SHADOW_RANGE_BLOCKS, max(1u, n_blocks * gridDim.y / SHADOW_MIN_RANGES));
    if (static_cast<unsigned long long>(blockIdx.x) * blocks_per_wg >= n_blocks) return;      // (also: no rays)
    uint32_t next = blockIdx.x * blocks_per_wg * BLOCK;                                        // the next entry of the range to look at
    const uint32_t range_end = static_cast<uint32_t>(min(static_cast<unsigned long long>(n_rays), static_cast<unsigned long long>(next) + static_cast<unsigned long long>(blocks_per_wg) * BLOCK));
    const uint32_t light = first_light + blockIdx.y;
    const float* __restrict__ L = sc.lights + light * 8;
    const bool point = L[0] != 0.0f;
    const V3 lv{L[1], L[2], L[3]};
    const uint32_t light_bit = 1u << (light - group_first);
    const uint32_t* __restrict__ list = st.list[(bounce + 1) & 1u];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_prims = PRIM == 1 ? sc.n_ell : sc.n_sph;
    const float4* light_row = (PRIM == 1 && point) ? sc.ell_light + static_cast<size_t>(light - sc.has_directional) * sc.n_ell : nullptr;
    uint32_t n_open = 0;
    unsigned long long wave_tests = 0;
    for (;;) {
        while (n_open < CAPACITY && next < range_end) {       // uniform over the workgroup
            const uint32_t i = next + threadIdx.x;
            uint32_t slot = 0;
            bool is_open = false;
            if (i < range_end) { slot = list[i]; is_open = (st.blocked[slot] & light_bit) == 0u; }
            const Mask mask = ballot(is_open);
            if (lane == 0) wave_open[wave] = static_cast<uint32_t>(__popcll(mask));
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < BLOCK / 64; ++w) { const uint32_t c = wave_open[w]; before += w < wave ? c : 0u; total += c; }
            if (is_open) open_slot[n_open + before + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)))] = slot;
            __syncthreads();
            n_open += total;
            next += BLOCK;
        }
        if (n_open == 0) break;
        const uint32_t take = min(n_open, CAPACITY);
        ShadowRays<R> s;
        bool tame = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t idx = r * BLOCK + threadIdx.x;
            s.valid[r] = idx < take;
            s.slot[r] = open_slot[s.valid[r] ? idx : 0u];          // lanes without a ray repeat the first one and store nothing
            const ShadowRay ray = make_shadow_ray(st, s.slot[r], point, lv);
            s.o[r] = ray.o; s.d[r] = ray.d; s.limit[r] = ray.limit;
            s.blocked[r] = false;
            tame = tame && ray_is_tame_for_ellipsoids(ray.o, ray.d);
        }
        const bool staged_wave = __all(tame);
        bool done = false;
        for (uint32_t first = 0; first < n_prims; first += TILE_PRIMS) {
            const uint32_t count = min(static_cast<uint32_t>(TILE_PRIMS), n_prims - first);
            if (__syncthreads_and(done)) break;
            if (PRIM == 1) {
                const bool as_written = __syncthreads_or(stage_ellipsoid_tile(lds, sc, first, count, false, V3{0.0f, 0.0f, 0.0f}, light_row));
                if (!done) {
                    wave_tests += shadow_live_tests<R>(s, count);
                    if (as_written || !staged_wave) {      // rare: the records straight from memory
                        for (uint32_t k = 0; k < count; ++k) {
                            const float4* rec = sc.ell + static_cast<size_t>(first + k) * 6;
                            const float4 i0 = rec[0], i1 = rec[1], i2 = rec[2];
                            const float4 l2 = light_row ? light_row[first + k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                            for (int r = 0; r < R; ++r) s.blocked[r] |= ellipsoid_blocks(s.o[r], s.d[r], i0, i1, i2, point, xyz(l2));
                        }
                    } else {
                        ellipsoid_tile_shadow<R>(lds, count, s, point);
                    }
                }
            } else {
                stage_tile<1>(lds, sc.sph, first, count);
                __syncthreads();
                if (!done) {
                    wave_tests += shadow_live_tests<R>(s, count);
                    for (uint32_t k = 0; k < count; ++k) {
                        const float4 sp = lds[k];
                        SphereFront front[R];
                        Mask missed = ~0ull;
#pragma unroll
                        for (int r = 0; r < R; ++r) { front[r] = sphere_front(s.o[r], s.d[r], sp); missed &= ballot(front[r].disc < 0.0f); }
                        if (missed == ballot(true)) continue;      // no lane of the wave has a real root
#pragma unroll
                        for (int r = 0; r < R; ++r) s.blocked[r] |= sphere_roots(front[r]) < s.limit[r];
                    }
                }
            }
            if (!done) done = shadow_wave_done<R>(s);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (s.valid[r] && s.blocked[r]) atomicOr(&st.blocked[s.slot[r]], light_bit);
        // the rays collected beyond this batch move to the front
        __syncthreads();
        const uint32_t left = n_open - take;
        uint32_t keep = 0;
        if (threadIdx.x < left) keep = open_slot[take + threadIdx.x];
        __syncthreads();
        if (threadIdx.x < left) open_slot[threadIdx.x] = keep;
        __syncthreads();
        n_open = left;
    }
    if (lane == 0 && wave_tests) atomicAdd(stat_slot(st, PRIM == 1 ? ST_SHADOW_TESTS_ELL : ST_SHADOW_TESTS_SPH), wave_tests);
}

// ------------------------------------------------------------------------------------------
// k_shade: lights [first_light, first_light + n_group) in order; the last group also reflects
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_shade(DevScene sc, State st, uint32_t bounce, uint32_t first_light, uint32_t n_group, uint32_t last_group) {
    const uint32_t n_rays = st.counts[bounce + 1];
    const uint32_t base = blockIdx.x * BLOCK;
    if (base >= n_rays) return;
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < n_rays;            // lanes past the end repeat the first ray and store nothing
    const uint32_t s = st.list[(bounce + 1) & 1u][valid ? i : base];
    const uint32_t id = st.hit_id[s];
    const uint32_t kind = id >> 30, index = id & 0x3FFFFFFFu;
    const float* mat = kind == 0 ? sc.tri_mat : (kind == 1 ? sc.sph_mat : sc.ell_mat);
    const uint32_t mat_count = kind == 0 ? sc.n_tri_mat : (kind == 1 ? sc.n_sph_mat : sc.n_ell_mat);
    float m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (index < mat_count) for (int k = 0; k < 10; ++k) m[k] = mat[index * 10 + k];

    const V3 p{st.px[s], st.py[s], st.pz[s]};
    const V3 n{st.nx[s], st.ny[s], st.nz[s]};
    const V3 d{st.dx[s], st.dy[s], st.dz[s]};
    float c[3] = {st.cr[s], st.cg[s], st.cb[s]};
    float w[3] = {st.wr[s], st.wg[s], st.wb[s]};
    const V3 to_eye = scale(-1.0f, d);                                    // ray_tracing.cpp:684
    const uint32_t blocked = st.blocked[s];
    block_add_stat(stat_slot(st, ST_SHADOW_RAYS), valid ? n_group : 0u);
    block_add_stat(stat_slot(st, ST_SHADOW_BLOCKED), valid ? __popc(blocked & (n_group >= 32 ? 0xFFFFFFFFu : ((1u << n_group) - 1u))) : 0);

    for (uint32_t g = 0; g < n_group; ++g) {
        if (blocked & (1u << g)) continue;
        const float* L = sc.lights + (first_light + g) * 8;
        const V3 lv{L[1], L[2], L[3]};
        V3 to_light;
        float scale_r = L[4], scale_g = L[5], scale_b = L[6];
        if (L[0] != 0.0f) {                                               // point light, ray_tracing.cpp:709-722
            const V3 full = sub(lv, p);
            const float dist = sqrt_(sdot(full, full));
            to_light = divide(full, dist);
            const float linear_term = sc.atten[1] * dist;                 // ray_tracing.cpp:328-334
            const float quadratic_term = sc.atten[2] * dist * dist;
            const float att = 1.0f / (sc.atten[0] + linear_term + quadratic_term);
            scale_r = att * L[4]; scale_g = att * L[5]; scale_b = att * L[6];
        } else {                                                          // directional, ray_tracing.cpp:690-700
            to_light = scale(-1.0f, lv);
        }
        const float di = fp_max(sdot(n, to_light), 0.0f);
        const V3 half = normalise(add(to_eye, to_light));
        const float si = fp_pow(fp_max(sdot(n, half), 0.0f), m[9]);
        c[0] += w[0] * (scale_r * (di * m[0] + si * m[3]));
        c[1] += w[1] * (scale_g * (di * m[1] + si * m[4]));
        c[2] += w[2] * (scale_b * (di * m[2] + si * m[5]));
    }
    if (!valid) return;
    st.cr[s] = c[0]; st.cg[s] = c[1]; st.cb[s] = c[2];
    if (last_group) {
        st.blocked[s] = 0u;
        st.wr[s] = w[0] * m[3]; st.wg[s] = w[1] * m[4]; st.wb[s] = w[2] * m[5];   // ray_tracing.cpp:726
    } else {
        st.blocked[s] = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// k_resolve: one thread per pixel of the chunk's tiles
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_resolve(State st, uint32_t first_tile_slot, uint8_t* __restrict__ tiles_out) {
    const uint32_t pixel = blockIdx.x * BLOCK + threadIdx.x;  // chunk-local pixel; the grid is exact (256 pixels per tile)
    const uint32_t s0 = pixel * RT_SAMPLES_PER_PIXEL;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < RT_SAMPLES_PER_PIXEL; ++k) { r += st.cr[s0 + k]; g += st.cg[s0 + k]; b += st.cb[s0 + k]; }
    r = 0.0625f * r; g = 0.0625f * g; b = 0.0625f * b;                   // render_work_queue.cpp:51
    // truncating conversion, low byte kept: what the x86 build of render_work_queue.cpp:53-55 does
    const uint32_t in_tile = pixel % 256u;   // tile buffers are row-major, the samples are not (tile_pixel_x)
    uint8_t* out = tiles_out + (static_cast<size_t>(first_tile_slot) * 256 + (pixel - in_tile) + tile_pixel_y(in_tile) * RT_TILE + tile_pixel_x(in_tile)) * 3;
    out[0] = static_cast<uint8_t>(static_cast<int>(r * 255.0f) & 0xFF);
    out[1] = static_cast<uint8_t>(static_cast<int>(g * 255.0f) & 0xFF);
    out[2] = static_cast<uint8_t>(static_cast<int>(b * 255.0f) & 0xFF);
}

// Tile buffers (rank-major) -> row-major image of the rectangle.
__global__ void __launch_bounds__(BLOCK) k_assemble(const uint8_t* __restrict__ tiles, uint32_t stride_tiles, uint32_t world,
                                                     uint32_t rect_w, uint32_t rect_h, uint32_t tiles_x, uint8_t* __restrict__ rgb) {
    const uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= rect_w * rect_h) return;
    const uint32_t x = idx % rect_w, y = idx / rect_w;
    const uint32_t tile = tile_number(x / RT_TILE, y / RT_TILE, tiles_x);
    const uint32_t rank = tile % world, slot = tile / world;
    const uint8_t* src = tiles + ((static_cast<size_t>(rank) * stride_tiles + slot) * 256 + (y % RT_TILE) * RT_TILE + x % RT_TILE) * 3;
    uint8_t* dst = rgb + static_cast<size_t>(idx) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// Chunk-local primary ids -> [y][x][sample] of the rectangle.
__global__ void __launch_bounds__(BLOCK) k_scatter_ids(const uint32_t* __restrict__ chunk_ids, uint32_t first_tile_slot, uint32_t n_samples,
                                                        Frame f, uint32_t* __restrict__ out) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n_samples) return;
    const uint32_t tile = (first_tile_slot + s / SAMPLES_PER_TILE) * f.tile_world + f.tile_rank;
    const uint32_t in_tile = s % SAMPLES_PER_TILE;
    const uint32_t pixel = in_tile / RT_SAMPLES_PER_PIXEL, sample = in_tile % RT_SAMPLES_PER_PIXEL;
    const uint32_t rx = tile_column(tile, f.tiles_x) * RT_TILE + tile_pixel_x(pixel);
    const uint32_t ry = tile_row(tile, f.tiles_x) * RT_TILE + tile_pixel_y(pixel);
    if (rx < f.rect_w && ry < f.rect_h) out[(static_cast<size_t>(ry) * f.rect_w + rx) * RT_SAMPLES_PER_PIXEL + sample] = chunk_ids[s];
}

// ------------------------------------------------------------------------------------------
// k_debug_eval: the device routines one at a time, for known-answer tests (rt_debug_eval)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_debug_eval(int op, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t j = i < n ? i : 0u;          // lanes past the end repeat item 0 and store nothing (wave-wide ballots need every lane)
    const bool store = i < n;
    switch (op) {
    case RT_DEBUG_FP_POW: { const float v = fp_pow(a[j], b[j]); if (store) out[j] = v; break; }
    case RT_DEBUG_FP_MIN: { const float v = fp_min(a[j], b[j]); if (store) out[j] = v; break; }
    case RT_DEBUG_FP_MAX: { const float v = fp_max(a[j], b[j]); if (store) out[j] = v; break; }
    case RT_DEBUG_DIVIDE: { const float v = a[j] / b[j]; if (store) out[j] = v; break; }
    case RT_DEBUG_DIVIDE_IN_RANGE: { const float v = divide_in_range(a[j], b[j]); if (store) out[j] = v; break; }
    case RT_DEBUG_PRIMARY_RAY: {   // a: eye, i, j, k, half_world (14), half_w_px, half_h_px; b: x, y, x offset, y offset per item
        Frame f;
        for (int q = 0; q < 3; ++q) { f.eye[q] = a[q]; f.bi[q] = a[3 + q]; f.bj[q] = a[6 + q]; f.bk[q] = a[9 + q]; }
        f.hw[0] = a[12]; f.hw[1] = a[13]; f.half_w_px = a[14]; f.half_h_px = a[15];
        const V3 d = primary_direction(f, static_cast<uint32_t>(b[j * 4]), static_cast<uint32_t>(b[j * 4 + 1]), b[j * 4 + 2], b[j * 4 + 3]);
        if (store) { out[j * 3] = d.x; out[j * 3 + 1] = d.y; out[j * 3 + 2] = d.z; }
        break;
    }
    case RT_DEBUG_SCENE_BOX: {     // a: box (6); b: origin, direction per item
        float lo, hi;
        scene_box_distances(V3{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, V3{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]}, a, lo, hi);
        if (store) { out[j * 2] = lo; out[j * 2 + 1] = hi; }
        break;
    }
    case RT_DEBUG_TRIANGLE:
    case RT_DEBUG_TRIANGLE_STAGED: {   // a: a, a_to_b, a_to_c per item (9); b: origin, direction per item
        float4 rec[4];
        make_tri_record(V3{a[j * 9], a[j * 9 + 1], a[j * 9 + 2]}, V3{a[j * 9 + 3], a[j * 9 + 4], a[j * 9 + 5]}, V3{a[j * 9 + 6], a[j * 9 + 7], a[j * 9 + 8]}, rec);
        const V3 o{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, d{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]};
        float t = hit_triangle(o, d, rec[0], rec[1], rec[2], rec[3].x);
        if (op == RT_DEBUG_TRIANGLE_STAGED && ray_is_tame(o, d) && rec[3].w == rec[3].w) {   // (the scans take rays and records outside the staged form's range through the test as written)
            Mask out_mask;
            const TriFront front = tri_front(o, d, rec[0], rec[1], rec[2], out_mask);
            const TriHead head = tri_head(o, d, front, rec[0], rec[1], rec[2], rec[3].w, out_mask);
            bool unsure;
            const float staged = tri_tail(head, ((out_mask >> (threadIdx.x & 63u)) & 1ull) != 0ull, rec[0], rec[1], rec[2], rec[3].y, unsure);
            // an unsure pair reports NaN: the scans run the plain form for it, and the test counts such pairs
            t = unsure ? __builtin_nanf("") : staged;
        }
        if (store) out[j] = t;
        break;
    }
    case RT_DEBUG_TRIANGLE_PRETEST: {   // a, b as for TRIANGLE; out: (result, the step that decided the pair) per item
        // the default form of the brute-force scans for one pair: sign test, first and second pre-test, the test as written
        float4 rec[4], pre[4];
        const V3 ta{a[j * 9], a[j * 9 + 1], a[j * 9 + 2]}, tab{a[j * 9 + 3], a[j * 9 + 4], a[j * 9 + 5]}, tac{a[j * 9 + 6], a[j * 9 + 7], a[j * 9 + 8]};
        make_tri_record(ta, tab, tac, rec);
        make_pretest_record(ta, tab, tac, rec, pre);
        const V3 o{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, d{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]};
        const float o_max = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
        float t = FLT_MAX, step = 0.0f;
        bool behind;
        Pre stage;
        const PreFront front = pretest_front(o, d, pre[0], pre[1], behind);
        if (!behind) {
            step = 1.0f;
            if (pretest_stage1_rest(front, d, o_max, pre[0], pre[1], pre[2], stage)) {
                step = 2.0f;
                if (pretest_stage2(stage, pre[3])) { step = 3.0f; t = hit_triangle(o, d, rec[0], rec[1], rec[2], rec[3].x); }
            }
        }
        // rays of one origin: the record's a - o and (a - o).n, no sign test (stage_pretest_tile); must decide the same way from the first pre-test on
        const V3 ao = sub(xyz(pre[1]), o);
        Pre shared_stage;
        const bool kept_shared = pretest_stage1_shared(d, o_max, make_float4(pre[0].x, pre[0].y, pre[0].z, vdot(ao, xyz(pre[0]))), make_float4(ao.x, ao.y, ao.z, pre[0].w),
                                                       make_float4(pre[2].x, pre[2].y, pre[2].z, pre[1].w), shared_stage) && pretest_stage2(shared_stage, pre[3]);
        if (store) { out[j * 3] = t; out[j * 3 + 1] = step; out[j * 3 + 2] = kept_shared ? 1.0f : 0.0f; }
        break;
    }
    case RT_DEBUG_SPHERE: {        // a: centre, radius per item (4); b: origin, direction per item
        const float t = hit_sphere(V3{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, V3{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]}, make_float4(a[j * 4], a[j * 4 + 1], a[j * 4 + 2], a[j * 4 + 3]));
        if (store) out[j] = t;
        break;
    }
    case RT_DEBUG_ELLIPSOID:
    case RT_DEBUG_ELLIPSOID_STAGED: {   // a: rows of the inverse transform, rows of the transform per item (24); b: origin, direction per item
        float4 rec[6];
        for (int row = 0; row < 6; ++row) rec[row] = make_float4(a[j * 24 + row * 4], a[j * 24 + row * 4 + 1], a[j * 24 + row * 4 + 2], a[j * 24 + row * 4 + 3]);
        const V3 o{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, d{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]};
        float t = hit_ellipsoid(o, d, rec[0], rec[1], rec[2], rec[3], rec[4], rec[5]);
        if (op == RT_DEBUG_ELLIPSOID) { if (store) out[j] = t; break; }
        float stage = 3.0f;     // 3: the test as written (the scans take records and rays outside the staged form's range through it)
        if (ell_record_is_regular(rec) && ray_is_tame_for_ellipsoids(o, d)) {
            const EllParts e = ell_parts(rec[0], rec[1], rec[2], o, d);
            if (ell_certain_miss(e, ell_margin_term(e.oo))) { t = FLT_MAX; stage = 0.0f; }
            else {
                const EllRoot root = ell_root(e);
                if (ell_root_is_out(root)) { t = FLT_MAX; stage = 1.0f; }
                else { t = ell_world_distance(o, e, root, rec[3], rec[4], rec[5]); stage = 2.0f; }
            }
        }
        if (store) { out[j * 2] = t; out[j * 2 + 1] = stage; }
        break;
    }
    case RT_DEBUG_ELLIPSOID_SHADOW:
    case RT_DEBUG_ELLIPSOID_SHADOW_STAGED: {   // a: rows of the inverse transform (12), light position (3), 1 for a point light else 0, per item; b: origin, direction per item
        float4 rec[6];
        for (int row = 0; row < 3; ++row) rec[row] = make_float4(a[j * 16 + row * 4], a[j * 16 + row * 4 + 1], a[j * 16 + row * 4 + 2], a[j * 16 + row * 4 + 3]);
        rec[3] = make_float4(1.0f, 0.0f, 0.0f, 0.0f); rec[4] = make_float4(0.0f, 1.0f, 0.0f, 0.0f); rec[5] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);   // (the shadow test does not use the transform)
        const bool point = a[j * 16 + 15] != 0.0f;
        const V3 l2 = vmat_point(rec[0], rec[1], rec[2], V3{a[j * 16 + 12], a[j * 16 + 13], a[j * 16 + 14]});     // as k_prep_ellipsoid_lights
        const V3 o{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, d{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]};
        bool blocked = ellipsoid_blocks(o, d, rec[0], rec[1], rec[2], point, l2);
        if (op == RT_DEBUG_ELLIPSOID_SHADOW) { if (store) out[j] = blocked ? 1.0f : 0.0f; break; }
        float stage = 3.0f;
        const bool light_fits = !point || fmaxf(fabsf(l2.x), fmaxf(fabsf(l2.y), fabsf(l2.z))) <= ELL_LIGHT_COORD;
        if (ell_record_is_regular(rec) && light_fits && ray_is_tame_for_ellipsoids(o, d)) {
            const EllParts e = ell_parts(rec[0], rec[1], rec[2], o, d);
            if (ell_certain_miss(e, ell_margin_term(e.oo))) { blocked = false; stage = 0.0f; }
            else { blocked = ell_root_blocks(e, ell_root(e), point, l2); stage = 1.0f; }
        }
        if (store) { out[j * 2] = blocked ? 1.0f : 0.0f; out[j * 2 + 1] = stage; }
        break;
    }
    case RT_DEBUG_ELLIPSOID_PARTS: {    // a: rows of the inverse transform per item (12); b: origin, direction per item
        const float4 i0 = make_float4(a[j * 12], a[j * 12 + 1], a[j * 12 + 2], a[j * 12 + 3]), i1 = make_float4(a[j * 12 + 4], a[j * 12 + 5], a[j * 12 + 6], a[j * 12 + 7]),
                     i2 = make_float4(a[j * 12 + 8], a[j * 12 + 9], a[j * 12 + 10], a[j * 12 + 11]);
        const V3 o{b[j * 6], b[j * 6 + 1], b[j * 6 + 2]}, d{b[j * 6 + 3], b[j * 6 + 4], b[j * 6 + 5]};
        const V3 o2 = vmat_point(i0, i1, i2, o);                         // Mat3x4AVX * Vec3AVX, ray_tracing.cpp:105-126
        const V3 d2 = vmat_dir(i0, i1, i2, d);                           // transform_direction, :208-227
        const bool absent = (d2.x < TOL) & (d2.y < TOL) & (d2.z < TOL);  // ellipsoid_is_absent, :336-344
        const V3 unit = scale(1.0f / sqrt_(vdot(d2, d2)), d2);           // :564-566
        const float t2 = hit_unit_sphere(o2, unit);                      // intersect_with_unit_sphere, :229-263
        if (store) { float* q = out + j * 8; q[0] = o2.x; q[1] = o2.y; q[2] = o2.z; q[3] = d2.x; q[4] = d2.y; q[5] = d2.z; q[6] = absent ? 1.0f : 0.0f; q[7] = t2; }
        break;
    }
    default: break;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Chunks of a frame are traced on MAX_LANES "lanes" side by side: each has its own pair of streams and its
// own ray-state arena, so the thinly filled launches of one chunk's late bounces share the chip with
// the other chunk's early ones.
#ifndef RT_LANES
#define RT_LANES 2
#endif
constexpr uint32_t MAX_LANES = RT_LANES;
struct Lane {
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;      // the next bounce's closest-hit scans run here, beside the shadow scans
    std::vector<hipEvent_t> sync_events;    // fork / join between the two streams (no timing)
    void* state_mem = nullptr;              // per-chunk sample state, grown on demand
    size_t state_samples = 0;
    uint32_t* counts = nullptr;             // 64 words
    uint32_t* host_counts = nullptr;        // pinned copy of `counts`, filled word by word as the host asks (fetch_count)
    hipEvent_t count_ready = nullptr;       // the word last asked for has arrived
};
// What one frame in the making needs: lanes (streams + ray-state arenas), the events between them, work counters.
// A device has FRAME_SLOTS of them, so that the first launches of frame k+1 can fill the chip while the thin last
// launches of frame k drain (rt_render_tiles_begin / rt_render_tiles_end); the plain render calls use slot 0.
constexpr int FRAME_SLOTS = 2;
struct FrameSlot {
    Lane lanes[MAX_LANES];
    hipEvent_t lanes_fork = nullptr, lanes_join[MAX_LANES] = {}, lanes_started[MAX_LANES] = {};
    unsigned long long* stats = nullptr;    // STAT_COPIES x ST_COUNT words
    std::vector<hipEvent_t> events;         // timing events, reused from call to call
    std::mutex guard;                       // one frame at a time in this slot
    // a frame started with rt_render_tiles_begin: driven by a host thread of its own
    std::thread worker;
    bool busy = false;
    int rc = 0;
    std::string error;
    rt_stats stats_out;
    // a frame begun and never ended is still completed: its thread is joined when the library is unloaded
    ~FrameSlot() { if (worker.joinable()) worker.join(); }
};
struct Library {
    bool ready = false;
    int device = -1;                        // this context's index (g_devices[device]); the HIP device of that number unless hip_device says otherwise
    int hip_device = -1;                    // the HIP device the context lives on (rt_init_devices can put several contexts on one GPU: tests)
    FrameSlot slots[FRAME_SLOTS];
    hipStream_t stream = nullptr;           // = slots[0].lanes[0].stream: uploads, scene preparation, assembly
    unsigned char* scratch = nullptr;       // device address of `scratch` on this device (hipGetSymbolAddress)
    std::mutex guard;                       // scene creation, assembly, the test hook: one at a time
    std::mutex frames_guard;                // rt_render_tiles_begin / _end: which slot holds a begun frame
    // multi-GPU (rt_comm_*, rt_init_devices): this device's rank in a communicator, a stream for the exchange, tile buffers
    void* comm = nullptr;                   // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready = nullptr;        // the tiles to send are complete
    // rt_render_multi*: this device's tile buffers (two frames may be under way: one being traced, one being exchanged), an
    // event per buffer that says its exchange is complete, and on device 0 the gathered buffers and rt_render_multi's image
    uint8_t* multi_tiles[2] = {nullptr, nullptr};
    uint8_t* multi_gathered[2] = {nullptr, nullptr};
    uint8_t* multi_image = nullptr;
    size_t multi_tiles_bytes[2] = {0, 0}, multi_gathered_bytes[2] = {0, 0}, multi_image_bytes = 0;
    hipEvent_t multi_sent[2] = {nullptr, nullptr};
    hipEvent_t multi_exchange_begin[2] = {nullptr, nullptr}, multi_exchange_end[2] = {nullptr, nullptr};   // device 0: around gather + assembly (timed)
};
// One context per HIP device.  A process that renders on one GPU (rt_init) has one; rt_init_devices creates one per
// device of a single process.  Calls use the context their thread was bound to (bind_device), or the first one created.
constexpr int MAX_DEVICES = 16;
Library g_devices[MAX_DEVICES];
std::atomic<Library*> g_first_device{nullptr};
std::atomic<int> g_multi_devices{0};        // > 0: rt_init_devices was called with this many devices
thread_local Library* t_device = nullptr;
Library& current_device() {
    if (t_device) return *t_device;
    Library* first = g_first_device.load(std::memory_order_acquire);
    return first ? *first : g_devices[0];
}
// Binds the calling thread to a device context (and makes its HIP device current) for the life of the object.
// The thread's HIP device is put back as it was found, whether or not the thread had a binding before: a caller's thread
// that was never bound resolves to the first context (current_device) and must find that context's device current again.
struct DeviceBinding {
    Library* before;
    int device_before = -1;
    bool ok;
    explicit DeviceBinding(Library& lib) : before(t_device) {
        if (hipGetDevice(&device_before) != hipSuccess) device_before = -1;
        ok = hipSetDevice(lib.hip_device) == hipSuccess;
        t_device = &lib;
    }
    ~DeviceBinding() {
        t_device = before;
        const int back = before ? before->hip_device : device_before;
        if (back >= 0) (void)hipSetDevice(back);
    }
};

#ifndef RT_RAYS_PER_LANE
#define RT_RAYS_PER_LANE 2
#endif
constexpr int RAYS_PER_LANE = RT_RAYS_PER_LANE;   // brute-force scans over spheres and ellipsoids
// The staged triangle scans skip a stage only when every ray of the wave agrees, so fewer rays per wave skip more:
// one ray per lane (scene7 at 1280x720, closest hit 47.7 -> 49.5 % of the FP32 peak, shadow 37.4 -> 43.2 %; four: 41.3 / 28.2 %).
#ifndef RT_TRI_RAYS_PER_LANE
#define RT_TRI_RAYS_PER_LANE 1
#endif
constexpr int TRI_RAYS_PER_LANE = RT_TRI_RAYS_PER_LANE;
// The pre-test form reads three 16-byte words of LDS per step for 18 issue slots of arithmetic (rays with one origin): with one
// ray per lane the LDS, shared by the four SIMDs of a CU, is the limit; two rays per lane share each read.
#ifndef RT_PRETEST_RAYS_PER_LANE
#define RT_PRETEST_RAYS_PER_LANE 2
#endif
constexpr int PRETEST_RAYS_PER_LANE = RT_PRETEST_RAYS_PER_LANE;
#ifndef RT_PRETEST_SHADOW_RAYS_PER_LANE
#define RT_PRETEST_SHADOW_RAYS_PER_LANE 2
#endif
constexpr int PRETEST_SHADOW_RAYS_PER_LANE = RT_PRETEST_SHADOW_RAYS_PER_LANE;
#ifndef RT_SORT_RAYS_MIN_TRIANGLES
#define RT_SORT_RAYS_MIN_TRIANGLES 2048
#endif
#ifndef RT_SORT_SHADOW_ORIGINS
#define RT_SORT_SHADOW_ORIGINS 1
#endif
constexpr uint32_t SORT_RAYS_MIN_TRIANGLES = RT_SORT_RAYS_MIN_TRIANGLES;   // brute-force scans of fewer triangles take reflected rays in list order
// Rays per lane of the hierarchy scans.  A wave's life is a chain of dependent L2 round trips whose
// length grows with its number of rays, and a launch ends with its slowest wave: one ray per lane
// is best while the top level is short (scene7: 391 top nodes).  Walking a long top level once per
// wave costs more than that, so big scenes (the synthetic one: 3907 top nodes) take two, and start
// the walk one level higher, at the super nodes (synthetic scene: 593 -> 561 ms; scene7 would lose: 25 -> 30 ms).
constexpr uint32_t TOPS_FOR_DEEP_WALK = 1024;
// Kinds whose eye rays are walked once per wave (k_closest_packet): bit 0 triangles, 1 ellipsoids, 2 spheres.
// It pays where the leaf test needs little data (scene5, 1000 spheres, at 3840x2160: 26.4 -> 23.4 ms); for
// triangles the chain of dependent loads down to the 64-byte records of every leaf the wave can reach makes
// it lose (scene7: 24.8 -> 34.5 ms).  RT_AMD_EYE_PACKETS in the environment overrides it (tuning).
constexpr uint32_t EYE_PACKET_KINDS = 6;
bool packet_walk_for_eye_rays(int prim) {
    static const uint32_t kinds = [] { const char* e = std::getenv("RT_AMD_EYE_PACKETS"); return e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : EYE_PACKET_KINDS; }();
    return (kinds >> prim) & 1u;
}
// A kind with at most this many slots is scanned by brute force even in the default mode: the walk of
// a hierarchy costs more than it saves there (measured crossover: 32 to 64 slots for every kind;
// scene6, 40 + 8 + 8 slots and 1.15 G rays at 3840x2160: 208 ms walking, 141 ms brute force).
// RT_AMD_BRUTE_BELOW in the environment overrides it (tuning); RT_FLAG_WALK_ALL walks regardless.
constexpr uint32_t BRUTE_FORCE_BELOW = 64;
uint32_t brute_force_below() {
    static const uint32_t value = [] {
        const char* e = std::getenv("RT_AMD_BRUTE_BELOW");
        return e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : BRUTE_FORCE_BELOW;
    }();
    return value;
}
constexpr uint32_t DEFAULT_CHUNK_TILES = 32768;   // 134 M samples, 18.8 GB of ray state: a 3840x2160 frame in one chunk (long launches hide the tail of the slowest waves)
constexpr uint32_t MAX_BOUNCES = 62;
constexpr uint32_t MIN_LANE_TILES = 256;   // a frame is cut into chunks of at least this many tiles, one per lane

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

struct Rect { uint32_t x0, y0, w, h, tiles_x, tiles_y, tiles; };
Rect rect_of(const rt_render_params& p) {
    Rect r;
    if (p.rect_w) { r.x0 = p.rect_x0; r.y0 = p.rect_y0; r.w = p.rect_w; r.h = p.rect_h; }
    else { r.x0 = 0; r.y0 = 0; r.w = p.width; r.h = p.height; }
    r.tiles_x = ceil_div(r.w, RT_TILE);
    r.tiles_y = ceil_div(r.h, RT_TILE);
    r.tiles = r.tiles_x * r.tiles_y;
    return r;
}
uint32_t rank_tiles(const rt_render_params& p) {
    const Rect r = rect_of(p);
    const uint32_t world = p.tile_world ? p.tile_world : 1, rank = p.tile_rank;
    if (rank >= world) return 0;
    return (r.tiles + world - 1 - rank) / world;
}

}  // namespace

struct rt_scene {
    DevScene dev;
    int device = -1;                   // the HIP device these allocations live on
    rt_scene* next = nullptr;          // the replica on the next device of a single-process multi-GPU set-up (rt_init_devices)
    std::vector<void*> allocations;
    // bound of ray origins the box records were built for; grows (records rebuilt) when a camera's eye lies outside it.
    // Every frame reads the records under a shared lock for as long as it is traced; a rebuild takes the lock exclusively,
    // i.e. waits for the frames in flight on any slot, and is complete on the device before it lets go.
    mutable float box_bound = 0.0f;
    mutable std::shared_mutex boxes;
    bool directional_is_unit = true;   // |direction|^2 <= 1 + 20 * 2^-24: the shadow scans toward it may use the pre-tests
};

namespace {

// bytes of per-sample state for a chunk of `samples` samples (a multiple of 4096):
// 24 float arrays + hit_id + blocked + 2 lists + primary ids + three 64-bit keys = 35 words per sample
// + per 256 samples: 4 wave ballots and one offset (ordered compaction) + the block totals of the two-level scan
constexpr size_t state_bytes(size_t samples) { return samples * 35 * sizeof(float) + samples / BLOCK * 40 + SCAN_THREADS * sizeof(uint32_t); }

// Scratch of rt_render_tiles_noalloc: the per-sample state of one chunk of at most NOALLOC_MAX_TILES tiles, then 64 list
// lengths, then the work counters.  Device code and host code reach it only through the address
// rt_init obtains with hipGetSymbolAddress (Library::scratch): the name below is not a device address on the host side.
constexpr uint32_t NOALLOC_MAX_TILES = RT_NOALLOC_MAX_TILES;
constexpr size_t NOALLOC_STATE_BYTES = (state_bytes(static_cast<size_t>(NOALLOC_MAX_TILES) * SAMPLES_PER_TILE) + 255) / 256 * 256;
constexpr size_t NOALLOC_COUNTS_BYTES = 256;                                      // 64 words
constexpr size_t NOALLOC_STATS_BYTES = STAT_COPIES * ST_COUNT * sizeof(unsigned long long);
constexpr size_t NOALLOC_BYTES = NOALLOC_STATE_BYTES + NOALLOC_COUNTS_BYTES + NOALLOC_STATS_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

int ensure_state(Lane& lane, size_t samples) {
    if (lane.state_samples >= samples) return 0;
    if (lane.state_mem) { HIP_TRY(hipFree(lane.state_mem)); lane.state_mem = nullptr; lane.state_samples = 0; }
    HIP_TRY(hipMalloc(&lane.state_mem, state_bytes(samples)));
    lane.state_samples = samples;
    return 0;
}

State make_state(void* state_mem, uint32_t* counts, unsigned long long* stats, size_t samples, bool want_ids) {
    State st;
    float* f = static_cast<float*>(state_mem);
    float** fields[] = {&st.ox, &st.oy, &st.oz, &st.dx, &st.dy, &st.dz, &st.wr, &st.wg, &st.wb, &st.cr, &st.cg, &st.cb,
                        &st.px, &st.py, &st.pz, &st.nx, &st.ny, &st.nz, &st.nox, &st.noy, &st.noz, &st.ndx, &st.ndy, &st.ndz};
    size_t k = 0;
    for (float** field : fields) *field = f + samples * k++;
    uint32_t* u = reinterpret_cast<uint32_t*>(f + samples * k);
    st.hit_id = u; st.blocked = u + samples; st.list[0] = u + 2 * samples; st.list[1] = u + 3 * samples;
    st.primary_id = want_ids ? u + 4 * samples : nullptr;
    st.key = reinterpret_cast<unsigned long long*>(u + 5 * samples);   // 8-byte aligned: samples is a multiple of 4096
    st.key_ell = st.key + samples;
    st.key_sph = st.key_ell + samples;
    st.flag_bits = st.key_sph + samples;                               // 16-byte aligned likewise
    st.wg_offset = reinterpret_cast<uint32_t*>(st.flag_bits + samples / BLOCK * 4);
    st.scan_block = st.wg_offset + samples / BLOCK + samples / BLOCK % 2;   // after the offsets (8 bytes of the 40 per workgroup are spare)
    st.counts = counts;
    st.stats = stats;
    return st;
}

struct Timer {
    // pairs of events around launches of one kernel family; resolved after the stream is idle
    // families: 0 closest-hit scans of a bounce (all kinds), 1 shadow scans of a light group (all kinds),
    // 2 + kind: one kind's closest-hit scan, 5 + kind: one kind's shadow scan (kind: 0 triangles, 1 spheres, 2 ellipsoids)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans[8];
    FrameSlot* owner = nullptr;  // whose pool of events is used
    size_t next_event = 0;
    std::mutex guard;   // the lanes of a frame are driven by a host thread each
    hipEvent_t take() {
        if (next_event == owner->events.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            owner->events.push_back(e);
        }
        return owner->events[next_event++];
    }
};

template <typename Launch>
void timed(Timer* timer, int family, hipStream_t stream, Launch&& launch) {
    if (!timer) { launch(); return; }
    hipEvent_t a, b;
    { std::lock_guard<std::mutex> lock(timer->guard); a = timer->take(); b = timer->take(); }
    if (a) (void)hipEventRecord(a, stream);
    launch();
    if (b) (void)hipEventRecord(b, stream);
    if (a && b) { std::lock_guard<std::mutex> lock(timer->guard); timer->spans[family].push_back({a, b}); }
}

// event number i of a lane's fork/join pool
hipEvent_t sync_event(Lane& lane, size_t i) {
    while (lane.sync_events.size() <= i) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        lane.sync_events.push_back(e);
    }
    return lane.sync_events[i];
}

// (Re)builds the box records of every kind for ray origins within `bound` (max norm) of 0.
void build_boxes(const rt_scene* scene, float bound, hipStream_t stream) {
    const DevScene& sc = scene->dev;
    for (int prim = 0; prim < 3; ++prim) {
        const Hierarchy& h = prim == 0 ? sc.th : (prim == 1 ? sc.eh : sc.sh);
        if (h.n == 0) continue;
        float4* leaf = const_cast<float4*>(h.leaf_box);
        float4* cl = const_cast<float4*>(h.cl_box);
        float4* top = const_cast<float4*>(h.top_box);
        float4* super = const_cast<float4*>(h.super_box);
        float4* root = const_cast<float4*>(h.root_box);
        const dim3 leaf_grid(ceil_div(h.n, BLOCK)), cl_grid(ceil_div(h.n_clusters, BLOCK)), top_grid(ceil_div(h.n_tops, BLOCK)), super_grid(ceil_div(h.n_supers, BLOCK)), root_grid(ceil_div(h.n_roots, BLOCK));
        if (prim == 0) hipLaunchKernelGGL(k_prep_tri_box, leaf_grid, dim3(BLOCK), 0, stream, sc.tri, h.k_of, h.n, bound, leaf);
        else if (prim == 1) hipLaunchKernelGGL(k_prep_ell_box, leaf_grid, dim3(BLOCK), 0, stream, sc.ell, h.k_of, h.n, bound, leaf);
        else hipLaunchKernelGGL(k_prep_sph_box, leaf_grid, dim3(BLOCK), 0, stream, sc.sph, h.k_of, h.n, bound, leaf);
        hipLaunchKernelGGL(k_prep_cluster_box, cl_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(leaf), h.n, h.n_clusters, cl);
        hipLaunchKernelGGL(k_prep_cluster_box, top_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(cl), h.n_clusters, h.n_tops, top);
        hipLaunchKernelGGL(k_prep_cluster_box, super_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(top), h.n_tops, h.n_supers, super);
        hipLaunchKernelGGL(k_prep_packed_children, cl_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(leaf), h.n, h.n_clusters, const_cast<uint4*>(h.cl_nodes));
        hipLaunchKernelGGL(k_prep_packed_children, top_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(cl), h.n_clusters, h.n_tops, const_cast<uint4*>(h.top_nodes));
        hipLaunchKernelGGL(k_prep_packed_children, super_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(top), h.n_tops, h.n_supers, const_cast<uint4*>(h.super_nodes));
        hipLaunchKernelGGL(k_prep_cluster_box, root_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(super), h.n_supers, h.n_roots, root);
        hipLaunchKernelGGL(k_prep_packed_children, root_grid, dim3(BLOCK), 0, stream, static_cast<const float4*>(super), h.n_supers, h.n_roots, const_cast<uint4*>(h.root_nodes));
        hipLaunchKernelGGL(k_prep_packed_children, dim3(ceil_div(h.n_apex, BLOCK)), dim3(BLOCK), 0, stream, static_cast<const float4*>(root), h.n_roots, h.n_apex, const_cast<uint4*>(h.apex_nodes));
    }
    scene->box_bound = bound;
}
const Hierarchy& hierarchy_of(const DevScene& sc, int prim) { return prim == 0 ? sc.th : (prim == 1 ? sc.eh : sc.sh); }
// box-hierarchy scans of one kind, one or two rays per lane
// box-hierarchy scans of one kind.  `deep` (more than TOPS_FOR_DEEP_WALK top nodes): two rays per lane and the walk
// starts at the super nodes; otherwise one ray per lane from the top nodes.
#ifndef RT_DFS_MAX_ROUNDS
#define RT_DFS_MAX_ROUNDS 32
#endif
#ifndef RT_DFS_FILL_BLOCKS
#define RT_DFS_FILL_BLOCKS (256 * 6 * 4)
#endif
// rays per lane over a wave's life, at most (8: 208 ms, 16: 199 ms, 32: 195.5 ms on the 1 M-triangle scene); workgroups per round:
// a launch of fewer keeps one ray per lane, so that it still fills the chip
constexpr uint32_t DFS_MAX_ROUNDS = RT_DFS_MAX_ROUNDS, DFS_FILL_BLOCKS = RT_DFS_FILL_BLOCKS;
template <int PRIM>
void launch_closest_scan(bool deep, dim3 grid, hipStream_t on, const DevScene& sc, const State& sb, uint32_t b) {
    if (deep) hipLaunchKernelGGL((k_closest_scan<DEEP_RAYS, 4, PRIM>), grid, dim3(BLOCK), 0, on, sc, sb, b);
    else hipLaunchKernelGGL((k_closest_scan<1, 3, PRIM>), grid, dim3(BLOCK), 0, on, sc, sb, b);
}
template <int PRIM>
void launch_shadow_scan(bool deep, bool depth_first, dim3 grid, uint32_t ray_blocks, hipStream_t on, const DevScene& sc, const State& sb, uint32_t b, uint32_t first_light, uint32_t group_first) {
    // Large hierarchies: each lane walks the whole hierarchy for its ray, depth first; a wave takes 64 * rounds rays, and
    // there are enough workgroups to fill the chip several times over.  (1 M-triangle scene, 92 % of the shadow rays
    // blocked: 262 -> 199 ms per frame.  Small hierarchies keep the queued walk, whose first level is shared by the
    // wave: scene7, 19 % blocked, 16.9 ms against 17.6.)
    // The other kinds of such a scene follow the triangles' scan and find most rays blocked already: walked depth first
    // as well, their waves fill up with the rays that are still open (204 -> 199 ms).
    if (RT_SHADOW_DFS && depth_first) {
        const uint32_t rounds = std::min(DFS_MAX_ROUNDS, std::max(1u, ray_blocks * grid.y / DFS_FILL_BLOCKS));
        hipLaunchKernelGGL(k_shadow_dfs<PRIM>, dim3(ceil_div(ray_blocks, rounds), grid.y), dim3(BLOCK), 0, on, sc, sb, b, first_light, group_first, rounds);
        return;
    }
    if (deep) hipLaunchKernelGGL((k_shadow_scan<DEEP_RAYS, 4, PRIM>), grid, dim3(BLOCK), 0, on, sc, sb, b, first_light, group_first);
    else hipLaunchKernelGGL((k_shadow_scan<1, 3, PRIM>), grid, dim3(BLOCK), 0, on, sc, sb, b, first_light, group_first);
}

int render_impl(const rt_scene* scene, const rt_camera_desc* cam, const rt_render_params* prm, uint8_t* tiles_dev,
                uint32_t* ids_dev /* rect-shaped, nullable */, rt_stats* stats, bool no_alloc = false, int slot_index = 0) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    DeviceBinding here(lib);      // a caller's thread that has never made this device current (HIP's current device is per thread)
    if (!here.ok) return rt_fail(RT_ERR_DEVICE, "hipSetDevice failed");
    FrameSlot& slot = lib.slots[slot_index];
    std::lock_guard<std::mutex> one_at_a_time(slot.guard);
    if (!scene || !cam || !prm) return rt_fail(RT_ERR_ARGUMENT, "null argument");
    if (prm->width == 0 || prm->height == 0) return rt_fail(RT_ERR_ARGUMENT, "image size is zero");
    if (prm->max_bounce_count < 0 || static_cast<uint32_t>(prm->max_bounce_count) > MAX_BOUNCES) return rt_fail(RT_ERR_LIMIT, "max_bounce_count out of range (0..62)");
    const Rect rect = rect_of(*prm);
    if (rect.x0 + rect.w > prm->width || rect.y0 + rect.h > prm->height || rect.w == 0 || rect.h == 0) return rt_fail(RT_ERR_ARGUMENT, "rectangle outside the image");
    const uint32_t world = prm->tile_world ? prm->tile_world : 1;
    if (prm->tile_rank >= world) return rt_fail(RT_ERR_ARGUMENT, "tile_rank >= tile_world");
    const uint32_t my_tiles = rank_tiles(*prm);
    uint32_t chunk_tiles = prm->chunk_tiles ? prm->chunk_tiles : DEFAULT_CHUNK_TILES;
    if (chunk_tiles > 65536) chunk_tiles = 65536;
    if (chunk_tiles > my_tiles) chunk_tiles = my_tiles ? my_tiles : 1;
    // unless the caller fixed the chunk size, a frame that would be one chunk is cut into one chunk per lane
    // (not in brute-force mode, whose launches are timed one by one)
    if (prm->chunk_tiles == 0 && (prm->flags & RT_FLAG_NO_FILTER) == 0 && my_tiles <= chunk_tiles && my_tiles >= 2 * MIN_LANE_TILES) chunk_tiles = ceil_div(my_tiles, std::min(MAX_LANES, my_tiles / MIN_LANE_TILES));
    if (no_alloc) {   // one chunk on one lane, its state in the scratch array: nothing is allocated, created or freed below
        if (my_tiles > NOALLOC_MAX_TILES) return rt_fail(RT_ERR_LIMIT, "rt_render_tiles_noalloc: more than " RT_STRINGIFY(RT_NOALLOC_MAX_TILES) " tiles of 16x16 pixels for this rank");
        chunk_tiles = my_tiles ? my_tiles : 1;
    }
    const size_t chunk_samples = static_cast<size_t>(chunk_tiles) * SAMPLES_PER_TILE;
    const uint32_t n_lanes = std::min(MAX_LANES, ceil_div(my_tiles, chunk_tiles));
    if (!no_alloc) for (uint32_t l = 0; l < n_lanes; ++l) if (int rc = ensure_state(slot.lanes[l], chunk_samples)) return rc;
    unsigned long long* const stats_dev = no_alloc ? reinterpret_cast<unsigned long long*>(lib.scratch + NOALLOC_STATE_BYTES + NOALLOC_COUNTS_BYTES) : slot.stats;
    auto counts_of = [&](uint32_t l) { return no_alloc ? reinterpret_cast<uint32_t*>(lib.scratch + NOALLOC_STATE_BYTES) : slot.lanes[l].counts; };

    while (scene && scene->device != lib.device) scene = scene->next;   // the replica on this device
    if (!scene) return rt_fail(RT_ERR_ARGUMENT, "the scene has no replica on this device");
    DevScene sc = scene->dev;
    hipStream_t stream = slot.lanes[0].stream;
    std::shared_lock<std::shared_mutex> reading(scene->boxes, std::defer_lock);   // held until this frame's launches have completed
    {   // the box records must hold for the eye as an origin too
        float need = sc.origin_bound;
        for (int i = 0; i < 3; ++i) need = std::fmax(need, 1.001f * std::fabs(cam->eye[i]) + 1.0e-3f);
        if (!std::isfinite(need)) return rt_fail(RT_ERR_ARGUMENT, "camera eye is not finite");
        for (;;) {
            reading.lock();
            if (need <= scene->box_bound) break;
            reading.unlock();
            std::unique_lock<std::shared_mutex> rebuilding(scene->boxes);
            if (!(need <= scene->box_bound)) {
                build_boxes(scene, need, stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(stream));
            }
        }
        sc.box_bound = scene->box_bound;
    }
    Frame f;
    std::memcpy(f.eye, cam->eye, sizeof f.eye);
    std::memcpy(f.bi, cam->basis_i, sizeof f.bi);
    std::memcpy(f.bj, cam->basis_j, sizeof f.bj);
    std::memcpy(f.bk, cam->basis_k, sizeof f.bk);
    std::memcpy(f.hw, cam->half_world, sizeof f.hw);
    f.half_w_px = 0.5f * static_cast<float>(prm->width);      // render_work_queue.cpp:32-35
    f.half_h_px = 0.5f * static_cast<float>(prm->height);
    f.rect_x0 = rect.x0; f.rect_y0 = rect.y0; f.rect_w = rect.w; f.rect_h = rect.h;
    f.tiles_x = rect.tiles_x; f.tile_rank = prm->tile_rank; f.tile_world = world;

    State lane_state[MAX_LANES];
    for (uint32_t l = 0; l < n_lanes; ++l)
        lane_state[l] = make_state(no_alloc ? static_cast<void*>(lib.scratch) : slot.lanes[l].state_mem, counts_of(l), stats_dev, chunk_samples, ids_dev != nullptr);
    HIP_TRY(hipMemsetAsync(stats_dev, 0, STAT_COPIES * ST_COUNT * sizeof(unsigned long long), stream));

    Timer timer_storage;
    timer_storage.owner = &slot;
    Timer* timer = stats ? &timer_storage : nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    if (timer) { ev_begin = timer->take(); ev_end = timer->take(); if (ev_begin) HIP_TRY(hipEventRecord(ev_begin, stream)); }

    const uint32_t depth = static_cast<uint32_t>(prm->max_bounce_count);
    const bool filter = (prm->flags & RT_FLAG_NO_FILTER) == 0;
    const bool plain_test = (prm->flags & RT_FLAG_PLAIN_TEST) != 0;
    const bool exact_stages = (prm->flags & RT_FLAG_EXACT_STAGES) != 0;
    // per kind (0 triangles, 1 ellipsoids, 2 spheres): walk its hierarchy, or scan it by brute force
    const uint32_t below = (prm->flags & RT_FLAG_WALK_ALL) ? 0u : brute_force_below();
    const bool force_deep = (prm->flags & RT_FLAG_DEEP_WALK) != 0;
    const bool walk[3] = {filter && sc.th.n > below, filter && sc.eh.n > below, filter && sc.sh.n > below};
    if (n_lanes > 1) {   // the other lanes start when the box records and the cleared counters are ready
        HIP_TRY(hipEventRecord(slot.lanes_fork, slot.lanes[0].stream));
        for (uint32_t l = 1; l < n_lanes; ++l) HIP_TRY(hipStreamWaitEvent(slot.lanes[l].stream, slot.lanes_fork, 0));
    }
    // One chunk; every launch of it goes to the streams of its lane.
    // The lanes start one after the other: lane l + 1 begins when lane l has generated its rays, so that
    // the short launches between the scans of one lane fall beside the scans of the other (scene7: 20.5 -> 20.1 ms;
    // starting it after the first scans of lane l gains nothing).
    std::atomic<uint32_t> started_count{0};   // lanes that have recorded (or given up recording) their lanes_started event
    auto run_chunk = [&](uint32_t first, uint32_t chunk) -> int {
        Lane& lane = slot.lanes[chunk % n_lanes];
        const hipStream_t stream = lane.stream;       // everything of this chunk goes to its lane
        if (chunk > 0 && chunk < n_lanes) {
            while (started_count.load(std::memory_order_acquire) < chunk) std::this_thread::yield();   // the event must have been recorded
            HIP_TRY(hipStreamWaitEvent(stream, slot.lanes_started[chunk - 1], 0));
        }
        const State st = lane_state[chunk % n_lanes];
        const uint32_t n_tiles = (my_tiles - first < chunk_tiles) ? my_tiles - first : chunk_tiles;
        const uint32_t n_samples = n_tiles * SAMPLES_PER_TILE;
        HIP_TRY(hipMemsetAsync(st.counts, 0, 64 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_primary, dim3(n_samples / BLOCK), dim3(BLOCK), 0, stream, f, sc, st, first);
        const uint32_t scan_grid = ceil_div(n_samples / BLOCK, SCAN_THREADS);
        hipLaunchKernelGGL(k_scan_flags, dim3(scan_grid), dim3(SCAN_THREADS), 0, stream, st, static_cast<const uint32_t*>(nullptr), n_samples);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_THREADS), 0, stream, st, static_cast<const uint32_t*>(nullptr), n_samples, st.counts, -1, static_cast<int>(ST_LIVE));
        hipLaunchKernelGGL(k_fill_list, dim3(n_samples / BLOCK), dim3(BLOCK), 0, stream, st, static_cast<const uint32_t*>(nullptr), n_samples,
                           static_cast<const uint32_t*>(nullptr), st.list[0]);
        // The host follows the length of the chunk's ray list, one word per bounce over a pinned buffer, and
        // sizes the grids by it: a grid sized for the chunk's capacity is mostly workgroups that find nothing
        // to do (scene7: two thirds of the samples miss the scene box, one ray in five is reflected), and
        // dispatching those cost a tenth of the frame.  `bound` is what the host knows: the exact length for
        // the first bounce (one wait on an idle lane per chunk), the length of the bounce before for the
        // others (that word arrives while the scans of the bounce before are still queued: no idle gap).
        // A list can only shrink, so an older length is a valid bound; a chunk whose list is empty stops.
        // (Waiting for the exact length of every bounce instead, on an idle lane each time, measured the same.)
        uint32_t bound = n_samples;
        auto fetch_count = [&](uint32_t b) -> int {
            HIP_TRY(hipMemcpyAsync(lane.host_counts + b, st.counts + b, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(lane.count_ready, stream));
            return 0;
        };
        auto wait_count = [&](uint32_t b) -> int {
            HIP_TRY(hipEventSynchronize(lane.count_ready));
            bound = std::min(bound, lane.host_counts[b]);
            return 0;
        };
        if (int rc = fetch_count(0)) return rc;
        if (chunk + 1 < n_lanes) {
            const hipError_t recorded = hipEventRecord(slot.lanes_started[chunk], stream);
            started_count.store(chunk + 1, std::memory_order_release);   // whatever happened: the next lane's thread is waiting for this
            HIP_TRY(recorded);
        }
        if (int rc = wait_count(0)) return rc;
        auto ray_blocks = [&] { return ceil_div(std::max(bound, 1u), BLOCK); };
        auto scan_blocks = [&] { return ceil_div(std::max(bound, 1u), BLOCK * RAYS_PER_LANE); };
        auto tri_scan_blocks = [&] { return ceil_div(std::max(bound, 1u), BLOCK * TRI_RAYS_PER_LANE); };
        auto blocks_for = [&](int rays_per_lane) {   // ray blocks, and room for the split of a small launch
            const uint32_t blocks = ceil_div(std::max(bound, 1u), static_cast<uint32_t>(BLOCK * rays_per_lane));
            return blocks > SEG_TARGET ? blocks : SEG_TARGET;
        };
        // state as bounce b sees it: the two ray sets swap roles every bounce
        auto state_of = [&](uint32_t b) {
            State sb = st;
            if (b & 1u) {
                std::swap(sb.ox, sb.nox); std::swap(sb.oy, sb.noy); std::swap(sb.oz, sb.noz);
                std::swap(sb.dx, sb.ndx); std::swap(sb.dy, sb.ndy); std::swap(sb.dz, sb.ndz);
            }
            return sb;
        };
        // closest-hit scans of bounce b, all kinds, on stream `on`
        auto closest_scans = [&](uint32_t b, hipStream_t on) {
            const State sb = state_of(b);
            auto hierarchy_scan = [&](int prim) {   // one or two rays per lane, three or four levels (TOPS_FOR_DEEP_WALK)
                if (b == 0 && packet_walk_for_eye_rays(prim)) {   // rays from the eye: one walk per wave
                    const dim3 grid(blocks_for(1));
                    if (prim == 0) hipLaunchKernelGGL(k_closest_packet<0>, grid, dim3(BLOCK), 0, on, sc, sb, b);
                    else if (prim == 1) hipLaunchKernelGGL(k_closest_packet<1>, grid, dim3(BLOCK), 0, on, sc, sb, b);
                    else hipLaunchKernelGGL(k_closest_packet<2>, grid, dim3(BLOCK), 0, on, sc, sb, b);
                    return;
                }
                const bool deep = force_deep || hierarchy_of(sc, prim).n_tops > TOPS_FOR_DEEP_WALK;
                const dim3 grid(blocks_for(deep ? DEEP_RAYS : 1));
                if (prim == 0) launch_closest_scan<0>(deep, grid, on, sc, sb, b);
                else if (prim == 1) launch_closest_scan<1>(deep, grid, on, sc, sb, b);
                else launch_closest_scan<2>(deep, grid, on, sc, sb, b);
            };
            timed(timer, 0, on, [&] {
                if (sc.n_tri) timed(timer, 2, on, [&] {
                    if (walk[0]) hierarchy_scan(0);
                    else if (plain_test) hipLaunchKernelGGL((k_closest_tri_brute<TRI_RAYS_PER_LANE, 0>), dim3(tri_scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b, static_cast<const uint32_t*>(sb.list[b & 1u]));
                    else {
                        // reflected rays of a long scan are taken in the order of their directions (launches follow each other on one stream in this mode;
                        // the other list buffer is free until this bounce's compaction, the scan words between two compactions)
                        const uint32_t* list = sb.list[b & 1u];
                        if (!filter && b >= 1 && sc.n_tri >= SORT_RAYS_MIN_TRIANGLES) {
                            uint32_t* counts = sb.scan_block, *cursors = sb.scan_block + DIR_BINS;
                            (void)hipMemsetAsync(counts, 0, DIR_BINS * sizeof(uint32_t), on);
                            hipLaunchKernelGGL(k_bin_count<0>, dim3(ray_blocks()), dim3(BLOCK), 0, on, sc, sb, b, counts);
                            hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(DIR_BINS), 0, on, static_cast<const uint32_t*>(counts), cursors);
                            hipLaunchKernelGGL(k_bin_scatter<0>, dim3(ray_blocks()), dim3(BLOCK), 0, on, sc, sb, b, cursors, sb.list[(b + 1) & 1u]);
                            list = sb.list[(b + 1) & 1u];
                        }
                        if (exact_stages) hipLaunchKernelGGL((k_closest_tri_brute<TRI_RAYS_PER_LANE, 1>), dim3(tri_scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b, list);
                        else hipLaunchKernelGGL((k_closest_tri_brute<PRETEST_RAYS_PER_LANE, 2>), dim3(ceil_div(std::max(bound, 1u), BLOCK * PRETEST_RAYS_PER_LANE)), dim3(BLOCK), 0, on, sc, sb, b, list);
                    }
                });
                if (sc.n_sph) timed(timer, 3, on, [&] {
                    if (!walk[2] && b == 0 && !plain_test) hipLaunchKernelGGL(k_closest_sph_eye<RAYS_PER_LANE>, dim3(scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b);
                    else if (!walk[2]) hipLaunchKernelGGL(k_closest_sph<RAYS_PER_LANE>, dim3(scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b);
                    else hierarchy_scan(2);
                });
                if (sc.n_ell) timed(timer, 4, on, [&] {
                    if (!walk[1] && plain_test) hipLaunchKernelGGL(k_closest_ell<RAYS_PER_LANE>, dim3(scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b);
                    else if (!walk[1]) hipLaunchKernelGGL(k_closest_ell_staged<RAYS_PER_LANE>, dim3(scan_blocks()), dim3(BLOCK), 0, on, sc, sb, b);
                    else hierarchy_scan(1);
                });
            });
        };
        // box-hierarchy shadow scan of one kind toward the n lights first_light..
        auto shadow_scan = [&](const State& sb, int prim, uint32_t n, uint32_t b, uint32_t first_light, uint32_t group_first) {
            const bool deep = force_deep || hierarchy_of(sc, prim).n_tops > TOPS_FOR_DEEP_WALK;
            const bool depth_first = deep || (walk[0] && sc.th.n_tops > TOPS_FOR_DEEP_WALK);
            const dim3 grid(blocks_for(deep ? DEEP_RAYS : 1), n);
            if (prim == 0) launch_shadow_scan<0>(deep, depth_first, grid, ray_blocks(), stream, sc, sb, b, first_light, group_first);
            else if (prim == 1) launch_shadow_scan<1>(deep, depth_first, grid, ray_blocks(), stream, sc, sb, b, first_light, group_first);
            else launch_shadow_scan<2>(deep, depth_first, grid, ray_blocks(), stream, sc, sb, b, first_light, group_first);
        };

        // Per bounce: closest-hit scans -> k_hit (which also writes the reflected ray) -> list compaction ->
        // shadow scans + k_shade.  The closest-hit scans of bounce b+1 need only the reflected rays and the
        // list, so they run on the side stream beside the shadow scans of bounce b (their tails overlap);
        // k_hit of bounce b+1 waits for both.
        if (depth && bound) closest_scans(0, stream);
        for (uint32_t b = 0; b < depth && bound; ++b) {
            const State sb = state_of(b);
            hipLaunchKernelGGL(k_hit, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sc, sb, b);
            hipLaunchKernelGGL(k_scan_flags, dim3(scan_grid), dim3(SCAN_THREADS), 0, stream, sb, static_cast<const uint32_t*>(sb.counts + b), 0u);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_THREADS), 0, stream, sb, static_cast<const uint32_t*>(sb.counts + b), 0u, sb.counts + b + 1,
                               static_cast<int>(ST_CLOSEST_RAYS), static_cast<int>(ST_HITS));
            hipLaunchKernelGGL(k_fill_list, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sb, static_cast<const uint32_t*>(sb.counts + b), 0u,
                               static_cast<const uint32_t*>(sb.list[b & 1u]), sb.list[(b + 1) & 1u]);
            if (int rc = fetch_count(b + 1)) return rc;
            const bool overlap = filter && b + 1 < depth && !no_alloc;   // brute force is timed launch by launch: no overlap there; the fork / join events are created on demand
            hipEvent_t forked = overlap ? sync_event(lane, 2 * b) : nullptr, joined = overlap ? sync_event(lane, 2 * b + 1) : nullptr;
            if (overlap && forked && joined) {
                HIP_TRY(hipEventRecord(forked, stream));
                HIP_TRY(hipStreamWaitEvent(lane.side_stream, forked, 0));
                closest_scans(b + 1, lane.side_stream);
                HIP_TRY(hipEventRecord(joined, lane.side_stream));
            }
            bool hits_in_cell_order = false;     // this bounce's hit points have been copied in the order of their cells
            if (sc.n_lights == 0) {
                hipLaunchKernelGGL(k_shade, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sc, sb, b, 0u, 0u, 1u);
            } else {
                for (uint32_t l0 = 0; l0 < sc.n_lights; l0 += 32) {
                    const uint32_t group = (sc.n_lights - l0 < 32) ? sc.n_lights - l0 : 32;
                    // the directional light, if it is in this group, is light 0
                    const uint32_t n_dir = (l0 == 0 && sc.has_directional) ? 1u : 0u;
                    const uint32_t n_pt = group - n_dir;
                    // brute-force shadow scan of one kind toward lights first.. (n of them)
                    auto brute_force = [&](int prim, uint32_t n, uint32_t first_light) {
                        const dim3 grid(prim == 0 ? ray_blocks() : scan_blocks(), n);    // the triangle scan takes one ray per lane (and cuts the list into ranges of its own: a workgroup beyond the last range returns)
                        if (prim == 0 && plain_test) hipLaunchKernelGGL((k_shadow_tri_brute<0>), grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0, static_cast<const uint32_t*>(sb.list[(b + 1) & 1u]));
                        else if (prim == 0) {
                            // a long scan takes the hit points in the order of their cells (the list buffer of this bounce's rays is free after its compaction)
                            const uint32_t* list = sb.list[(b + 1) & 1u];
                            if (RT_SORT_SHADOW_ORIGINS && !filter && sc.n_tri >= SORT_RAYS_MIN_TRIANGLES) {
                                if (!hits_in_cell_order) {
                                    uint32_t* counts = sb.scan_block, *cursors = sb.scan_block + DIR_BINS;
                                    (void)hipMemsetAsync(counts, 0, DIR_BINS * sizeof(uint32_t), stream);
                                    hipLaunchKernelGGL(k_bin_count<1>, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sc, sb, b, counts);
                                    hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(DIR_BINS), 0, stream, static_cast<const uint32_t*>(counts), cursors);
                                    hipLaunchKernelGGL(k_bin_scatter<1>, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sc, sb, b, cursors, sb.list[b & 1u]);
                                    hits_in_cell_order = true;
                                }
                                list = sb.list[b & 1u];
                            }
                            if (exact_stages) hipLaunchKernelGGL((k_shadow_tri_brute<1>), grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0, list);
                            // two rays per lane when the launch is long enough for every workgroup to refill its lanes many times over (ranges of 16
                            // capacities); a short launch keeps one ray per lane, whose 256-ray workgroups fill and agree better (scene7 crop: 0.78 against 0.63)
                            else if (PRETEST_SHADOW_RAYS_PER_LANE == 2 && static_cast<unsigned long long>(ray_blocks()) * n >= 16ull * SHADOW_MIN_RANGES)
                                hipLaunchKernelGGL(k_shadow_tri_pretest<2>, grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0, list);
                            else hipLaunchKernelGGL(k_shadow_tri_pretest<1>, grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0, list);
                        }
                        else if (prim == 1 && plain_test) hipLaunchKernelGGL(k_shadow_ell<RAYS_PER_LANE>, grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0);
                        else if (prim == 2 && plain_test) hipLaunchKernelGGL(k_shadow_sph<RAYS_PER_LANE>, grid, dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0);
                        // (these two cut the ray list into ranges of their own, like the triangle scan: a workgroup beyond the last range returns)
                        else if (prim == 1) hipLaunchKernelGGL((k_shadow_brute<RAYS_PER_LANE, 1>), dim3(ray_blocks(), n), dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0);
                        else hipLaunchKernelGGL((k_shadow_brute<RAYS_PER_LANE, 2>), dim3(ray_blocks(), n), dim3(BLOCK), 0, stream, sc, sb, b, first_light, l0);
                    };
                    timed(timer, 1, stream, [&] {
                        const uint32_t count[3] = {sc.n_tri, sc.n_ell, sc.n_sph};
                        const int family_of[3] = {5, 7, 6};   // prim (0 triangles, 1 ellipsoids, 2 spheres) -> timer family (triangles, spheres, ellipsoids)
                        for (int prim = 0; prim < 3; ++prim) {
                            if (count[prim] == 0) continue;
                            timed(timer, family_of[prim], stream, [&] {
                                if (!walk[prim]) {
                                    brute_force(prim, group, l0);
                                } else if (n_dir && !scene->directional_is_unit) {
                                    // the pre-tests need a direction of unit length; the directional light's is the caller's
                                    brute_force(prim, 1, l0);
                                    if (n_pt) shadow_scan(sb, prim, n_pt, b, l0 + 1, l0);
                                } else {
                                    shadow_scan(sb, prim, group, b, l0, l0);
                                }
                            });
                        }
                    });
                    hipLaunchKernelGGL(k_shade, dim3(ray_blocks()), dim3(BLOCK), 0, stream, sc, sb, b, l0, group, (l0 + group == sc.n_lights) ? 1u : 0u);
                }
            }
            if (overlap && forked && joined) HIP_TRY(hipStreamWaitEvent(stream, joined, 0));
            else if (b + 1 < depth) closest_scans(b + 1, stream);
            if (b + 1 < depth) if (int rc = wait_count(b + 1)) return rc;
        }
        hipLaunchKernelGGL(k_resolve, dim3(n_tiles), dim3(BLOCK), 0, stream, st, first, tiles_dev);
        if (ids_dev) hipLaunchKernelGGL(k_scatter_ids, dim3(n_samples / BLOCK), dim3(BLOCK), 0, stream, st.primary_id, first, n_samples, f, ids_dev);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    // Chunk c belongs to lane c % n_lanes.  A lane waits for its own counts, so each is driven by a host
    // thread of its own: the caller's thread takes lane 0.
    const uint32_t n_chunks = ceil_div(my_tiles, chunk_tiles);
    auto run_lane = [&](uint32_t l) -> int {
        int rc = 0;
        for (uint32_t chunk = l; chunk < n_chunks && rc == 0; chunk += n_lanes) rc = run_chunk(chunk * chunk_tiles, chunk);
        // a lane that failed before it recorded its event must not keep the next one waiting (lanes start in order, so l lanes have started)
        uint32_t seen = started_count.load(std::memory_order_acquire);
        while (seen < l + 1 && !started_count.compare_exchange_weak(seen, l + 1, std::memory_order_acq_rel)) {}
        return rc;
    };
    int lane_rc[MAX_LANES] = {};
    std::string lane_error[MAX_LANES];
    {
        std::vector<std::thread> helpers;
        for (uint32_t l = 1; l < n_lanes; ++l)
            helpers.emplace_back([&, l] {
                DeviceBinding bound(lib);
                lane_rc[l] = bound.ok ? run_lane(l) : rt_fail(RT_ERR_DEVICE, "hipSetDevice failed in the thread of a lane");
                if (lane_rc[l]) lane_error[l] = g_last_error;   // the message is per thread
            });
        lane_rc[0] = run_lane(0);
        for (std::thread& t : helpers) t.join();
    }
    // A lane that failed leaves launches of the other lanes in flight, writing the caller's buffers: drain the device before returning.
    for (uint32_t l = 0; l < n_lanes; ++l) {
        if (!lane_rc[l]) continue;
        if (l == 0) lane_error[0] = g_last_error;
        (void)hipDeviceSynchronize();
        return rt_fail(lane_rc[l], lane_error[l].c_str());
    }
    for (uint32_t l = 1; l < n_lanes; ++l) {
        HIP_TRY(hipEventRecord(slot.lanes_join[l], slot.lanes[l].stream));
        HIP_TRY(hipStreamWaitEvent(stream, slot.lanes_join[l], 0));
    }
    if (timer && ev_end) HIP_TRY(hipEventRecord(ev_end, stream));
    HIP_TRY(hipStreamSynchronize(stream));

    if (stats) {
        unsigned long long copies[STAT_COPIES * ST_COUNT], h[ST_COUNT] = {};
        HIP_TRY(hipMemcpy(copies, stats_dev, sizeof copies, hipMemcpyDeviceToHost));
        for (int c = 0; c < STAT_COPIES; ++c) for (int k = 0; k < ST_COUNT; ++k) h[k] += copies[c * ST_COUNT + k];
        std::memset(stats, 0, sizeof *stats);
        // samples inside the rectangle, from the tile geometry; the bounding box culled those that are not live
        unsigned long long inside = 0;
        for (uint32_t t = prm->tile_rank; t < rect.tiles; t += world) {
            const uint32_t tx = tile_column(t, rect.tiles_x), ty = tile_row(t, rect.tiles_x);
            const uint32_t w = rect.w - tx * RT_TILE < RT_TILE ? rect.w - tx * RT_TILE : RT_TILE;
            const uint32_t hh = rect.h - ty * RT_TILE < RT_TILE ? rect.h - ty * RT_TILE : RT_TILE;
            inside += static_cast<unsigned long long>(w) * hh * RT_SAMPLES_PER_PIXEL;
        }
        stats->samples = inside; stats->culled = inside - h[ST_LIVE];
        stats->closest_rays = h[ST_CLOSEST_RAYS]; stats->hits = h[ST_HITS];
        stats->shadow_rays = h[ST_SHADOW_RAYS]; stats->shadow_blocked = h[ST_SHADOW_BLOCKED];
        const uint64_t real_counts[3] = {sc.n_tri_mat, sc.n_sph_mat, sc.n_ell_mat};
        for (int k = 0; k < 3; ++k) stats->closest_tests[k] = stats->closest_rays * real_counts[k];
        stats->shadow_tests[0] = h[ST_SHADOW_TESTS_TRI]; stats->shadow_tests[1] = h[ST_SHADOW_TESTS_SPH]; stats->shadow_tests[2] = h[ST_SHADOW_TESTS_ELL];
        stats->full_tests = h[ST_FULL_TESTS]; stats->pair_pretests[0] = h[ST_PAIR_PRETESTS]; stats->pair_pretests[1] = h[ST_PAIR_PRETESTS_SHADOW];
        stats->queued_pairs[0] = h[ST_QUEUED]; stats->queued_pairs[1] = h[ST_QUEUED_SHADOW];
        stats->first_level_nodes[0] = h[ST_FIRST_NODES]; stats->first_level_nodes[1] = h[ST_FIRST_NODES_SHADOW];
        stats->walk_steps[0] = h[ST_WALK_BATCHES]; stats->walk_steps[1] = h[ST_WALK_BATCHES_SHADOW];
        stats->walk_waves[0] = h[ST_WALK_WAVES]; stats->walk_waves[1] = h[ST_WALK_WAVES_SHADOW];
        stats->executed_flop[0] = h[ST_FLOP]; stats->executed_flop[1] = h[ST_FLOP_SHADOW];
        for (int k = 0; k < 4; ++k) { stats->brute_wave_steps[0][k] = h[ST_BRUTE_STEPS + k]; stats->brute_wave_steps[1][k] = h[ST_BRUTE_STEPS_SHADOW + k]; }
        float ms = 0.0f;
        if (ev_begin && ev_end && hipEventElapsedTime(&ms, ev_begin, ev_end) == hipSuccess) stats->device_ms = ms;
        double family_ms[8] = {};
        for (int fam = 0; fam < 8; ++fam)
            for (auto& span : timer->spans[fam]) if (hipEventElapsedTime(&ms, span.first, span.second) == hipSuccess) family_ms[fam] += ms;
        stats->closest_ms = family_ms[0]; stats->shadow_ms = family_ms[1];
        stats->closest_launches = timer->spans[0].size(); stats->shadow_launches = timer->spans[1].size();
        for (int k = 0; k < 3; ++k) {
            stats->closest_kind_ms[k] = family_ms[2 + k]; stats->closest_kind_launches[k] = timer->spans[2 + k].size();
            stats->shadow_kind_ms[k] = family_ms[5 + k]; stats->shadow_kind_launches[k] = timer->spans[5 + k].size();
        }
    }
    return 0;
}

// Cluster order of the primitives of one kind: the scan indices in the leaf order of a kd tree over
// their centroids, degenerate and non-finite primitives last.  A range is split near its middle at a
// multiple of CLUSTER^3, CLUSTER^2 or CLUSTER (the largest it exceeds), along the axis that leaves
// the two halves the smallest boxes (half area x count, over the primitives' own bounds), so that
// every aligned group of CLUSTER positions is one leaf (a compact patch), every aligned group of
// CLUSTER clusters one top node, and so on upward.  Only the box records and the triangles' pre-test
// records are stored in this order; it decides which primitives share a box and nothing else, so
// any permutation gives the same image.
struct ClusterOrder {
    const std::vector<float>& centroid;   // 3 floats per primitive
    const std::vector<float>& bounds;     // 6 floats per primitive: low corner, high corner
    std::vector<uint32_t> order;

    static constexpr uint32_t THREAD_LEVELS = 4;        // the first levels of the tree give their left halves to a thread of their own
    static constexpr uint32_t THREAD_MIN = 1u << 12;    // ... where a half is worth a thread

    static constexpr int BINS = 32;
    static constexpr uint32_t EXACT_BELOW = 4096;
    struct Bin {
        uint32_t count = 0;
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        void add(const float* b) {
            ++count;
            for (int q = 0; q < 3; ++q) { lo[q] = b[q] < lo[q] ? b[q] : lo[q]; hi[q] = b[3 + q] > hi[q] ? b[3 + q] : hi[q]; }
        }
        void merge(const Bin& o) {
            count += o.count;
            for (int q = 0; q < 3; ++q) { lo[q] = o.lo[q] < lo[q] ? o.lo[q] : lo[q]; hi[q] = o.hi[q] > hi[q] ? o.hi[q] : hi[q]; }
        }
        double half_area() const {
            if (count == 0) return 0.0;
            const double e[3] = {static_cast<double>(hi[0]) - lo[0], static_cast<double>(hi[1]) - lo[1], static_cast<double>(hi[2]) - lo[2]};
            return e[0] * e[1] + e[1] * e[2] + e[0] * e[2];
        }
    };

    // orders [begin, end) of `order`; `trial` is scratch of the calling thread
    void split(uint32_t begin, uint32_t end, uint32_t level, std::vector<uint32_t>& trial) {
        const uint32_t count = end - begin;
        if (count <= CLUSTER) return;
        const uint32_t align = count > CLUSTER * CLUSTER * CLUSTER ? CLUSTER * CLUSTER * CLUSTER : (count > CLUSTER * CLUSTER ? CLUSTER * CLUSTER : CLUSTER);
        uint32_t left = ((count / 2 + align / 2) / align) * align;
        if (left == 0) left = align;
        if (left >= count) left = count / 2;
        // Candidate planes: BINS - 1 per axis, evenly spaced over the centroids' extent.  The cost of one
        // is half area x count of the two sides; the cheapest decides the axis and, rounded to the
        // alignment, how many primitives go left.
        float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = begin; i < end; ++i)
            for (int q = 0; q < 3; ++q) {
                const float v = centroid[static_cast<size_t>(order[i]) * 3 + q];
                clo[q] = v < clo[q] ? v : clo[q]; chi[q] = v > chi[q] ? v : chi[q];
            }
        int best_axis = 0;
        for (int q = 1; q < 3; ++q) if (chi[q] - clo[q] > chi[best_axis] - clo[best_axis]) best_axis = q;
        double best_cost = DBL_MAX;
        for (int axis = 0; axis < 3; ++axis) {
            const double width = static_cast<double>(chi[axis]) - clo[axis];
            if (!(width > 0.0) || !std::isfinite(width)) continue;
            Bin bins[BINS];
            for (uint32_t i = begin; i < end; ++i) {
                const uint32_t k = order[i];
                int b = static_cast<int>((static_cast<double>(centroid[static_cast<size_t>(k) * 3 + axis]) - clo[axis]) / width * BINS);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                bins[b].add(&bounds[static_cast<size_t>(k) * 6]);
            }
            double right_area[BINS];
            Bin sweep;
            for (int b = BINS - 1; b > 0; --b) { sweep.merge(bins[b]); right_area[b] = sweep.half_area(); }
            sweep = Bin();
            for (int b = 1; b < BINS; ++b) {
                sweep.merge(bins[b - 1]);
                if (sweep.count == 0 || sweep.count == count) continue;
                uint32_t at = ((sweep.count + align / 2) / align) * align;
                if (at == 0) at = align;
                if (at >= count) continue;
                const double cost = sweep.half_area() * sweep.count + right_area[b] * (count - sweep.count);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; left = at; }
            }
        }
        if (count <= EXACT_BELOW) {
            // small ranges: every aligned position of every axis, with the exact boxes of its two sides
            best_cost = DBL_MAX;
            std::vector<double> right_area(count / align + 2);
            for (int axis = 0; axis < 3; ++axis) {
                trial.assign(order.begin() + begin, order.begin() + end);
                std::sort(trial.begin(), trial.end(), [&](uint32_t x, uint32_t y) {
                    const float cx = centroid[static_cast<size_t>(x) * 3 + axis], cy = centroid[static_cast<size_t>(y) * 3 + axis];
                    return cx < cy || (cx == cy && x < y);
                });
                Bin sweep;
                for (uint32_t i = count; i-- > align;) {
                    sweep.add(&bounds[static_cast<size_t>(trial[i]) * 6]);
                    if (i % align == 0) right_area[i / align] = sweep.half_area();
                }
                sweep = Bin();
                for (uint32_t i = 0; i + 1 < count; ++i) {
                    sweep.add(&bounds[static_cast<size_t>(trial[i]) * 6]);
                    const uint32_t at = i + 1;
                    if (at % align) continue;
                    const double cost = sweep.half_area() * at + right_area[at / align] * (count - at);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; left = at; }
                }
            }
        }
        {
            const int axis = best_axis;
            std::nth_element(order.begin() + begin, order.begin() + begin + left, order.begin() + end, [&](uint32_t x, uint32_t y) {
                const float cx = centroid[static_cast<size_t>(x) * 3 + axis], cy = centroid[static_cast<size_t>(y) * 3 + axis];
                return cx < cy || (cx == cy && x < y);
            });
        }
        if (level < THREAD_LEVELS && left >= THREAD_MIN) {   // the halves are disjoint ranges of `order`
            std::thread other([this, begin, left, level] {
                std::vector<uint32_t> scratch;
                split(begin, begin + left, level + 1, scratch);
            });
            split(begin + left, end, level + 1, trial);
            other.join();
        } else {
            split(begin, begin + left, level + 1, trial);
            split(begin + left, end, level + 1, trial);
        }
    }
};

std::vector<uint32_t> cluster_order(const std::vector<float>& centroid, const std::vector<float>& bounds, const std::vector<uint8_t>& good) {
    const uint32_t n = static_cast<uint32_t>(good.size());
    ClusterOrder work{centroid, bounds, {}};
    std::vector<uint32_t> rejects, trial;
    work.order.reserve(n);
    for (uint32_t k = 0; k < n; ++k) (good[k] ? work.order : rejects).push_back(k);
    work.split(0u, static_cast<uint32_t>(work.order.size()), 0u, trial);
    work.order.insert(work.order.end(), rejects.begin(), rejects.end());
    return std::move(work.order);
}

// centroids of the triangles in scan order (k = lane * batches + batch), and which of them are proper triangles
void triangle_centroids(const float* tri8, uint32_t nb, std::vector<float>& centroid, std::vector<float>& bounds, std::vector<uint8_t>& good) {
    const uint32_t n = 8u * nb;
    centroid.assign(static_cast<size_t>(n) * 3, 0.0f);
    bounds.assign(static_cast<size_t>(n) * 6, 0.0f);
    good.assign(n, 0);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t j = k / nb, i = k % nb;
        const float* p = tri8 + static_cast<size_t>(i) * 72 + j;
        const double a[3] = {p[0], p[8], p[16]}, ab[3] = {p[24], p[32], p[40]}, ac[3] = {p[48], p[56], p[64]};
        const double nx = ab[1] * ac[2] - ab[2] * ac[1], ny = ab[2] * ac[0] - ab[0] * ac[2], nz = ab[0] * ac[1] - ab[1] * ac[0];
        const double c[3] = {a[0] + (ab[0] + ac[0]) / 3.0, a[1] + (ab[1] + ac[1]) / 3.0, a[2] + (ab[2] + ac[2]) / 3.0};
        const double nn = nx * nx + ny * ny + nz * nz;
        good[k] = std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]) && std::isfinite(nn) && nn > 0.0;
        if (good[k]) for (int q = 0; q < 3; ++q) {
            centroid[static_cast<size_t>(k) * 3 + q] = static_cast<float>(c[q]);
            bounds[static_cast<size_t>(k) * 6 + q] = static_cast<float>(a[q] + std::fmin(0.0, std::fmin(ab[q], ac[q])));
            bounds[static_cast<size_t>(k) * 6 + 3 + q] = static_cast<float>(a[q] + std::fmax(0.0, std::fmax(ab[q], ac[q])));
        }
    }
}
// centres of the ellipsoids in scan order: the translation column of the forward transform (Mat3x4x8, ray_tracing.h:62-69)
void ellipsoid_centroids(const float* fwd8, const float* inv8, uint32_t nb, std::vector<float>& centroid, std::vector<float>& bounds, std::vector<uint8_t>& good) {
    const uint32_t n = 8u * nb;
    centroid.assign(static_cast<size_t>(n) * 3, 0.0f);
    bounds.assign(static_cast<size_t>(n) * 6, 0.0f);
    good.assign(n, 0);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t j = k / nb, i = k % nb;
        const float* pf = fwd8 + static_cast<size_t>(i) * 96 + j;
        const float* pi = inv8 + static_cast<size_t>(i) * 96 + j;
        double c[3], reach[3], scale = 0.0;
        for (int row = 0; row < 3; ++row) {
            c[row] = pf[(row * 4 + 3) * 8];
            double rr = 0.0;   // the image of the unit sphere reaches the length of the row along this axis
            for (int col = 0; col < 3; ++col) {
                scale += std::fabs(static_cast<double>(pi[(row * 4 + col) * 8]));
                rr += static_cast<double>(pf[(row * 4 + col) * 8]) * pf[(row * 4 + col) * 8];
            }
            reach[row] = std::sqrt(rr);
        }
        good[k] = std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]) && std::isfinite(scale) && scale > 0.0 && std::isfinite(reach[0] + reach[1] + reach[2]);
        if (good[k]) for (int q = 0; q < 3; ++q) {
            centroid[static_cast<size_t>(k) * 3 + q] = static_cast<float>(c[q]);
            bounds[static_cast<size_t>(k) * 6 + q] = static_cast<float>(c[q] - reach[q]);
            bounds[static_cast<size_t>(k) * 6 + 3 + q] = static_cast<float>(c[q] + reach[q]);
        }
    }
}

// centres of the spheres in scan order (Sphere8: x[8], y[8], z[8], radius[8])
void sphere_centroids(const float* sph8, uint32_t nb, std::vector<float>& centroid, std::vector<float>& bounds, std::vector<uint8_t>& good) {
    const uint32_t n = 8u * nb;
    centroid.assign(static_cast<size_t>(n) * 3, 0.0f);
    bounds.assign(static_cast<size_t>(n) * 6, 0.0f);
    good.assign(n, 0);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t j = k / nb, i = k % nb;
        const float* p = sph8 + static_cast<size_t>(i) * 32 + j;
        good[k] = std::isfinite(p[0]) && std::isfinite(p[8]) && std::isfinite(p[16]) && std::isfinite(p[24]);
        if (good[k]) for (int q = 0; q < 3; ++q) {
            centroid[static_cast<size_t>(k) * 3 + q] = p[q * 8];
            bounds[static_cast<size_t>(k) * 6 + q] = p[q * 8] - std::fabs(p[24]);
            bounds[static_cast<size_t>(k) * 6 + 3 + q] = p[q * 8] + std::fabs(p[24]);
        }
    }
}

template <typename T>
int upload(rt_scene* s, const T* host, size_t count, T** dev) {
    *dev = nullptr;
    if (count == 0) return 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dev), count * sizeof(T)));
    s->allocations.push_back(*dev);
    if (host) HIP_TRY(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, current_device().stream));
    return 0;
}

// No C++ exception crosses the C boundary (std::thread and std::vector can throw): it becomes an error code.
template <typename Body>
int guarded(Body&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return rt_fail(RT_ERR_LIMIT, "out of host memory");
    } catch (const std::exception& e) {
        return rt_fail(RT_ERR_DEVICE, e.what());
    } catch (...) {
        return rt_fail(RT_ERR_DEVICE, "unknown C++ exception");
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// Creates the context of one device (streams, events, counters); idempotent.
static int init_device(int device, int hip_device = -1) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (hip_device < 0) hip_device = device;
    if (device < 0 || hip_device >= n || device >= MAX_DEVICES) return rt_fail(RT_ERR_ARGUMENT, "rt_init: no such device");
    Library& lib = g_devices[device];
    if (lib.ready) return 0;
    HIP_TRY(hipSetDevice(hip_device));
    for (FrameSlot& slot : lib.slots) {
        for (Lane& lane : slot.lanes) {
            HIP_TRY(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&lane.side_stream, hipStreamNonBlocking));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&lane.counts), 64 * sizeof(uint32_t)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&lane.host_counts), 64 * sizeof(uint32_t), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&lane.count_ready, hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&slot.lanes_fork, hipEventDisableTiming));
        for (hipEvent_t& e : slot.lanes_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : slot.lanes_started) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot.stats), STAT_COPIES * ST_COUNT * sizeof(unsigned long long)));
    }
    lib.stream = lib.slots[0].lanes[0].stream;
    HIP_TRY(hipStreamCreateWithFlags(&lib.comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&lib.comm_ready, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&lib.multi_sent[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&lib.multi_exchange_begin[k]));
        HIP_TRY(hipEventCreate(&lib.multi_exchange_end[k]));
    }
    // this device's copy of the scratch array of rt_render_tiles_noalloc: its address, once, with the device current
    HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&lib.scratch), HIP_SYMBOL(scratch)));
    if (!lib.scratch) return rt_fail(RT_ERR_DEVICE, "rt_init: the scratch array has no device address");
    lib.device = device;
    lib.hip_device = hip_device;
    lib.ready = true;
    Library* none = nullptr;
    g_first_device.compare_exchange_strong(none, &lib);
    return 0;
}

int rt_init(int device) {
    Library* first = g_first_device.load(std::memory_order_acquire);
    if (first && first->device == device) return 0;
    if (first) return rt_fail(RT_ERR_ARGUMENT, "rt_init: already initialised on another device");
    return init_device(device);
}

void rt_scene_destroy(rt_scene* scene) {
    while (scene) {
        rt_scene* next = scene->next;
        if (scene->device >= 0 && scene->device < MAX_DEVICES && g_devices[scene->device].ready) {
            DeviceBinding bound(g_devices[scene->device]);
            for (void* p : scene->allocations) (void)hipFree(p);
        }
        delete scene;
        scene = next;
    }
}

// cluster orders of the three kinds (host work, the same for every device)
struct SceneOrders { std::vector<uint32_t> of[3]; bool ready[3] = {false, false, false}; };
static int scene_create_impl(const rt_scene_desc* desc, SceneOrders& orders, rt_scene** out);
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out) {
    if (!current_device().ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    if (!desc || !out) return rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: null argument");
    *out = nullptr;
    return guarded([&] {
        SceneOrders orders;
        const int n_multi = g_multi_devices.load();
        if (n_multi <= 1) {
            Library& lib = current_device();
            DeviceBinding here(lib);
            if (!here.ok) return rt_fail(RT_ERR_DEVICE, "rt_scene_create: hipSetDevice failed");
            std::lock_guard<std::mutex> one_at_a_time(lib.guard);
            return scene_create_impl(desc, orders, out);
        }
        // one replica per device, device 0 first; the cluster orders are computed once
        rt_scene* head = nullptr;
        rt_scene** tail = &head;
        for (int d = 0; d < n_multi; ++d) {
            DeviceBinding bound(g_devices[d]);
            std::lock_guard<std::mutex> one_at_a_time(g_devices[d].guard);
            const int rc = bound.ok ? scene_create_impl(desc, orders, tail) : rt_fail(RT_ERR_DEVICE, "rt_scene_create: hipSetDevice failed");
            if (rc) { rt_scene_destroy(head); return rc; }
            tail = &(*tail)->next;
        }
        *out = head;
        return 0;
    });
}
static int scene_create_impl(const rt_scene_desc* desc, SceneOrders& orders, rt_scene** out) {
    Library& lib = current_device();
    if (desc->n_point_lights && !desc->point_light_sources) return rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: point_light_sources is null");
    if ((desc->n_triangles && !desc->triangle_materials) || (desc->n_spheres && !desc->sphere_materials) || (desc->n_ellipsoids && !desc->ellipsoid_materials))
        return rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: a material array is null");
    if (desc->n_triangle8 > (1u << 26) || desc->n_sphere8 > (1u << 26) || desc->n_ellipsoid8 > (1u << 26)) return rt_fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^29 primitives of one kind");
    if (desc->n_triangles > desc->n_triangle8 * 8 || desc->n_spheres > desc->n_sphere8 * 8 || desc->n_ellipsoids > desc->n_ellipsoid8 * 8) return rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: more materials than primitive slots");
    if (desc->n_point_lights > 4096) return rt_fail(RT_ERR_LIMIT, "rt_scene_create: more than 4096 point lights");
    if (desc->n_triangle8 > (1u << 21) || desc->n_sphere8 > (1u << 21) || desc->n_ellipsoid8 > (1u << 21)) return rt_fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^24 slots of one primitive kind");
    rt_scene* s = new rt_scene();
    s->device = lib.device;
    auto fail = [&](int rc) { rt_scene_destroy(s); return rc; };
    DevScene& d = s->dev;
    std::memset(&d, 0, sizeof d);
    hipStream_t stream = lib.stream;
    int rc;

    d.tri_nb = static_cast<uint32_t>(desc->n_triangle8); d.n_tri = 8u * d.tri_nb;
    d.sph_nb = static_cast<uint32_t>(desc->n_sphere8); d.n_sph = 8u * d.sph_nb;
    d.ell_nb = static_cast<uint32_t>(desc->n_ellipsoid8); d.n_ell = 8u * d.ell_nb;
    d.n_tri_mat = static_cast<uint32_t>(desc->n_triangles);
    d.n_sph_mat = static_cast<uint32_t>(desc->n_spheres);
    d.n_ell_mat = static_cast<uint32_t>(desc->n_ellipsoids);

    float *raw_tri, *raw_sph, *raw_inv, *raw_fwd, *raw_pl;
    if ((rc = upload(s, desc->triangle8s, static_cast<size_t>(d.tri_nb) * 72, &raw_tri))) return fail(rc);
    if ((rc = upload(s, desc->sphere8s, static_cast<size_t>(d.sph_nb) * 32, &raw_sph))) return fail(rc);
    if ((rc = upload(s, desc->ellipsoid8_inverse_transforms, static_cast<size_t>(d.ell_nb) * 96, &raw_inv))) return fail(rc);
    if ((rc = upload(s, desc->ellipsoid8_transforms, static_cast<size_t>(d.ell_nb) * 96, &raw_fwd))) return fail(rc);
    if ((rc = upload(s, desc->point_light_sources, static_cast<size_t>(desc->n_point_lights) * 6, &raw_pl))) return fail(rc);
    float *tri_mat, *sph_mat, *ell_mat;
    if ((rc = upload(s, desc->triangle_materials, static_cast<size_t>(d.n_tri_mat) * 10, &tri_mat))) return fail(rc);
    if ((rc = upload(s, desc->sphere_materials, static_cast<size_t>(d.n_sph_mat) * 10, &sph_mat))) return fail(rc);
    if ((rc = upload(s, desc->ellipsoid_materials, static_cast<size_t>(d.n_ell_mat) * 10, &ell_mat))) return fail(rc);
    d.tri_mat = tri_mat; d.sph_mat = sph_mat; d.ell_mat = ell_mat;

    float4 *tri, *trif, *sph, *ell, *ell_light;
    float* ell_ok;
    if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(d.n_tri) * 4, &tri))) return fail(rc);
    if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(d.n_tri) * 4, &trif))) return fail(rc);
    if ((rc = upload<float4>(s, nullptr, d.n_sph, &sph))) return fail(rc);
    if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(d.n_ell) * 6, &ell))) return fail(rc);
    if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(d.n_ell) * desc->n_point_lights, &ell_light))) return fail(rc);
    if ((rc = upload<float>(s, nullptr, d.n_ell, &ell_ok))) return fail(rc);
    d.tri = tri; d.trif = trif; d.sph = sph; d.ell = ell; d.ell_light = ell_light; d.ell_ok = ell_ok;
    {
        float b = 0.0f;
        for (int i = 0; i < 6; ++i) b = std::fmax(b, std::fabs(desc->bounding_box[i]));
        d.origin_bound = 1.001f * b + 1.0e-3f;
    }
    if (d.n_tri) hipLaunchKernelGGL(k_prep_triangles, dim3(ceil_div(d.n_tri, BLOCK)), dim3(BLOCK), 0, stream, raw_tri, d.tri_nb, tri, trif);
    if (d.n_sph) hipLaunchKernelGGL(k_prep_spheres, dim3(ceil_div(d.n_sph, BLOCK)), dim3(BLOCK), 0, stream, raw_sph, d.sph_nb, sph);
    if (d.n_ell) hipLaunchKernelGGL(k_prep_ellipsoids, dim3(ceil_div(d.n_ell, BLOCK)), dim3(BLOCK), 0, stream, raw_inv, raw_fwd, d.ell_nb, ell, ell_ok);
    if (d.n_ell && desc->n_point_lights)
        hipLaunchKernelGGL(k_prep_ellipsoid_lights, dim3(ceil_div(d.n_ell, BLOCK)), dim3(BLOCK), 0, stream, ell, d.n_ell, raw_pl, static_cast<uint32_t>(desc->n_point_lights), ell_light);

    // box hierarchy of every kind (see Hierarchy): cluster order on the host; the boxes themselves by build_boxes below
    for (int prim = 0; prim < 3; ++prim) {
        Hierarchy& h = prim == 0 ? d.th : (prim == 1 ? d.eh : d.sh);
        h.n = prim == 0 ? d.n_tri : (prim == 1 ? d.n_ell : d.n_sph);
        h.n_clusters = ceil_div(h.n, CLUSTER);
        h.n_tops = ceil_div(h.n_clusters, CLUSTER);
        h.n_supers = ceil_div(h.n_tops, CLUSTER);
        h.n_roots = ceil_div(h.n_supers, CLUSTER);
        h.n_apex = ceil_div(h.n_roots, CLUSTER);
        if (h.n == 0) continue;
        if (!orders.ready[prim]) {
            std::vector<float> centroid, bounds;
            std::vector<uint8_t> good;
            if (prim == 0) {
                if (!desc->triangle8s) return fail(rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: triangle8s is null"));
                triangle_centroids(desc->triangle8s, d.tri_nb, centroid, bounds, good);
            } else if (prim == 1) {
                if (!desc->ellipsoid8_transforms || !desc->ellipsoid8_inverse_transforms) return fail(rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: ellipsoid transforms are null"));
                ellipsoid_centroids(desc->ellipsoid8_transforms, desc->ellipsoid8_inverse_transforms, d.ell_nb, centroid, bounds, good);
            } else {
                if (!desc->sphere8s) return fail(rt_fail(RT_ERR_ARGUMENT, "rt_scene_create: sphere8s is null"));
                sphere_centroids(desc->sphere8s, d.sph_nb, centroid, bounds, good);
            }
            orders.of[prim] = cluster_order(centroid, bounds, good);
            orders.ready[prim] = true;
        }
        uint32_t* k_of;
        float4 *leaf_box, *cl_box, *top_box, *super_box, *root_box;
        uint4 *cl_nodes, *top_nodes, *super_nodes, *root_nodes, *apex_nodes;
        if ((rc = upload(s, orders.of[prim].data(), orders.of[prim].size(), &k_of))) return fail(rc);
        if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n) * 2, &leaf_box))) return fail(rc);
        if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n_clusters) * 2, &cl_box))) return fail(rc);
        if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n_tops) * 2, &top_box))) return fail(rc);
        if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n_supers) * 2, &super_box))) return fail(rc);
        if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n_roots) * 2, &root_box))) return fail(rc);
        h.k_of = k_of;
        if ((rc = upload<uint4>(s, nullptr, static_cast<size_t>(h.n_clusters) * 8, &cl_nodes))) return fail(rc);
        if ((rc = upload<uint4>(s, nullptr, static_cast<size_t>(h.n_tops) * 8, &top_nodes))) return fail(rc);
        if ((rc = upload<uint4>(s, nullptr, static_cast<size_t>(h.n_supers) * 8, &super_nodes))) return fail(rc);
        if ((rc = upload<uint4>(s, nullptr, static_cast<size_t>(h.n_roots) * 8, &root_nodes))) return fail(rc);
        if ((rc = upload<uint4>(s, nullptr, static_cast<size_t>(h.n_apex) * 8, &apex_nodes))) return fail(rc);
        h.leaf_box = leaf_box; h.cl_box = cl_box; h.top_box = top_box; h.super_box = super_box; h.root_box = root_box;
        h.cl_nodes = cl_nodes; h.top_nodes = top_nodes; h.super_nodes = super_nodes; h.root_nodes = root_nodes; h.apex_nodes = apex_nodes;
        if (prim == 0) {
            float4* trif_pos;
            if ((rc = upload<float4>(s, nullptr, static_cast<size_t>(h.n) * 4, &trif_pos))) return fail(rc);
            hipLaunchKernelGGL(k_reorder_pretest_records, dim3(ceil_div(h.n, BLOCK)), dim3(BLOCK), 0, stream, static_cast<const float4*>(trif), h.k_of, h.n, trif_pos);
            d.trif_pos = trif_pos;
        }
    }

    // light table: the directional light first (ray_tracing.cpp:687), then the point lights in file order (:704)
    std::vector<float> lights;
    d.has_directional = desc->has_directional_light ? 1u : 0u;
    if (d.has_directional) {
        const float* L = desc->directional_light;
        lights.insert(lights.end(), {0.0f, L[0], L[1], L[2], L[3], L[4], L[5], 0.0f});
        const double len2 = static_cast<double>(L[0]) * L[0] + static_cast<double>(L[1]) * L[1] + static_cast<double>(L[2]) * L[2];
        s->directional_is_unit = len2 <= 1.0 + 20.0 * 5.9604644775390625e-8;   // shorter is harmless: every pre-test only gets weaker
    }
    for (uint64_t l = 0; l < desc->n_point_lights; ++l) {
        const float* L = desc->point_light_sources + l * 6;
        lights.insert(lights.end(), {1.0f, L[0], L[1], L[2], L[3], L[4], L[5], 0.0f});
    }
    d.n_lights = static_cast<uint32_t>(lights.size() / 8);
    float* lights_dev;
    if ((rc = upload(s, lights.data(), lights.size(), &lights_dev))) return fail(rc);
    d.lights = lights_dev;
    std::memcpy(d.bb, desc->bounding_box, sizeof d.bb);
    std::memcpy(d.ambient, desc->ambient, sizeof d.ambient);
    std::memcpy(d.atten, desc->attenuation, sizeof d.atten);

    build_boxes(s, d.origin_bound, stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail(rt_fail(RT_ERR_DEVICE, "rt_scene_create: scene preparation kernels failed"));
    *out = s;
    return 0;
}

uint32_t rt_rank_tile_count(const rt_render_params* params) { return params ? rank_tiles(*params) : 0; }
uint32_t rt_total_tile_count(const rt_render_params* params) { return params ? rect_of(*params).tiles : 0; }

int rt_render_tiles_device(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* tiles_dev, rt_stats* stats) {
    if (!tiles_dev) return rt_fail(RT_ERR_ARGUMENT, "rt_render_tiles_device: null output");
    return guarded([&] { return render_impl(scene, camera, params, static_cast<uint8_t*>(tiles_dev), nullptr, stats); });
}

int rt_render_tiles_noalloc(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* tiles_dev) {
    if (!tiles_dev) return rt_fail(RT_ERR_ARGUMENT, "rt_render_tiles_noalloc: null output");
    return guarded([&] { return render_impl(scene, camera, params, static_cast<uint8_t*>(tiles_dev), nullptr, nullptr, true); });
}

// Frames in flight.  A frame ends with thin launches (the late bounces, the last chunk's tail) that leave most of the chip
// idle, and the shorter the frame - one rank's share of a split frame - the larger that part.  begin starts a frame on a free
// slot and returns; a host thread of the library drives it (it has to wait for the ray counts that size the launches).  With two
// slots the first launches of the next frame run beside the last ones of this frame.  end waits for a frame and returns its status.
int rt_render_tiles_begin(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* tiles_dev, int* ticket) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    if (!scene || !camera || !params || !tiles_dev || !ticket) return rt_fail(RT_ERR_ARGUMENT, "rt_render_tiles_begin: null argument");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> frames(lib.frames_guard);     // begin and end from any threads: one at a time
        int free_slot = -1;
        for (int i = 0; i < FRAME_SLOTS; ++i) if (!lib.slots[i].busy) { free_slot = i; break; }
        if (free_slot < 0) return rt_fail(RT_ERR_LIMIT, "rt_render_tiles_begin: two frames are in flight already (rt_render_tiles_end)");
        {
            const rt_scene* replica = scene;
            while (replica && replica->device != lib.device) replica = replica->next;
            if (!replica) return rt_fail(RT_ERR_ARGUMENT, "the scene has no replica on this device");
            for (int i = 0; i < 3; ++i) if (!std::isfinite(camera->eye[i])) return rt_fail(RT_ERR_ARGUMENT, "camera eye is not finite");
        }
        FrameSlot& slot = lib.slots[free_slot];
        slot.busy = true;
        slot.rc = 0;
        const rt_camera_desc cam_copy = *camera;
        const rt_render_params prm_copy = *params;
        slot.worker = std::thread([&lib, &slot, free_slot, scene, cam_copy, prm_copy, tiles_dev] {
            DeviceBinding bound(lib);
            slot.rc = bound.ok ? guarded([&] { return render_impl(scene, &cam_copy, &prm_copy, static_cast<uint8_t*>(tiles_dev), nullptr, &slot.stats_out, false, free_slot); })
                               : rt_fail(RT_ERR_DEVICE, "rt_render_tiles_begin: hipSetDevice failed in the frame's thread");
            if (slot.rc) slot.error = g_last_error;
        });
        *ticket = free_slot;
        return 0;
    });
}

int rt_render_tiles_end(int ticket, rt_stats* stats) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    std::lock_guard<std::mutex> frames(lib.frames_guard);
    if (ticket < 0 || ticket >= FRAME_SLOTS || !lib.slots[ticket].busy) return rt_fail(RT_ERR_ARGUMENT, "rt_render_tiles_end: no frame in flight under this ticket");
    FrameSlot& slot = lib.slots[ticket];
    if (slot.worker.joinable()) slot.worker.join();
    slot.busy = false;
    if (stats) *stats = slot.stats_out;
    return slot.rc ? rt_fail(slot.rc, slot.error.c_str()) : 0;
}

int rt_assemble_tiles_device(const void* gathered, uint32_t stride_tiles, const rt_render_params* params, void* rgb_dev) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    DeviceBinding here(lib);
    std::lock_guard<std::mutex> one_at_a_time(lib.guard);
    if (!gathered || !params || !rgb_dev) return rt_fail(RT_ERR_ARGUMENT, "rt_assemble_tiles_device: null argument");
    const Rect rect = rect_of(*params);
    const uint32_t world = params->tile_world ? params->tile_world : 1;
    hipLaunchKernelGGL(k_assemble, dim3(ceil_div(rect.w * rect.h, BLOCK)), dim3(BLOCK), 0, lib.stream,
                       static_cast<const uint8_t*>(gathered), stride_tiles, world, rect.w, rect.h, rect.tiles_x, static_cast<uint8_t*>(rgb_dev));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(lib.stream));
    return 0;
}

int rt_render(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, uint8_t* rgb_out, uint32_t* primary_id_out, rt_stats* stats) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    if (!params || !rgb_out) return rt_fail(RT_ERR_ARGUMENT, "rt_render: null argument");
    DeviceBinding here(lib);      // the buffers below belong to the device of this context, whatever the calling thread's HIP device was
    const Rect rect = rect_of(*params);
    const uint32_t world = params->tile_world ? params->tile_world : 1;
    const uint32_t my_tiles = rank_tiles(*params);
    const size_t n_px = static_cast<size_t>(rect.w) * rect.h;
    uint8_t *tiles_dev = nullptr, *rgb_dev = nullptr;
    uint32_t* ids_dev = nullptr;
    int rc = 0;
    auto cleanup = [&]() { if (tiles_dev) (void)hipFree(tiles_dev); if (rgb_dev) (void)hipFree(rgb_dev); if (ids_dev) (void)hipFree(ids_dev); };
    if (hipMalloc(reinterpret_cast<void**>(&tiles_dev), static_cast<size_t>(my_tiles ? my_tiles : 1) * 768) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&rgb_dev), n_px * 3) != hipSuccess ||
        (primary_id_out && hipMalloc(reinterpret_cast<void**>(&ids_dev), n_px * 16 * sizeof(uint32_t)) != hipSuccess)) {
        cleanup();
        return rt_fail(RT_ERR_DEVICE, "rt_render: out of device memory");
    }
    if (primary_id_out && hipMemsetAsync(ids_dev, 0xFF, n_px * 16 * sizeof(uint32_t), lib.stream) != hipSuccess) { cleanup(); return rt_fail(RT_ERR_DEVICE, "rt_render: memset failed"); }
    rc = guarded([&] { return render_impl(scene, camera, params, tiles_dev, ids_dev, stats); });
    if (rc == 0) {
        if (world == 1) {
            rc = rt_assemble_tiles_device(tiles_dev, my_tiles, params, rgb_dev);
            if (rc == 0 && hipMemcpy(rgb_out, rgb_dev, n_px * 3, hipMemcpyDeviceToHost) != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_render: copy back failed");
        } else {
            // only this rank's tiles are defined: copy them pixel by pixel into the caller's image
            std::vector<uint8_t> tiles(static_cast<size_t>(my_tiles) * 768);
            if (my_tiles && hipMemcpy(tiles.data(), tiles_dev, tiles.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_render: copy back failed");
            for (uint32_t slot = 0; rc == 0 && slot < my_tiles; ++slot) {
                const uint32_t tile = slot * world + params->tile_rank;
                const uint32_t tx = tile_column(tile, rect.tiles_x), ty = tile_row(tile, rect.tiles_x);
                for (uint32_t py = 0; py < RT_TILE; ++py)
                    for (uint32_t px = 0; px < RT_TILE; ++px) {
                        const uint32_t x = tx * RT_TILE + px, y = ty * RT_TILE + py;
                        if (x < rect.w && y < rect.h) std::memcpy(rgb_out + (static_cast<size_t>(y) * rect.w + x) * 3, tiles.data() + (static_cast<size_t>(slot) * 256 + py * RT_TILE + px) * 3, 3);
                    }
            }
        }
        if (rc == 0 && primary_id_out && hipMemcpy(primary_id_out, ids_dev, n_px * 16 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_render: id copy back failed");
    }
    cleanup();
    return rc;
}

int rt_render_file_info(const rt_file_info* info, uint8_t* rgb_out, rt_stats* stats) {
    if (!info || !rgb_out) return rt_fail(RT_ERR_ARGUMENT, "rt_render_file_info: null argument");
    rt_camera_desc cam;
    rt_camera_setup(info->camera, &cam);
    rt_scene* scene = nullptr;
    int rc = rt_scene_create(&info->desc, &scene);
    if (rc) return rc;
    rt_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = info->width; p.height = info->height; p.max_bounce_count = info->max_recursion_depth;
    rc = rt_render(scene, &cam, &p, rgb_out, nullptr, stats);
    rt_scene_destroy(scene);
    return rc;
}

int rt_debug_eval(int op, uint32_t n, const float* a, const float* b, float* out) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    static const struct { int op; uint32_t a_per_item, a_fixed, b_per_item, out_per_item; } shapes[] = {
        {RT_DEBUG_FP_POW, 1, 0, 1, 1}, {RT_DEBUG_FP_MIN, 1, 0, 1, 1}, {RT_DEBUG_FP_MAX, 1, 0, 1, 1}, {RT_DEBUG_DIVIDE, 1, 0, 1, 1},
        {RT_DEBUG_DIVIDE_IN_RANGE, 1, 0, 1, 1}, {RT_DEBUG_PRIMARY_RAY, 0, 16, 4, 3}, {RT_DEBUG_SCENE_BOX, 0, 6, 6, 2},
        {RT_DEBUG_TRIANGLE, 9, 0, 6, 1}, {RT_DEBUG_TRIANGLE_STAGED, 9, 0, 6, 1}, {RT_DEBUG_SPHERE, 4, 0, 6, 1},
        {RT_DEBUG_ELLIPSOID, 24, 0, 6, 1}, {RT_DEBUG_ELLIPSOID_STAGED, 24, 0, 6, 2}, {RT_DEBUG_ELLIPSOID_SHADOW, 16, 0, 6, 1},
        {RT_DEBUG_ELLIPSOID_SHADOW_STAGED, 16, 0, 6, 2}, {RT_DEBUG_ELLIPSOID_PARTS, 12, 0, 6, 8}, {RT_DEBUG_TRIANGLE_PRETEST, 9, 0, 6, 3}};
    const auto* shape = &shapes[0];
    bool found = false;
    for (const auto& sh : shapes) if (sh.op == op) { shape = &sh; found = true; }
    if (!found) return rt_fail(RT_ERR_ARGUMENT, "rt_debug_eval: unknown operation");
    if (!a || !b || !out || n == 0 || n > (1u << 26)) return rt_fail(RT_ERR_ARGUMENT, "rt_debug_eval: null argument or item count out of range");
    const size_t na = static_cast<size_t>(shape->a_per_item) * n + shape->a_fixed, nb = static_cast<size_t>(shape->b_per_item) * n, no = static_cast<size_t>(shape->out_per_item) * n;
    DeviceBinding here(lib);
    float* dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), (na + nb + no) * sizeof(float)));
    int rc = 0;
    hipStream_t stream = lib.stream;
    if (hipMemcpyAsync(dev, a, na * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(dev + na, b, nb * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_debug_eval: upload failed");
    if (rc == 0) {
        hipLaunchKernelGGL(k_debug_eval, dim3(ceil_div(n, BLOCK)), dim3(BLOCK), 0, stream, op, n, static_cast<const float*>(dev), static_cast<const float*>(dev + na), dev + na + nb);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, dev + na + nb, no * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, "rt_debug_eval: kernel or copy back failed");
    }
    (void)hipFree(dev);
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// multi-GPU: image tiles over devices, tile buffers gathered on the first one by RCCL (xGMI)
// ------------------------------------------------------------------------------------------
// This stands where the reference hands scanlines to the threads of one machine
// (Source/render_work_queue.cpp:59-80, Source/main.cpp:77-90): tile t of the rectangle belongs to device
// t % n, the scene is replicated, and the only exchange is every device sending its compact tile buffer to
// the first device (grouped ncclSend / ncclRecv: one hop per peer over xGMI), which scatters the tiles into
// the row-major image (k_assemble).  RCCL is loaded on first use, so hosts that render on one GPU never load it.
namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_guard;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_guard);
    if (g_rccl.ok) return 0;
    // a copy the process has loaded already (a host that brings its own ROCm libraries) comes first
    void* handle = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* name : names) if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!handle) return rt_fail(RT_ERR_DEVICE, "RCCL (librccl.so) could not be loaded");
    auto sym = [&](const char* name) { return dlsym(handle, name); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.Send || !g_rccl.Recv ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString) return rt_fail(RT_ERR_DEVICE, "RCCL lacks an entry point this library needs");
    g_rccl.ok = true;
    return 0;
}

#define RCCL_TRY(expr)                                                                        \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            char buf_[512];                                                                   \
            std::snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return rt_fail(RT_ERR_DEVICE, buf_);                                              \
        }                                                                                     \
    } while (0)

// Every device of `devices` sends `bytes` bytes of its tile buffer to the first one, which receives them rank-major
// (its own included: a send to itself, so that one device alone exercises the same calls).  Enqueued on the
// devices' exchange streams; the caller synchronises the first device's stream.
int enqueue_gather(Library* const* devices, const uint8_t* const* send, int n, size_t bytes, uint8_t* gathered_on_first, int only_rank = -1) {
    RCCL_TRY(g_rccl.GroupStart());
    for (int d = 0; d < n; ++d) {
        if (only_rank >= 0 && d != only_rank) continue;     // one process per GPU: this process issues its own rank's part
        Library& lib = *devices[d];
        const ncclComm_t comm = static_cast<ncclComm_t>(lib.comm);
        if (bytes) RCCL_TRY(g_rccl.Send(send[d], bytes, ncclUint8, 0, comm, lib.comm_stream));
        if (lib.comm_rank == 0 && bytes)
            for (int r = 0; r < lib.comm_world; ++r) RCCL_TRY(g_rccl.Recv(gathered_on_first + static_cast<size_t>(r) * bytes, bytes, ncclUint8, r, comm, lib.comm_stream));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

int grow(uint8_t** buffer, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*buffer) { HIP_TRY(hipFree(*buffer)); *buffer = nullptr; *have = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(buffer), need));
    *have = need;
    return 0;
}

// One host thread per device, started by rt_init_devices and kept until the library is unloaded (the reference starts its
// workers once and joins them once, Source/main.cpp:77-90): a frame hands every device its share and waits for all of them.
struct DeviceWorker {
    std::thread thread;
    std::mutex guard;
    std::condition_variable wake;
    std::function<void()> job;
    bool has_job = false, stop = false;
};
struct WorkerPool {
    DeviceWorker workers[MAX_DEVICES];
    int n = 0;
    void start(int count) {
        for (int d = n; d < count; ++d) {
            DeviceWorker& w = workers[d];
            w.thread = std::thread([&w, d] {
                DeviceBinding bound(g_devices[d]);
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lock(w.guard);
                        w.wake.wait(lock, [&w] { return w.has_job || w.stop; });
                        if (w.stop) return;
                        job = std::move(w.job);
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lock(w.guard);
                        w.has_job = false;
                    }
                    w.wake.notify_all();
                }
            });
        }
        n = std::max(n, count);
    }
    // runs body(d) on the thread of every device d < count and returns when all are done
    template <typename Body>
    void on_every_device(int count, Body&& body) {
        for (int d = 0; d < count; ++d) {
            DeviceWorker& w = workers[d];
            {
                std::lock_guard<std::mutex> lock(w.guard);
                w.job = [&body, d] { body(d); };
                w.has_job = true;
            }
            w.wake.notify_all();
        }
        for (int d = 0; d < count; ++d) {
            DeviceWorker& w = workers[d];
            std::unique_lock<std::mutex> lock(w.guard);
            w.wake.wait(lock, [&w] { return !w.has_job; });
        }
    }
    ~WorkerPool() {
        for (int d = 0; d < n; ++d) {
            { std::lock_guard<std::mutex> lock(workers[d].guard); workers[d].stop = true; }
            workers[d].wake.notify_all();
            if (workers[d].thread.joinable()) workers[d].thread.join();
        }
    }
};
WorkerPool g_pool;          // (after g_devices: destroyed before the device contexts)
std::mutex g_multi_guard;   // one rt_render_multi_begin / _end at a time
struct MultiFrame { bool busy = false; int n = 0; rt_stats stats[MAX_DEVICES]; bool timed = false; };
MultiFrame g_multi_frames[2];
int g_multi_next = 0;       // the buffer set the next frame takes
bool g_copy_gather = false; // several contexts on one GPU (RT_AMD_DEVICES_ON_ONE_GPU): gather by device-to-device copies

}  // namespace

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the identifier is RCCL's");
    if (!id) return rt_fail(RT_ERR_ARGUMENT, "rt_comm_unique_id: null argument");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId uid;
    RCCL_TRY(g_rccl.GetUniqueId(&uid));
    std::memcpy(id, uid.internal, RT_COMM_ID_BYTES);
    return 0;
}

int rt_comm_init_rank(const uint8_t id[RT_COMM_ID_BYTES], int rank, int world) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    if (!id || world < 1 || rank < 0 || rank >= world) return rt_fail(RT_ERR_ARGUMENT, "rt_comm_init_rank: bad rank / world / identifier");
    if (lib.comm) return rt_fail(RT_ERR_ARGUMENT, "rt_comm_init_rank: this device already belongs to a communicator");
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(lib.hip_device));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, RT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, uid, rank));
    lib.comm = comm; lib.comm_rank = rank; lib.comm_world = world;
    return 0;
}

int rt_comm_gather_tiles(const void* tiles_dev, uint32_t tiles_per_rank_stride, void* gathered_dev) {
    Library& lib = current_device();
    if (!lib.ready || !lib.comm) return rt_fail(RT_ERR_DEVICE, "rt_comm_gather_tiles: no communicator (rt_comm_init_rank)");
    if (!tiles_dev || (lib.comm_rank == 0 && !gathered_dev)) return rt_fail(RT_ERR_ARGUMENT, "rt_comm_gather_tiles: null buffer");
    Library* devices[MAX_DEVICES] = {};
    const uint8_t* send[MAX_DEVICES] = {};
    devices[lib.comm_rank % MAX_DEVICES] = &lib;          // only this rank's entry is used
    send[lib.comm_rank % MAX_DEVICES] = static_cast<const uint8_t*>(tiles_dev);
    if (lib.comm_world > MAX_DEVICES) return rt_fail(RT_ERR_LIMIT, "rt_comm_gather_tiles: more ranks than this library supports");
    return enqueue_gather(devices, send, lib.comm_world, static_cast<size_t>(tiles_per_rank_stride) * 768, static_cast<uint8_t*>(gathered_dev), lib.comm_rank);
}

int rt_comm_wait(void) {
    Library& lib = current_device();
    if (!lib.ready) return rt_fail(RT_ERR_DEVICE, "rt_init has not been called");
    HIP_TRY(hipStreamSynchronize(lib.comm_stream));
    return 0;
}

void rt_comm_destroy(void) {
    for (Library& lib : g_devices) {
        if (!lib.comm) continue;
        if (g_rccl.ok) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(lib.comm));
        lib.comm = nullptr; lib.comm_rank = 0; lib.comm_world = 1;
    }
}

int rt_init_devices(int n_devices) {
    if (n_devices < 1 || n_devices > MAX_DEVICES) return rt_fail(RT_ERR_ARGUMENT, "rt_init_devices: between 1 and 16 devices");
    if (g_multi_devices.load() == n_devices) return 0;
    if (g_multi_devices.load() != 0) return rt_fail(RT_ERR_ARGUMENT, "rt_init_devices: already initialised with another number of devices");
    Library* first = g_first_device.load(std::memory_order_acquire);
    if (first && first->device != 0) return rt_fail(RT_ERR_ARGUMENT, "rt_init_devices: rt_init chose a device other than 0 before");
    // RT_AMD_DEVICES_ON_ONE_GPU in the environment (tests): every context lives on HIP device 0, so that the n-device host logic -
    // replicas of the scene, one worker thread and one share of the tiles per context, rank-major gather, assembly - runs on a
    // machine with one GPU.  RCCL refuses two ranks on one GPU: the gather is then n device-to-device copies.
    const bool on_one_gpu = std::getenv("RT_AMD_DEVICES_ON_ONE_GPU") != nullptr;
    for (int d = 0; d < n_devices; ++d) if (int rc = init_device(d, on_one_gpu ? 0 : d)) return rc;
    HIP_TRY(hipSetDevice(0));
    g_copy_gather = on_one_gpu && n_devices > 1;
    if (!g_copy_gather && (n_devices > 1 || std::getenv("RT_AMD_COMM_ALWAYS"))) {   // one device needs no exchange (the variable makes it go through RCCL all the same: tests)
        for (int d = 0; d < n_devices; ++d) if (g_devices[d].comm) return rt_fail(RT_ERR_ARGUMENT, "rt_init_devices: a device already belongs to a communicator");
        if (int rc = load_rccl()) return rc;
        ncclComm_t comms[MAX_DEVICES];
        int list[MAX_DEVICES];
        for (int d = 0; d < n_devices; ++d) list[d] = d;
        RCCL_TRY(g_rccl.CommInitAll(comms, n_devices, list));
        for (int d = 0; d < n_devices; ++d) { g_devices[d].comm = comms[d]; g_devices[d].comm_rank = d; g_devices[d].comm_world = n_devices; }
        HIP_TRY(hipSetDevice(0));
    }
    g_pool.start(n_devices);
    g_multi_devices.store(n_devices);
    return 0;
}


// A frame over all devices in two halves, with the result left in device memory.  begin: every device traces its share into
// a tile buffer of its own (one host thread per device; returns when all shares are complete) and the exchange is enqueued
// on the exchange streams - every device sends its buffer to device 0 (RCCL, grouped send / receive), which scatters the
// tiles into rgb_dev (memory of device 0, rect_w x rect_h RGB8).  end: waits for that exchange.  Two frames may be under
// way, so that the tiles of frame k travel while frame k + 1 is traced:  begin(k + 1); end(k); begin(k + 2); ...
int rt_render_multi_begin(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* rgb_dev, int want_stats, int* ticket) {
    const int n = g_multi_devices.load();
    if (n < 1) return rt_fail(RT_ERR_DEVICE, "rt_render_multi: rt_init_devices has not been called");
    if (!scene || !camera || !params || !rgb_dev || !ticket) return rt_fail(RT_ERR_ARGUMENT, "rt_render_multi: null argument");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> one_at_a_time(g_multi_guard);
        const int set = g_multi_next;
        MultiFrame& frame = g_multi_frames[set];
        if (frame.busy) return rt_fail(RT_ERR_LIMIT, "rt_render_multi_begin: two frames are under way already (rt_render_multi_end)");
        const Rect rect = rect_of(*params);
        if (rect.w == 0 || rect.h == 0) return rt_fail(RT_ERR_ARGUMENT, "rt_render_multi: empty rectangle");
        const uint32_t stride = ceil_div(rect.tiles, static_cast<uint32_t>(n));
        const size_t bytes = static_cast<size_t>(stride) * 768;
        int rc_of[MAX_DEVICES] = {};
        std::string error_of[MAX_DEVICES];
        g_pool.on_every_device(n, [&](int d) {
            Library& lib = g_devices[d];
            int rc = grow(&lib.multi_tiles[set], &lib.multi_tiles_bytes[set], bytes);
            if (!rc && d == 0) rc = grow(&lib.multi_gathered[set], &lib.multi_gathered_bytes[set], bytes * n);
            if (!rc) {
                rt_render_params mine = *params;
                mine.tile_rank = static_cast<uint32_t>(d);
                mine.tile_world = static_cast<uint32_t>(n);
                rc = guarded([&] { return render_impl(scene, camera, &mine, lib.multi_tiles[set], nullptr, want_stats ? &frame.stats[d] : nullptr); });   // returns with the device's streams idle
            }
            rc_of[d] = rc;
            if (rc) error_of[d] = g_last_error;
        });
        for (int d = 0; d < n; ++d) if (rc_of[d]) return rt_fail(rc_of[d], error_of[d].c_str());
        Library& root = g_devices[0];
        DeviceBinding bound(root);
        HIP_TRY(hipEventRecord(root.multi_exchange_begin[set], root.comm_stream));
        const uint8_t* gathered = root.multi_tiles[set];
        if (root.comm) {
            Library* devices[MAX_DEVICES];
            const uint8_t* send[MAX_DEVICES];
            for (int d = 0; d < n; ++d) { devices[d] = &g_devices[d]; send[d] = g_devices[d].multi_tiles[set]; }
            if (int rc = enqueue_gather(devices, send, n, bytes, root.multi_gathered[set])) return rc;
            gathered = root.multi_gathered[set];
        } else if (g_copy_gather) {
            for (int d = 0; d < n; ++d)
                if (bytes) HIP_TRY(hipMemcpyAsync(root.multi_gathered[set] + static_cast<size_t>(d) * bytes, g_devices[d].multi_tiles[set], bytes, hipMemcpyDeviceToDevice, root.comm_stream));
            gathered = root.multi_gathered[set];
        }
        hipLaunchKernelGGL(k_assemble, dim3(ceil_div(rect.w * rect.h, BLOCK)), dim3(BLOCK), 0, root.comm_stream,
                           gathered, stride, static_cast<uint32_t>(n), rect.w, rect.h, rect.tiles_x, static_cast<uint8_t*>(rgb_dev));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(root.multi_exchange_end[set], root.comm_stream));
        for (int d = 1; d < n; ++d) {      // the senders: their tile buffers may be written again once this has happened
            DeviceBinding other(g_devices[d]);
            HIP_TRY(hipEventRecord(g_devices[d].multi_sent[set], g_copy_gather ? root.comm_stream : g_devices[d].comm_stream));
        }
        frame.busy = true; frame.n = n; frame.timed = want_stats != 0;
        g_multi_next = 1 - set;
        *ticket = set;
        return 0;
    });
}

int rt_render_multi_end(int ticket, rt_stats* stats_per_device, double* exchange_ms) {
    if (ticket < 0 || ticket > 1) return rt_fail(RT_ERR_ARGUMENT, "rt_render_multi_end: no such ticket");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> one_at_a_time(g_multi_guard);
        MultiFrame& frame = g_multi_frames[ticket];
        if (!frame.busy) return rt_fail(RT_ERR_ARGUMENT, "rt_render_multi_end: no frame under way with this ticket");
        frame.busy = false;
        Library& root = g_devices[0];
        DeviceBinding bound(root);      // the events of the exchange are device 0's
        HIP_TRY(hipEventSynchronize(root.multi_exchange_end[ticket]));
        for (int d = 1; d < frame.n; ++d) HIP_TRY(hipEventSynchronize(g_devices[d].multi_sent[ticket]));
        if (exchange_ms) {
            float ms = 0.0f;
            *exchange_ms = hipEventElapsedTime(&ms, root.multi_exchange_begin[ticket], root.multi_exchange_end[ticket]) == hipSuccess ? ms : 0.0;
        }
        if (stats_per_device) {
            if (frame.timed) std::memcpy(stats_per_device, frame.stats, sizeof(rt_stats) * static_cast<size_t>(frame.n));
            else std::memset(stats_per_device, 0, sizeof(rt_stats) * static_cast<size_t>(frame.n));
        }
        return 0;
    });
}

int rt_render_multi(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, uint8_t* rgb_out, rt_stats* stats_per_device) {
    const int n = g_multi_devices.load();
    if (n < 1) return rt_fail(RT_ERR_DEVICE, "rt_render_multi: rt_init_devices has not been called");
    if (!scene || !camera || !params || !rgb_out) return rt_fail(RT_ERR_ARGUMENT, "rt_render_multi: null argument");
    const Rect rect = rect_of(*params);
    const size_t image_bytes = static_cast<size_t>(rect.w) * rect.h * 3;
    Library& root = g_devices[0];
    {
        DeviceBinding bound(root);
        std::lock_guard<std::mutex> one_at_a_time(root.guard);
        if (int rc = grow(&root.multi_image, &root.multi_image_bytes, image_bytes ? image_bytes : 1)) return rc;
    }
    int ticket = -1;
    if (int rc = rt_render_multi_begin(scene, camera, params, root.multi_image, stats_per_device ? 1 : 0, &ticket)) return rc;
    if (int rc = rt_render_multi_end(ticket, stats_per_device, nullptr)) return rc;
    DeviceBinding bound(root);
    HIP_TRY(hipMemcpy(rgb_out, root.multi_image, image_bytes, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
