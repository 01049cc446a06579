/*
 * rt_amd.h - C ABI of the MI355X (gfx950) per-pixel render loop.
 *
 * This is the drop-in boundary for the reference renderer's hot path.  The
 * reference (LewisBray/RayTracer) has no FFI: its whole program is one
 * translation unit of static functions (Source/main.cpp:18-23).  The entry
 * points below are what a maintainer would bind in place of
 *   - the per-row driver   render_scanline(...)      Source/render_work_queue.h:28-36
 *   - the work queue       thread_proc / push_entry  Source/render_work_queue.cpp:3-13,59-80
 *   - the tracing kernel   intersect(Ray,Scene,int)  Source/ray_tracing.h:137
 *   - primary rays         ray_direction_through_pixel  Source/ray_tracing.h:127-135
 *   - the AABB cull        intersect(Ray,AABB)       Source/ray_tracing.h:51
 * and, for the front end that feeds them,
 *   - parse_input_file(const char*)                  Source/input_parsing.h:20
 *   - the camera set-up in main()                    Source/main.cpp:44-57
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All functions return 0 on success and a non-zero code on failure, in which
 * case rt_last_error() returns a static / thread-local message (the reference
 * reports errors the same way: a `const char* error`, Source/input_parsing.h:15-18).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene data: exactly the reference's in-memory layout ------------------
 * Every pointer addresses host memory laid out like the matching std::vector
 * in the reference's `Scene` (Source/ray_tracing.h:88-106), so a binding passes
 * `scene.triangle8s.data()` etc. unchanged:
 *   Triangle8  = 9 x float[8]  (a.x a.y a.z a_to_b.x .. a_to_c.z), 288 B   ray_tracing.h:18-22
 *   Sphere8    = 4 x float[8]  (centre.x centre.y centre.z radius), 128 B   ray_tracing.h:24-27
 *   Mat3x4x8   = float[3][4][8], 384 B                                     ray_tracing.h:14-16
 *   Material   = 10 floats (diffuse rgb, specular rgb, emission rgb, shininess), 40 B  ray_tracing.h:72-77
 *   PointLightSource = 6 floats (position xyz, colour rgb)                 ray_tracing.h:67-70
 * The last batch of 8 is zero padded, as the reference's parser leaves it
 * (Source/input_parsing.cpp:260-262,354-357,418-420).
 */
typedef struct rt_scene_desc {
    const float* triangle8s;                    /* n_triangle8 * 72 floats */
    const float* triangle_materials;            /* n_triangles * 10 floats */
    const float* sphere8s;                      /* n_sphere8 * 32 floats */
    const float* sphere_materials;              /* n_spheres * 10 floats */
    const float* ellipsoid8_inverse_transforms; /* n_ellipsoid8 * 96 floats */
    const float* ellipsoid8_transforms;         /* n_ellipsoid8 * 96 floats */
    const float* ellipsoid_materials;           /* n_ellipsoids * 10 floats */
    const float* point_light_sources;           /* n_point_lights * 6 floats */
    uint64_t n_triangle8;
    uint64_t n_triangles;
    uint64_t n_sphere8;
    uint64_t n_spheres;
    uint64_t n_ellipsoid8;
    uint64_t n_ellipsoids;
    uint64_t n_point_lights;
    float bounding_box[6];        /* min_x max_x min_y max_y min_z max_z  (maths.h:55-62) */
    float ambient[3];
    float attenuation[3];         /* constant, linear, quadratic (ray_tracing.h:79-83) */
    float directional_light[6];   /* unit direction xyz, colour rgb (ray_tracing.h:62-65) */
    int32_t has_directional_light;
    int32_t reserved_;
} rt_scene_desc;

/* Camera as render_scanline receives it: the eye, the basis main() derives
 * (Source/main.cpp:44-52) and tan(fov/2) per axis (Source/main.cpp:54-57). */
typedef struct rt_camera_desc {
    float eye[3];
    float basis_i[3];
    float basis_j[3];
    float basis_k[3];
    float half_world[2];          /* width, height */
} rt_camera_desc;

/* What to render.  width/height are the full image (they define the rays);
 * the rectangle [rect_x0, rect_x0+rect_w) x [rect_y0, rect_y0+rect_h) is the
 * part that is traced and returned (rect_w == 0 means the whole image).
 * The rectangle is cut into 16x16-pixel tiles, tiles_x = ceil(rect_w / 16) per row.  Tile number t lies in row
 * t / tiles_x at column (t % tiles_x - row) mod tiles_x: rows are rotated against each other, so that the tiles of one
 * rank lie on diagonals.  The call renders tiles t with  t % tile_world == tile_rank  (tile_world == 0 or 1:
 * all of them).  This replaces the scanline queue of
 * Source/render_work_queue.cpp:59-80. */
typedef struct rt_render_params {
    uint32_t width;
    uint32_t height;
    int32_t  max_bounce_count;    /* FileInfo::max_recursion_depth */
    uint32_t rect_x0, rect_y0, rect_w, rect_h;
    uint32_t tile_rank, tile_world;
    uint32_t chunk_tiles;         /* tiles traced per batch of launches; 0 = default */
    uint32_t flags;               /* RT_FLAG_* */
    uint32_t reserved_;
} rt_render_params;

#define RT_TILE 16u
#define RT_SAMPLES_PER_PIXEL 16u  /* SQRT_SAMPLES_PER_PIXEL^2, render_work_queue.cpp:24-29 */

#define RT_FLAG_NO_FILTER 1u      /* brute force: every ray is tested against every primitive, no bounding hierarchy */
#define RT_FLAG_WALK_ALL  2u      /* walk the bounding hierarchy of every primitive kind, also of kinds so small that the
                                     library would scan them by brute force (same results; for tests and tuning) */
#define RT_FLAG_PLAIN_TEST 8u     /* brute-force scans evaluate every test exactly as written (triangles: three IEEE divisions per pair) instead of
                                     deciding most pairs by a conservative pre-test first (same results; for measurements) */
#define RT_FLAG_EXACT_STAGES 16u  /* brute-force triangle scans evaluate the test in exact stages (the division without its range handling, comparisons of
                                     the numerators) instead of conservative pre-tests followed by the test as written (same results; for measurements) */
#define RT_FLAG_DEEP_WALK 4u      /* walk every hierarchy the way the library walks big ones (closest-hit rays from the super
                                     nodes, shadow rays depth first) whatever its size (same results; for tests and tuning) */

/* Primary-hit id of one sample, as stored in the id buffer. */
#define RT_ID_KIND_TRIANGLE  0u
#define RT_ID_KIND_SPHERE    1u
#define RT_ID_KIND_ELLIPSOID 2u
#define RT_ID_MISS   0xFFFFFFFFu  /* ray entered the scene box and hit nothing */
#define RT_ID_CULLED 0xFFFFFFFEu  /* rejected by the scene bounding box (render_work_queue.cpp:45-46) */
#define RT_ID_MAKE(kind, index) (((uint32_t)(kind) << 30) | (uint32_t)(index))

/* Exact work counters of one render call. */
typedef struct rt_stats {
    uint64_t samples;             /* samples generated */
    uint64_t culled;              /* rejected by the bounding box */
    uint64_t closest_rays;        /* ray segments scanned for the closest hit (all bounces) */
    uint64_t hits;                /* of which hit something */
    uint64_t shadow_rays;         /* shadow ray segments */
    uint64_t shadow_blocked;      /* of which blocked */
    uint64_t closest_tests[3];    /* ray-primitive PAIRS of the closest-hit search: rays x unpadded count of triangles, spheres, ellipsoids.
                                     With RT_FLAG_NO_FILTER every pair is visited by a brute-force scan; in the default mode the
                                     bounding hierarchy decides most pairs without visiting them (pair_pretests, queued_pairs and
                                     full_tests count what was executed) */
    uint64_t shadow_tests[3];     /* ray-primitive tests, shadow rays, up to the point the ray's wave stopped scanning (brute-force
                                     triangle scan: up to the tile after which the ray's lane was given another ray) */
    double   device_ms;           /* HIP-event time of all launches of this call */
    double   closest_ms;          /* HIP-event time summed over the closest-hit scan launches */
    double   shadow_ms;           /* HIP-event time summed over the shadow scan launches */
    uint64_t closest_launches;    /* launches timed in closest_ms */
    uint64_t shadow_launches;     /* launches timed in shadow_ms */
    uint64_t full_tests;          /* ray-triangle pairs that went through the full test after the two-stage pre-test (hierarchy scans only) */
    uint64_t pair_pretests[2];    /* per-ray box tests of the hierarchy scans (top nodes the per-wave test kept, and children of kept nodes): closest-hit scans, shadow scans */
    uint64_t queued_pairs[2];     /* (ray, primitive) pairs the boxes left over, i.e. sent to the two-stage pre-test (triangles) or the full test: closest-hit scans, shadow scans */
    uint64_t executed_flop[2];    /* FLOP the hierarchy scans executed in per-ray tests (box test 24, two-stage pre-test 40, full test 54 triangle / 19 sphere / 95 or 71 ellipsoid): closest-hit scans, shadow scans */
    double   closest_kind_ms[3];  /* HIP-event time of the closest-hit scan launches per kind: triangle, sphere, ellipsoid (their sum is closest_ms
                                     only when launches do not overlap: RT_FLAG_NO_FILTER runs them one after the other) */
    double   shadow_kind_ms[3];   /* the same for the shadow scans */
    uint64_t closest_kind_launches[3];
    uint64_t shadow_kind_launches[3];
    uint64_t walk_waves[2];         /* hierarchy scans, closest-hit / shadow: wavefronts that walked (64 rays each; a wavefront of the depth-first
                                       shadow walk of large hierarchies takes up to 2048 rays, 64 at a time) */
    uint64_t first_level_nodes[2];  /* nodes of the level a walk starts at whose per-ray box tests ran, summed over those wavefronts
                                       (the depth-first walk has no such level: it adds nothing here) */
    uint64_t walk_steps[2];         /* steps of up to 64 (ray, node) pairs, one pair per lane, below that level: 64 * walk_steps against the
                                       pairs walked ((pair_pretests - per-ray tests of the first level) / 16) is the fill of the lanes; for the
                                       depth-first walk a step serves either the lanes that hold a node or those that hold a leaf */
    uint64_t brute_wave_steps[2][4]; /* brute-force triangle scans in their default form, closest-hit / shadow: (wavefront, triangle) steps in units of 64 pairs (a step of a
                                       scan that takes two rays per lane counts twice) in all, of which went past the sign test of the two dot
                                       products (rays of one origin skip that test), of which reached the second pre-test, of which reached the test as written */
} rt_stats;

/* Thread safety: a GPU has one set of streams and ray-state arenas per frame slot; rt_scene_create, the render calls,
 * rt_render_tiles_begin / _end, rt_render_multi_begin / _end and rt_assemble_tiles_device take locks of the library, so
 * concurrent calls from any threads are safe and run one after the other (the reference's render_scanline is re-entrant;
 * parallelism here is inside a call).  Bounding records that have to be rebuilt for a camera far outside the scene are
 * rebuilt when no frame that reads them is in flight.  A frame begun and never ended is completed when the library is unloaded.
 * HIP's current device is a property of the thread: a calling thread need not have made the library's device current, every
 * entry point that works on the device does so for the length of the call and puts the thread's device back as it found it. */
typedef struct rt_scene rt_scene;          /* opaque: device-resident scene */
typedef struct rt_file_info rt_file_info;  /* opaque: a parsed .test file */

/* ---- library ---------------------------------------------------------------*/
/* Select the HIP device this process renders on and create its stream. */
int rt_init(int device);
/* Message of the last failing call on this thread. */
const char* rt_last_error(void);
/* Number of HIP devices visible, or -1. */
int rt_device_count(void);

/* ---- scene upload (replaces nothing in the reference: it is the H2D step) ---*/
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* ---- the render loop ---------------------------------------------------------
 * rt_render: replacement for the loop `for y: render_scanline(...)` driven by
 * thread_proc (Source/render_work_queue.cpp:15-80).  rgb_out (host) receives
 * rect_h rows of rect_w RGB8 pixels, row 0 on top, as Image::pixels does
 * (render_work_queue.cpp:52-55).  primary_id_out (host, nullable) receives
 * rect_w*rect_h*16 ids, index (y*rect_w + x)*16 + sample.
 * With tile_world > 1 only this rank's tiles are written; other pixels are left untouched. */
int rt_render(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params,
              uint8_t* rgb_out, uint32_t* primary_id_out, rt_stats* stats);

/* Same, but the result stays in device memory as a compact tile buffer:
 * tile s of this rank (global tile t = s*tile_world + tile_rank) occupies bytes
 * [s*768, (s+1)*768) as 16 rows of 16 RGB8 pixels.  tiles_dev must hold
 * rt_rank_tile_count(params) * 768 bytes.  Used by the multi-GPU path, whose
 * RCCL gather consumes device buffers, and by the benchmark. */
int rt_render_tiles_device(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params,
                           void* tiles_dev, rt_stats* stats);
/* Frames in flight: rt_render_tiles_device in two halves.  begin starts the frame and returns a ticket; end waits for it and
 * returns its status and statistics.  Up to two frames may be in flight per GPU (each has its own streams and ray-state arena):
 * the first launches of one frame then run beside the thin last launches of the other, which matters when a frame is short
 * (one rank's share of a split frame).  tiles_dev must stay valid until end; the frames' results do not depend on each other. */
int rt_render_tiles_begin(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* tiles_dev, int* ticket);
int rt_render_tiles_end(int ticket, rt_stats* stats);
/* rt_render_tiles_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
 * runs between the captures of HIP graphs); it has no counterpart in the reference.  The call allocates, creates and frees nothing,
 * but it does synchronise: it launches on the library's own streams, reads the ray count back after every bounce and
 * returns with those streams idle, so it cannot itself be captured into a HIP graph.
 * the per-sample state lives in a fixed device array of the library (about 37 MB per GPU, held until the process ends),
 * so this rank's share of the rectangle may be at most RT_NOALLOC_MAX_TILES tiles (e.g. 128x128 pixels); a larger one is
 * refused with an error.  Same tile buffer, same bytes as rt_render_tiles_device; no statistics.  Every call on a GPU
 * uses the same array: calls are serialised inside the library, and the array's contents do not outlive a call. */
#define RT_NOALLOC_MAX_TILES 64
int rt_render_tiles_noalloc(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params, void* tiles_dev);
/* Number of tiles rank `tile_rank` of `tile_world` renders for these params. */
uint32_t rt_rank_tile_count(const rt_render_params* params);
/* Total tiles of the rectangle. */
uint32_t rt_total_tile_count(const rt_render_params* params);
/* Scatter gathered tile buffers (rank-major: rank r's buffer at offset
 * r * tiles_per_rank_stride * 768 bytes) into a row-major rect_w x rect_h RGB8
 * image.  Both pointers are device memory. */
int rt_assemble_tiles_device(const void* gathered_tiles_dev, uint32_t tiles_per_rank_stride,
                             const rt_render_params* params, void* rgb_dev);

/* ---- more than one GPU -----------------------------------------------------------
 * The reference hands scanlines to the threads of one machine (Source/render_work_queue.cpp:59-80, Source/main.cpp:77-90).
 * Here tile t of the rectangle belongs to GPU t % n, the scene is replicated, and the only exchange is every GPU sending
 * its compact tile buffer to the first one over xGMI (RCCL, grouped send / receive), which scatters them into the image.
 * Two ways to set it up:
 *  (1) ONE PROCESS, n GPUs: rt_init_devices(n) (devices 0..n-1, one host thread and one set of streams per device, started
 *      once and kept, as the reference starts its workers once, Source/main.cpp:77-90; one RCCL communicator from
 *      ncclCommInitAll); rt_scene_create then replicates the scene on every device, and rt_render_multi stands where
 *      rt_render does: same image bytes.  stats_per_device: n entries, or null.
 *      rt_render_multi_begin / rt_render_multi_end are the same frame in two halves with the image left in memory of
 *      device 0 (rgb_dev: rect_w x rect_h RGB8): begin returns when every device has traced its share and the exchange
 *      (gather to device 0 + scatter into rgb_dev) is enqueued; end waits for that exchange and returns the per-device
 *      statistics (if begin was given want_stats) and the device time of the exchange.  Up to two frames may be under way,
 *      so that the tiles of frame k travel while frame k + 1 is traced: begin(k + 1); end(k); begin(k + 2); ...
 *      (For tests on a machine with one GPU: with RT_AMD_DEVICES_ON_ONE_GPU set in the environment rt_init_devices(n) creates
 *      its n device contexts on HIP device 0 and the gather is n device-to-device copies, RCCL taking one rank per GPU.)
 *  (2) ONE PROCESS PER GPU (rank r renders with tile_rank = r, tile_world = n): rank 0 calls rt_comm_unique_id and hands
 *      the identifier to the others by any means the host has; every rank calls rt_comm_init_rank after rt_init;
 *      after rt_render_tiles_device has returned, rt_comm_gather_tiles enqueues the exchange of the tile buffers on a
 *      stream of its own (gathered_dev, world * stride * 768 bytes, is needed on rank 0 only) and returns at once, so the next
 *      frame can be traced meanwhile; rt_comm_wait waits for it; rank 0 then calls rt_assemble_tiles_device. */
#define RT_COMM_ID_BYTES 128
int rt_init_devices(int n_devices);
int rt_render_multi(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params,
                    uint8_t* rgb_out, rt_stats* stats_per_device);
int rt_render_multi_begin(const rt_scene* scene, const rt_camera_desc* camera, const rt_render_params* params,
                          void* rgb_dev, int want_stats, int* ticket);
int rt_render_multi_end(int ticket, rt_stats* stats_per_device, double* exchange_ms);
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_init_rank(const uint8_t id[RT_COMM_ID_BYTES], int rank, int world);
int rt_comm_gather_tiles(const void* tiles_dev, uint32_t tiles_per_rank_stride, void* gathered_dev);
int rt_comm_wait(void);
void rt_comm_destroy(void);

/* ---- front end (host) --------------------------------------------------------*/
/* parse_input_file (Source/input_parsing.cpp:96-603).  On a parse error returns
 * non-zero and rt_last_error() is the reference's message for that error. */
int rt_parse_input_file(const char* filename, rt_file_info** out);
void rt_file_info_destroy(rt_file_info* info);
/* Views into the parsed FileInfo (Source/input_parsing.h:8-13); valid until destroy. */
const rt_scene_desc* rt_file_info_scene(const rt_file_info* info);
uint32_t rt_file_info_width(const rt_file_info* info);
uint32_t rt_file_info_height(const rt_file_info* info);
int32_t rt_file_info_max_recursion_depth(const rt_file_info* info);
const char* rt_file_info_output_filename(const rt_file_info* info);
/* camera[11] = eye xyz, look_at xyz, up xyz, field_of_view x, y (ray_tracing.h:29-39) */
void rt_file_info_camera(const rt_file_info* info, float camera[11]);
/* Override the image size after parsing, recomputing field_of_view.x the way the
 * `camera` command does (Source/input_parsing.cpp:215-216). */
void rt_file_info_set_size(rt_file_info* info, uint32_t width, uint32_t height);

/* Camera basis and half image extents, main.cpp:44-57. camera11 as above. */
void rt_camera_setup(const float camera11[11], rt_camera_desc* out);

/* Whole reference flow for one file: parse, set up the camera, upload, render
 * (Source/main.cpp:33-91).  rgb_out holds width*height*3 bytes. */
int rt_render_file_info(const rt_file_info* info, uint8_t* rgb_out, rt_stats* stats);

/* Write an RGB8 image as PNG (stands where main.cpp:93 calls stbi_write_png;
 * pixel-identical, not byte-identical): Sub filter, deflate on up to 16 host threads. */
int rt_write_png(const char* filename, uint32_t width, uint32_t height, const uint8_t* rgb);
/* The same behind the caller's back: begin copies the pixels and returns while a thread encodes and writes them, so that
 * the next frame can be traced meanwhile; end waits for the file and returns the writer's status.  One image at a time. */
int rt_write_png_begin(const char* filename, uint32_t width, uint32_t height, const uint8_t* rgb);
int rt_write_png_end(void);

/* ---- test hook -----------------------------------------------------------------
 * Evaluates ONE device routine of the render loop on n items, so that the GPU forms of the
 * reference's scalar and AVX routines can be compared bit for bit with known answers
 * (maths.cpp:9-159 fp_pow / fp_min / fp_max; ray_tracing.cpp:460-473, :265-297, :129-165, :169-206, :105-126, :208-263, :336-344, :555-579).
 * All pointers are host memory.  Layout per operation (floats):
 *   FP_POW, FP_MIN, FP_MAX, DIVIDE, DIVIDE_IN_RANGE: a[n], b[n] -> out[n]
 *   PRIMARY_RAY: a = eye, basis i, j, k, half_world w h, half width and half height in pixels (16);
 *                b = pixel x, pixel y, x offset, y offset per item -> out = direction (3 per item)
 *   SCENE_BOX:   a = min_x max_x min_y max_y min_z max_z; b = origin, direction per item -> out = min, max distance
 *   TRIANGLE, TRIANGLE_STAGED: a = a, a_to_b, a_to_c per item (9); b = origin, direction per item -> out[n]
 *                (distance or FLT_MAX; STAGED evaluates the staged form the brute-force scans use and reports
 *                NaN where that form declares itself unsure and the scans fall back to the plain form)
 *   TRIANGLE_PRETEST: a, b as for TRIANGLE -> 3 floats per item: the result of the default form of the brute-force scans for the pair
 *                (FLT_MAX where a step decided that the pair is a certain miss), the step that decided (0 sign test, 1 first pre-test,
 *                2 second pre-test, 3 the test as written), and 1 / 0 for whether the form for rays of one origin keeps the pair up to the test as written
 *   SPHERE:      a = centre, radius per item (4); b = origin, direction per item -> out[n]
 *   ELLIPSOID, ELLIPSOID_STAGED: a = rows of the inverse transform, rows of the transform per item (24); b = origin, direction
 *                per item -> distance or FLT_MAX (closest-hit body, ray_tracing.cpp:555-579); STAGED evaluates the form the
 *                brute-force scans use and returns 2 floats per item: the result (FLT_MAX where a stage decided that the pair
 *                cannot produce a distance below FLT_MAX) and the stage that decided (0, 1, 2; 3 = the test as written,
 *                for records and rays outside the staged form's range)
 *   ELLIPSOID_SHADOW, ELLIPSOID_SHADOW_STAGED: a = rows of the inverse transform (12), light position (3), 1 for a point
 *                light / 0 for a directional one, per item; b = origin, direction per item -> 1 if the ellipsoid blocks the ray
 *                (ray_tracing.cpp:376-401, :432-454), else 0; STAGED returns (result, stage) as above
 *   ELLIPSOID_PARTS: a = rows of the inverse transform per item (12); b = origin, direction per item -> 8 floats per item:
 *                origin in ellipsoid space (Mat3x4AVX * Vec3AVX, ray_tracing.cpp:105-126), direction in ellipsoid space
 *                (transform_direction, :208-227), ellipsoid_is_absent as 1 / 0 (:336-344), intersect_with_unit_sphere of
 *                that origin and the normalised direction (:229-263) */
#define RT_DEBUG_FP_POW 0
#define RT_DEBUG_FP_MIN 1
#define RT_DEBUG_FP_MAX 2
#define RT_DEBUG_DIVIDE 3
#define RT_DEBUG_DIVIDE_IN_RANGE 4
#define RT_DEBUG_PRIMARY_RAY 5
#define RT_DEBUG_SCENE_BOX 6
#define RT_DEBUG_TRIANGLE 7
#define RT_DEBUG_TRIANGLE_STAGED 8
#define RT_DEBUG_SPHERE 9
#define RT_DEBUG_ELLIPSOID 10
#define RT_DEBUG_ELLIPSOID_STAGED 11
#define RT_DEBUG_ELLIPSOID_SHADOW 12
#define RT_DEBUG_ELLIPSOID_SHADOW_STAGED 13
#define RT_DEBUG_ELLIPSOID_PARTS 14
#define RT_DEBUG_TRIANGLE_PRETEST 15
int rt_debug_eval(int op, uint32_t n, const float* a, const float* b, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
