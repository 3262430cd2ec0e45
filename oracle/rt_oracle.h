/*
 * rt_oracle.h -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the parity tests,
 * the smoke test and the CPU-baseline leg of the benchmark compare against.
 * Nothing in the shipped GPU path links, loads or calls it.
 *
 * It restates, in scalar C with every f32 operation rounded separately
 * (build with -ffp-contract=off), the algorithm of the reference Go backend
 * (paths relative to the reference checkout, src/backend/):
 *   models/ray.go:23-42        NewRay
 *   models/aabb.go:62-99       AABB.RayIntersect
 *   models/triangle.go:45-103  NewTriangle, TriangleSorter, RayIntersect
 *   models/bvh.go:31-212       BuildBVH, WalkNode, GetSplitPlaneSAH
 *   process/trace.go:20-187    Trace, rayCast, getMaterialParameters
 *   utility/utility.go:34-95   ClampColor, MultiplyColor, RandomInHemisphere
 *   models/arealight.go:42-56  AreaLight.Sample
 *   models/camera.go:43-129    Camera.Initialize, GetCameraRay
 *   models/context.go:172-226  area-light discovery
 *   main.go:131-182            pixel loop + resolve
 *
 * Pinning status: the reference ships known-answer tests for the ray/AABB
 * slab test only (models/aabb_test.go:10-84); tests/golden/aabb_kat.json
 * holds those five cases and this file reproduces all of them.  Everything
 * else (triangle test, traversal, shading, BVH build, camera, light) has no
 * golden vector in the reference and the reference cannot be built in this
 * environment (no Go toolchain, js/wasm-only sources): PARITY UNPINNED for
 * those parts.  Vector arithmetic follows github.com/go-gl/mathgl v1.0.0
 * (mgl32: Dot/Cross/Len/Normalize/TransformCoordinate/QuatRotate) and the
 * float64 helpers follow Go's pure-Go math package (Cos, Pow, Exp, Log),
 * both restated from their published algorithms.
 *
 * Randomness: the reference draws from Go's global math/rand and from gonum's
 * Owen-scrambled Halton tables, neither reproducible here.  The restatement
 * takes the sample tables as input data and uses a counter-based splitmix64
 * stream per camera sample (see orc_rng_* below); the GPU path uses the same
 * definition, so RGB parity between the two is bit-exact while parity with a
 * real Go run is statistical only.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODE_F32 0  /* the reference's arithmetic: float32, unfused */
#define ORC_MODE_F64 1  /* same control flow evaluated in float64 */

typedef struct orc_scene orc_scene;

typedef struct orc_counters {
    uint64_t rays;            /* rayCast calls (trace.go:122) */
    uint64_t nodes_visited;   /* interior nodes whose children were tested */
    uint64_t box_tests;       /* AABB.RayIntersect calls */
    uint64_t tris_fetched;    /* leaf triangles iterated (before the cull) */
    uint64_t tris_tested;     /* Triangle.RayIntersect calls */
} orc_counters;

/* Same field order and meaning as the GPU library's pass description. */
typedef struct orc_camera {
    float transform[16];      /* column-major (mgl32.Mat4) */
    float projection_plane_distance;
    int32_t rays_per_pixel;
    int32_t projection;       /* 0 perspective, 1 orthographic */
    float field_of_view;      /* vertical, degrees */
    float ortographic_size;
} orc_camera;

typedef struct orc_settings {
    int32_t draw_surface_normal;
    int32_t gamma_correction;
    float gamma;
    int32_t light_sample_rays;
    int32_t bounce_limit;
    float light_intensity;
    float debug_light_size;
    int32_t force_debug_light;
    int32_t debug_light_at_camera;
    float debug_light_transform[16];
} orc_settings;

typedef struct orc_pass {
    orc_camera camera;
    int32_t total_width, total_height;
    int32_t task_id;
    int32_t pad0;
    int64_t rng_seed;
    int32_t x_offset, y_offset;
    int32_t width, height;
    orc_settings settings;
    int32_t render_key;
    int32_t pad1;
} orc_pass;

typedef struct orc_light {
    float transform[16];
    float size[2];
    float emission[3];
    float normal[3];
    int32_t found;
} orc_light;

/* ---- scalar primitives (known-answer tests) ---- */
void orc_aabb_intersect(const float bmin[3], const float bmax[3], const float o[3],
                        const float d[3], int mode, int *hit, double *tmin, double *tmax);
void orc_tri_intersect(const float v0[3], const float v1[3], const float v2[3],
                       const float o[3], const float d[3], int mode,
                       double *t, double *u, double *v);
void orc_tri_normal(const float v0[3], const float v1[3], const float v2[3], float n[3]);

/* ---- Go math restatements (float64) ---- */
double orc_go_cos(double x);
double orc_go_exp(double x);
double orc_go_log(double x);
double orc_go_pow(double x, double y);
double orc_go_min(double x, double y);
double orc_go_max(double x, double y);

/* ---- counter RNG shared by definition with the GPU path ---- */
uint64_t orc_rng_stream(int64_t seed, uint64_t path_index);
float orc_rng_f32(uint64_t stream, uint32_t counter);

/* ---- scene ---- */
/* verts: n*9 floats (v0,v1,v2); material: n ids into kd; group_index: the
 * per-group running index the reference stores in Triangle.Index
 * (context.go:118,158,164).  kd: m*3; is_light: m flags (material named
 * "Light"). */
orc_scene *orc_scene_create(const float *verts, const int32_t *material,
                            const int32_t *group_index, int32_t n,
                            const float *kd, const uint8_t *is_light, int32_t m);
void orc_scene_destroy(orc_scene *s);
/* models/context.go:138-156: n*6 floats (uv0, uv1, uv2 per triangle, input order). */
void orc_scene_set_texcoords(orc_scene *s, const float *uv);
/* models/texture.go:24-46 + context.go:85-103: the image.RGBA behind one
 * material's map_Kd (RGBA8 rows).  Returns 0, or -1 for a bad argument. */
int32_t orc_scene_set_texture(orc_scene *s, int32_t material, const uint8_t *rgba,
                              int32_t w, int32_t h);

/* literal BuildBVH (bvh.go:31-114,146-212); returns node count */
int32_t orc_build_bvh(orc_scene *s, int32_t use_bvh, int32_t max_leaf, int32_t max_depth);
/* install a tree given in preorder (Index order): per node is_leaf, start,
 * end (inclusive), bounds min/max; order[] = original triangle id at each
 * sorted array position. */
int32_t orc_set_bvh(orc_scene *s, int32_t n_nodes, const int32_t *is_leaf,
                    const int32_t *start, const int32_t *end, const float *bmin,
                    const float *bmax, const int32_t *order);
int32_t orc_node_count(const orc_scene *s);
/* preorder export; any output may be NULL */
void orc_export_bvh(const orc_scene *s, int32_t *is_leaf, int32_t *start, int32_t *end,
                    int32_t *depth, float *bmin, float *bmax, float *split, int32_t *order);

void orc_scene_light(const orc_scene *s, orc_light *out);
void orc_set_sample_tables(orc_scene *s, const double *camera_tab, int32_t n_camera,
                           const double *light_tab, int32_t n_light);

/* batch rayCast (trace.go:121-152).  rays: n*6 (origin, direction); tmax: n.
 * tri_pos = position in the sorted triangle array or -1. */
void orc_trace_rays(const orc_scene *s, const float *rays, const float *tmax, int64_t n,
                    int mode, int32_t *tri_pos, float *t, float *u, float *v,
                    orc_counters *counters);

/* camera.go:43-85 -> top_left[3], steps[2] (horizontal, vertical) */
void orc_camera_setup(const orc_camera *cam, int32_t total_w, int32_t total_h,
                      float top_left[3], float steps[2]);
/* camera.go:99-129 with the jitter taken from the camera table at
 * (path_index mod table length) */
void orc_camera_ray(const orc_scene *s, const orc_pass *pass, int32_t x, int32_t y,
                    uint64_t path_index, float ray[6]);

/* main.go:93-188: render one tile.  accum (w*h*3, may be NULL) receives the
 * per-pixel sums before the resolve; rgba (w*h*4) the resolved bytes.
 * sample_begin/sample_count select a sub-range of the spp loop (the
 * incremental mode of main.go:234-306 calls with count 1); path indices are
 * i*spp+j for the one-shot mode (incremental=0) and s*pixels+i otherwise.
 * radiance (w*h*sample_count*3, may be NULL) receives each Trace() result. */
void orc_render_tile(orc_scene *s, const orc_pass *pass, int32_t sample_begin,
                     int32_t sample_count, int32_t incremental, float *accum,
                     uint8_t *rgba, float *radiance, orc_counters *counters);
/* main.go:158-182 on an accumulated buffer */
void orc_resolve(const float *accum, int64_t pixels, int32_t divisor, int32_t gamma_on,
                 float gamma, uint8_t *rgba);

/* Light-table row per Sample() call: 0 = hashed per camera sample (default,
 * what the kernels use), 1 = the reference's single advancing index per light
 * (arealight.go:42-56; serial, for comparing image statistics). */
void orc_set_light_index_mode(orc_scene *s, int32_t mode);

/* main.go:117-154: the "trace" progress events of one render() call, each with
 * the value of context.Rays at that moment; returns how many there are. */
int32_t orc_render_progress(orc_scene *s, const orc_pass *pass, uint64_t base_rays,
                            float *progress, uint64_t *rays, int32_t max_events);

#ifdef __cplusplus
}
#endif
#endif
