/*
 * rt_amd.h -- C ABI of the MI355X-native ray-casting / path-tracing library
 * (libraytracer_amd.so).
 *
 * The library replaces ONE path of the reference Go backend: the per-tile
 * render loop and everything under it (ray generation, BVH traversal,
 * ray/box and ray/triangle tests, the shadow-ray / bounce loop and the
 * resolve).  Each entry point cites the reference interface it stands in for
 * (paths relative to the reference checkout, src/backend/).  All signatures
 * are plain C: pointers, sizes, PODs.  Functions return RT_OK (0) or a
 * negative rt_status; rt_last_error() gives the message of the calling
 * thread's last failure.  Nothing aborts across this boundary (the reference
 * maps failures to exitCode -1 + message, main.go:308-325).
 *
 * Threading: an rt_bvh is immutable after creation and may be shared by
 * threads.  An rt_scene may be shared too: its geometry is immutable, the
 * rt_trace_rays* calls are independent of each other, and the render calls,
 * which share the scene's path workspace, take turns (a lock on the host; on
 * the GPU each waits for the previous one, whatever streams they were given).
 * The rt_scene_set_* calls must not run while another thread uses the scene.
 * One scene lives on one GPU; multi-GPU callers
 * create one scene per device (the reference replicates scene + BVH per
 * WebWorker, frontend Renderer.js:213-228,281-297).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rt_status;
#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / malformed input */
#define RT_ERR_DEVICE (-2)    /* HIP runtime failure */
#define RT_ERR_NOMEM (-3)
#define RT_ERR_PARSE (-4)     /* JSON / OBJ / MTL / PNG syntax */
#define RT_ERR_STATE (-5)     /* call order (e.g. render before loadBVH) */

/* arithmetic modes of the traversal kernels */
#define RT_MODE_STRICT_F32 0  /* reference arithmetic and visit order; bit-exact */
#define RT_MODE_FP64 1        /* same control flow in float64 (validation) */
#define RT_MODE_FAST_F32 2    /* f32, near-child-first ordered traversal */
/* The same ordered traversal for rays that START within one scene diagonal of
 * the scene's bounding box (rt_scene_info: near_lo, near_hi): the step decodes
 * a record's planes relative to the ray origin, one packed instruction less per
 * plane pair, and the planes are kept far enough outside their boxes that the
 * box tests stay on the safe side of the reference's for such rays.  The
 * caller promises the region; a ray from farther away may miss a hit that
 * grazes a box.  The render entry points choose this form by themselves when
 * they are given RT_MODE_FAST_F32 and every camera ray starts in the region
 * (all other rays of a path start on the scene's surfaces). */
#define RT_MODE_FAST_NEAR_F32 3

const char *rt_last_error(void);
/* library / build identification, e.g. "raytracer_amd 0.3 gfx950" */
const char *rt_version(void);

/* ------------------------------------------------------------------ */
/* Geometry input: what RenderContext.Initialize produces              */
/* (models/context.go:105-168, models/triangle.go:45-65)               */
/* ------------------------------------------------------------------ */
typedef struct rt_mesh {
    const float *verts;           /* n_tris * 9: v0, v1, v2 */
    const int32_t *material;      /* n_tris ids into kd[] (NULL: all 0) */
    const int32_t *group_index;   /* n_tris: Triangle.Index, the per-group
                                     running index (NULL: 0..n-1) */
    const float *texcoords;       /* n_tris * 6 (NULL: none) */
    int32_t n_tris;
    const float *kd;              /* n_materials * 3 diffuse colours */
    const uint8_t *is_light;      /* n_materials flags: material named "Light" */
    const int32_t *texture;       /* n_materials texture ids or -1 (NULL: none) */
    int32_t n_materials;
} rt_mesh;

/* ------------------------------------------------------------------ */
/* BVH (host side).  Replaces models/bvh.go BuildBVH / BVH.Load and    */
/* the JSON hand-over of main.go:65-90.                                */
/* ------------------------------------------------------------------ */
typedef struct rt_bvh rt_bvh;

/* bvh.go:31-114,146-212: full-sweep SAH on 3 axes with the reference's cost
 * formula, split-plane rule, median fallback and depth / leaf limits.  The
 * result (topology, bounds, triangle order) equals the reference algorithm's;
 * the implementation is O(n log n) and multi-threaded. */
rt_status rt_bvh_build(const rt_mesh *mesh, int32_t use_bvh, int32_t max_leaf_size,
                       int32_t max_depth, rt_bvh **out);
/* main.go:78-90 + bvh.go:40-54: parse the reference's BVH JSON and re-sort
 * the triangles node by node from the stored split planes. */
rt_status rt_bvh_from_json(const rt_mesh *mesh, const char *json, size_t len, rt_bvh **out);
/* main.go:65-76: serialise as the reference's JSON ({"Root":{...}}).  Returns
 * a malloc'd NUL-terminated string in *out (free with rt_free). */
rt_status rt_bvh_to_json(const rt_bvh *bvh, char **out, size_t *len);
/* compact binary cache of the same tree (the reference caches the JSON in the
 * browser's IndexedDB, frontend Renderer.js:235-274) */
rt_status rt_bvh_save(const rt_bvh *bvh, const char *path);
rt_status rt_bvh_load(const rt_mesh *mesh, const char *path, rt_bvh **out);
int32_t rt_bvh_node_count(const rt_bvh *bvh);
int32_t rt_bvh_tri_count(const rt_bvh *bvh);
int32_t rt_bvh_max_depth(const rt_bvh *bvh);
/* Preorder (BVHNode.Index order) export; any pointer may be NULL.
 * is_leaf/start/end/depth: n_nodes; bmin/bmax: n_nodes*3; split: n_nodes*4;
 * order: n_tris, original triangle id at each position of the sorted array. */
rt_status rt_bvh_export(const rt_bvh *bvh, int32_t *is_leaf, int32_t *start, int32_t *end,
                        int32_t *depth, float *bmin, float *bmax, float *split,
                        int32_t *order);
void rt_bvh_destroy(rt_bvh *bvh);
void rt_free(void *p);

/* The flattening step on its own (host memory): what rt_scene_create uploads.
 * Replaces the pointer tree of bvh.go:17-29 and the []*Triangle of
 * triangle.go:12-31.  nodes: n_nodes*8 floats -- per node, in preorder,
 * (min.xyz, bits a)(max.xyz, bits b) with a = index of the first node after
 * the subtree and b = -1 for interior nodes, a = first triangle and
 * b = triangle count for leaves.  tris: n_tris*12 floats in sorted-array
 * order -- (Normal, V0, Edge0, Edge2*-1).  tri_material: material id, bit 30
 * set for "Light".  Any output may be NULL. */
rt_status rt_flatten(const rt_mesh *mesh, const rt_bvh *bvh, float *nodes, float *tris,
                     int32_t *tri_material);

/* The second flat view used by RT_MODE_FAST_F32: a 4-wide tree collapsed from
 * the binary one (leaves of more than four triangles, and smaller ones where
 * it pays, cut in two by array position first), one 64-byte record of 16 dwords per interior node:
 *   [0..2]  origin of the record's frame, [3..5] step per axis (extent / 255, nudged up),
 *   [6..8]  lower planes x, y, z -- child c's 8-bit offset in byte c,
 *   [9..11] upper planes x, y, z likewise (a plane is fma(offset, step, origin);
 *           decoded boxes contain the binary tree's boxes),
 *   [12..15] child refs = record index, or for a leaf
 *           0x80000000 | count << 24 | first triangle; unused slot = 0x80000000.
 * Order of the records: the 1024 with the largest frames (a top of the tree)
 * first, largest first -- the traversal kernel keeps a prefix of the array in
 * LDS --, the others depth first, a record's first interior child behind it.
 * Call with records == NULL to get the count.  *root_ref is the ref traversal
 * starts from. */
rt_status rt_flatten_ordered(const rt_mesh *mesh, const rt_bvh *bvh, float *records,
                             size_t capacity_records, size_t *n_records, uint32_t *root_ref);

/* ------------------------------------------------------------------ */
/* Scene (device side): flattened node / triangle buffers in HBM.      */
/* Replaces the pointer tree + []*Triangle of models/bvh.go:13-29 and  */
/* models/triangle.go:12-31 that WalkNode chases.                      */
/* ------------------------------------------------------------------ */
typedef struct rt_scene rt_scene;

typedef struct rt_light {           /* models/arealight.go:10-20 */
    float transform[16];            /* column-major */
    float size[2];
    float emission[3];
    float normal[3];
    int32_t found;                  /* 0: no "Light" material -> debug light */
} rt_light;

typedef struct rt_scene_info {
    int32_t n_tris, n_nodes, max_depth, n_materials;
    int32_t device;
    int32_t has_light;
    uint64_t node_bytes, tri_bytes, total_device_bytes;
    uint64_t ordered_bytes;          /* size of the 4-wide ordered view (0: none) */
    /* which flat views of the tree the scene holds besides the plain preorder
     * array: the linked reference-order view (needs leaves of at most 127
     * triangles and triangle positions below 2^24; without it the reference-order
     * kernel reads the preorder array), and the 4-wide ordered view (needs at
     * most 2^24-2 triangles; without it RT_MODE_FAST_F32 runs the reference
     * order) */
    int32_t linked_view, ordered_view;
    /* the tile renderer's current bound on camera samples traced together (what
     * rt_scene_set_max_paths set, lowered for good by a render call whose batch
     * did not fit device memory or the workspace limit), and the bytes of path
     * workspace the scene holds right now */
    uint64_t max_paths_in_flight, workspace_bytes;
    /* the box ray origins must lie in for RT_MODE_FAST_NEAR_F32 */
    float near_lo[3], near_hi[3];
} rt_scene_info;

/* context.go:172-226 on the host, no device needed: the area light of a mesh.
 * The reference finds it inside RenderContext.Initialize (context.go:180),
 * before buildBVH / loadBVH sort the triangles (main.go:65-90), so the
 * triangles are visited in input (OBJ) order and the last "Light" triangle in
 * that order sets normal and size.  `bvh` is ignored (kept for binary
 * compatibility; pass NULL). */
rt_status rt_mesh_light(const rt_mesh *mesh, const rt_bvh *bvh, rt_light *out);

/* Copies mesh + tree to device `device` (host memory stays the caller's). */
rt_status rt_scene_create(const rt_mesh *mesh, const rt_bvh *bvh, int32_t device,
                          rt_scene **out);
rt_status rt_scene_info_get(const rt_scene *scene, rt_scene_info *out);
/* context.go:172-226: the rectangular area light found from the triangles
 * whose material is named "Light". */
rt_status rt_scene_light(const rt_scene *scene, rt_light *out);
/* Sample tables consumed by camera.go:87-97 (n_camera x 2) and
 * arealight.go:42-56 (n_light x 2), values in [0,1).  The reference fills
 * them with gonum's Owen-scrambled Halton points (12345 / 100001 rows); by
 * default the library installs tables from rt_default_sample_table. */
rt_status rt_scene_set_sample_tables(rt_scene *scene, const double *camera_tab,
                                     int32_t n_camera, const double *light_tab,
                                     int32_t n_light);
/* Build-defined scrambled Halton (bases 2, 3) table: out[n*2]. */
rt_status rt_default_sample_table(int32_t n, uint64_t seed, double *out);
/* PNG bytes -> RGBA8 rows as NewTexture holds them (models/texture.go:24-46):
 * opaque images unchanged, images with alpha premultiplied the way image/draw
 * converts NRGBA to RGBA.  Call with rgba == NULL to get the size. */
rt_status rt_png_decode(const uint8_t *data, size_t size, uint8_t *rgba, size_t capacity,
                        int32_t *width, int32_t *height);
/* Texture for material lookups (models/texture.go:24-59): RGBA8 rows, top
 * row first, as image.RGBA stores them. */
rt_status rt_scene_set_texture(rt_scene *scene, int32_t texture_id, const uint8_t *rgba,
                               int32_t width, int32_t height);
/* Upper bound on camera samples traced together by the tile renderer (sets
 * the size of its device workspace, rt_workspace_bytes_per_path() per sample:
 * 360 bytes at 4 bounces and 1 light sample); 64 .. 2^29, default 67108864. */
rt_status rt_scene_set_max_paths(rt_scene *scene, uint64_t max_paths);
/* Upper bound in bytes on the device memory the tile renderer's workspace may
 * take on this scene (0 = none, the default).  A batch whose workspace would
 * be larger is halved, like one that does not fit free device memory; results
 * do not depend on the batch shape.  A workspace that is already larger than
 * the new bound is released at once.  For GPUs shared with other work. */
rt_status rt_scene_set_workspace_limit(rt_scene *scene, uint64_t max_bytes);
/* Bytes of workspace one camera sample in flight takes at these settings
 * (two ray queues of LightSampleRays+1 rays and hits, per-vertex terms for
 * BounceLimit+1 vertices, per-path words). */
uint64_t rt_workspace_bytes_per_path(int32_t light_sample_rays, int32_t bounce_limit);
/* Launch parameters of the traversal kernels, for experiments: entries of the
 * per-lane stack kept in LDS (8, 12, 16 or 24; deeper entries spill), the
 * active-lane count below which a wave refills its idle lanes (1..64),
 * persistent blocks per CU, and the number of still-descending lanes below
 * which a wave of the ordered kernel turns to the lanes that wait on a leaf,
 * and the number of lanes inside a leaf below which it turns back (both
 * 1..64).  0 for any of the three lane counts selects the kernel's own default
 * (they differ between the ordered and the reference-order kernel).  Results
 * do not depend on them. */
rt_status rt_scene_set_tuning(rt_scene *scene, int32_t stack_entries, int32_t refill_below,
                              int32_t blocks_per_cu, int32_t descend_min, int32_t leaf_min);
/* When enabled, render calls that return rt_stats run the counting variant of
 * the traversal kernel and fill nodes_visited / box_tests / tris_fetched /
 * tris_tested (slower; for the roofline byte model, not for timed runs). */
rt_status rt_scene_set_counters(rt_scene *scene, int32_t enabled);
void rt_scene_destroy(rt_scene *scene);

/* ------------------------------------------------------------------ */
/* Batch ray cast.  Replaces process/trace.go:121-152 rayCast (and the */
/* WalkNode / RayIntersect calls under it) for n rays at once.         */
/* ------------------------------------------------------------------ */
typedef struct rt_ray {             /* models/ray.go:9-21; 32 bytes */
    float origin[3];
    float tmax;                     /* rayCast's initialTmin */
    float dir[3];
    float pad;
} rt_ray;

typedef struct rt_hit {             /* process/trace.go:11-17; 16 bytes */
    int32_t tri_pos;                /* position in the sorted triangle array, -1 none */
    float t, u, v;                  /* t == tmax when nothing closer was hit */
} rt_hit;

typedef struct rt_stats {
    uint64_t rays;                  /* rayCast count (context.Rays, trace.go:122) */
    uint64_t nodes_visited;         /* filled only when counters are requested */
    uint64_t box_tests;
    uint64_t tris_fetched;
    uint64_t tris_tested;
    double kernel_ms;               /* device time of the traversal kernels */
    double total_ms;                /* device time of the whole call */
    uint64_t kernel_launches;       /* traversal launches kernel_ms is the sum of */
} rt_stats;

/* Host buffers in, host buffers out (copies through the device). */
rt_status rt_trace_rays(rt_scene *scene, const rt_ray *rays, size_t n, rt_hit *hits,
                        int32_t mode, int32_t want_counters, rt_stats *stats);
/* Device buffers already resident in HBM; asynchronous on `stream`
 * (a hipStream_t; NULL = the default stream).  No host synchronisation. */
rt_status rt_trace_rays_device(rt_scene *scene, const void *d_rays, size_t n, void *d_hits,
                               int32_t mode, void *stream);
/* Same with per-launch traversal counters accumulated into d_counters
 * (8 x uint64 on the device: rays, nodes_visited, box_tests, tris_fetched,
 * tris_tested, then three wave-level iteration counts of the ordered kernel:
 * descend steps, leaf steps, lanes active in leaf steps). */
rt_status rt_trace_rays_device_counted(rt_scene *scene, const void *d_rays, size_t n,
                                       void *d_hits, int32_t mode, void *d_counters,
                                       void *stream);

/* ------------------------------------------------------------------ */
/* Tile render.  Replaces main.go:93-188 render() minus the JSON.      */
/* ------------------------------------------------------------------ */
typedef struct rt_camera {          /* models/camera.go:19-31 */
    float transform[16];
    float projection_plane_distance;
    int32_t rays_per_pixel;
    int32_t projection;             /* 0 perspective, 1 orthographic */
    float field_of_view;
    float ortographic_size;
} rt_camera;

typedef struct rt_settings {        /* models/settings.go:5-16 */
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
} rt_settings;

typedef struct rt_pass {            /* models/context.go:40-52 */
    rt_camera camera;
    int32_t total_width, total_height;
    int32_t task_id;
    int32_t pad0;
    int64_t rng_seed;
    int32_t x_offset, y_offset;
    int32_t width, height;
    rt_settings settings;
    int32_t render_key;
    int32_t pad1;
} rt_pass;

/* camera.go:43-85: projection-plane corner and pixel steps of a pass. */
rt_status rt_camera_setup(const rt_camera *cam, int32_t total_width, int32_t total_height,
                          float top_left[3], float steps[2]);

/* context.go:245-275 (RenderPass.Initialize: identity camera fallback, debug
 * light, emission = LightIntensity) + camera.go:43-85, evaluated on the host:
 * fills the opaque parameter block the kernels receive for a pass (at most 512
 * bytes).  For tools and tests; the render calls do this themselves. */
rt_status rt_pass_params(const rt_light *scene_light, const rt_pass *pass, void *out,
                         size_t out_size);

/* One whole tile: RaysPerPixel samples per pixel, accumulate, resolve.
 * accum_rgb (width*height*3 floats, may be NULL) receives the per-pixel sums
 * before the resolve, rgba (width*height*4) the image.RGBA bytes.  Host
 * pointers. */
rt_status rt_render_tile(rt_scene *scene, const rt_pass *pass, int32_t mode,
                         float *accum_rgb, uint8_t *rgba, rt_stats *stats);
/* Device-resident variant: d_accum (width*height*3 f32, zeroed by the call)
 * and d_rgba (width*height*4) are device pointers; asynchronous on `stream`
 * except for the ray-count read-back when stats != NULL. */
rt_status rt_render_tile_device(rt_scene *scene, const rt_pass *pass, int32_t mode,
                                void *d_accum, void *d_rgba, rt_stats *stats, void *stream);
/* Several tiles of one frame in one call: what a worker does by calling
 * render() once per task it pulls (frontend Renderer.js:565-568), batched so
 * that small tiles still fill the GPU.  The passes must share camera, settings
 * and frame size; offsets, sizes, TaskID and RNGSeed differ.  Each tile's
 * result equals its own rt_render_tile_device call bit for bit.  Pixels of tile
 * i start at offset sum_{j<i} width_j*height_j in d_accum (x3 floats) and
 * d_rgba (x4 bytes). */
rt_status rt_render_tiles_device(rt_scene *scene, const rt_pass *passes, int32_t n_passes,
                                 int32_t mode, void *d_accum, void *d_rgba, rt_stats *stats,
                                 void *stream);
/* main.go:234-306 incrementalRender: trace samples [sample_begin,
 * sample_begin+sample_count) of every pixel into the caller-owned device
 * accumulation buffer (NOT zeroed), then resolve with divisor
 * sample_begin+sample_count into d_rgba (may be NULL to skip). */
rt_status rt_render_accumulate(rt_scene *scene, const rt_pass *pass, int32_t mode,
                               int32_t sample_begin, int32_t sample_count, void *d_accum,
                               void *d_rgba, rt_stats *stats, void *stream);
/* One tile like rt_render_tile_device, for callers that cannot allocate: code
 * under HIP graph capture, tight serving loops that render one small tile per
 * request, harnesses that forbid allocation.  It has no counterpart in the
 * reference.  The call allocates nothing (no hipMalloc / hipFree, no event or
 * stream creation): the path workspace is a static device array of
 * RT_STATIC_WORKSPACE_BYTES that every GPU the library runs on holds until
 * the process ends.  The tile is traced in as many sample batches as that
 * array allows; a tile whose pixel count times
 * rt_workspace_bytes_per_path(LightSampleRays, BounceLimit) exceeds it is
 * refused with RT_ERR_INVALID.  Results equal rt_render_tile_device's bit for
 * bit.  No statistics are returned (they would need a read-back).  All calls
 * on one GPU share that GPU's array: calls from several threads take turns on
 * the host and each waits on the GPU for the one before it, but calls issued
 * while an earlier one is still being CAPTURED into a graph, or replays of
 * captured graphs that overlap, race on it.  d_accum (width*height*3 f32,
 * zeroed by the call) and d_rgba (width*height*4) are device pointers. */
#define RT_STATIC_WORKSPACE_BYTES (64u << 20)
rt_status rt_render_tile_static(rt_scene *scene, const rt_pass *pass, int32_t mode,
                                void *d_accum, void *d_rgba, void *stream);
/* main.go:158-182 on a device accumulation buffer. */
rt_status rt_resolve_device(const void *d_accum, size_t pixels, int32_t divisor,
                            int32_t gamma_on, float gamma, void *d_rgba, void *stream);

/* Measurement aid (no counterpart in the reference): the rate, in records
 * per second, at which every lane of the GPU can chase random 64-byte records
 * through a table of table_bytes (dependent loads, four 16-byte loads per
 * record, full occupancy) -- the access pattern of tree traversal in
 * isolation, and so the ceiling the cache hierarchy sets for it.  Allocates
 * and frees the table; takes a few milliseconds. */
rt_status rt_gather_probe(int32_t device, size_t table_bytes, double *records_per_s);

#ifdef __cplusplus
}
#endif
#endif
