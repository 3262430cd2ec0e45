// capi.cpp -- the extern "C" surface declared in include/rt_amd.h.  Thin
// argument checking + forwarding; no exception or abort crosses this layer.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "backend.h"
#include "rt_device.h"
#include "rt_internal.h"

namespace rt {
rt_status bvh_to_json(const Bvh &bvh, std::string &out);
rt_status bvh_from_json(const rt_mesh &mesh, const char *json, size_t len, Bvh &out);
rt_status bvh_save(const Bvh &bvh, const char *path);
rt_status bvh_load(const rt_mesh &mesh, const char *path, Bvh &out);
}  // namespace rt

using rt::fail;

#define RT_GUARD_BEGIN try {
#define RT_GUARD_END                                                        \
    }                                                                       \
    catch (const std::bad_alloc &) {                                        \
        return fail(RT_ERR_NOMEM, "out of host memory");                    \
    }                                                                       \
    catch (const std::exception &e) {                                       \
        return fail(RT_ERR_INVALID, std::string("internal error: ") + e.what()); \
    }                                                                       \
    catch (...) {                                                           \
        return fail(RT_ERR_INVALID, "internal error");                      \
    }

extern "C" {

const char *rt_last_error(void) { return rt::last_error_text().c_str(); }
const char *rt_version(void) { return "raytracer_amd 0.3 gfx950"; }

void rt_free(void *p) { std::free(p); }

rt_status rt_bvh_build(const rt_mesh *mesh, int32_t use_bvh, int32_t max_leaf_size,
                       int32_t max_depth, rt_bvh **out) {
    RT_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    std::unique_ptr<rt_bvh> b(new rt_bvh);
    st = rt::build_bvh(*mesh, use_bvh != 0, max_leaf_size, max_depth, b->bvh);
    if (st != RT_OK) return st;
    *out = b.release();
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_bvh_from_json(const rt_mesh *mesh, const char *json, size_t len, rt_bvh **out) {
    RT_GUARD_BEGIN
    if (!out || !json) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    std::unique_ptr<rt_bvh> b(new rt_bvh);
    st = rt::bvh_from_json(*mesh, json, len, b->bvh);
    if (st != RT_OK) return st;
    *out = b.release();
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_bvh_to_json(const rt_bvh *bvh, char **out, size_t *len) {
    RT_GUARD_BEGIN
    if (!bvh || !out) return fail(RT_ERR_INVALID, "NULL argument");
    std::string s;
    rt_status st = rt::bvh_to_json(bvh->bvh, s);
    if (st != RT_OK) return st;
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (!p) return fail(RT_ERR_NOMEM, "out of host memory");
    std::memcpy(p, s.c_str(), s.size() + 1);
    *out = p;
    if (len) *len = s.size();
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_bvh_save(const rt_bvh *bvh, const char *path) {
    RT_GUARD_BEGIN
    if (!bvh || !path) return fail(RT_ERR_INVALID, "NULL argument");
    return rt::bvh_save(bvh->bvh, path);
    RT_GUARD_END
}

rt_status rt_bvh_load(const rt_mesh *mesh, const char *path, rt_bvh **out) {
    RT_GUARD_BEGIN
    if (!out || !path) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    std::unique_ptr<rt_bvh> b(new rt_bvh);
    st = rt::bvh_load(*mesh, path, b->bvh);
    if (st != RT_OK) return st;
    *out = b.release();
    return RT_OK;
    RT_GUARD_END
}

int32_t rt_bvh_node_count(const rt_bvh *bvh) { return bvh ? bvh->bvh.n_nodes : 0; }
int32_t rt_bvh_tri_count(const rt_bvh *bvh) { return bvh ? bvh->bvh.n_tris : 0; }
int32_t rt_bvh_max_depth(const rt_bvh *bvh) { return bvh ? bvh->bvh.max_depth : 0; }

rt_status rt_bvh_export(const rt_bvh *bvh, int32_t *is_leaf, int32_t *start, int32_t *end,
                        int32_t *depth, float *bmin, float *bmax, float *split, int32_t *order) {
    RT_GUARD_BEGIN
    if (!bvh) return fail(RT_ERR_INVALID, "bvh is NULL");
    const rt::Bvh &b = bvh->bvh;
    const size_t nn = (size_t)b.n_nodes;
    if (is_leaf) std::memcpy(is_leaf, b.is_leaf.data(), nn * 4);
    if (start) std::memcpy(start, b.start.data(), nn * 4);
    if (end) std::memcpy(end, b.end.data(), nn * 4);
    if (depth) std::memcpy(depth, b.depth.data(), nn * 4);
    if (bmin) std::memcpy(bmin, b.bmin.data(), nn * 12);
    if (bmax) std::memcpy(bmax, b.bmax.data(), nn * 12);
    if (split) std::memcpy(split, b.split.data(), nn * 16);
    if (order) std::memcpy(order, b.order.data(), (size_t)b.n_tris * 4);
    return RT_OK;
    RT_GUARD_END
}

void rt_bvh_destroy(rt_bvh *bvh) { delete bvh; }

rt_status rt_flatten(const rt_mesh *mesh, const rt_bvh *bvh, float *nodes, float *tris,
                     int32_t *tri_material) {
    RT_GUARD_BEGIN
    if (!bvh) return fail(RT_ERR_INVALID, "bvh is NULL");
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    rt::FlatScene flat;
    st = rt::flatten(*mesh, bvh->bvh, flat);
    if (st != RT_OK) return st;
    if (nodes) std::memcpy(nodes, flat.nodes.data(), flat.nodes.size() * 4);
    if (tris) std::memcpy(tris, flat.tris.data(), flat.tris.size() * 4);
    if (tri_material) std::memcpy(tri_material, flat.tri_mat.data(), flat.tri_mat.size() * 4);
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_flatten_ordered(const rt_mesh *mesh, const rt_bvh *bvh, float *records,
                             size_t capacity_records, size_t *n_records, uint32_t *root_ref) {
    RT_GUARD_BEGIN
    if (!bvh || !n_records || !root_ref) return fail(RT_ERR_INVALID, "NULL argument");
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    rt::FlatScene flat;
    st = rt::flatten(*mesh, bvh->bvh, flat);
    if (st != RT_OK) return st;
    if (!flat.fast_ok) return fail(RT_ERR_INVALID, "tree does not fit the ordered-traversal encoding (more than 2^24-2 triangles, or a wide tree too deep for the traversal stack)");
    *n_records = flat.wide.size() / rtc::kWideFloats;
    *root_ref = flat.fast_root;
    if (records) {
        if (capacity_records < *n_records) return fail(RT_ERR_INVALID, "records buffer too small");
        std::memcpy(records, flat.wide.data(), flat.wide.size() * 4);
    }
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_mesh_light(const rt_mesh *mesh, const rt_bvh *bvh, rt_light *out) {
    RT_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    (void)bvh;   // the light does not depend on the tree: see the header
    rt::find_light(*mesh, *out);
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_pass_params(const rt_light *scene_light, const rt_pass *pass, void *out, size_t out_size) {
    RT_GUARD_BEGIN
    if (!scene_light || !pass || !out) return fail(RT_ERR_INVALID, "NULL argument");
    if (out_size < sizeof(rtc::PassParams)) return fail(RT_ERR_INVALID, "output block too small");
    rtc::PassParams pp;
    rt_status st = rt::prepare_pass(*scene_light, *pass, pp);
    if (st != RT_OK) return st;
    std::memcpy(out, &pp, sizeof pp);
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_scene_set_counters(rt_scene *scene, int32_t enabled) {
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    scene->impl->count_traversal = enabled != 0;
    return RT_OK;
}

rt_status rt_scene_set_tuning(rt_scene *scene, int32_t stack_entries, int32_t refill_below,
                              int32_t blocks_per_cu, int32_t descend_min, int32_t leaf_min) {
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    if (stack_entries != 8 && stack_entries != 12 && stack_entries != 16 && stack_entries != 24)
        return fail(RT_ERR_INVALID, "stack_entries must be 8, 12, 16 or 24");
    if (refill_below < 0 || refill_below > 64 || blocks_per_cu < 1 || blocks_per_cu > 64 ||
        descend_min < 0 || descend_min > 64 || leaf_min < 0 || leaf_min > 64)
        return fail(RT_ERR_INVALID, "tuning value out of range");
    scene->impl->tune_stack = stack_entries;
    scene->impl->tune_refill = refill_below;
    scene->impl->tune_blocks_per_cu = blocks_per_cu;
    scene->impl->tune_descend_min = descend_min;
    scene->impl->tune_leaf_min = leaf_min;
    return RT_OK;
}

rt_status rt_scene_set_max_paths(rt_scene *scene, uint64_t max_paths) {
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    if (max_paths < 64 || max_paths > ((uint64_t)1 << 29))
        return fail(RT_ERR_INVALID, "max_paths must be within [64, 2^29]");
    // render calls read the bound, and lower it when a batch does not fit, under this lock
    std::lock_guard<std::mutex> lock(scene->impl->render_mu);
    scene->impl->max_paths_in_flight = (size_t)max_paths;
    return RT_OK;
}

rt_status rt_scene_set_workspace_limit(rt_scene *scene, uint64_t max_bytes) {
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    return rt::scene_set_workspace_limit(*scene->impl, (size_t)max_bytes);
}

uint64_t rt_workspace_bytes_per_path(int32_t light_sample_rays, int32_t bounce_limit) {
    return (uint64_t)rt::workspace_bytes_per_path(light_sample_rays, bounce_limit & 0xff);
}

rt_status rt_scene_create(const rt_mesh *mesh, const rt_bvh *bvh, int32_t device, rt_scene **out) {
    RT_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!bvh) return fail(RT_ERR_STATE, "no BVH: build or load one before creating the scene");
    rt_status st = rt::check_mesh(mesh);
    if (st != RT_OK) return st;
    std::unique_ptr<rt_scene> s(new rt_scene);
    s->impl = nullptr;
    st = rt::scene_create(*mesh, bvh->bvh, device, &s->impl);
    if (st != RT_OK) return st;
    *out = s.release();
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_scene_info_get(const rt_scene *scene, rt_scene_info *out) {
    if (!scene || !scene->impl || !out) return fail(RT_ERR_INVALID, "NULL argument");
    rt::Scene &s = *scene->impl;
    std::memset(out, 0, sizeof *out);
    {
        std::lock_guard<std::mutex> lock(s.render_mu);
        out->max_paths_in_flight = s.max_paths_in_flight;
        out->workspace_bytes = s.ws_block_bytes;
    }
    out->n_tris = s.n_tris;
    out->n_nodes = s.n_nodes;
    out->max_depth = s.max_depth;
    out->n_materials = s.n_materials;
    out->device = s.device;
    out->has_light = s.light.found;
    out->node_bytes = s.node_bytes;
    out->tri_bytes = s.tri_bytes;
    out->ordered_bytes = s.wide_bytes;
    for (int a = 0; a < 3; ++a) {
        out->near_lo[a] = s.near_lo[a];
        out->near_hi[a] = s.near_hi[a];
    }
    out->linked_view = s.d_linked ? 1 : 0;
    out->ordered_view = s.d_wide ? 1 : 0;
    out->total_device_bytes = s.node_bytes * (s.d_linked ? 2 : 1) + s.wide_bytes + s.tri_bytes +
                              (uint64_t)s.n_tris * 16 + (uint64_t)(s.n_cam + s.n_light) * 16;
    return RT_OK;
}

rt_status rt_scene_light(const rt_scene *scene, rt_light *out) {
    if (!scene || !scene->impl || !out) return fail(RT_ERR_INVALID, "NULL argument");
    *out = scene->impl->light;
    return RT_OK;
}

rt_status rt_scene_set_sample_tables(rt_scene *scene, const double *camera_tab, int32_t n_camera,
                                     const double *light_tab, int32_t n_light) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    return rt::scene_set_tables(*scene->impl, camera_tab, n_camera, light_tab, n_light);
    RT_GUARD_END
}

rt_status rt_default_sample_table(int32_t n, uint64_t seed, double *out) {
    RT_GUARD_BEGIN
    if (n < 0 || (n > 0 && !out)) return fail(RT_ERR_INVALID, "bad table arguments");
    rt::default_sample_table(n, seed, out);
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_png_decode(const uint8_t *data, size_t size, uint8_t *rgba, size_t capacity,
                        int32_t *width, int32_t *height) {
    RT_GUARD_BEGIN
    if (!data || !width || !height) return fail(RT_ERR_INVALID, "NULL argument");
    std::vector<uint8_t> px;
    rt_status st = rt::decode_png(data, size, px, *width, *height);
    if (st != RT_OK) return st;
    if (rgba) {
        if (capacity < px.size()) return fail(RT_ERR_INVALID, "pixel buffer too small");
        std::memcpy(rgba, px.data(), px.size());
    }
    return RT_OK;
    RT_GUARD_END
}

rt_status rt_scene_set_texture(rt_scene *scene, int32_t texture_id, const uint8_t *rgba,
                               int32_t width, int32_t height) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    return rt::scene_set_texture(*scene->impl, texture_id, rgba, width, height);
    RT_GUARD_END
}

void rt_scene_destroy(rt_scene *scene) {
    if (!scene) return;
    rt::scene_destroy(scene->impl);
    delete scene;
}

rt_status rt_trace_rays(rt_scene *scene, const rt_ray *rays, size_t n, rt_hit *hits, int32_t mode,
                        int32_t want_counters, rt_stats *stats) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    return rt::trace_rays_host(*scene->impl, rays, n, hits, mode, want_counters != 0, stats);
    RT_GUARD_END
}

rt_status rt_trace_rays_device(rt_scene *scene, const void *d_rays, size_t n, void *d_hits,
                               int32_t mode, void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    if (n > 0 && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "device buffers are NULL");
    return rt::launch_trace(*scene->impl, static_cast<const rt_ray *>(d_rays), n, nullptr,
                            static_cast<rt_hit *>(d_hits), mode, nullptr,
                            static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_trace_rays_device_counted(rt_scene *scene, const void *d_rays, size_t n, void *d_hits,
                                       int32_t mode, void *d_counters, void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl) return fail(RT_ERR_INVALID, "scene is NULL");
    if (n > 0 && (!d_rays || !d_hits || !d_counters)) return fail(RT_ERR_INVALID, "device buffers are NULL");
    return rt::launch_trace(*scene->impl, static_cast<const rt_ray *>(d_rays), n, nullptr,
                            static_cast<rt_hit *>(d_hits), mode,
                            static_cast<unsigned long long *>(d_counters),
                            static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_camera_setup(const rt_camera *cam, int32_t total_width, int32_t total_height,
                          float top_left[3], float steps[2]) {
    if (!cam || !top_left || !steps) return fail(RT_ERR_INVALID, "NULL argument");
    rt::camera_setup(*cam, total_width, total_height, top_left, steps);
    return RT_OK;
}

rt_status rt_render_tile(rt_scene *scene, const rt_pass *pass, int32_t mode, float *accum_rgb,
                         uint8_t *rgba, rt_stats *stats) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl || !pass) return fail(RT_ERR_INVALID, "NULL argument");
    return rt::render_tile_host(*scene->impl, *pass, mode, accum_rgb, rgba, stats);
    RT_GUARD_END
}

rt_status rt_render_tile_device(rt_scene *scene, const rt_pass *pass, int32_t mode, void *d_accum,
                                void *d_rgba, rt_stats *stats, void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl || !pass) return fail(RT_ERR_INVALID, "NULL argument");
    return rt::render_accumulate(*scene->impl, pass, 1, mode, 0, pass->camera.rays_per_pixel, false,
                                 true, static_cast<float *>(d_accum),
                                 static_cast<uint8_t *>(d_rgba), stats,
                                 static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_render_accumulate(rt_scene *scene, const rt_pass *pass, int32_t mode,
                               int32_t sample_begin, int32_t sample_count, void *d_accum,
                               void *d_rgba, rt_stats *stats, void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl || !pass) return fail(RT_ERR_INVALID, "NULL argument");
    return rt::render_accumulate(*scene->impl, pass, 1, mode, sample_begin, sample_count, true, false,
                                 static_cast<float *>(d_accum), static_cast<uint8_t *>(d_rgba),
                                 stats, static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_render_tiles_device(rt_scene *scene, const rt_pass *passes, int32_t n_passes,
                                 int32_t mode, void *d_accum, void *d_rgba, rt_stats *stats,
                                 void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl || !passes) return fail(RT_ERR_INVALID, "NULL argument");
    if (n_passes < 1) return fail(RT_ERR_INVALID, "no pass given");
    return rt::render_accumulate(*scene->impl, passes, n_passes, mode, 0,
                                 passes[0].camera.rays_per_pixel, false, true,
                                 static_cast<float *>(d_accum), static_cast<uint8_t *>(d_rgba),
                                 stats, static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_render_tile_static(rt_scene *scene, const rt_pass *pass, int32_t mode, void *d_accum,
                                void *d_rgba, void *stream) {
    RT_GUARD_BEGIN
    if (!scene || !scene->impl || !pass) return fail(RT_ERR_INVALID, "NULL argument");
    return rt::render_tile_static(*scene->impl, *pass, mode, static_cast<float *>(d_accum),
                                  static_cast<uint8_t *>(d_rgba), static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

rt_status rt_gather_probe(int32_t device, size_t table_bytes, double *records_per_s) {
    RT_GUARD_BEGIN
    return rt::gather_probe(device, table_bytes, records_per_s);
    RT_GUARD_END
}

rt_status rt_resolve_device(const void *d_accum, size_t pixels, int32_t divisor, int32_t gamma_on,
                            float gamma, void *d_rgba, void *stream) {
    RT_GUARD_BEGIN
    return rt::resolve_device(static_cast<const float *>(d_accum), pixels, divisor, gamma_on, gamma,
                              static_cast<uint8_t *>(d_rgba), static_cast<hipStream_t>(stream));
    RT_GUARD_END
}

}  // extern "C"
